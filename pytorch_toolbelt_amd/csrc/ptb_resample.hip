// ptb_resample.hip -- bilinear resize for multiscale TTA (reference inference/tta.py:599-621, 645-689, which call
// torch.nn.functional.interpolate(mode="bilinear")).  4-tap gather, HBM/L2-bound; index and weight arithmetic
// follows ATen's area_pixel_compute_source_index / compute_source_index_and_lambda in fp32 so results match the
// reference's torch kernels to rounding.
#include <algorithm>

#include "ptb_view_device.h"

namespace ptb {

struct Taps { int i0, i1; float l0, l1; };

template <int ALIGN = -1>   // -1: run-time align_corners; 0 / 1: compile-time (no branch in the unrolled tap code)
__device__ __forceinline__ Taps taps(int dst, float scale, int n_in, bool align_corners) {
    float src;
    if (ALIGN < 0 ? align_corners : (ALIGN == 1)) {
        src = scale * (float)dst;
    } else {
        src = scale * ((float)dst + 0.5f) - 0.5f;
        src = src < 0.f ? 0.f : src;
    }
    Taps t;
    t.i0 = min((int)src, n_in - 1);
    t.i1 = t.i0 + (t.i0 < n_in - 1 ? 1 : 0);
    t.l1 = fminf(fmaxf(src - (float)t.i0, 0.f), 1.f);
    t.l0 = 1.f - t.l1;
    return t;
}

// one thread = 4 consecutive output columns of one output row (16 B store per lane)
__global__ __launch_bounds__(256) void resize_bilinear_kernel(const float* __restrict__ in, float* __restrict__ out, int planes,
                                                              int hin, int win, int hout, int wout, float sh, float sw,
                                                              int align_corners) {
    const int wq = (wout + 3) / 4;
    const long long total = (long long)planes * hout * wq;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
        const int q = (int)(idx % wq);
        const long long rest = idx / wq;
        const int oy = (int)(rest % hout);
        const long long p = rest / hout;
        const Taps ty = taps(oy, sh, hin, align_corners);
        const float* r0 = in + (p * hin + ty.i0) * (long long)win;
        const float* r1 = in + (p * hin + ty.i1) * (long long)win;
        float res[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int ox = 4 * q + m;
            if (ox < wout) {
                const Taps tx = taps(ox, sw, win, align_corners);
                const float top = tx.l0 * r0[tx.i0] + tx.l1 * r0[tx.i1];
                const float bot = tx.l0 * r1[tx.i0] + tx.l1 * r1[tx.i1];
                res[m] = ty.l0 * top + ty.l1 * bot;
            }
        }
        float* o = out + (p * hout + oy) * (long long)wout + 4 * q;
        if ((wout & 3) == 0) {
            out_store4(o, make_float4(res[0], res[1], res[2], res[3]));
        } else {
#pragma unroll
            for (int m = 0; m < 4; ++m) if (4 * q + m < wout) o[m] = res[m];
        }
    }
}

// ---------------------------------------------------------------------------------------------- fused multiscale
// ms_image_deaugment (tta.py:645-689) in ONE pass: every scale's prediction is sampled at the output grid (4-tap
// bilinear, or a straight 16 B load when the sizes already match) and reduced in registers -- the reference resizes
// each map to full size (write + read), stacks them (another copy) and reduces in further passes.
constexpr int MS_MAX = 8;
struct MsArgs {
    const float* in[MS_MAX];
    int h[MS_MAX], w[MS_MAX];
    float sh[MS_MAX], sw[MS_MAX];
    int n;           // number of scales
    int planes, hout, wout, align_corners, op;
    // row strips (one rank of a multi-GPU job computes output rows [row0, row0 + hout) of a hout_full-row result from
    // source row strips): hfull[s] = full source height (tap arithmetic), src0[s] = global index of the first source row
    // held in in[s] (which has h[s] rows).  Single GPU: row0 = 0, hout_full = hout, hfull = h, src0 = 0.
    int row0, hout_full;
    int hfull[MS_MAX], src0[MS_MAX];
};

// local row of global source row i inside the strip held for scale s (clamped: the caller provides every row the taps need)
__device__ __forceinline__ int ms_local(const MsArgs& a, int s, int i) { return min(max(i - a.src0[s], 0), a.h[s] - 1); }

constexpr float kMsEps = 1e-6f;
constexpr float kMsOneMinusEps = (float)(1.0 - 1e-6);

// v_log_f32 / v_exp_f32 based log/exp (absolute error ~1e-7 on the [0,1] probability range this path is used for, far
// inside the 1e-5 tolerance); the full-precision libm versions made this kernel ALU-bound.
__device__ __forceinline__ float ms_log(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }
__device__ __forceinline__ float ms_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }

__device__ __forceinline__ float ms_pre(float x, int op) {
    switch (op) {
        case PTB_RED_GMEAN: return ms_log(x);
        case PTB_RED_HMEAN: return 1.0f / (x < kMsEps ? kMsEps : x);
        case PTB_RED_HARMONIC1P: return 1.0f / (x + 1.0f);
        case PTB_RED_LOGODD: { const float p = x < kMsEps ? kMsEps : (x > kMsOneMinusEps ? kMsOneMinusEps : x); return logf(p / (1.0f - p)); }
        case PTB_RED_LOG1P: return log1pf(x);
        default: return x;
    }
}
__device__ __forceinline__ float ms_post(float s, int op, float n) {
    if (op == PTB_RED_SUM) return s;
    const float m = s / n;
    switch (op) {
        case PTB_RED_GMEAN: return ms_exp(m);
        case PTB_RED_HMEAN: return 1.0f / (m < kMsEps ? kMsEps : m);
        case PTB_RED_HARMONIC1P: return 1.0f / m - 1.0f;
        case PTB_RED_LOGODD: { const float e = expf(m); return e / (1.0f + e); }
        case PTB_RED_LOG1P: return expf(m) - 1.0f;
        default: return m;
    }
}

__global__ __launch_bounds__(256) void ms_reduce_kernel(const MsArgs a, float* __restrict__ out) {
    const int wq = (a.wout + 3) / 4;
    const long long total = (long long)a.planes * a.hout * wq;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const bool vec_out = (a.wout & 3) == 0;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
        const int q = (int)(idx % wq);
        const long long rest = idx / wq;
        const int oy = (int)(rest % a.hout);
        const long long p = rest / a.hout;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < a.n; ++s) {
            float v[4];
            const int hin = a.h[s], win = a.w[s];
            const int gy = oy + a.row0;
            if (a.hfull[s] == a.hout_full && win == a.wout) {  // same size: F.interpolate is skipped by the reference (offset 0)
                const float* r = a.in[s] + (p * hin + ms_local(a, s, gy)) * (long long)win + 4 * q;
                if (vec_out) { const float4 t = *reinterpret_cast<const float4*>(r); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
                else { for (int m = 0; m < 4; ++m) v[m] = 4 * q + m < a.wout ? r[m] : 1.f; }
            } else {
                const Taps ty = taps(gy, a.sh[s], a.hfull[s], a.align_corners);
                const float* r0 = a.in[s] + (p * hin + ms_local(a, s, ty.i0)) * (long long)win;
                const float* r1 = a.in[s] + (p * hin + ms_local(a, s, ty.i1)) * (long long)win;
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const int ox = 4 * q + m;
                    v[m] = 1.f;
                    if (ox < a.wout) {
                        const Taps tx = taps(ox, a.sw[s], win, a.align_corners);
                        const float top = tx.l0 * r0[tx.i0] + tx.l1 * r0[tx.i1];
                        const float bot = tx.l0 * r1[tx.i0] + tx.l1 * r1[tx.i1];
                        v[m] = ty.l0 * top + ty.l1 * bot;
                    }
                }
            }
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const float t = a.op >= PTB_RED_GMEAN ? ms_pre(v[m], a.op) : v[m];
                acc[m] = s ? acc[m] + t : t;
            }
        }
        float res[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) res[m] = ms_post(acc[m], a.op, (float)a.n);
        float* o = out + (p * a.hout + oy) * (long long)a.wout + 4 * q;
        if (vec_out) out_store4(o, make_float4(res[0], res[1], res[2], res[3]));
        else for (int m = 0; m < 4; ++m) if (4 * q + m < a.wout) o[m] = res[m];
    }
}

// Tiled variant: a workgroup owns a 64 x 16 output tile; for every resized scale it first copies the source window the
// tile's taps fall into (<= 24 x 96 floats for ratios up to 1.25) into LDS with coalesced 16-byte loads, then all
// 4-tap gathers hit LDS instead of issuing 16 scattered 4-byte global loads per lane and scale.  Windows that do not
// fit (strong down-sampling) fall back to direct global gathers for that scale.
// LDS row pitch MS_LP = 112 floats (= 16 mod 32 banks): the 2 x 16 lanes a ds_read serves per cycle belong to two
// output rows; with a pitch of 96 (= 0 mod 32) both rows hit the same banks (rocprof: 54 % of the LDS cycles were bank
// conflicts), with 16 mod 32 the two rows' stride-3 / stride-5 lane patterns (scale 0.75 / 1.25) interleave exactly.
constexpr int MS_TW = 64, MS_TH = 16, MS_LR = 24, MS_LC = 96, MS_LP = 112;

// OPK: 0 = sum / mean, 2 = gmean (branch-free: the run-time switch over all reductions costs more than the gathers),
// 1 = any other reduction
template <int OPK, int ALIGN>
__global__ __launch_bounds__(256) void ms_reduce_tiled_kernel(const MsArgs a, float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float lds[MS_LR * MS_LP];
    __shared__ __attribute__((aligned(16))) Taps ctap[MS_TW];  // the tile's 64 column taps, computed once per scale
    const int tiles_x = (a.wout + MS_TW - 1) / MS_TW, tiles_y = (a.hout + MS_TH - 1) / MS_TH;
    int bid = blockIdx.x;
    const int txi = bid % tiles_x;
    bid /= tiles_x;
    const int tyi = bid % tiles_y;
    const long long p = bid / tiles_y;
    const int tid = threadIdx.x, lx = tid & 15, ly = tid >> 4;
    const int ox0 = txi * MS_TW, oy0 = tyi * MS_TH;
    const int oy = oy0 + ly, ox = ox0 + 4 * lx;
    const bool row_ok = oy < a.hout;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < a.n; ++s) {
        float v[4] = {1.f, 1.f, 1.f, 1.f};
        const int hin = a.h[s], win = a.w[s], hfull = a.hfull[s], src0 = a.src0[s];
        auto local_row = [&](int i) { return min(max(i - src0, 0), hin - 1); };   // global source row -> row of this rank's strip
        const float* src = a.in[s] + p * (long long)hin * win;
        const int gy = oy + a.row0, gy0 = oy0 + a.row0;
        if (hfull == a.hout_full && win == a.wout) {
            const int ry = local_row(gy);
            if (row_ok && ox + 3 < a.wout) {
                const float4 t = *reinterpret_cast<const float4*>(src + (long long)ry * win + ox);
                v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
            } else if (row_ok) {
                for (int m = 0; m < 4; ++m) if (ox + m < a.wout) v[m] = src[(long long)ry * win + ox + m];
            }
        } else {
            // workgroup-uniform source window of this tile
            const int oy_last = min(oy0 + MS_TH, a.hout) - 1, ox_last = min(ox0 + MS_TW, a.wout) - 1;
            // window rows in GLOBAL source coordinates, then shifted into the strip this rank holds
            const int r_lo = taps<ALIGN>(gy0, a.sh[s], hfull, a.align_corners).i0, r_hi = taps<ALIGN>(oy_last + a.row0, a.sh[s], hfull, a.align_corners).i1;
            const int c_lo = taps<ALIGN>(ox0, a.sw[s], win, a.align_corners).i0 & ~3, c_hi = taps<ALIGN>(ox_last, a.sw[s], win, a.align_corners).i1;
            const int nr = r_hi - r_lo + 1, nc = c_hi - c_lo + 1;
            const bool staged = nr <= MS_LR && nc <= MS_LC && (win & 3) == 0;
            if (staged) {
                // 32 lanes x 16 B cover a window row (<= 24 float4), 8 rows per pass: no integer division per element
                const int q_per_row = (nc + 3) / 4;
                const int q4 = tid & 31, col = c_lo + 4 * q4;
                if (q4 < q_per_row && col < win) {  // win % 4 == 0 and col % 4 == 0: the whole float4 is inside the row
                    for (int rr = tid >> 5; rr < nr; rr += 8)
                        *reinterpret_cast<float4*>(&lds[rr * MS_LP + 4 * q4]) =
                            *reinterpret_cast<const float4*>(src + (long long)local_row(r_lo + rr) * win + col);
                }
                if (tid < MS_TW) ctap[tid] = taps<ALIGN>(min(ox0 + tid, a.wout - 1), a.sw[s], win, a.align_corners);
                __syncthreads();
            }
            if (row_ok) {
                // branch-free: columns past the right edge use the clamped tap of the last column (computed, never
                // stored), so all 16 gathers of the lane are issued back to back instead of one pixel at a time
                Taps ty = taps<ALIGN>(gy, a.sh[s], hfull, a.align_corners);
                Taps tx[4];
                float t[4][4];
                if (staged) {
#pragma unroll
                    for (int m = 0; m < 4; ++m) tx[m] = ctap[4 * lx + m];
                    const float* l0 = lds + (ty.i0 - r_lo) * MS_LP - c_lo;
                    const float* l1 = lds + (ty.i1 - r_lo) * MS_LP - c_lo;
#pragma unroll
                    for (int m = 0; m < 4; ++m) { t[m][0] = l0[tx[m].i0]; t[m][1] = l0[tx[m].i1]; t[m][2] = l1[tx[m].i0]; t[m][3] = l1[tx[m].i1]; }
                } else {
#pragma unroll
                    for (int m = 0; m < 4; ++m) tx[m] = taps<ALIGN>(min(ox + m, a.wout - 1), a.sw[s], win, a.align_corners);
                    const float* g0 = src + (long long)local_row(ty.i0) * win;
                    const float* g1 = src + (long long)local_row(ty.i1) * win;
#pragma unroll
                    for (int m = 0; m < 4; ++m) { t[m][0] = g0[tx[m].i0]; t[m][1] = g0[tx[m].i1]; t[m][2] = g1[tx[m].i0]; t[m][3] = g1[tx[m].i1]; }
                }
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    v[m] = ty.l0 * (tx[m].l0 * t[m][0] + tx[m].l1 * t[m][1]) + ty.l1 * (tx[m].l0 * t[m][2] + tx[m].l1 * t[m][3]);
            }
            if (staged) __syncthreads();  // the next scale reuses the LDS window
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const float t = OPK == 0 ? v[m] : (OPK == 2 ? ms_log(v[m]) : ms_pre(v[m], a.op));
            acc[m] = s ? acc[m] + t : t;
        }
    }
    if (!row_ok) return;
    float res[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) res[m] = OPK == 2 ? ms_exp(acc[m] / (float)a.n) : ms_post(acc[m], a.op, (float)a.n);
    float* o = out + (p * a.hout + oy) * (long long)a.wout + ox;
    if ((a.wout & 3) == 0 && ox + 3 < a.wout) out_store4(o, make_float4(res[0], res[1], res[2], res[3]));
    else for (int m = 0; m < 4; ++m) if (ox + m < a.wout) o[m] = res[m];
}

// ---------------------------------------------------------------------------------------------- fused flips + multiscale
// BASELINE configs[4]: every scale's model output is itself a flip-augmented batch [V*B, C, h_s, w_s] (tta.py:257-284,
// 319-341, 470-484), de-augmented per scale (tta.py:287-316, 344-365, 503-524) and then merged over the scales
// (tta.py:645-689).  As separate calls that is one pass per scale that writes the flip-reduced map and one more that reads it
// back; here ONE launch reads every view of every scale once and writes the merged map once:
//     out = outer_s( bilinear_s( inner_v( flip_v^-1( y_s[v] ) ) ) )
// A workgroup owns a 64 x TH output tile (TH = 32 by default: the 64 x 16 tiles of ms_reduce_tiled_kernel refetch 1.21x the source
// bytes -- halo rows / columns and 16-byte alignment of every window --, 64 x 32 tiles 1.13x, 64 x 64 tiles 1.08x).  Per resized scale the tile's source window
// (<= 88 x 96) is brought in with 16-byte loads, 4 window rows x V views in flight per lane, mirrored views read at mirrored
// addresses (a reversed row is still one contiguous segment); the inner reduction over the views happens ON THE WAY INTO LDS, so
// the window holds flip-reduced values exactly like the reference's intermediate tensor and the 4-tap gathers never see the
// views.  Scales that already have the output size are reduced straight from registers.  Only row-preserving views (flips
// of rows / columns) -- the groups the reference combines with multiscale TTA.
constexpr int FZ_T = 64, FZ_LR = 88, FZ_LR32 = 46, FZ_LR16 = 25, FZ_LC = 96, FZ_LP = 112, FZ_VMAX = 4;

struct FzArgs {
    const float* in[MS_MAX];           // view 0 of scale s: [planes, h, w]; view v lies v * planes * h * w elements further
    int h[MS_MAX], w[MS_MAX];
    float sh[MS_MAX], sw[MS_MAX];
    int n, nviews, codes;              // codes: 3 bits per view (bit 1 = flip rows, bit 2 = flip columns; bit 0 must be 0)
    int planes, hout, wout, align_corners, op_outer, op_inner;
    float inner_div;
    // IEEE divisions cost ~12 vector instructions each and the mean of V views / n scales needs one per value (PMC: they were 40 %
    // of this kernel's 2 142 vector instructions per wave).  inner_mul != 0: multiply by it instead (V a power of two: the
    // reciprocal is exact, bit-identical to the division); outer_mul != 0: likewise for the mean over the scales (used whenever a
    // scale is resized -- the interpolation arithmetic already differs from ATen's in the last bits; exact division otherwise).
    float inner_mul, outer_mul;
    int strip;                         // > 0: XCD-aware tile order, strips of this many tile columns (see the kernel); 0: row-major
    long long total_tiles;
    // Row strips (one rank of a multi-GPU multiscale merge): the launch produces output rows oy_base .. oy_base + hout - 1 of a
    // hout_full-row map from the source rows row0[s] .. row0[s] + hs[s] - 1 of every scale; h[s] stays the FULL height (the taps are
    // those of the full-size call, so the strips of the result concatenate to it bit for bit).  Whole maps: oy_base = 0, row0 = 0,
    // hs = h, hout_full = hout.  Views that flip rows do not come in strips.
    int hs[MS_MAX], row0[MS_MAX];
    int oy_base, hout_full;
};

// 4 consecutive de-augmented values of view `code` of a [h, w] plane at (row, col) (col % 4 == 0, w % 4 == 0)
__device__ __forceinline__ float4 fz_load(const float* __restrict__ plane, int h, int w, int row, int col, int code) {
    const int r = (code & 2) ? h - 1 - row : row;
    const int c = (code & 4) ? w - 4 - col : col;
    const float4 t = ld16<true>(plane + (long long)r * w + c);
    return (code & 4) ? make_float4(t.w, t.z, t.y, t.x) : t;
}

template <int NV, int INNER>
__device__ __forceinline__ float4 fz_inner(const float4 (&x)[NV], int nv, int op, float div, float mul) {
    float4 s;
    if (INNER == 0) {
        s = x[0];
#pragma unroll
        for (int k = 1; k < NV; ++k)
            if (k < nv) { s.x = __fadd_rn(s.x, x[k].x); s.y = __fadd_rn(s.y, x[k].y); s.z = __fadd_rn(s.z, x[k].z); s.w = __fadd_rn(s.w, x[k].w); }
        if (mul != 0.f) return make_float4(s.x * mul, s.y * mul, s.z * mul, s.w * mul);     // sum (mul = 1) or an exact reciprocal
        return make_float4(s.x / div, s.y / div, s.z / div, s.w / div);
    }
    if (INNER == 2 && NV == 2) {
        // gmean of exactly two views: exp(mean(log a, log b)) = sqrt(a b) -- ONE quarter-rate instruction per element instead of
        // three (2 x v_log + v_exp).  The product is formed as (a 2^64) b and the root scaled back by 2^-32, which keeps every
        // pair of probabilities down to ~1e-29 each exact to rounding; a wave that holds a product outside [2^-100, 2^120) --
        // vanishing probabilities, or inputs that are no probabilities at all -- takes sqrt(a) sqrt(b) instead (never on sane data).
        const float S = 0x1p64f, Si = 0x1p-32f, lo = 0x1p-100f, hi = 0x1p120f;
        const float4 t = make_float4(x[0].x * S * x[1].x, x[0].y * S * x[1].y, x[0].z * S * x[1].z, x[0].w * S * x[1].w);
        const bool odd = !(t.x >= lo && t.x < hi && t.y >= lo && t.y < hi && t.z >= lo && t.z < hi && t.w >= lo && t.w < hi);
        if (__any(odd))
            return make_float4(__builtin_amdgcn_sqrtf(x[0].x) * __builtin_amdgcn_sqrtf(x[1].x), __builtin_amdgcn_sqrtf(x[0].y) * __builtin_amdgcn_sqrtf(x[1].y),
                               __builtin_amdgcn_sqrtf(x[0].z) * __builtin_amdgcn_sqrtf(x[1].z), __builtin_amdgcn_sqrtf(x[0].w) * __builtin_amdgcn_sqrtf(x[1].w));
        return make_float4(__builtin_amdgcn_sqrtf(t.x) * Si, __builtin_amdgcn_sqrtf(t.y) * Si, __builtin_amdgcn_sqrtf(t.z) * Si, __builtin_amdgcn_sqrtf(t.w) * Si);
    }
    if (INNER == 2) {   // gmean, branch-free (the run-time switch over all reductions costs more than the loads it sits between)
        // in the log2 domain: exp2(mean(log2 x)) == exp(mean(log x)) without the two constant multiplies per value
        s = make_float4(__builtin_amdgcn_logf(x[0].x), __builtin_amdgcn_logf(x[0].y), __builtin_amdgcn_logf(x[0].z), __builtin_amdgcn_logf(x[0].w));
#pragma unroll
        for (int k = 1; k < NV; ++k)
            if (k < nv) { s.x += __builtin_amdgcn_logf(x[k].x); s.y += __builtin_amdgcn_logf(x[k].y); s.z += __builtin_amdgcn_logf(x[k].z); s.w += __builtin_amdgcn_logf(x[k].w); }
        const float inv = fast_rcp(div);
        return make_float4(__builtin_amdgcn_exp2f(s.x * inv), __builtin_amdgcn_exp2f(s.y * inv), __builtin_amdgcn_exp2f(s.z * inv), __builtin_amdgcn_exp2f(s.w * inv));
    }
    s = make_float4(red_pre<1>(x[0].x, op), red_pre<1>(x[0].y, op), red_pre<1>(x[0].z, op), red_pre<1>(x[0].w, op));
#pragma unroll
    for (int k = 1; k < NV; ++k)
        if (k < nv) {
            s.x += red_pre<1>(x[k].x, op); s.y += red_pre<1>(x[k].y, op); s.z += red_pre<1>(x[k].z, op); s.w += red_pre<1>(x[k].w, op);
        }
    return make_float4(red_post<1>(s.x, op, div), red_post<1>(s.y, op, div), red_post<1>(s.z, op, div), red_post<1>(s.w, op, div));
}

// (the mean / mean and gmean / gmean instances of the default tile height are pinned at 6 waves per SIMD: 80 VGPRs is an occupancy cliff, and a two-register
// drift of the allocator -- 82 VGPRs, 5 waves -- cost them 7 %; the other instances keep whatever the allocator chooses)
// TW = 128 (with TH = 16): the same 2048 output pixels per workgroup as the default 64 x 32 tile, but twice as wide -- the 16-byte
// alignment slack of every window row and the taps' extra column are paid once per 128 instead of per 64 columns, and a row of
// tiles is half as many source bytes, so the two halo rows a tile shares with the tile below it are still in the XCD's L2 when
// that one runs (ptb_set_tunable key 15).
template <int NV, int INNER, int OUTER, int ALIGN, int TH, int TW = 64>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((NV <= 2 && ((TH == 32 && TW == 64) || (TH == 16 && TW == 128)) && ((INNER == 2 && OUTER == 2) || (INNER == 0 && OUTER == 0))) ? 6 : 1)))
void ms_flip_reduce_kernel(const FzArgs a, float* __restrict__ out) {
    constexpr int LX = TW / 4, RPP = 256 / LX;          // lanes across a tile row (4 columns each), output rows per pass of the workgroup
    constexpr int R = TH / RPP;                         // output rows per thread (tile = TW columns x TH rows)
    constexpr int FZ_T = TW, FZ_LC = TW == 128 ? 168 : ptb::FZ_LC, FZ_LP = TW == 128 ? 176 : ptb::FZ_LP;
    static_assert(TH % RPP == 0 && R >= 1, "tile shape");
    constexpr int LR = TH == 64 ? FZ_LR : (TH == 32 ? FZ_LR32 : FZ_LR16);      // LDS window rows
    constexpr int FZ_U = NV <= 2 ? 3 : 2;   // window slots a lane has in flight at once (x NV views)
    __shared__ __attribute__((aligned(16))) float lds[LR * FZ_LP];
    __shared__ __attribute__((aligned(16))) Taps ctap[FZ_T];
    const int tiles_x = (a.wout + FZ_T - 1) / FZ_T, tiles_y = (a.hout + TH - 1) / TH;
    int txi, tyi;
    long long p;
    if (a.strip > 0) {
        // XCD-aware order: the dispatcher deals workgroups round-robin over the 8 XCDs, each with its own L2.  Workgroup b therefore
        // becomes tile  (b % 8) * ceil(N / 8) + b / 8  of a walk that goes down strips of `strip` tile columns: neighbouring tiles --
        // which share halo rows / columns and the 16-byte alignment overlap of their source windows -- run on the same XCD within a
        // few hundred workgroups of each other, so the second reader of those bytes finds them in that L2 instead of in HBM.
        const long long per_xcd = (a.total_tiles + 7) / 8;
        const long long L = (long long)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
        if (L >= a.total_tiles) return;
        const int tpp = tiles_x * tiles_y;
        p = L / tpp;
        const int r = (int)(L - p * tpp);
        const int per_strip = a.strip * tiles_y;
        const int st = r / per_strip, r2 = r - st * per_strip;
        const int width = min(a.strip, tiles_x - st * a.strip);
        tyi = r2 / width;
        txi = st * a.strip + (r2 - tyi * width);
    } else {
        int bid = blockIdx.x;
        txi = bid % tiles_x;
        bid /= tiles_x;
        tyi = bid % tiles_y;
        p = bid / tiles_y;
    }
    const int tid = threadIdx.x, lx = tid & (LX - 1), ly = tid / LX;
    const int ox0 = txi * FZ_T, oy0 = tyi * TH, ox = ox0 + 4 * lx;
    const bool col_ok = ox < a.wout;
    float acc[R][4];
#pragma unroll
    for (int j = 0; j < R; ++j)
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[j][m] = 0.f;
    for (int s = 0; s < a.n; ++s) {
        const int hin = a.h[s], win = a.w[s];
        const long long plane_sz = (long long)a.hs[s] * win, vstride = (long long)a.planes * plane_sz;
        const float* src = a.in[s] + p * plane_sz;
        float v[R][4];
#pragma unroll
        for (int j = 0; j < R; ++j)
#pragma unroll
            for (int m = 0; m < 4; ++m) v[j][m] = 1.f;
        if (hin == a.hout_full && win == a.wout) {
            // same size: the reference skips F.interpolate (offset 0) -- reduce the views straight from registers
            float4 xs[R][NV];
#pragma unroll
            for (int j = 0; j < R; ++j) {
                const int oy = oy0 + ly + RPP * j;
#pragma unroll
                for (int k = 0; k < NV; ++k) {
                    xs[j][k] = make_float4(1.f, 1.f, 1.f, 1.f);
                    if (k < a.nviews && col_ok && oy < a.hout) xs[j][k] = fz_load(src + k * vstride, a.hs[s], win, a.oy_base + oy - a.row0[s], ox, (a.codes >> (3 * k)) & 7);
                }
            }
#pragma unroll
            for (int j = 0; j < R; ++j) {
                const float4 r = fz_inner<NV, INNER>(xs[j], a.nviews, a.op_inner, a.inner_div, a.inner_mul);
                v[j][0] = r.x; v[j][1] = r.y; v[j][2] = r.z; v[j][3] = r.w;
            }
        } else {
            const int oy_last = min(oy0 + TH, a.hout) - 1, ox_last = min(ox0 + FZ_T, a.wout) - 1;
            const int r_lo = taps<ALIGN>(a.oy_base + oy0, a.sh[s], hin, a.align_corners).i0, r_hi = taps<ALIGN>(a.oy_base + oy_last, a.sh[s], hin, a.align_corners).i1;
            const int c_lo = taps<ALIGN>(ox0, a.sw[s], win, a.align_corners).i0 & ~3, c_hi = taps<ALIGN>(ox_last, a.sw[s], win, a.align_corners).i1;
            const int nr = min(r_hi - r_lo + 1, LR), nc = min(c_hi - c_lo + 1, FZ_LC);   // (the host only launches shapes that fit)
            // every lane takes 16-byte slots tid, tid + 256, ... of the window (row-major, q_per_row slots per row: all lanes busy,
            // whatever the window width); FZ_U slots x V views are requested together
            const int q_per_row = min((nc + 3) / 4, (win - c_lo + 3) / 4);
            const int total = nr * q_per_row;
            const unsigned recip = (65536u + q_per_row - 1) / q_per_row;      // slot / q_per_row == (slot * recip) >> 16 for slot < 2112
            for (int s0 = tid; s0 < total; s0 += 256 * FZ_U) {
                float4 x[FZ_U][NV];
                int row[FZ_U], qq[FZ_U];
#pragma unroll
                for (int u = 0; u < FZ_U; ++u) {
                    const int slot = s0 + 256 * u;
                    row[u] = (int)(((unsigned)slot * recip) >> 16);
                    qq[u] = slot - row[u] * q_per_row;
#pragma unroll
                    for (int k = 0; k < NV; ++k) {
                        x[u][k] = make_float4(1.f, 1.f, 1.f, 1.f);
                        if (k < a.nviews && slot < total) x[u][k] = fz_load(src + k * vstride, a.hs[s], win, r_lo + row[u] - a.row0[s], c_lo + 4 * qq[u], (a.codes >> (3 * k)) & 7);
                    }
                }
#pragma unroll
                for (int u = 0; u < FZ_U; ++u)
                    if (s0 + 256 * u < total) *reinterpret_cast<float4*>(&lds[row[u] * FZ_LP + 4 * qq[u]]) = fz_inner<NV, INNER>(x[u], a.nviews, a.op_inner, a.inner_div, a.inner_mul);
            }
            if (tid < FZ_T) ctap[tid] = taps<ALIGN>(min(ox0 + tid, a.wout - 1), a.sw[s], win, a.align_corners);
            __syncthreads();
            Taps tx[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) tx[m] = ctap[4 * lx + m];
#pragma unroll
            for (int j = 0; j < R; ++j) {
                const int oy = min(oy0 + ly + RPP * j, a.hout - 1);   // rows past the bottom edge: computed from the last row, never stored
                const Taps ty = taps<ALIGN>(a.oy_base + oy, a.sh[s], hin, a.align_corners);
                const float* l0 = lds + min(ty.i0 - r_lo, LR - 1) * FZ_LP - c_lo;
                const float* l1 = lds + min(ty.i1 - r_lo, LR - 1) * FZ_LP - c_lo;
                float t[4][4];
#pragma unroll
                for (int m = 0; m < 4; ++m) { t[m][0] = l0[tx[m].i0]; t[m][1] = l0[tx[m].i1]; t[m][2] = l1[tx[m].i0]; t[m][3] = l1[tx[m].i1]; }
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    v[j][m] = ty.l0 * (tx[m].l0 * t[m][0] + tx[m].l1 * t[m][1]) + ty.l1 * (tx[m].l0 * t[m][2] + tx[m].l1 * t[m][3]);
            }
            __syncthreads();  // the next scale reuses the LDS window
        }
#pragma unroll
        for (int j = 0; j < R; ++j)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const float t = OUTER == 0 ? v[j][m] : (OUTER == 2 ? __builtin_amdgcn_logf(v[j][m]) : ms_pre(v[j][m], a.op_outer));   // (gmean: log2 domain)
                acc[j][m] = s ? acc[j][m] + t : t;
            }
    }
    if (!col_ok) return;
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const int oy = oy0 + ly + RPP * j;
        if (oy < a.hout) {
            float r[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                if (OUTER == 2) r[m] = __builtin_amdgcn_exp2f(acc[j][m] * a.outer_mul);
                else if (OUTER == 0) r[m] = a.outer_mul != 0.f ? acc[j][m] * a.outer_mul : acc[j][m] / (float)a.n;
                else r[m] = ms_post(acc[j][m], a.op_outer, (float)a.n);
            }
            out_store4(out + (p * a.hout + oy) * (long long)a.wout + ox, make_float4(r[0], r[1], r[2], r[3]));
        }
    }
}

// ---------------------------------------------------------------------------------------------- nearest + backward passes
// TTA "respects gradient flow" (tta.py:3-4): the adjoints of the resize kernels.  A bilinear output pixel reads 4 taps, so its
// gradient is added to 4 source pixels; several output pixels share a source pixel, hence fp32 atomics (global_atomic_add_f32,
// like ATen's upsample_bilinear2d_backward -- the summation order, and with it the last bit, is not deterministic).
__global__ __launch_bounds__(256) void resize_bilinear_bwd_kernel(const float* __restrict__ gout, float* __restrict__ gin, int planes, int hin,
                                                                  int win, int hout, int wout, float sh, float sw, int align_corners) {
    const long long total = (long long)planes * hout * wout;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
        const int ox = (int)(idx % wout);
        const long long rest = idx / wout;
        const int oy = (int)(rest % hout);
        const long long p = rest / hout;
        const Taps ty = taps(oy, sh, hin, align_corners), tx = taps(ox, sw, win, align_corners);
        const float g = gout[idx];
        float* r0 = gin + (p * hin + ty.i0) * (long long)win;
        float* r1 = gin + (p * hin + ty.i1) * (long long)win;
        atomicAdd(r0 + tx.i0, g * ty.l0 * tx.l0);
        atomicAdd(r0 + tx.i1, g * ty.l0 * tx.l1);
        atomicAdd(r1 + tx.i0, g * ty.l1 * tx.l0);
        atomicAdd(r1 + tx.i1, g * ty.l1 * tx.l1);
    }
}

// F.interpolate(mode="nearest"): src = min(floor(dst * in / out), in - 1) (ATen nearest_neighbor_compute_source_index)
__device__ __forceinline__ int nearest_src(int dst, float scale, int n_in) { return min((int)floorf((float)dst * scale), n_in - 1); }

template <bool BWD>
__global__ __launch_bounds__(256) void resize_nearest_kernel(const float* __restrict__ in, float* __restrict__ out, int planes, int hin, int win,
                                                             int hout, int wout, float sh, float sw) {
    // forward: out[p, oy, ox] = in[p, sy, sx];  BWD: `in` = grad_out [planes, hout, wout], `out` = grad_in [planes, hin, win] (zeroed)
    const long long total = (long long)planes * hout * wout;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
        const int ox = (int)(idx % wout);
        const long long rest = idx / wout;
        const int oy = (int)(rest % hout);
        const long long p = rest / hout;
        const long long src = (p * hin + nearest_src(oy, sh, hin)) * (long long)win + nearest_src(ox, sw, win);
        if (BWD) atomicAdd(out + src, in[idx]);
        else out[idx] = in[src];
    }
}

// F.interpolate(mode="nearest-exact"): src = min(floor((dst + 0.5) * in / out), in - 1) (ATen nearest_neighbor_exact_compute_source_index)
__device__ __forceinline__ int nearest_exact_src(int dst, float scale, int n_in) { return min((int)floorf(((float)dst + 0.5f) * scale), n_in - 1); }

template <bool BWD>
__global__ __launch_bounds__(256) void resize_nearest_exact_kernel(const float* __restrict__ in, float* __restrict__ out, int planes, int hin, int win,
                                                                   int hout, int wout, float sh, float sw) {
    const long long total = (long long)planes * hout * wout;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
        const int ox = (int)(idx % wout);
        const long long rest = idx / wout;
        const int oy = (int)(rest % hout);
        const long long p = rest / hout;
        const long long src = (p * hin + nearest_exact_src(oy, sh, hin)) * (long long)win + nearest_exact_src(ox, sw, win);
        if (BWD) atomicAdd(out + src, in[idx]);
        else out[idx] = in[src];
    }
}

// F.interpolate(mode="area") == adaptive_avg_pool2d (aten AdaptiveAveragePooling): output o averages the input window
// [floor(o * in / out), ceil((o + 1) * in / out)) of each axis (integer arithmetic), rows outer, columns inner, one division by the
// window's element count.  BWD: every element of the window receives grad_out / count (atomic adds into the zeroed grad_in, like aten).
template <bool BWD>
__global__ __launch_bounds__(256) void resize_area_kernel(const float* __restrict__ in, float* __restrict__ out, int planes, int hin, int win, int hout,
                                                          int wout) {
    const long long total = (long long)planes * hout * wout;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
        const int ox = (int)(idx % wout);
        const long long rest = idx / wout;
        const int oy = (int)(rest % hout);
        const long long p = rest / hout;
        const int y0 = (int)(((long long)oy * hin) / hout), y1 = (int)((((long long)oy + 1) * hin + hout - 1) / hout);
        const int x0 = (int)(((long long)ox * win) / wout), x1 = (int)((((long long)ox + 1) * win + wout - 1) / wout);
        const float count = (float)((y1 - y0) * (x1 - x0));
        if (BWD) {
            const float g = in[idx] / count;
            for (int y = y0; y < y1; ++y)
                for (int x = x0; x < x1; ++x) atomicAdd(out + (p * hin + y) * (long long)win + x, g);
        } else {
            float sum = 0.f;
            for (int y = y0; y < y1; ++y) {
                const float* row = in + (p * hin + y) * (long long)win;
                for (int x = x0; x < x1; ++x) sum = __fadd_rn(sum, row[x]);
            }
            out[idx] = sum / count;
        }
    }
}

// ------------------------------------------------------------------------------------------------ bicubic (F.interpolate mode="bicubic")
// aten UpSampleBicubic2d: source coordinate WITHOUT the clamp at 0 of the linear modes, the 4 x 4 neighbours floor(src) - 1 .. + 2
// clamped into the image, cubic convolution weights with A = -0.75; rows are interpolated along x first, then the 4 results along y.
struct Cubic { int i[4]; float w[4]; };
__device__ __forceinline__ Cubic cubic_taps(int dst, float scale, int n_in, int align_corners) {
    const float src = align_corners ? scale * (float)dst : __fsub_rn(__fmul_rn(scale, (float)dst + 0.5f), 0.5f);
    const float fl = floorf(src);
    const float t = src - fl;
    const int base = (int)fl;
    constexpr float A = -0.75f;
    auto conv1 = [](float x) { return __fadd_rn(__fmul_rn(__fmul_rn(__fsub_rn(__fmul_rn(A + 2.0f, x), A + 3.0f), x), x), 1.0f); };
    auto conv2 = [](float x) { return __fsub_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fsub_rn(__fmul_rn(A, x), 5.0f * A), x), 8.0f * A), x), 4.0f * A); };
    Cubic c;
    c.w[0] = conv2(t + 1.0f); c.w[1] = conv1(t); c.w[2] = conv1(1.0f - t); c.w[3] = conv2(2.0f - t);
#pragma unroll
    for (int k = 0; k < 4; ++k) c.i[k] = min(max(base - 1 + k, 0), n_in - 1);
    return c;
}

template <bool BWD>
__global__ __launch_bounds__(256) void resize_bicubic_kernel(const float* __restrict__ in, float* __restrict__ out, int planes, int hin, int win,
                                                             int hout, int wout, float sh, float sw, int align_corners) {
    // forward: out[p, oy, ox] = sum_ij wy[i] wx[j] in[p, iy[i], ix[j]];  BWD: `in` = grad_out, `out` = grad_in (zeroed), atomic adds
    const long long total = (long long)planes * hout * wout;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
        const int ox = (int)(idx % wout);
        const long long rest = idx / wout;
        const int oy = (int)(rest % hout);
        const long long p = rest / hout;
        const Cubic cy = cubic_taps(oy, sh, hin, align_corners), cx = cubic_taps(ox, sw, win, align_corners);
        if (BWD) {
            float* dst = out + p * (long long)hin * win;
            const float g = in[idx];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) atomicAdd(dst + (long long)cy.i[i] * win + cx.i[j], g * cy.w[i] * cx.w[j]);
        } else {
            const float* src = in + p * (long long)hin * win;
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float* row = src + (long long)cy.i[i] * win;
                float r = __fmul_rn(row[cx.i[0]], cx.w[0]);
#pragma unroll
                for (int j = 1; j < 4; ++j) r = __fadd_rn(r, __fmul_rn(row[cx.i[j]], cx.w[j]));
                acc = i == 0 ? __fmul_rn(r, cy.w[0]) : __fadd_rn(acc, __fmul_rn(r, cy.w[i]));
            }
            out[idx] = acc;
        }
    }
}

// backward of ms_deaug_reduce: out = post(sum_s pre(v_s) / n), v_s = bilinear_s(x_s) (or x_s itself at the output size):
//   d out / d x_s[tap] = g * post'(out) / n * pre'(v_s) * tap weight      (linear reductions: g / n, or g for "sum")
struct MsGrads { float* g[MS_MAX]; };   // (kernel argument: the unrolled scale loop indexes it with compile-time constants)

__global__ __launch_bounds__(256) void ms_reduce_bwd_kernel(const MsArgs a, const float* __restrict__ fwd_out, const float* __restrict__ gout,
                                                            const MsGrads gin_) {
    float* const* gin = gin_.g;
    const long long total = (long long)a.planes * a.hout * a.wout;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const bool nonlin = a.op >= PTB_RED_GMEAN;
    const float inv_n = a.op == PTB_RED_SUM ? 1.0f : 1.0f / (float)a.n;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
        const int ox = (int)(idx % a.wout);
        const long long rest = idx / a.wout;
        const int oy = (int)(rest % a.hout);
        const long long p = rest / a.hout;
        float g = gout[idx] * inv_n;
        if (nonlin) g *= red_dpost(fwd_out[idx], a.op);
#pragma unroll
        for (int s = 0; s < MS_MAX; ++s) {
            if (s >= a.n) break;
            const int hin = a.h[s], win = a.w[s];
            const float* src = a.in[s] + p * (long long)hin * win;
            float* dst = gin[s] + p * (long long)hin * win;
            if (hin == a.hout && win == a.wout) {
                const long long o = (long long)oy * win + ox;
                dst[o] = nonlin ? g * red_dpre(src[o], a.op) : g;      // every source pixel is hit exactly once
            } else {
                const Taps ty = taps(oy, a.sh[s], hin, a.align_corners), tx = taps(ox, a.sw[s], win, a.align_corners);
                const long long o00 = (long long)ty.i0 * win + tx.i0, o01 = (long long)ty.i0 * win + tx.i1;
                const long long o10 = (long long)ty.i1 * win + tx.i0, o11 = (long long)ty.i1 * win + tx.i1;
                float c = g;
                if (nonlin) {
                    const float v = ty.l0 * (tx.l0 * src[o00] + tx.l1 * src[o01]) + ty.l1 * (tx.l0 * src[o10] + tx.l1 * src[o11]);
                    c *= red_dpre(v, a.op);
                }
                atomicAdd(dst + o00, c * ty.l0 * tx.l0);
                atomicAdd(dst + o01, c * ty.l0 * tx.l1);
                atomicAdd(dst + o10, c * ty.l1 * tx.l0);
                atomicAdd(dst + o11, c * ty.l1 * tx.l1);
            }
        }
    }
}

}  // namespace ptb

using namespace ptb;

extern "C" int ptb_resize_bilinear(const float* in, float* out, int64_t planes, int hin, int win, int hout, int wout,
                                   int align_corners, ptb_stream_t stream) {
    if (!in || !out || planes < 0 || hin < 1 || win < 1 || hout < 1 || wout < 1) return PTB_EINVAL;
    if (planes == 0) return PTB_OK;
    if (planes > 0x7fffffffLL) return PTB_EUNSUPPORTED;
    float sh, sw;
    if (align_corners) {
        sh = hout > 1 ? (float)(hin - 1) / (float)(hout - 1) : 0.f;
        sw = wout > 1 ? (float)(win - 1) / (float)(wout - 1) : 0.f;
    } else {
        sh = (float)hin / (float)hout;
        sw = (float)win / (float)wout;
    }
    const long long total = planes * hout * ((wout + 3) / 4);
    const long long want = (total + 255) / 256;
    const int blocks = (int)(want < 256 * 32 ? want : 256 * 32);
    hipLaunchKernelGGL(resize_bilinear_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, out, (int)planes, hin, win,
                       hout, wout, sh, sw, align_corners);
    return check_launch();
}

static int ms_reduce_impl(const float* const* inputs, const int* hs_full, const int* ws, const int* src_row0, const int* src_rows,
                          int n, float* out, int64_t planes, int hout_full, int wout, int out_row0, int out_rows, int align_corners,
                          int reduction, ptb_stream_t stream) {
    if (!inputs || !hs_full || !ws || !out || n < 1 || n > MS_MAX || planes < 0 || hout_full < 1 || wout < 1) return PTB_EINVAL;
    if (out_row0 < 0 || out_rows < 0 || out_row0 + out_rows > hout_full) return PTB_EINVAL;
    if (reduction < PTB_RED_SUM || reduction > PTB_RED_LOG1P) return PTB_EINVAL;
    if (planes == 0 || out_rows == 0) return PTB_OK;
    if (planes > 0x7fffffffLL) return PTB_EUNSUPPORTED;
    MsArgs a{};
    for (int s = 0; s < n; ++s) {
        const int r0 = src_row0 ? src_row0[s] : 0, nr = src_rows ? src_rows[s] : hs_full[s];
        if (!inputs[s] || hs_full[s] < 1 || ws[s] < 1 || r0 < 0 || nr < 1 || r0 + nr > hs_full[s]) return PTB_EINVAL;
        a.in[s] = inputs[s]; a.h[s] = nr; a.w[s] = ws[s]; a.hfull[s] = hs_full[s]; a.src0[s] = r0;
        if (hs_full[s] == hout_full && ws[s] == wout && (wout % 4 == 0) && !aligned16(inputs[s])) return PTB_EUNSUPPORTED;
        if (align_corners) {
            a.sh[s] = hout_full > 1 ? (float)(hs_full[s] - 1) / (float)(hout_full - 1) : 0.f;
            a.sw[s] = wout > 1 ? (float)(ws[s] - 1) / (float)(wout - 1) : 0.f;
        } else {
            a.sh[s] = (float)hs_full[s] / (float)hout_full;
            a.sw[s] = (float)ws[s] / (float)wout;
        }
    }
    const int hout = out_rows;
    a.n = n; a.planes = (int)planes; a.hout = hout; a.wout = wout; a.align_corners = align_corners; a.op = reduction;
    a.row0 = out_row0; a.hout_full = hout_full;
    const long long tiles = planes * ((hout + MS_TH - 1) / MS_TH) * ((wout + MS_TW - 1) / MS_TW);
    if (g_ms_tiled && tiles <= 0x7fffffffLL) {
        const dim3 grid((unsigned)tiles), block(256);
        hipStream_t st = (hipStream_t)stream;
#define PTB_MS(OPK) do { if (align_corners) hipLaunchKernelGGL((ms_reduce_tiled_kernel<OPK, 1>), grid, block, 0, st, a, out); \
                        else hipLaunchKernelGGL((ms_reduce_tiled_kernel<OPK, 0>), grid, block, 0, st, a, out); } while (0)
        if (reduction == PTB_RED_GMEAN) PTB_MS(2);
        else if (reduction >= PTB_RED_GMEAN) PTB_MS(1);
        else PTB_MS(0);
#undef PTB_MS
        return check_launch();
    }
    const long long total = planes * hout * ((wout + 3) / 4);
    const long long want = (total + 255) / 256;
    const int blocks = (int)(want < 256 * 32 ? want : 256 * 32);
    hipLaunchKernelGGL(ms_reduce_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, out);
    return check_launch();
}

extern "C" int ptb_ms_deaug_reduce(const float* const* inputs, const int* hs, const int* ws, int n, float* out, int64_t planes,
                                   int hout, int wout, int align_corners, int reduction, ptb_stream_t stream) {
    if (hout < 1) return PTB_EINVAL;
    return ms_reduce_impl(inputs, hs, ws, nullptr, nullptr, n, out, planes, hout, wout, 0, hout, align_corners, reduction, stream);
}

extern "C" int ptb_ms_deaug_reduce_strip(const float* const* inputs, const int* hs_full, const int* ws, const int* src_row0,
                                         const int* src_rows, int n, float* out, int64_t planes, int hout_full, int wout,
                                         int out_row0, int out_rows, int align_corners, int reduction, ptb_stream_t stream) {
    if (!src_row0 || !src_rows) return PTB_EINVAL;
    return ms_reduce_impl(inputs, hs_full, ws, src_row0, src_rows, n, out, planes, hout_full, wout, out_row0, out_rows, align_corners,
                          reduction, stream);
}

// ptb_set_tunable key 6: output tile height of the fused multiscale kernel.  Same box, cfg5 (fused mean / fused gmean / plain
// multiscale merge): 64 rows 417 / 475 / 234 us, 32 rows 412 / 432 / 224 us, 16 rows 405 / 434 / 239 us -- smaller tiles refetch more
// halo (1.08x / 1.13x / 1.21x) but put 4 / 7 / 8 workgroups on a CU, and this kernel waits more than it streams.
// ptb_set_tunable key 9: XCD-aware tile order, strip width in tile columns (0 = plain row-major over all XCDs).  Same box, cfg5 fused
// gmean / fused mean / plain merge: row-major 434 / 412 / 224 us; strips of 2: 467 / 446 / 240, 4: 456 / 431 / 232, 8: 468 / 446 / 239,
// 16: 428 / 399 / 217, 64 (= whole tile rows: every XCD walks its own contiguous eighth of the tiles row-major): 414 / 391 / 211.
// Keeping neighbours on one L2 pays; narrow strips cost more in DRAM page locality than the vertical halo reuse returns.
namespace ptb { int g_ms_tile_rows = 32; int g_ms_strip = 64; int g_ms_tile_w = 128; }   // key 15: 128 (default) = 128 x 16 output tiles when the windows fit (NV <= 2), 64 = 64 x g_ms_tile_rows

template <int NV, int INNER, int TH>
static void launch_fz_th(const FzArgs& a, float* out, unsigned blocks, hipStream_t st) {
    const dim3 grid(blocks), block(256);
#define PTB_FZ(OUTER) do { if (a.align_corners) hipLaunchKernelGGL((ms_flip_reduce_kernel<NV, INNER, OUTER, 1, TH>), grid, block, 0, st, a, out); \
                           else hipLaunchKernelGGL((ms_flip_reduce_kernel<NV, INNER, OUTER, 0, TH>), grid, block, 0, st, a, out); } while (0)
    if (a.op_outer == PTB_RED_GMEAN) PTB_FZ(2);
    else if (a.op_outer >= PTB_RED_GMEAN) PTB_FZ(1);
    else PTB_FZ(0);
#undef PTB_FZ
}

template <int NV, int INNER, int TH>
static void launch_fz_wide(const FzArgs& a, float* out, hipStream_t st) {   // 128 x TH tiles
    const long long tiles = (long long)a.planes * ((a.hout + TH - 1) / TH) * ((a.wout + 127) / 128);
    FzArgs b = a;
    b.strip = 64;
    b.total_tiles = tiles;
    const dim3 grid((unsigned)(8 * ((tiles + 7) / 8))), block(256);
#define PTB_FZW(OUTER) do { if (a.align_corners) hipLaunchKernelGGL((ms_flip_reduce_kernel<NV, INNER, OUTER, 1, TH, 128>), grid, block, 0, st, b, out); \
                            else hipLaunchKernelGGL((ms_flip_reduce_kernel<NV, INNER, OUTER, 0, TH, 128>), grid, block, 0, st, b, out); } while (0)
    if (a.op_outer == PTB_RED_GMEAN) PTB_FZW(2);
    else if (a.op_outer >= PTB_RED_GMEAN) PTB_FZW(1);
    else PTB_FZW(0);
#undef PTB_FZW
}

template <int NV, int INNER>
static void launch_fz(const FzArgs& a, float* out, int th, hipStream_t st) {
    const long long tiles = (long long)a.planes * ((a.hout + th - 1) / th) * ((a.wout + FZ_T - 1) / FZ_T);
    FzArgs b = a;
    b.strip = g_ms_strip;
    b.total_tiles = tiles;
    const long long blocks = b.strip > 0 ? 8 * ((tiles + 7) / 8) : tiles;
    if (th == 16) launch_fz_th<NV, INNER, 16>(b, out, (unsigned)blocks, st);
    else if (th == 32) launch_fz_th<NV, INNER, 32>(b, out, (unsigned)blocks, st);
    else launch_fz_th<NV, INNER, 64>(b, out, (unsigned)blocks, st);
}

static int ms_flip_impl(const float* const* inputs, const int* hs, const int* ws, const int* src_row0, const int* src_rows, int n, int V,
                        const int* views, int inner_reduction, float* out, int64_t planes, int hout_full, int wout, int out_row0, int out_rows,
                        int align_corners, int reduction, ptb_stream_t stream);

extern "C" int ptb_ms_flip_deaug_reduce(const float* const* inputs, const int* hs, const int* ws, int n, int V, const int* views,
                                        int inner_reduction, float* out, int64_t planes, int hout, int wout, int align_corners,
                                        int reduction, ptb_stream_t stream) {
    return ms_flip_impl(inputs, hs, ws, nullptr, nullptr, n, V, views, inner_reduction, out, planes, hout, wout, 0, hout, align_corners, reduction, stream);
}

// One rank's rows of the same pass: inputs[s] = rows src_row0[s] .. src_row0[s] + src_rows[s] - 1 of scale s's [V * planes, hs_full[s], ws[s]]
// views (every view plane holds the same row range); out = rows out_row0 .. out_row0 + out_rows - 1 of the [planes, hout_full, wout] result.
// The strips must contain every source row the taps of those output rows touch (checked: PTB_EBOUNDS); row-flipping views are refused.
extern "C" int ptb_ms_flip_deaug_reduce_strip(const float* const* inputs, const int* hs_full, const int* ws, const int* src_row0,
                                              const int* src_rows, int n, int V, const int* views, int inner_reduction, float* out,
                                              int64_t planes, int hout_full, int wout, int out_row0, int out_rows, int align_corners,
                                              int reduction, ptb_stream_t stream) {
    if (!src_row0 || !src_rows) return PTB_EINVAL;
    return ms_flip_impl(inputs, hs_full, ws, src_row0, src_rows, n, V, views, inner_reduction, out, planes, hout_full, wout, out_row0, out_rows,
                        align_corners, reduction, stream);
}

static Taps host_taps(int dst, float scale, int n_in, bool align_corners) {     // the device's taps(), evaluated on the host (same fp32 ops)
    float src;
    if (align_corners) src = scale * (float)dst;
    else { src = scale * ((float)dst + 0.5f) - 0.5f; src = src < 0.f ? 0.f : src; }
    Taps t;
    t.i0 = std::min((int)src, n_in - 1);
    t.i1 = t.i0 + (t.i0 < n_in - 1 ? 1 : 0);
    t.l1 = 0.f; t.l0 = 0.f;
    return t;
}

static int ms_flip_impl(const float* const* inputs, const int* hs, const int* ws, const int* src_row0, const int* src_rows, int n, int V,
                        const int* views, int inner_reduction, float* out, int64_t planes, int hout_full, int wout, int out_row0, int out_rows,
                        int align_corners, int reduction, ptb_stream_t stream) {
    const int hout = hout_full;     // (the scales and the "same size" test are those of the full-size call)
    if (!inputs || !hs || !ws || !out || !views || n < 1 || n > MS_MAX || V < 1 || V > FZ_VMAX || planes < 0 || hout < 1 || wout < 1) return PTB_EINVAL;
    if (out_row0 < 0 || out_rows < 0 || out_row0 + out_rows > hout_full) return PTB_EINVAL;
    if (out_rows == 0) return PTB_OK;
    if (reduction < PTB_RED_SUM || reduction > PTB_RED_LOG1P || inner_reduction < PTB_RED_SUM || inner_reduction > PTB_RED_LOG1P) return PTB_EINVAL;
    if (planes == 0) return PTB_OK;
    if (planes > 0x7fffffffLL) return PTB_EUNSUPPORTED;
    FzArgs a{};
    bool wide_ok = true;
    for (int k = 0; k < V; ++k) {
        if (views[k] < 0 || views[k] > 7) return PTB_EINVAL;
        if (views[k] & 1) return PTB_EUNSUPPORTED;           // transposing views: not combined with multiscale here
        a.codes |= (views[k] & 7) << (3 * k);
    }
    if (g_force_scalar || wout % 4 || !aligned16(out)) return PTB_EUNSUPPORTED;
    for (int s = 0; s < n; ++s) {
        if (!inputs[s] || hs[s] < 1 || ws[s] < 1) return PTB_EINVAL;
        if (ws[s] % 4 || !aligned16(inputs[s])) return PTB_EUNSUPPORTED;
        a.in[s] = inputs[s]; a.h[s] = hs[s]; a.w[s] = ws[s];
        if (align_corners) {
            a.sh[s] = hout > 1 ? (float)(hs[s] - 1) / (float)(hout - 1) : 0.f;
            a.sw[s] = wout > 1 ? (float)(ws[s] - 1) / (float)(wout - 1) : 0.f;
        } else {
            a.sh[s] = (float)hs[s] / (float)hout;
            a.sw[s] = (float)ws[s] / (float)wout;
        }
        if (hs[s] == hout && ws[s] == wout) continue;
        // the source window of a 64 x 64 tile must fit the LDS window (conservative bounds: + 2 taps, + 3 alignment, + 1 rounding)
        // (a window never exceeds the map itself, so small maps fit whatever the ratio)
        const int th_ = g_ms_tile_rows;
        const int need_r = std::min((int)ceilf(th_ * a.sh[s]) + 3, hs[s]), need_c = std::min((int)ceilf(FZ_T * a.sw[s]) + 6, ws[s] + 3);
        if (need_r > (th_ == 16 ? FZ_LR16 : (th_ == 32 ? FZ_LR32 : FZ_LR)) || need_c > FZ_LC) return PTB_EUNSUPPORTED;
        wide_ok = wide_ok && std::min((int)ceilf(16 * a.sh[s]) + 3, hs[s]) <= FZ_LR16 && std::min((int)ceilf(128 * a.sw[s]) + 6, ws[s] + 3) <= 168;
    }
    a.n = n; a.nviews = V; a.planes = (int)planes; a.hout = out_rows; a.wout = wout; a.align_corners = align_corners;
    a.hout_full = hout_full; a.oy_base = out_row0;
    for (int s = 0; s < n; ++s) {
        a.hs[s] = src_rows ? src_rows[s] : hs[s];
        a.row0[s] = src_row0 ? src_row0[s] : 0;
        if (!src_row0) continue;
        if (a.codes & 0x492) return PTB_EUNSUPPORTED;          // bit 1 of some view: rows flipped -- such views do not come in strips
        if (a.hs[s] < 1 || a.row0[s] < 0 || a.row0[s] + a.hs[s] > hs[s]) return PTB_EINVAL;
        const bool same = hs[s] == hout && ws[s] == wout;
        const int lo = same ? out_row0 : host_taps(out_row0, a.sh[s], hs[s], align_corners != 0).i0;
        const int hi = same ? out_row0 + out_rows - 1 : host_taps(out_row0 + out_rows - 1, a.sh[s], hs[s], align_corners != 0).i1;
        if (lo < a.row0[s] || hi >= a.row0[s] + a.hs[s]) return PTB_EBOUNDS;     // the strip lacks a row the taps read
    }
    a.op_outer = reduction; a.op_inner = inner_reduction;
    a.inner_div = inner_reduction == PTB_RED_SUM ? 1.0f : (float)V;
    a.inner_mul = inner_reduction == PTB_RED_SUM ? 1.0f : ((V & (V - 1)) == 0 ? 1.0f / (float)V : 0.f);
    bool resized = false;
    for (int s = 0; s < n; ++s) resized = resized || hs[s] != hout || ws[s] != wout;
    a.outer_mul = reduction == PTB_RED_SUM ? 1.0f : ((resized || reduction == PTB_RED_GMEAN || (n & (n - 1)) == 0) ? 1.0f / (float)n : 0.f);
    const int th = g_ms_tile_rows;
    if (planes * ((out_rows + th - 1) / th) * ((wout + FZ_T - 1) / FZ_T) > 0x7fffffffLL) return PTB_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int inner = inner_reduction == PTB_RED_GMEAN ? 2 : (inner_reduction > PTB_RED_GMEAN ? 1 : 0);
    if (g_ms_tile_w == 128 && wide_ok && V <= 2 && planes * ((out_rows + 15) / 16) * ((wout + 127) / 128) <= 0x7fffffffLL) {
        // (128 x 32 tiles measured slower: 363 / 520 us mean / gmean at cfg5 against 337 / 385 us for 128 x 16)
        if (V == 1) launch_fz_wide<1, 0, 16>(a, out, st);
        else if (inner == 2) launch_fz_wide<2, 2, 16>(a, out, st);
        else if (inner) launch_fz_wide<2, 1, 16>(a, out, st);
        else launch_fz_wide<2, 0, 16>(a, out, st);
        return check_launch();
    }
    if (V == 1) launch_fz<1, 0>(a, out, th, st);
    else if (V == 2) { if (inner == 2) launch_fz<2, 2>(a, out, th, st); else if (inner) launch_fz<2, 1>(a, out, th, st); else launch_fz<2, 0>(a, out, th, st); }
    else { if (inner == 2) launch_fz<4, 2>(a, out, th, st); else if (inner) launch_fz<4, 1>(a, out, th, st); else launch_fz<4, 0>(a, out, th, st); }
    return check_launch();
}


static int grid_1d(long long total) {
    const long long want = (total + 255) / 256;
    return (int)(want < 1 ? 1 : (want < 256 * 32 ? want : 256 * 32));
}

static void resize_scales(int hin, int win, int hout, int wout, int align_corners, float& sh, float& sw) {
    if (align_corners) {
        sh = hout > 1 ? (float)(hin - 1) / (float)(hout - 1) : 0.f;
        sw = wout > 1 ? (float)(win - 1) / (float)(wout - 1) : 0.f;
    } else {
        sh = (float)hin / (float)hout;
        sw = (float)win / (float)wout;
    }
}

extern "C" int ptb_resize_bilinear_bwd(const float* grad_out, float* grad_in, int64_t planes, int hin, int win, int hout, int wout,
                                       int align_corners, ptb_stream_t stream) {
    if (!grad_out || !grad_in || planes < 0 || hin < 1 || win < 1 || hout < 1 || wout < 1) return PTB_EINVAL;
    if (planes == 0) return PTB_OK;
    if (planes > 0x7fffffffLL) return PTB_EUNSUPPORTED;
    float sh, sw;
    resize_scales(hin, win, hout, wout, align_corners, sh, sw);
    hipLaunchKernelGGL(resize_bilinear_bwd_kernel, dim3(grid_1d(planes * hout * wout)), dim3(256), 0, (hipStream_t)stream, grad_out, grad_in,
                       (int)planes, hin, win, hout, wout, sh, sw, align_corners);
    return check_launch();
}

extern "C" int ptb_resize_nearest(const float* in, float* out, int64_t planes, int hin, int win, int hout, int wout, int backward,
                                  ptb_stream_t stream) {
    if (!in || !out || planes < 0 || hin < 1 || win < 1 || hout < 1 || wout < 1) return PTB_EINVAL;
    if (planes == 0) return PTB_OK;
    if (planes > 0x7fffffffLL) return PTB_EUNSUPPORTED;
    const float sh = (float)hin / (float)hout, sw = (float)win / (float)wout;
    const dim3 grid(grid_1d(planes * hout * wout)), block(256);
    if (backward) hipLaunchKernelGGL(resize_nearest_kernel<true>, grid, block, 0, (hipStream_t)stream, in, out, (int)planes, hin, win, hout, wout, sh, sw);
    else hipLaunchKernelGGL(resize_nearest_kernel<false>, grid, block, 0, (hipStream_t)stream, in, out, (int)planes, hin, win, hout, wout, sh, sw);
    return check_launch();
}

extern "C" int ptb_resize_nearest_exact(const float* in, float* out, int64_t planes, int hin, int win, int hout, int wout, int backward,
                                        ptb_stream_t stream) {
    if (!in || !out || planes < 0 || hin < 1 || win < 1 || hout < 1 || wout < 1) return PTB_EINVAL;
    if (planes == 0) return PTB_OK;
    if (planes > 0x7fffffffLL) return PTB_EUNSUPPORTED;
    const float sh = (float)hin / (float)hout, sw = (float)win / (float)wout;
    const dim3 grid(grid_1d(planes * hout * wout)), block(256);
    if (backward) hipLaunchKernelGGL(resize_nearest_exact_kernel<true>, grid, block, 0, (hipStream_t)stream, in, out, (int)planes, hin, win, hout, wout, sh, sw);
    else hipLaunchKernelGGL(resize_nearest_exact_kernel<false>, grid, block, 0, (hipStream_t)stream, in, out, (int)planes, hin, win, hout, wout, sh, sw);
    return check_launch();
}

extern "C" int ptb_resize_area(const float* in, float* out, int64_t planes, int hin, int win, int hout, int wout, int backward, ptb_stream_t stream) {
    if (!in || !out || planes < 0 || hin < 1 || win < 1 || hout < 1 || wout < 1) return PTB_EINVAL;
    if (planes == 0) return PTB_OK;
    if (planes > 0x7fffffffLL) return PTB_EUNSUPPORTED;
    const dim3 grid(grid_1d(planes * hout * wout)), block(256);
    if (backward) hipLaunchKernelGGL(resize_area_kernel<true>, grid, block, 0, (hipStream_t)stream, in, out, (int)planes, hin, win, hout, wout);
    else hipLaunchKernelGGL(resize_area_kernel<false>, grid, block, 0, (hipStream_t)stream, in, out, (int)planes, hin, win, hout, wout);
    return check_launch();
}

extern "C" int ptb_resize_bicubic(const float* in, float* out, int64_t planes, int hin, int win, int hout, int wout, int align_corners,
                                  int backward, ptb_stream_t stream) {
    if (!in || !out || planes < 0 || hin < 1 || win < 1 || hout < 1 || wout < 1) return PTB_EINVAL;
    if (planes == 0) return PTB_OK;
    if (planes > 0x7fffffffLL) return PTB_EUNSUPPORTED;
    float sh, sw;
    resize_scales(hin, win, hout, wout, align_corners, sh, sw);
    const dim3 grid(grid_1d(planes * hout * wout)), block(256);
    if (backward) hipLaunchKernelGGL(resize_bicubic_kernel<true>, grid, block, 0, (hipStream_t)stream, in, out, (int)planes, hin, win, hout, wout, sh, sw, align_corners);
    else hipLaunchKernelGGL(resize_bicubic_kernel<false>, grid, block, 0, (hipStream_t)stream, in, out, (int)planes, hin, win, hout, wout, sh, sw, align_corners);
    return check_launch();
}

extern "C" int ptb_ms_deaug_reduce_bwd(const float* const* inputs, const int* hs, const int* ws, int n, const float* fwd_out,
                                       const float* grad_out, float* const* grad_inputs, int64_t planes, int hout, int wout,
                                       int align_corners, int reduction, ptb_stream_t stream) {
    if (!inputs || !hs || !ws || !grad_out || !grad_inputs || n < 1 || n > MS_MAX || planes < 0 || hout < 1 || wout < 1) return PTB_EINVAL;
    if (reduction < PTB_RED_SUM || reduction > PTB_RED_LOG1P) return PTB_EINVAL;
    if (reduction >= PTB_RED_GMEAN && !fwd_out) return PTB_EINVAL;
    if (planes == 0) return PTB_OK;
    if (planes > 0x7fffffffLL) return PTB_EUNSUPPORTED;
    MsArgs a{};
    MsGrads g{};
    for (int s = 0; s < n; ++s) {
        if (!inputs[s] || !grad_inputs[s] || hs[s] < 1 || ws[s] < 1) return PTB_EINVAL;
        a.in[s] = inputs[s]; a.h[s] = hs[s]; a.w[s] = ws[s]; a.hfull[s] = hs[s];
        resize_scales(hs[s], ws[s], hout, wout, align_corners, a.sh[s], a.sw[s]);
        g.g[s] = grad_inputs[s];
    }
    a.n = n; a.planes = (int)planes; a.hout = hout; a.wout = wout; a.align_corners = align_corners; a.op = reduction; a.hout_full = hout;
    hipLaunchKernelGGL(ms_reduce_bwd_kernel, dim3(grid_1d(planes * hout * wout)), dim3(256), 0, (hipStream_t)stream, a, fwd_out, grad_out, g);
    return check_launch();
}
