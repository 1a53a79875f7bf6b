// Device-side pieces shared by the view kernels (ptb_views.hip) and the loop-edge kernels (ptb_edges.hip):
// launch descriptors, TTA reductions, the XOR-swizzled LDS tile and the "one chunk -> all V views" scatter.
#pragma once
#include <initializer_list>

#include "ptb_common.h"

namespace ptb {

constexpr int MAX_VIEWS = 8;
constexpr int MAX_T = 4;       // LDS tiles: at most 4 of 8 D4 views transpose
constexpr int MAX_CELLS = 64;  // per launch (kernarg budget)
constexpr int MAX_COVER = 4;   // tiles covering one cell
constexpr int MAX_GROUP = 64;  // tiles per launch in accumulate mode
constexpr int CW = 64;         // chunk columns (256 B row segments)

// number of LDS transpose tiles a kernel instantiation needs (CODES < 0: view codes known only at run time)
constexpr int lds_tiles(int nv, int codes) {
    if (codes < 0) return nv < MAX_T ? nv : MAX_T;
    int n = 0;
    for (int k = 0; k < nv; ++k) n += (codes >> (3 * k)) & 1;
    return n;
}

struct Cell {
    int ox, oy, w, h;     // rectangle in accumulator coordinates
    int chunk_end;        // exclusive prefix of chunk counts (cells sorted heavy-first)
    int ntiles;
    int fresh;            // 1: no element of this cell was ever written -> store instead of read-modify-write
    int final_;           // 1: every planned tile of this cell is in after this launch -> write (sum / norm) to `merged`
    int tile[MAX_COVER];  // group-local tile indices, ascending batch order
};

struct CellArgs {
    Cell cells[MAX_CELLS];
    int tile_x[MAX_GROUP];
    int tile_y[MAX_GROUP];
    int tile_id[MAX_GROUP];  // index into the batch (source tile)
};

struct ViewArgs {
    const float* src;
    float* dst;           // plain output, or the accumulator image
    float* norm;          // accumulate mode
    const float* weight;  // accumulate mode, [H, W] of the tile
    float* merged;            // planned accumulate: the merge result [C, H, W] (same strides as dst) ...
    const float* norm_full;   // ... and the complete normaliser [H, W] it is divided by
    int H, W, C;          // output-tile rows / cols, channels
    long long src_view_stride;  // elements between consecutive views of one tile (B*C*H*W)
    long long src_tile_stride;  // C*H*W
    long long dst_tile_stride;
    long long dst_chan_stride;
    int dst_row_stride;
    int nviews;
    int codes;           // 3 bits per view
    int tiles_per_view;  // per-view mode: out tile t uses view t / tiles_per_view ...
    int src_tile_mod;    // ... and source tile t % src_tile_mod
    int op;              // PTB_RED_*
    float divisor;       // linear ops: out = sum / divisor (1 for sum); non-linear: mean divisor
    float scale;         // per-view mode multiplier
    int chunks_x, chunks_y;  // plain modes: chunks per tile
    int ncells, total_chunks;
    int in_dtype;        // element type of src: PTB_F32 | PTB_F16 | PTB_BF16 (reduce / accumulate kernels)
    int keep_acc;        // planned accumulate: a finalised cell ALSO stores its weighted sum in the accumulator (PTB_PLANNED_KEEP_SUMS)
    int round_src;       // PTB_ROUND_SRC: the reduced value is rounded to the (half / bf16) source type before it is blended
    int chan_loop;       // band plan kernel, identity view: one workgroup per work item walks all channels (ptb_set_tunable key 27)
    int lds_db;          // band plan kernel, prefetching instances: alternate between two sets of LDS tiles (ptb_set_tunable key 25)
    int rot_views;       // band plan kernel (A/B, ptb_set_tunable key 22): odd work items issue their view loads starting at view NV / 2
};

enum { MODE_REDUCE = 0, MODE_PERVIEW = 1, MODE_ACCUM = 2 };

// ------------------------------------------------------------------------------------------------ reductions
constexpr float kEps = 1e-6f;
constexpr float kOneMinusEps = (float)(1.0 - 1e-6);

// v_log_f32 / v_exp_f32 / v_rcp_f32 (1 ulp each) instead of libm: on probabilities and their logs the results stay within
// ~1e-7 (relative for the reciprocal forms) of the libm values, far inside the 1e-5 parity tolerance, while the libm
// versions made the non-linear reductions ALU-bound (log1p: 213 us instead of 52 us per 8-tile d4 batch).
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ float fast_log(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

template <int OPK>
__device__ __forceinline__ float red_pre(float x, int op) {
    if (OPK == 0) return x;
    switch (op) {
        case PTB_RED_GMEAN: return fast_log(x);                                  // functional.py:261
        case PTB_RED_HMEAN: return fast_rcp(x < kEps ? kEps : x);                // functional.py:275
        case PTB_RED_HARMONIC1P: return fast_rcp(x + 1.0f);                      // functional.py:292
        case PTB_RED_LOGODD: {                                                   // functional.py:311-312
            float p = x < kEps ? kEps : (x > kOneMinusEps ? kOneMinusEps : x);
            return fast_log(p * fast_rcp(1.0f - p));
        }
        case PTB_RED_LOG1P: return fast_log(1.0f + x);                           // functional.py:330
        default: return x;
    }
}

// s / divisor for a wave-uniform divisor (the number of views).  An IEEE division is ~12 vector instructions; when the divisor is
// a power of two (d4: 8 views, d2: 4, flips of one axis: 2) multiplying by its exact reciprocal 2^-k gives the same bits for
// every s (both are exact scalings, rounded once), so the mean of the views stays bit-identical to the reference's `sum / V`.
__device__ __forceinline__ float div_views(float s, float divisor) {
    const unsigned bits = __float_as_uint(divisor);
    if ((bits & 0x807FFFFFu) == 0 && bits != 0) return s * __uint_as_float(0x7F000000u - bits);
    return s / divisor;
}

template <int OPK>
__device__ __forceinline__ float red_post(float s, int op, float divisor) {
    if (OPK == 0) return divisor == 1.0f ? s : div_views(s, divisor);
    const float m = div_views(s, divisor);
    switch (op) {
        case PTB_RED_GMEAN: return fast_exp(m);
        case PTB_RED_HMEAN: return fast_rcp(m < kEps ? kEps : m);
        case PTB_RED_HARMONIC1P: return fast_rcp(m) - 1.0f;
        case PTB_RED_LOGODD: { const float e = fast_exp(m); return e * fast_rcp(1.0f + e); }
        case PTB_RED_LOG1P: return fast_exp(m) - 1.0f;
        default: return m;
    }
}

// d post / d m expressed through the forward output, and d pre / d x, for the backward of the non-linear reductions
__device__ __forceinline__ float red_dpost(float out, int op) {
    switch (op) {
        case PTB_RED_GMEAN: return out;                                         // out = exp(m)
        case PTB_RED_HMEAN: return out >= 1.0f / kEps ? 0.f : -out * out;        // out = 1 / max(m, eps)
        case PTB_RED_HARMONIC1P: return -(out + 1.0f) * (out + 1.0f);           // out = 1/m - 1
        case PTB_RED_LOGODD: return out * (1.0f - out);                         // out = sigmoid(m)
        case PTB_RED_LOG1P: return out + 1.0f;                                  // out = exp(m) - 1
        default: return 1.0f;
    }
}
__device__ __forceinline__ float red_dpre(float x, int op) {
    switch (op) {
        case PTB_RED_GMEAN: return 1.0f / x;
        case PTB_RED_HMEAN: return x < kEps ? 0.f : -1.0f / (x * x);
        case PTB_RED_HARMONIC1P: return -1.0f / ((x + 1.0f) * (x + 1.0f));
        case PTB_RED_LOGODD: return (x < kEps || x > kOneMinusEps) ? 0.f : 1.0f / (x * (1.0f - x));
        case PTB_RED_LOG1P: return 1.0f / (1.0f + x);
        default: return 1.0f;
    }
}

// XOR-swizzled [CH][64] fp32 LDS tile: 16-byte slots of a row are permuted by (row/4) so that the transposing
// scatter (lanes walk rows) spreads over banks and the b128 gather (16 lanes per row) stays conflict-free.
__device__ __forceinline__ int swz(int i, int j) { return i * CW + ((((j >> 2) ^ (i >> 2)) & 15) << 2) + (j & 3); }

// The same tile as the transposing views of gather_reduce write it.  A ds_write_b32 is serviced in two 32-lane groups over 32 banks:
// with 64-row blocks (16 lanes per source row, rows 4 apart in i for consecutive lanes) lanes qq and qq + 8 of a group fall on one
// bank under swz -- a 2-way conflict on every scalar store of the four transposing d4 views (round 4 PMC: SQ_LDS_BANK_CONFLICT /
// SQ_LDS_IDX_ACTIVE = 40 % in band_plan_kernel).  Rows 32..63 therefore keep the two halves of their 16-byte slots swapped
// (position-in-slot bit 1 ^= row bit 5): the 32 lanes of a group then cover 32 banks, the ds_read_b128 still fetches whole slots
// (its bank pattern is unchanged) and un-swaps in registers (unswz4).  32-row blocks (8 lanes per source row) never had the conflict.
// (Used by the 64-row instances whose view codes are compile-time constants -- every TTA group; the run-time-codes instance sits at its
// 128-register ceiling and keeps the plain swizzle, its code untouched.)
__device__ __forceinline__ int swzg(int i, int j) {
    return i * CW + ((((j >> 2) ^ (i >> 2)) & 15) << 2) + ((j & 3) ^ ((i >> 4) & 2));
}
__device__ __forceinline__ float4 unswz4(const float4 t, int i) {
    return (i & 32) ? make_float4(t.z, t.w, t.x, t.y) : t;
}

typedef float v4f __attribute__((ext_vector_type(4)));

// 16-byte global load; NT = non-temporal (streamed once: do not keep the line in L2 / Infinity Cache, which is left to
// the accumulator read-modify-writes).  On MI355X nt streaming reads measured +8..15 % HBM bandwidth (tools/bw_probe).
template <bool NT>
__device__ __forceinline__ float4 ld16(const float* p) {
    const v4f* q = reinterpret_cast<const v4f*>(p);
    const v4f v = NT ? __builtin_nontemporal_load(q) : *q;
    return make_float4(v.x, v.y, v.z, v.w);
}

// 4 consecutive source elements at element offset `off` of `base`, as fp32.  LD: 0 = fp32, 1 = fp32 non-temporal,
// 2 = fp16, 3 = bf16 (both non-temporal, 8 B per lane; the conversions are exact, so a half-precision model output gives
// bit for bit what its .float() copy would -- without that copy ever being written to HBM).
template <int LD>
__device__ __forceinline__ float4 ld4(const float* base, long long off) {
    if constexpr (LD <= 1) {
        return ld16<LD == 1>(base + off);
    } else if constexpr (LD == 2) {
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        const h4 v = __builtin_nontemporal_load(reinterpret_cast<const h4*>(reinterpret_cast<const _Float16*>(base) + off));
        return make_float4((float)v.x, (float)v.y, (float)v.z, (float)v.w);
    } else {
        typedef unsigned short u4 __attribute__((ext_vector_type(4)));
        const u4 v = __builtin_nontemporal_load(reinterpret_cast<const u4*>(reinterpret_cast<const unsigned short*>(base) + off));
        return make_float4(__uint_as_float((unsigned)v.x << 16), __uint_as_float((unsigned)v.y << 16),
                           __uint_as_float((unsigned)v.z << 16), __uint_as_float((unsigned)v.w << 16));
    }
}

// Four source elements as they lie in memory -- 8 bytes (two registers) for half / bf16, the float4 itself for fp32: what a PREFETCHED
// tile is kept as until its turn (gather_load_raw / gather_widen below); widen4 is ld4's conversion.
typedef unsigned int raw2 __attribute__((ext_vector_type(2)));
template <int LD> struct RawOf { typedef raw2 type; };
template <> struct RawOf<0> { typedef v4f type; };
template <> struct RawOf<1> { typedef v4f type; };
template <int LD>
__device__ __forceinline__ typename RawOf<LD>::type ld4_raw(const float* base, long long off) {
    if constexpr (LD <= 1) {
        const v4f* q = reinterpret_cast<const v4f*>(base + off);
        return LD == 1 ? __builtin_nontemporal_load(q) : *q;
    } else {
        return __builtin_nontemporal_load(reinterpret_cast<const raw2*>(reinterpret_cast<const unsigned short*>(base) + off));
    }
}
template <int LD>
__device__ __forceinline__ float4 widen4(const typename RawOf<LD>::type r) {
    if constexpr (LD <= 1) {
        return make_float4(r.x, r.y, r.z, r.w);
    } else if constexpr (LD == 2) {
        const unsigned lo = r.x, hi = r.y;     // (element-wise: a bit cast of the 32-bit halves to _Float16 pairs converted the wrong lanes)
        return make_float4((float)__builtin_bit_cast(_Float16, (unsigned short)(lo & 0xffffu)), (float)__builtin_bit_cast(_Float16, (unsigned short)(lo >> 16)),
                           (float)__builtin_bit_cast(_Float16, (unsigned short)(hi & 0xffffu)), (float)__builtin_bit_cast(_Float16, (unsigned short)(hi >> 16)));
    } else {
        return make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xFFFF0000u), __uint_as_float(r.y << 16), __uint_as_float(r.y & 0xFFFF0000u));
    }
}

// PTB_ROUND_SRC: what `tta.*_image_deaugment(half tensor)` hands to `integrate_batch` is a HALF tensor -- the reduced value rounded to
// the source type (round to nearest even, torch's `.to(dtype)`), which integrate_batch then widens exactly (tiles.py:334-335).  The
// fused launch of a lazy de-augmentation handle reproduces that rounding in registers: one conversion there and back per value
// (v_cvt_f16_f32 / gfx950's v_cvt_pk_bf16_f32), nothing for fp32 sources.
template <int LD>
__device__ __forceinline__ float round_src1(float x) {
    if constexpr (LD == 2) {
        return (float)(_Float16)x;
    } else if constexpr (LD == 3) {
        return (float)(__bf16)x;      // gfx950's v_cvt_pk_bf16_f32: round to nearest even like torch's conversion (a NaN stays a NaN)
    } else {
        return x;
    }
}
template <int LD>
__device__ __forceinline__ float4 round_src4(const float4 v, int on) {
    if constexpr (LD <= 1) return v;
    else return on ? make_float4(round_src1<LD>(v.x), round_src1<LD>(v.y), round_src1<LD>(v.z), round_src1<LD>(v.w)) : v;
}

__device__ __forceinline__ float comp(const float4& v, int m) { return m == 0 ? v.x : (m == 1 ? v.y : (m == 2 ? v.z : v.w)); }

// One source chunk (this thread's float4 `v` at chunk-local (r, 4q); chunk origin (x0, y0), extent cw x ch of tile b,
// channel c) written to all a.nviews views of the chunk-major output [V*B, C, H, W].  Row-preserving views are stored
// straight from the registers at mirrored addresses; the transposing views share one transposed, XOR-swizzled LDS copy
// of the chunk (ST[c][r] = S[r][c]) read back with ds_read_b128.  `st` = CW*CH floats of LDS.
template <int CH, bool NONLIN>
__device__ __forceinline__ void scatter_chunk(const ViewArgs& a, int B, int b, int c, int x0, int y0, int cw, int ch, float4 v,
                                              float* st, int tid) {
    constexpr int SL = CH / 4;  // 16-byte slots per LDS row
    const int q = tid & 15, r = tid >> 4;
    const bool act = r < ch && 4 * q < cw;
    const long long plane = (long long)a.H * a.W;
    bool any_t = false;
    for (int k = 0; k < a.nviews; ++k) {
        const int code = (a.codes >> (3 * k)) & 7;
        if (code & 1) { any_t = true; continue; }
        if (act) {   // out[i][j] = src[fr ? H-1-i : i][fc ? W-1-j : j]  <=>  src (R, Cc) lands at i = R or H-1-R, j = Cc or W-1-Cc
            const int i = (code & 2) ? a.H - 1 - (y0 + r) : y0 + r;
            const int j = (code & 4) ? a.W - 4 - (x0 + 4 * q) : x0 + 4 * q;
            const long long off = (((long long)k * B + b) * a.C + c) * plane + (long long)i * a.W + j;
            float4 w = (code & 4) ? make_float4(v.w, v.z, v.y, v.x) : v;
            if (NONLIN) {
                const float4 x4 = ld16<true>(a.weight + off);
                w.x *= red_dpre(x4.x, a.op); w.y *= red_dpre(x4.y, a.op); w.z *= red_dpre(x4.z, a.op); w.w *= red_dpre(x4.w, a.op);
            }
            out_store4(a.dst + off, w);
        }
    }
    if (!any_t) return;
    if (act) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int col = 4 * q + m;  // source column -> LDS row
            st[col * CH + ((((r >> 2) ^ (col >> 2)) & (SL - 1)) << 2) + (r & 3)] = comp(v, m);
        }
    }
    __syncthreads();
    const int ii = tid / SL, qq = tid % SL;  // LDS row (source column) and slot (4 source rows)
    if (ii < cw && 4 * qq < ch) {
        const float4 t = *reinterpret_cast<const float4*>(&st[ii * CH + (((qq ^ (ii >> 2)) & (SL - 1)) << 2)]);
        for (int k = 0; k < a.nviews; ++k) {
            const int code = (a.codes >> (3 * k)) & 7;
            if (!(code & 1)) continue;
            // out[i][j] = src[fr ? N-1-j : j][fc ? N-1-i : i]: source column Cc = x0+ii gives the out row, source rows give out cols
            const int i = (code & 4) ? a.W - 1 - (x0 + ii) : x0 + ii;
            const int j = (code & 2) ? a.H - 4 - (y0 + 4 * qq) : y0 + 4 * qq;
            const long long off = (((long long)k * B + b) * a.C + c) * plane + (long long)i * a.W + j;
            float4 w = (code & 2) ? make_float4(t.w, t.z, t.y, t.x) : t;
            if (NONLIN) {
                const float4 x4 = ld16<true>(a.weight + off);
                w.x *= red_dpre(x4.x, a.op); w.y *= red_dpre(x4.y, a.op); w.z *= red_dpre(x4.z, a.op); w.w *= red_dpre(x4.w, a.op);
            }
            out_store4(a.dst + off, w);
        }
    }
}

// ------------------------------------------------------------------------------------------------ view codes of the TTA groups
constexpr int pack_codes(std::initializer_list<int> l) {
    int v = 0, k = 0;
    for (int c : l) v |= c << (3 * k++);
    return v;
}
constexpr int CODES_ID = 0;
constexpr int CODES_FLIPLR = pack_codes({0, 4});
constexpr int CODES_FLIPUD = pack_codes({0, 2});
constexpr int CODES_FLIPS = pack_codes({0, 4, 2});
constexpr int CODES_D2 = pack_codes({0, 4, 2, 6});
constexpr int CODES_D4 = pack_codes({0, 5, 6, 3, 1, 4, 7, 2});  // inverse views of d4_image_deaugment, tta.py:455-466


// The second half of gather_reduce: this thread's NV loaded float4 (row-preserving views already in output orientation, transposing
// views as read along the source rows) -> transposing views through the LDS tile -> reduction over the views.
template <int CH, int NV, int CODES, int OPK>
__device__ __forceinline__ float4 gather_tail(float4 (&v)[NV], int nv_rt, int codes_rt, int cw, int ch, int op, float divisor, float* lds, int tid,
                                              bool more_entries) {
    constexpr int QPR = CH / 4;  // float4 per source row of a transposed block
    const int q = tid & 15, r = tid >> 4;
    const int rr = tid / QPR, qq = tid % QPR;
    const bool act = (r < ch) && (4 * q < cw);
    const bool tact = (rr < cw) && (4 * qq < ch);
    const int nv = CODES >= 0 ? NV : nv_rt;
    const int codes = CODES >= 0 ? CODES : codes_rt;
    int tb = 0;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        if (k < nv && ((codes >> (3 * k)) & 1)) {
            const int code = (codes >> (3 * k)) & 7;
            float* buf = lds + tb * (CW * CH);
            ++tb;
            if (tact) {
                const int jl = (code & 2) ? cw - 1 - rr : rr;
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const int cc = 4 * qq + m;
                    const int il = (code & 4) ? ch - 1 - cc : cc;
                    if constexpr (CH == 64 && CODES >= 0) buf[swzg(il, jl)] = comp(v[k], m);
                    else buf[swz(il, jl)] = comp(v[k], m);
                }
            }
        }
    }
    if (tb) {
        __syncthreads();
        tb = 0;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            if (k < nv && ((codes >> (3 * k)) & 1)) {
                const float* buf = lds + tb * (CW * CH);
                ++tb;
                v[k] = act ? *reinterpret_cast<const float4*>(buf + swz(r, 4 * q)) : make_float4(1.f, 1.f, 1.f, 1.f);
                if constexpr (CH == 64 && CODES >= 0) v[k] = unswz4(v[k], r);
            }
        }
        if (more_entries) __syncthreads();  // LDS tiles are reused by the next covering tile
    }

    float4 s = make_float4(red_pre<OPK>(v[0].x, op), red_pre<OPK>(v[0].y, op), red_pre<OPK>(v[0].z, op), red_pre<OPK>(v[0].w, op));
#pragma unroll
    for (int k = 1; k < NV; ++k) {
        if (k < nv) {
            s.x = __fadd_rn(s.x, red_pre<OPK>(v[k].x, op));
            s.y = __fadd_rn(s.y, red_pre<OPK>(v[k].y, op));
            s.z = __fadd_rn(s.z, red_pre<OPK>(v[k].z, op));
            s.w = __fadd_rn(s.w, red_pre<OPK>(v[k].w, op));
        }
    }
    return make_float4(red_post<OPK>(s.x, op, divisor), red_post<OPK>(s.y, op, divisor), red_post<OPK>(s.z, op, divisor),
                       red_post<OPK>(s.w, op, divisor));
}


// Reduced value of one (tile, channel) for this thread's float4 at chunk-local (r, 4q); the chunk starts at
// tile-local (ly, lx) and spans ch x cw.  `plane` = view 0 of this tile & channel.
template <int CH, int NV, int CODES, int OPK, int LD>
__device__ __forceinline__ float4 gather_reduce(const float* __restrict__ src, long long plane, long long view_stride, int nv_rt,
                                                int codes_rt, int H, int W, int lx, int ly, int cw, int ch, int op,
                                                float divisor, float* lds, int tid, bool more_entries) {
    constexpr int QPR = CH / 4;  // float4 per source row of a transposed block
    const int q = tid & 15, r = tid >> 4;
    const int rr = tid / QPR, qq = tid % QPR;
    const bool act = (r < ch) && (4 * q < cw);
    const bool tact = (rr < cw) && (4 * qq < ch);
    const int nv = CODES >= 0 ? NV : nv_rt;
    const int codes = CODES >= 0 ? CODES : codes_rt;

    float4 v[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        v[k] = make_float4(1.f, 1.f, 1.f, 1.f);
        if (k < nv) {
            const int code = (codes >> (3 * k)) & 7;
            const long long p = plane + (long long)k * view_stride;   // element offset of view k
            if (!(code & 1)) {
                if (act) {
                    const int i = ly + r, j = lx + 4 * q;
                    const int row = (code & 2) ? H - 1 - i : i;
                    const int col = (code & 4) ? W - 4 - j : j;
                    const float4 t = ld4<LD>(src, p + (long long)row * W + col);
                    v[k] = (code & 4) ? make_float4(t.w, t.z, t.y, t.x) : t;
                }
            } else if (tact) {
                const int R0 = (code & 2) ? H - lx - cw : lx;  // H == W for transposing views
                const int C0 = (code & 4) ? W - ly - ch : ly;
                v[k] = ld4<LD>(src, p + (long long)(R0 + rr) * W + C0 + 4 * qq);
            }
        }
    }

    return gather_tail<CH, NV, CODES, OPK>(v, nv_rt, codes_rt, cw, ch, op, divisor, lds, tid, more_entries);
}

// gather_reduce in two steps: the loads of one covering tile as raw values (8 bytes per view for half / bf16 -- 16 registers for the
// eight d4 views -- 16 bytes for fp32), held while the PREVIOUS tile is still being transposed, reduced and blended, and their
// widening into gather_tail's input.  band_plan_kernel<.., PF> requests tile e + 1 before it finishes tile e.
// ROT: the loads are ISSUED starting at view ROT (then ROT + 1, ..., wrapping) -- which register receives which view, and so every bit of
// the result, is unchanged; an A/B on whether workgroups that all walk the views in one order collide in DRAM (ptb_set_tunable key 22).
template <int CH, int NV, int CODES, int LD, int ROT = 0>
__device__ __forceinline__ void gather_load_raw(const float* __restrict__ src, long long plane, long long view_stride, int nv_rt, int codes_rt, int H,
                                                int W, int lx, int ly, int cw, int ch, int tid, typename RawOf<LD>::type (&raw)[NV]) {
    constexpr int QPR = CH / 4;
    const int q = tid & 15, r = tid >> 4;
    const int rr = tid / QPR, qq = tid % QPR;
    const bool act = (r < ch) && (4 * q < cw);
    const bool tact = (rr < cw) && (4 * qq < ch);
    const int nv = CODES >= 0 ? NV : nv_rt;
    const int codes = CODES >= 0 ? CODES : codes_rt;
#pragma unroll
    for (int kk = 0; kk < NV; ++kk) {
        const int k = (kk + ROT) % NV;
        raw[k] = typename RawOf<LD>::type{};
        if (k < nv) {
            const int code = (codes >> (3 * k)) & 7;
            const long long p = plane + (long long)k * view_stride;
            if (!(code & 1)) {
                if (act) {
                    const int i = ly + r, j = lx + 4 * q;
                    const int row = (code & 2) ? H - 1 - i : i;
                    const int col = (code & 4) ? W - 4 - j : j;
                    raw[k] = ld4_raw<LD>(src, p + (long long)row * W + col);
                }
            } else if (tact) {
                const int R0 = (code & 2) ? H - lx - cw : lx;
                const int C0 = (code & 4) ? W - ly - ch : ly;
                raw[k] = ld4_raw<LD>(src, p + (long long)(R0 + rr) * W + C0 + 4 * qq);
            }
        }
    }
}
template <int CH, int NV, int CODES, int LD>
__device__ __forceinline__ void gather_widen(const typename RawOf<LD>::type (&raw)[NV], int nv_rt, int codes_rt, int cw, int ch, int tid, float4 (&v)[NV]) {
    constexpr int QPR = CH / 4;
    const int q = tid & 15, r = tid >> 4;
    const int rr = tid / QPR, qq = tid % QPR;
    const bool act = (r < ch) && (4 * q < cw);
    const bool tact = (rr < cw) && (4 * qq < ch);
    const int nv = CODES >= 0 ? NV : nv_rt;
    const int codes = CODES >= 0 ? CODES : codes_rt;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        v[k] = make_float4(1.f, 1.f, 1.f, 1.f);
        if (k < nv) {
            const int code = (codes >> (3 * k)) & 7;
            if (!(code & 1)) {
                if (act) {
                    const float4 t = widen4<LD>(raw[k]);
                    v[k] = (code & 4) ? make_float4(t.w, t.z, t.y, t.x) : t;
                }
            } else if (tact) {
                v[k] = widen4<LD>(raw[k]);
            }
        }
    }
}

}  // namespace ptb
