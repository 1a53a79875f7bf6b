// Device- and host-side pieces shared by the loss translation units (ptb_losses.hip, ptb_focal_softmax.hip): launch arguments,
// 16-byte streaming loads, wave / workgroup reductions into the slotted fp64 sums, the branch-free focal configuration and
// the wave-group walk over [B, C, HW] logits.  Not part of the C ABI.
#pragma once
#include <initializer_list>

#include "ptb_common.h"

// the losses are tolerance-checked (1e-5), not bit-exact: let the compiler fuse multiply-adds in the loss translation units
#pragma clang fp contract(fast)

namespace ptb {

enum {
    SEG_FOCAL = 1,             // accumulate sigmoid-focal sums
    SEG_STATS = 2,             // accumulate per-class region statistics
    SEG_HAS_IGNORE = 4,
    SEG_HAS_ALPHA = 8,
    SEG_REDUCED = 16,          // reduced focal loss (threshold)
    SEG_MASK_FOCAL_TERM = 32,  // normalised focal: ignored elements contribute 0 to sums[1]
    SEG_ELEMWISE = 64,         // also write the per-element focal loss
    SEG_NO_TERM = 128,         // the caller will not read sums[1] (focal loss without normalized=True): kernels may skip it
    SEG_NT_STORES = 256,       // (internal) non-temporal stores of the gradient
};
enum { PROB_SOFTMAX = 0, PROB_SIGMOID = 1, PROB_IDENTITY = 2 };

struct SegArgs {
    const float* logits;
    const long long* labels;     // [B, HW] or null
    const float* dense;          // [B, C, HW] or null
    const float* class_weights;  // [C] or null
    double* sums;                // [2 + 3*C]: focal loss, focal term, I[C], P[C], T[C]
    float* elem_out;             // [B, C, HW] when SEG_ELEMWISE
    int* error_flag;             // set to 1 on a label outside [0, C) that is not ignore_index
    int B, C;
    long long HW;
    int flags, prob;
    float gamma, alpha, threshold, ignore_value;
    long long ignore_label;
    // In-launch tail (ptb_region_loss_fwd): the workgroup that arrives LAST adds up the slots, evaluates the [C]-sized scalar
    // epilogue and its derivative, and leaves slots / counter / flag zeroed for the next call -- no memset, finalize or epilogue
    // launches around the streaming kernel.  tail_counter == null: the plain kernel (slots are added up by a later launch).
    unsigned int* tail_counter;  // [SUM_SLOTS + 1] arrival tickets; `sums`, these and `error_flag` live in a persistent all-zero workspace
    int* tail_error_out;         // [1] the call's label-error flag as the host reads it (written by the last workgroup), or null
    const unsigned char* tail_class_mask;
    float* tail_loss;            // [1]
    float* tail_coef;            // [2 + 2C]
    float tail_focal_scale, tail_dice_w, tail_jacc_w, tail_smooth, tail_eps;
    int tail_log_loss, tail_n_selected;
};

// ------------------------------------------------------------------------------------------------ small helpers
template <int PIX>
__device__ __forceinline__ void load_px(const float* __restrict__ p, float (&x)[PIX], bool ok) {
    if (!ok) return;
    if constexpr (PIX == 4) {
        // streamed once: non-temporal (leaves L2 / Infinity Cache to the labels and class weights); +2..3 % measured
        typedef float v4f __attribute__((ext_vector_type(4)));
        const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p));
        x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
    } else if constexpr (PIX == 2) {
        typedef float v2f __attribute__((ext_vector_type(2)));
        const v2f v = __builtin_nontemporal_load(reinterpret_cast<const v2f*>(p));
        x[0] = v.x; x[1] = v.y;
    } else {
#pragma unroll
        for (int k = 0; k < PIX; ++k) x[k] = p[k];
    }
}
template <int PIX>
__device__ __forceinline__ void store_px(float* __restrict__ p, const float (&x)[PIX], bool ok) {
    if (!ok) return;
    if constexpr (PIX == 4) {
        *reinterpret_cast<float4*>(p) = make_float4(x[0], x[1], x[2], x[3]);
    } else if constexpr (PIX == 2) {
        *reinterpret_cast<float2*>(p) = make_float2(x[0], x[1]);
    } else {
#pragma unroll
        for (int k = 0; k < PIX; ++k) p[k] = x[k];
    }
}

// streamed-once results (gradients): non-temporal stores, so that 0.5 GB of output does not displace what the kernel still reads
template <int PIX>
__device__ __forceinline__ void store_px_nt(float* __restrict__ p, const float (&x)[PIX], bool ok) {
    if (!ok) return;
    if constexpr (PIX == 4) {
        typedef float v4f __attribute__((ext_vector_type(4)));
        __builtin_nontemporal_store(v4f{x[0], x[1], x[2], x[3]}, reinterpret_cast<v4f*>(p));
    } else if constexpr (PIX == 2) {
        typedef float v2f __attribute__((ext_vector_type(2)));
        __builtin_nontemporal_store(v2f{x[0], x[1]}, reinterpret_cast<v2f*>(p));
    } else {
        store_px<PIX>(p, x, ok);
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }     // v_rcp_f32, 1 ulp
__device__ __forceinline__ float ex2(float x) { return __builtin_amdgcn_exp2f(x); }    // v_exp_f32 (2^x), no range fix-ups
__device__ __forceinline__ float lg2(float x) { return __builtin_amdgcn_logf(x); }     // v_log_f32 (log2 x)
constexpr float kLog2e = 1.4426950408889634f, kLn2 = 0.6931471805599453f;
__device__ __forceinline__ float fexp(float x) { return ex2(x * kLog2e); }
// exp(x - m) as 2^(x * log2e - M) with M = m * log2e rounded ONCE per pixel: one fma + v_exp per element.  Every class of a
// pixel is scaled by the same 2^(m * log2e - M) (1 +- 2e-6), which cancels in the softmax ratio and in u / (u + 2^-M).
__device__ __forceinline__ float fexp_sub(float x, float M) { return ex2(__builtin_fmaf(x, kLog2e, -M)); }

// x^g for x >= 0 without branches: 2^(g log2 x); x = 0 gives 0 for g > 0 (g == 0 is patched by the caller)
__device__ __forceinline__ float pow_pos(float base, float g) { return ex2(g * lg2(base)); }

// sigmoid(x) and log(1 + exp(-|x|)) from one exp (absolute error ~1e-7, far inside the 1e-5 loss tolerance)
struct Sig { float p, log1pe; };
__device__ __forceinline__ Sig sigmoid_parts(float x) {
    const float e = fexp(-fabsf(x));
    const float s1 = 1.0f + e;
    const float inv = rcp(s1);
    Sig s;
    s.p = x >= 0.f ? inv : e * inv;
    s.log1pe = lg2(s1) * kLn2;
    return s;
}

// Wave-uniform focal configuration, built once per kernel so the per-element math has no branches.
struct FocalCfg {
    float gamma, gm1;      // gamma, gamma - 1
    float a1, a0;          // alpha weight = a1 * t + a0 (a1 = 0, a0 = 1 when alpha is None)
    float thr, sc;         // reduced focal: f = 1 where pt < thr, base scaled by sc = 1 / (1 - thr); (-inf, 1) otherwise
    float f_when_g0;       // 1 if gamma == 0 (x^0 = 1 even at x = 0) else NaN marker unused
    float term_mask;       // multiplier of the focal term of ignored elements (0 when normalised, else 1)
    bool g0, g1;
};
__device__ __forceinline__ FocalCfg focal_cfg(const SegArgs& a) {
    FocalCfg c;
    c.gamma = a.gamma; c.gm1 = a.gamma - 1.0f;
    const bool ha = a.flags & SEG_HAS_ALPHA;
    c.a1 = ha ? 2.0f * a.alpha - 1.0f : 0.0f;
    c.a0 = ha ? 1.0f - a.alpha : 1.0f;
    const bool red = a.flags & SEG_REDUCED;
    c.thr = red ? a.threshold : -INFINITY;
    c.sc = red ? 1.0f / (1.0f - a.threshold) : 1.0f;
    c.f_when_g0 = 1.0f;
    c.term_mask = (a.flags & SEG_MASK_FOCAL_TERM) ? 0.0f : 1.0f;
    c.g0 = a.gamma == 0.0f; c.g1 = a.gamma == 1.0f;
    return c;
}

// BCE, focal term f, d f / d x and sigmoid of one element (functional.py:61-94).  G2 = gamma is exactly 2.
template <bool G2, bool GRAD>
__device__ __forceinline__ void focal_parts(float x, float t, const FocalCfg& c, float& ce, float& f, float& df, float& p) {
    const Sig s = sigmoid_parts(x);
    p = s.p;
    ce = fmaxf(x, 0.f) - x * t + s.log1pe;                 // BCE with logits
    const float pt = p * t + (1.f - p) * (1.f - t);
    const float base = fmaxf(1.f - pt, 0.f) * c.sc;
    const bool below = pt < c.thr;
    if (G2) f = base * base;
    else { f = pow_pos(base, c.gamma); f = c.g0 ? 1.0f : f; }
    f = below ? 1.0f : f;
    df = 0.f;
    if (GRAD) {
        const float dpt = p * (1.f - p) * (2.f * t - 1.f);
        float pw;                                           // base^(gamma-1)
        if (G2) pw = base;
        else { pw = pow_pos(base, c.gm1); pw = c.g1 ? 1.0f : pw; }
        df = -c.gamma * pw * c.sc * dpt;
        df = (below || (!G2 && c.g0)) ? 0.f : df;
    }
}

// Same-address device atomics serialise (~12 ns each): 8192 waves adding into ONE double costs more than the whole
// streaming pass.  So each workgroup reduces its 4 waves in LDS first and adds into one of PTB_SUM_SLOTS copies of the
// sums (slot = blockIdx % PTB_SUM_SLOTS); the caller adds the slots up (a [64, n] -> [n] sum).
constexpr int SUM_SLOTS = 64;

// fp64 add into a slot, in the RETURNING form: the value comes back from where the add was performed, so once a wave has waited
// for it (s_waitcnt vmcnt(0) in region_tail) the add is visible to an atomic read from any other workgroup / XCD.
__device__ __forceinline__ void slot_add(double* p, double v) {
    const double old = __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("" ::"v"(old));
}

__device__ __forceinline__ void block_add2(double v0, double v1, double* dst /* slot base */, int lane, int wave) {
    __shared__ double red[2][4];
    v0 = wave_sum(v0);
    v1 = wave_sum(v1);
    if (lane == 0) { red[0][wave] = v0; red[1][wave] = v1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        slot_add(&dst[0], red[0][0] + red[0][1] + red[0][2] + red[0][3]);
        slot_add(&dst[1], red[1][0] + red[1][1] + red[1][2] + red[1][3]);
    }
}

// ------------------------------------------------------------------------------------------------ in-launch scalar epilogue
__device__ __forceinline__ void tail_score_loss(float score, bool active, int log_loss, float eps, float& loss, float& dscore) {
    if (log_loss) {
        const float cl = fmaxf(score, eps);
        loss = -logf(cl);
        dscore = score >= eps ? -1.0f / cl : 0.f;
    } else {
        loss = 1.0f - score;
        dscore = -1.0f;
    }
    if (!active) { loss = 0.f; dscore = 0.f; }
}

// Called by every thread of every workgroup at the very end of a forward statistics kernel, after the workgroup's slot atomics.
// Everything that crosses workgroups here is an ATOMIC on device memory (the slot sums are fp64 atomic adds, the label flag an
// atomic or, the ticket an atomic add), i.e. performed at the memory side where all eight XCDs see it -- so the hand-off needs no
// cache write-back: each wave waits for its own atomics to be acknowledged (vmcnt), the workgroup synchronises, one lane draws a
// ticket.  The last arriver reads every slot value with an atomic EXCHANGE against zero: coherent wherever the adds came from,
// and the workspace is clean again for the next launch without a memset.  `scratch` = the kernel's dynamic LDS (>= 24 C + 80 bytes).
__device__ __forceinline__ void region_tail(const SegArgs& a, float* scratch) {
    const int C = a.C, row = 2 + 3 * C;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    scratch = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(scratch) + 7) & ~(uintptr_t)7);   // (doubles live here below)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int* flagw = reinterpret_cast<int*>(scratch);
    if (threadIdx.x == 0) {
        // Two-level arrival count: 2048 workgroups finishing together on ONE counter would serialise (~12 ns per same-address
        // atomic = 25 us); they draw tickets from SUM_SLOTS group counters instead, and only each group's last arriver -- by then
        // every add of its group has been performed -- goes on to the top-level counter.
        const unsigned group = blockIdx.x % SUM_SLOTS;
        const unsigned group_size = (gridDim.x - group + SUM_SLOTS - 1) / SUM_SLOTS;
        const unsigned n_groups = gridDim.x < (unsigned)SUM_SLOTS ? gridDim.x : (unsigned)SUM_SLOTS;
        int last_one = 0;
        if (__hip_atomic_fetch_add(a.tail_counter + group, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == group_size - 1) {
            __hip_atomic_store(a.tail_counter + group, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last_one = __hip_atomic_fetch_add(a.tail_counter + SUM_SLOTS, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == n_groups - 1;
        }
        flagw[0] = last_one;
    }
    __syncthreads();
    const bool last = flagw[0] != 0;
    __syncthreads();
    if (!last) return;
    double* tot = reinterpret_cast<double*>(scratch);          // [row] totals, then [8] reduction scratch
    unsigned long long* slots = reinterpret_cast<unsigned long long*>(a.sums);
    // The label flag and the top-level counter first (one lane), so that their round trip overlaps the slot exchanges below.
    int bad = 0;
    if (threadIdx.x == 0) {
        __hip_atomic_store(a.tail_counter + SUM_SLOTS, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bad = a.error_flag ? __hip_atomic_exchange(a.error_flag, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    }
    // wave w takes the values v = w, w + 4, ...; lane = slot.  Sixteen exchanges in flight per lane (one round trip up to C = 20),
    // then sixteen wave reductions.
    for (int v0 = wave; v0 < row; v0 += 64) {
        double x[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int v = v0 + 4 * u;
            x[u] = 0.0;
            if (v < row) x[u] = __longlong_as_double((long long)__hip_atomic_exchange(&slots[(size_t)lane * row + v], 0ull, __ATOMIC_RELAXED,
                                                                                      __HIP_MEMORY_SCOPE_AGENT));
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int v = v0 + 4 * u;
            if (v0 + 4 * u - wave >= row) break;      // (wave-uniform: no value of this wave is left)
            const double t = wave_sum(x[u]);
            if (lane == 0 && v < row) tot[v] = t;
        }
    }
    if (threadIdx.x == 0 && a.tail_error_out) a.tail_error_out[0] = bad;     // (thread 0 keeps `bad` for the loss below)
    __syncthreads();
    double dsum = 0.0, jsum = 0.0;
    const float inv_n = 1.0f / (float)a.tail_n_selected;
    for (int c = threadIdx.x; c < C; c += 256) {          // fp32 arithmetic in the reference's order (dice.py:112-131, jaccard.py:95-113)
        const float I = (float)tot[2 + c], P = (float)tot[2 + C + c], T = (float)tot[2 + 2 * C + c];
        const bool sel = !a.tail_class_mask || a.tail_class_mask[c];
        const bool active = T > 0.f;
        float gI = 0.f, gP = 0.f;
        if (a.tail_dice_w != 0.f) {
            const float num = 2.0f * I + a.tail_smooth, card = P + T + a.tail_smooth, den = fmaxf(card, a.tail_eps);
            float l, ds;
            tail_score_loss(num / den, active, a.tail_log_loss, a.tail_eps, l, ds);
            if (sel) {
                dsum += (double)l;
                gI += a.tail_dice_w * ds * (2.0f / den);
                gP += a.tail_dice_w * ds * (card >= a.tail_eps ? -num / (den * den) : 0.f);
            }
        }
        if (a.tail_jacc_w != 0.f) {
            const float num = I + a.tail_smooth, uni = P + T - I + a.tail_smooth, den = fmaxf(uni, a.tail_eps);
            float l, ds;
            tail_score_loss(num / den, active, a.tail_log_loss, a.tail_eps, l, ds);
            if (sel) {
                jsum += (double)l;
                const float dden = uni >= a.tail_eps ? num / (den * den) : 0.f;
                gI += a.tail_jacc_w * ds * (1.0f / den + dden);
                gP += a.tail_jacc_w * ds * (-dden);
            }
        }
        a.tail_coef[2 + c] = gI * inv_n;
        a.tail_coef[2 + C + c] = gP * inv_n;
    }
    dsum = wave_sum(dsum);
    jsum = wave_sum(jsum);
    double* red = tot + row;
    if (lane == 0) { red[wave] = dsum; red[4 + wave] = jsum; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double d = red[0] + red[1] + red[2] + red[3], j = red[4] + red[5] + red[6] + red[7];
        const float focal = a.tail_focal_scale != 0.f ? a.tail_focal_scale * (float)tot[0] : 0.f;
        const float dice = a.tail_dice_w != 0.f ? a.tail_dice_w * ((float)d * inv_n) : 0.f;
        const float jacc = a.tail_jacc_w != 0.f ? a.tail_jacc_w * ((float)j * inv_n) : 0.f;
        a.tail_loss[0] = bad ? __builtin_nanf("") : focal + dice + jacc;     // a label outside [0, C) poisons the loss
        a.tail_coef[0] = a.tail_focal_scale;
        a.tail_coef[1] = 0.f;
    }
}

// Label outside [0, C): raised with an atomic (the in-launch tail reads it from another workgroup, possibly another XCD).
__device__ __forceinline__ void raise_label_error(int* flag) {
    __hip_atomic_fetch_or(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One wave-group: image b, first pixel i0 of this lane, validity; labels of the lane's PIX pixels.
template <int PIX>
struct Group {
    int b;
    long long i0;
    bool ok;                // PIX == 4: all-or-nothing (HW % 4 == 0); PIX == 1: the single pixel
    long long lab[PIX];
    bool ign[PIX];
};

template <int PIX>
__device__ __forceinline__ Group<PIX> make_group(long long g, long long per_img, int lane, long long HW, const long long* __restrict__ labels,
                                                 bool has_ignore, long long ignore_label, int C, int* error_flag) {
    Group<PIX> G;
    G.b = (int)(g / per_img);
    G.i0 = (g - (long long)G.b * per_img) * 64 * PIX + (long long)lane * PIX;
    G.ok = G.i0 < HW;
#pragma unroll
    for (int k = 0; k < PIX; ++k) { G.lab[k] = -1; G.ign[k] = false; }
    if (labels && G.ok) {
        const long long* lp = labels + (long long)G.b * HW + G.i0;
        if constexpr (PIX == 4) {
            const longlong2 a = *reinterpret_cast<const longlong2*>(lp);
            const longlong2 b2 = *reinterpret_cast<const longlong2*>(lp + 2);
            G.lab[0] = a.x; G.lab[1] = a.y; G.lab[2] = b2.x; G.lab[3] = b2.y;
        } else if constexpr (PIX == 2) {
            const longlong2 a = *reinterpret_cast<const longlong2*>(lp);
            G.lab[0] = a.x; G.lab[1] = a.y;
        } else {
            G.lab[0] = lp[0];
        }
#pragma unroll
        for (int k = 0; k < PIX; ++k) {
            G.ign[k] = has_ignore && G.lab[k] == ignore_label;
            if (error_flag && !G.ign[k] && (G.lab[k] < 0 || G.lab[k] >= C)) raise_label_error(error_flag);
        }
    }
    return G;
}


// ------------------------------------------------------------------------------------------------ host side
// The slotted sums (SUM_SLOTS rows of `row` doubles) and the label flag start every forward from zero: zeroed here, on the launch
// stream, so that callers hand in plain uninitialised workspaces (no framework fill kernels on the loss path).
static inline int zero_sums(double* sums, int row, int* error_flag, hipStream_t s) {
    hipError_t e = hipMemsetAsync(sums, 0, (size_t)SUM_SLOTS * row * sizeof(double), s);
    if (e == hipSuccess && error_flag) e = hipMemsetAsync(error_flag, 0, sizeof(int), s);
    if (e != hipSuccess) { set_hip_error(e); return PTB_ELAUNCH; }
    return PTB_OK;
}

// Measured on MI355X (tools/ab_losses.py): the streaming kernels (focal fwd/bwd, softmax focal) gain ~10 % from an
// oversubscribed grid (32 workgroups per CU, one pixel group per wave), the statistics kernels lose from it (more
// per-workgroup LDS reductions + atomics), so they keep 8 per CU.
constexpr int kGridStream = 256 * 32, kGridStats = 256 * 8;
static inline int grid_for_groups(long long groups, int dflt) {
    const long long want = (groups + 3) / 4;
    const long long cap = g_loss_grid_cap > 0 ? g_loss_grid_cap : dflt;
    return (int)(want < 1 ? 1 : (want < cap ? want : cap));
}

static inline int fill_seg(SegArgs& a, const float* logits, const int64_t* labels, const float* dense, const float* class_weights,
                    int B, int C, int64_t HW, int flags, int prob, float gamma, float alpha, float threshold,
                    int64_t ignore_label, float ignore_value) {
    if (!logits || (!labels && !dense) || B < 0 || C < 1 || HW < 0) return PTB_EINVAL;
    if (prob < PROB_SOFTMAX || prob > PROB_IDENTITY) return PTB_EINVAL;
    a.logits = logits; a.labels = (const long long*)labels; a.dense = dense; a.class_weights = class_weights;
    a.B = B; a.C = C; a.HW = HW; a.flags = flags; a.prob = prob;
    a.gamma = gamma; a.alpha = alpha; a.threshold = threshold; a.ignore_label = ignore_label; a.ignore_value = ignore_value;
    return PTB_OK;
}

static inline bool vec_ok(int64_t HW, std::initializer_list<const void*> ptrs) {
    if (HW % 4 != 0 || g_force_scalar) return false;
    for (const void* p : ptrs) if (p && !aligned16(p)) return false;
    return true;
}


}  // namespace ptb
