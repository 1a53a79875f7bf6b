// ptb_losses.hip -- fused segmentation-loss reductions for gfx950 (MI355X).
//
// Reference: losses/functional.py:19-107 (focal_loss_with_logits), :110-173 (softmax_focal_loss_with_logits),
// :188-247 (soft_jaccard_score / soft_dice_score), losses/focal.py:77-105, losses/dice.py:59-131, losses/jaccard.py:48-103.
// The reference evaluates each loss as 12-15 full-tensor torch ops (and materialises an int64 one-hot, 1 GiB at
// [32,16,512,512]).  Here ONE pass over logits + labels produces every scalar the losses need:
//     sums[0] = sum of focal losses, sums[1] = sum of focal terms (normalised focal),
//     per class c: I_c = sum p*t, P_c = sum p, T_c = sum t   (p = softmax / sigmoid / given probabilities, masked)
// from which Dice, Jaccard and their log variants are [C]-sized scalar algebra.  Backward kernels recompute the
// activations from the logits (one more read, one gradient write) instead of storing any intermediate.
//
// Layout: logits [B, C, HW] fp32.  A wave owns 64*PIX consecutive pixels of one image (lane = PIX pixels, 16 B loads
// when PIX = 4) and walks the C planes, so every global access is a coalesced 256 B..1 KiB row segment.  Per-class
// sums are reduced across the 64 lanes with shuffles, accumulated per wave in LDS, and leave the workgroup as one fp64
// atomic per (statistic, class).  Memory/transcendental-bound elementwise work: no MFMA.
#include "ptb_common.h"

namespace ptb {

enum {
    SEG_FOCAL = 1,         // accumulate sigmoid-focal sums
    SEG_STATS = 2,         // accumulate per-class region statistics
    SEG_HAS_IGNORE = 4,
    SEG_HAS_ALPHA = 8,
    SEG_REDUCED = 16,      // reduced focal loss (threshold)
    SEG_MASK_FOCAL_TERM = 32,  // normalised focal: ignored elements contribute 0 to sums[1]
    SEG_ELEMWISE = 64,     // also write the per-element focal loss
};
enum { PROB_SOFTMAX = 0, PROB_SIGMOID = 1, PROB_IDENTITY = 2 };

struct SegArgs {
    const float* logits;
    const long long* labels;   // [B, HW] or null
    const float* dense;        // [B, C, HW] or null
    const float* class_weights;  // [C] or null
    double* sums;              // [2 + 3*C]: focal loss, focal term, I[C], P[C], T[C]
    float* elem_out;           // [B, C, HW] when SEG_ELEMWISE
    int* error_flag;           // set to 1 on a label outside [0, C) that is not ignore_index
    int B, C;
    long long HW;
    int flags, prob;
    float gamma, alpha, threshold, ignore_value;
    long long ignore_label;
};

// PIX consecutive floats of one lane: one 16-byte load when PIX == 4 (the host guarantees 16 B alignment then)
template <int PIX>
__device__ __forceinline__ void load_px(const float* __restrict__ p, float (&x)[PIX], bool ok) {
    if (!ok) return;
    if constexpr (PIX == 4) {
        const float4 v = *reinterpret_cast<const float4*>(p);
        x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
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
    } else {
#pragma unroll
        for (int k = 0; k < PIX; ++k) p[k] = x[k];
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__device__ __forceinline__ float powf_pos(float base, float g) {  // base >= 0
    if (g == 2.0f) return base * base;
    if (g == 1.0f) return base;
    if (g == 0.0f) return 1.0f;
    return base > 0.f ? __expf(g * __logf(base)) : 0.f;
}

// sigmoid focal term and loss of one element (functional.py:61-94).  hard = target is exactly 0 or 1.
__device__ __forceinline__ void focal_elem(float x, float t, const SegArgs& a, float cw, float& loss, float& term) {
    const float e = __expf(-fabsf(x));
    const float inv = 1.0f / (1.0f + e);
    const float p = x >= 0.f ? inv : e * inv;                    // sigmoid(x)
    const float ce = fmaxf(x, 0.f) - x * t + log1pf(e);          // BCE with logits
    const float pt = p * t + (1.f - p) * (1.f - t);
    float f;
    if (a.flags & SEG_REDUCED) {
        f = pt < a.threshold ? 1.0f : powf_pos((1.f - pt) / (1.f - a.threshold), a.gamma);
    } else {
        f = powf_pos(fmaxf(1.f - pt, 0.f), a.gamma);
    }
    float l = f * ce;
    if (a.flags & SEG_HAS_ALPHA) l *= a.alpha * t + (1.f - a.alpha) * (1.f - t);
    l *= cw;
    loss = l;
    term = f;
}

template <int PIX>
__global__ __launch_bounds__(256) void seg_loss_fwd_kernel(const SegArgs a) {
    extern __shared__ float lds[];  // [4 waves][3][C]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int C = a.C;
    float* wl = lds + wave * 3 * C;
    if (a.flags & SEG_STATS) for (int k = lane; k < 3 * C; k += 64) wl[k] = 0.f;
    double f_loss = 0.0, f_term = 0.0;
    const long long per_img = (a.HW + 64 * PIX - 1) / (64 * PIX);
    const long long groups = per_img * a.B;
    const bool ign = a.flags & SEG_HAS_IGNORE;
    for (long long g = (long long)blockIdx.x * 4 + wave; g < groups; g += (long long)gridDim.x * 4) {
        const int b = (int)(g / per_img);
        const long long i0 = (g % per_img) * 64 * PIX + (long long)lane * PIX;
        bool valid[PIX];
        long long lab[PIX];
        bool ignored[PIX];
#pragma unroll
        for (int k = 0; k < PIX; ++k) {
            valid[k] = i0 + k < a.HW;
            lab[k] = -1;
            ignored[k] = false;
        }
        if (a.labels) {
#pragma unroll
            for (int k = 0; k < PIX; ++k) {
                if (valid[k]) {
                    lab[k] = a.labels[(long long)b * a.HW + i0 + k];
                    ignored[k] = ign && lab[k] == a.ignore_label;
                    if (!ignored[k] && (lab[k] < 0 || lab[k] >= C)) *a.error_flag = 1;
                }
            }
        }
        // softmax statistics need the per-pixel log-sum-exp first
        float mx[PIX], den[PIX];
        if ((a.flags & SEG_STATS) && a.prob == PROB_SOFTMAX) {
#pragma unroll
            for (int k = 0; k < PIX; ++k) { mx[k] = -INFINITY; den[k] = 0.f; }
            for (int c = 0; c < C; ++c) {
                float xv[PIX];
                load_px<PIX>(a.logits + ((long long)b * C + c) * a.HW + i0, xv, valid[0]);
#pragma unroll
                for (int k = 0; k < PIX; ++k) {
                    if (valid[k]) {
                        const float x = xv[k];
                        const float m2 = fmaxf(mx[k], x);
                        den[k] = den[k] * __expf(mx[k] - m2) + __expf(x - m2);
                        mx[k] = m2;
                    }
                }
            }
        }
        for (int c = 0; c < C; ++c) {
            const long long off = ((long long)b * C + c) * a.HW + i0;
            float sI = 0.f, sP = 0.f, sT = 0.f;
            const float cw = a.class_weights ? a.class_weights[c] : 1.0f;
            float xv[PIX], tv[PIX], lv[PIX];
            load_px<PIX>(a.logits + off, xv, valid[0]);
            if (!a.labels) load_px<PIX>(a.dense + off, tv, valid[0]);
#pragma unroll
            for (int k = 0; k < PIX; ++k) {
                lv[k] = 0.f;
                if (!valid[k]) continue;
                const float x = xv[k];
                float t;
                bool ig = ignored[k];
                if (a.labels) {
                    t = lab[k] == c ? 1.f : 0.f;
                } else {
                    t = tv[k];
                    if (ign && t == a.ignore_value) ig = true;
                }
                if (a.flags & SEG_FOCAL) {
                    float l, f;
                    focal_elem(x, ig ? 0.f : t, a, cw, l, f);
                    if (ig) { l = 0.f; if (a.flags & SEG_MASK_FOCAL_TERM) f = 0.f; }
                    f_loss += (double)l;
                    f_term += (double)f;
                    lv[k] = l;
                }
                if (a.flags & SEG_STATS) {
                    float p;
                    if (a.prob == PROB_SOFTMAX) p = __expf(x - mx[k]) / den[k];
                    else if (a.prob == PROB_SIGMOID) { const float e = __expf(-fabsf(x)); p = (x >= 0.f ? 1.f : e) / (1.f + e); }
                    else p = x;
                    if (ig) { p = 0.f; t = 0.f; }   // p*mask, t*mask (dice.py:85-111)
                    sI += p * t; sP += p; sT += t;
                }
            }
            if (a.flags & SEG_ELEMWISE) store_px<PIX>(a.elem_out + off, lv, valid[0]);
            if (a.flags & SEG_STATS) {
                sI = wave_sum(sI); sP = wave_sum(sP); sT = wave_sum(sT);
                if (lane == 0) { wl[c] += sI; wl[C + c] += sP; wl[2 * C + c] += sT; }
            }
        }
    }
    // workgroup -> global: fp64 atomics, one per statistic per class per workgroup
    if (a.flags & SEG_FOCAL) {
        double l = f_loss, f = f_term;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { l += __shfl_xor(l, o); f += __shfl_xor(f, o); }
        if (lane == 0) { atomicAdd(&a.sums[0], l); atomicAdd(&a.sums[1], f); }
    }
    if (a.flags & SEG_STATS) {
        __syncthreads();
        for (int k = threadIdx.x; k < 3 * C; k += 256) {
            const double v = (double)lds[k] + (double)lds[3 * C + k] + (double)lds[6 * C + k] + (double)lds[9 * C + k];
            if (v != 0.0) atomicAdd(&a.sums[2 + k], v);
        }
    }
}

// ---------------------------------------------------------------------------------------------- focal backward
// d(loss)/dx for the sigmoid focal loss.  coef[0] multiplies dL_j/dx_j, coef[1] multiplies dF_j/dx_j (normalised
// variant: -sum(L)/N^2), both already scaled by the upstream gradient and the reduction's 1/numel; coef lives on the
// device so no host synchronisation is needed.  grad_elem (optional) is a per-element upstream gradient (reduction none).
__global__ __launch_bounds__(256) void focal_bwd_kernel(const SegArgs a, const float* __restrict__ coef,
                                                        const float* __restrict__ grad_elem, float* __restrict__ grad) {
    const long long n = (long long)a.B * a.C * a.HW;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const bool ign = a.flags & SEG_HAS_IGNORE;
    const float k1 = coef[0], k2 = coef[1];
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const long long plane = i / a.HW;
        const int c = (int)(plane % a.C);
        const long long b = plane / a.C;
        const float x = a.logits[i];
        float t;
        bool ig = false;
        if (a.labels) {
            const long long lab = a.labels[b * a.HW + (i - plane * a.HW)];
            ig = ign && lab == a.ignore_label;
            t = lab == c ? 1.f : 0.f;
        } else {
            t = a.dense[i];
            ig = ign && t == a.ignore_value;
        }
        float gx = 0.f;
        if (!ig) {
            const float e = __expf(-fabsf(x));
            const float inv = 1.0f / (1.0f + e);
            const float p = x >= 0.f ? inv : e * inv;
            const float ce = fmaxf(x, 0.f) - x * t + log1pf(e);
            const float pt = p * t + (1.f - p) * (1.f - t);
            const float dpt = p * (1.f - p) * (2.f * t - 1.f);
            float f, df;
            if (a.flags & SEG_REDUCED) {
                if (pt < a.threshold) { f = 1.f; df = 0.f; }
                else {
                    const float s = 1.f / (1.f - a.threshold);
                    const float base = (1.f - pt) * s;
                    f = powf_pos(base, a.gamma);
                    df = a.gamma == 0.f ? 0.f : -a.gamma * powf_pos(base, a.gamma - 1.f) * s * dpt;
                }
            } else {
                const float base = fmaxf(1.f - pt, 0.f);
                f = powf_pos(base, a.gamma);
                df = a.gamma == 0.f ? 0.f : -a.gamma * (a.gamma == 1.f ? 1.f : powf_pos(base, a.gamma - 1.f)) * dpt;
            }
            float w = a.class_weights ? a.class_weights[c] : 1.f;
            if (a.flags & SEG_HAS_ALPHA) w *= a.alpha * t + (1.f - a.alpha) * (1.f - t);
            const float dL = w * (df * ce + f * (p - t));
            const float g1 = grad_elem ? k1 * grad_elem[i] : k1;
            gx = g1 * dL + k2 * df;
        }
        grad[i] = gx;
    }
}

// ---------------------------------------------------------------------------------------------- region-stat backward
// Given dLoss/dI_c = gI[c] and dLoss/dP_c = gP[c] (device arrays, from the [C]-sized scalar epilogue), write
// dLoss/dlogits.  With G_c = (gI[c]*t_c + gP[c]) * mask:  softmax: p_k (G_k - sum_c G_c p_c);  sigmoid: G p (1-p);
// identity: G.
template <int PIX>
__global__ __launch_bounds__(256) void seg_stats_bwd_kernel(const SegArgs a, const float* __restrict__ gI,
                                                            const float* __restrict__ gP, float* __restrict__ grad) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int C = a.C;
    const long long per_img = (a.HW + 64 * PIX - 1) / (64 * PIX);
    const long long groups = per_img * a.B;
    const bool ign = a.flags & SEG_HAS_IGNORE;
    for (long long g = (long long)blockIdx.x * 4 + wave; g < groups; g += (long long)gridDim.x * 4) {
        const int b = (int)(g / per_img);
        const long long i0 = (g % per_img) * 64 * PIX + (long long)lane * PIX;
        bool valid[PIX], ignored[PIX];
        long long lab[PIX];
#pragma unroll
        for (int k = 0; k < PIX; ++k) {
            valid[k] = i0 + k < a.HW;
            lab[k] = -1;
            ignored[k] = false;
            if (valid[k] && a.labels) {
                lab[k] = a.labels[(long long)b * a.HW + i0 + k];
                ignored[k] = ign && lab[k] == a.ignore_label;
            }
        }
        float mx[PIX], den[PIX], dot[PIX];
        if (a.prob == PROB_SOFTMAX) {
#pragma unroll
            for (int k = 0; k < PIX; ++k) { mx[k] = -INFINITY; den[k] = 0.f; dot[k] = 0.f; }
            for (int c = 0; c < C; ++c) {
                const float* row = a.logits + ((long long)b * C + c) * a.HW + i0;
#pragma unroll
                for (int k = 0; k < PIX; ++k) if (valid[k]) {
                    const float x = row[k];
                    const float m2 = fmaxf(mx[k], x);
                    den[k] = den[k] * __expf(mx[k] - m2) + __expf(x - m2);
                    mx[k] = m2;
                }
            }
            for (int c = 0; c < C; ++c) {  // dot = sum_c G_c p_c
                const long long off = ((long long)b * C + c) * a.HW + i0;
#pragma unroll
                for (int k = 0; k < PIX; ++k) if (valid[k] && !ignored[k]) {
                    const float p = __expf(a.logits[off + k] - mx[k]) / den[k];
                    const float t = a.labels ? (lab[k] == c ? 1.f : 0.f) : a.dense[off + k];
                    dot[k] += (gI[c] * t + gP[c]) * p;
                }
            }
        }
        for (int c = 0; c < C; ++c) {
            const long long off = ((long long)b * C + c) * a.HW + i0;
#pragma unroll
            for (int k = 0; k < PIX; ++k) if (valid[k]) {
                const float x = a.logits[off + k];
                float t = a.labels ? (lab[k] == c ? 1.f : 0.f) : a.dense[off + k];
                bool ig = ignored[k];
                if (!a.labels && ign && t == a.ignore_value) ig = true;
                float gx = 0.f;
                if (!ig) {
                    const float G = gI[c] * t + gP[c];
                    if (a.prob == PROB_SOFTMAX) { const float p = __expf(x - mx[k]) / den[k]; gx = p * (G - dot[k]); }
                    else if (a.prob == PROB_SIGMOID) { const float e = __expf(-fabsf(x)); const float p = (x >= 0.f ? 1.f : e) / (1.f + e); gx = G * p * (1.f - p); }
                    else gx = G;
                }
                grad[off + k] = gx;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------- softmax focal
// softmax_focal_loss_with_logits (functional.py:110-173): per pixel sum_c pt_c^gamma * BCE(x_c, onehot_c) * w_c, masked by
// label != ignore_index.  sums[0] = sum of pixel losses, sums[1] = sum of ALL focal terms (the reference does not
// mask them, functional.py:161-164).  pixel_out (optional) receives the unreduced [B, HW] map.
struct SmfArgs {
    const float* logits; const long long* labels; const float* class_weights;
    double* sums; float* pixel_out; int* error_flag;
    int B, C; long long HW;
    int reduced; float gamma, threshold; long long ignore_label;
};

__device__ __forceinline__ float smf_term(float pt, const SmfArgs& a) {
    if (a.reduced) return pt < a.threshold ? 1.0f : powf_pos(pt / a.threshold, a.gamma);
    return powf_pos(pt, a.gamma);
}

__global__ __launch_bounds__(256) void softmax_focal_fwd_kernel(const SmfArgs a) {
    const long long n = (long long)a.B * a.HW;
    const long long stride = (long long)gridDim.x * blockDim.x;
    double s_loss = 0.0, s_term = 0.0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const long long b = i / a.HW, px = i - b * a.HW;
        const long long lab = a.labels[i];
        const bool ig = lab == a.ignore_label;
        if (!ig && (lab < 0 || lab >= a.C)) *a.error_flag = 1;
        const long long tgt = ig ? 0 : lab;   // masked_fill(target, ignore, 0), functional.py:139
        const float* row = a.logits + b * a.C * a.HW + px;
        float mx = -INFINITY, den = 0.f;
        for (int c = 0; c < a.C; ++c) {
            const float x = row[(long long)c * a.HW];
            const float m2 = fmaxf(mx, x);
            den = den * __expf(mx - m2) + __expf(x - m2);
            mx = m2;
        }
        float loss = 0.f, term = 0.f;
        for (int c = 0; c < a.C; ++c) {
            const float x = row[(long long)c * a.HW];
            const float p = __expf(x - mx) / den;
            const float t = c == tgt ? 1.f : 0.f;
            const float pt = (1.f - t) * p + t * (1.f - p);
            const float f = smf_term(pt, a);
            const float bce = fmaxf(x, 0.f) - x * t + log1pf(__expf(-fabsf(x)));
            loss += f * bce * (a.class_weights ? a.class_weights[c] : 1.f);
            term += f;
        }
        if (ig) loss = 0.f;
        if (a.pixel_out) a.pixel_out[i] = loss;
        s_loss += (double)loss;
        s_term += (double)term;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s_loss += __shfl_xor(s_loss, o); s_term += __shfl_xor(s_term, o); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&a.sums[0], s_loss); atomicAdd(&a.sums[1], s_term); }
}

// grad_k = k1 * g_i * [ p_k (h_k - sum_c h_c p_c) + w_k f_k (sigmoid(x_k) - t_k) ] + k2 * p_k (d_k - sum_c d_c p_c)
// with h_c = w_c * bce_c * df_c/dp_c, d_c = df_c/dp_c; k1 (x per-pixel upstream g_i, optional) and k2 as in focal_bwd.
__global__ __launch_bounds__(256) void softmax_focal_bwd_kernel(const SmfArgs a, const float* __restrict__ coef,
                                                                const float* __restrict__ grad_pix, float* __restrict__ grad) {
    const long long n = (long long)a.B * a.HW;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const float k1 = coef[0], k2 = coef[1];
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const long long b = i / a.HW, px = i - b * a.HW;
        const long long lab = a.labels[i];
        const bool ig = lab == a.ignore_label;
        const long long tgt = ig ? 0 : lab;
        const float* row = a.logits + b * a.C * a.HW + px;
        float* grow = grad + b * a.C * a.HW + px;
        float mx = -INFINITY, den = 0.f;
        for (int c = 0; c < a.C; ++c) {
            const float x = row[(long long)c * a.HW];
            const float m2 = fmaxf(mx, x);
            den = den * __expf(mx - m2) + __expf(x - m2);
            mx = m2;
        }
        const float g1 = ig ? 0.f : (grad_pix ? k1 * grad_pix[i] : k1);
        float dot_h = 0.f, dot_d = 0.f;
        for (int c = 0; c < a.C; ++c) {
            const float x = row[(long long)c * a.HW];
            const float p = __expf(x - mx) / den;
            const float t = c == tgt ? 1.f : 0.f;
            const float pt = (1.f - t) * p + t * (1.f - p);
            float df;  // d f / d p
            if (a.reduced) df = pt < a.threshold ? 0.f : a.gamma * powf_pos(pt / a.threshold, a.gamma - 1.f) / a.threshold * (1.f - 2.f * t);
            else df = a.gamma == 0.f ? 0.f : a.gamma * (a.gamma == 1.f ? 1.f : powf_pos(pt, a.gamma - 1.f)) * (1.f - 2.f * t);
            const float bce = fmaxf(x, 0.f) - x * t + log1pf(__expf(-fabsf(x)));
            const float w = a.class_weights ? a.class_weights[c] : 1.f;
            dot_h += w * bce * df * p;
            dot_d += df * p;
        }
        for (int c = 0; c < a.C; ++c) {
            const float x = row[(long long)c * a.HW];
            const float p = __expf(x - mx) / den;
            const float t = c == tgt ? 1.f : 0.f;
            const float pt = (1.f - t) * p + t * (1.f - p);
            float df;
            if (a.reduced) df = pt < a.threshold ? 0.f : a.gamma * powf_pos(pt / a.threshold, a.gamma - 1.f) / a.threshold * (1.f - 2.f * t);
            else df = a.gamma == 0.f ? 0.f : a.gamma * (a.gamma == 1.f ? 1.f : powf_pos(pt, a.gamma - 1.f)) * (1.f - 2.f * t);
            const float f = smf_term(pt, a);
            const float e = __expf(-fabsf(x));
            const float sg = (x >= 0.f ? 1.f : e) / (1.f + e);
            const float bce = fmaxf(x, 0.f) - x * t + log1pf(e);
            const float w = a.class_weights ? a.class_weights[c] : 1.f;
            const float gl = p * (w * bce * df - dot_h) + w * f * (sg - t);
            const float gf = p * (df - dot_d);
            grow[(long long)c * a.HW] = g1 * gl + k2 * gf;
        }
    }
}

static int grid_for(long long work_items, int per_block) {
    const long long want = (work_items + per_block - 1) / per_block;
    const long long cap = 256LL * 8;
    return (int)(want < 1 ? 1 : (want < cap ? want : cap));
}

static int fill_seg(SegArgs& a, const float* logits, const int64_t* labels, const float* dense, const float* class_weights,
                    int B, int C, int64_t HW, int flags, int prob, float gamma, float alpha, float threshold,
                    int64_t ignore_label, float ignore_value) {
    if (!logits || (!labels && !dense) || B < 0 || C < 1 || HW < 0) return PTB_EINVAL;
    if (prob < PROB_SOFTMAX || prob > PROB_IDENTITY) return PTB_EINVAL;
    a.logits = logits; a.labels = (const long long*)labels; a.dense = dense; a.class_weights = class_weights;
    a.B = B; a.C = C; a.HW = HW; a.flags = flags; a.prob = prob;
    a.gamma = gamma; a.alpha = alpha; a.threshold = threshold; a.ignore_label = ignore_label; a.ignore_value = ignore_value;
    return PTB_OK;
}

}  // namespace ptb

using namespace ptb;

extern "C" int ptb_seg_loss_fwd(const float* logits, const int64_t* labels, const float* dense, const float* class_weights,
                                double* sums, float* elem_out, int* error_flag, int B, int C, int64_t HW, int flags, int prob,
                                float gamma, float alpha, float threshold, int64_t ignore_label, float ignore_value,
                                ptb_stream_t stream) {
    SegArgs a{};
    if (int rc = fill_seg(a, logits, labels, dense, class_weights, B, C, HW, flags, prob, gamma, alpha, threshold, ignore_label, ignore_value)) return rc;
    if (!sums || !error_flag || ((flags & SEG_ELEMWISE) && !elem_out)) return PTB_EINVAL;
    if (C > 1024) return PTB_EUNSUPPORTED;
    a.sums = sums; a.elem_out = elem_out; a.error_flag = error_flag;
    if ((long long)B * HW == 0) return PTB_OK;
    const size_t shmem = (size_t)4 * 3 * C * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
    if (HW % 4 == 0 && aligned16(logits) && (!dense || aligned16(dense)) && (!elem_out || aligned16(elem_out))) {
        const long long groups = (HW + 255) / 256 * B;
        hipLaunchKernelGGL(seg_loss_fwd_kernel<4>, dim3(grid_for(groups, 4)), dim3(256), shmem, s, a);
    } else {
        const long long groups = (HW + 63) / 64 * B;
        hipLaunchKernelGGL(seg_loss_fwd_kernel<1>, dim3(grid_for(groups, 4)), dim3(256), shmem, s, a);
    }
    return check_launch();
}

extern "C" int ptb_focal_bwd(const float* logits, const int64_t* labels, const float* dense, const float* class_weights,
                             const float* coef, const float* grad_elem, float* grad, int B, int C, int64_t HW, int flags,
                             float gamma, float alpha, float threshold, int64_t ignore_label, float ignore_value,
                             ptb_stream_t stream) {
    SegArgs a{};
    if (int rc = fill_seg(a, logits, labels, dense, class_weights, B, C, HW, flags, PROB_SIGMOID, gamma, alpha, threshold, ignore_label, ignore_value)) return rc;
    if (!coef || !grad) return PTB_EINVAL;
    const long long n = (long long)B * C * HW;
    if (n == 0) return PTB_OK;
    hipLaunchKernelGGL(focal_bwd_kernel, dim3(grid_for(n, 256 * 4)), dim3(256), 0, (hipStream_t)stream, a, coef, grad_elem, grad);
    return check_launch();
}

extern "C" int ptb_seg_stats_bwd(const float* logits, const int64_t* labels, const float* dense, const float* gI, const float* gP,
                                 float* grad, int B, int C, int64_t HW, int flags, int prob, int64_t ignore_label,
                                 float ignore_value, ptb_stream_t stream) {
    SegArgs a{};
    if (int rc = fill_seg(a, logits, labels, dense, nullptr, B, C, HW, flags, prob, 0.f, 0.f, 0.f, ignore_label, ignore_value)) return rc;
    if (!gI || !gP || !grad) return PTB_EINVAL;
    if ((long long)B * HW == 0) return PTB_OK;
    hipStream_t s = (hipStream_t)stream;
    if (HW % 4 == 0) {
        const long long groups = (HW + 255) / 256 * B;
        hipLaunchKernelGGL(seg_stats_bwd_kernel<4>, dim3(grid_for(groups, 4)), dim3(256), 0, s, a, gI, gP, grad);
    } else {
        const long long groups = (HW + 63) / 64 * B;
        hipLaunchKernelGGL(seg_stats_bwd_kernel<1>, dim3(grid_for(groups, 4)), dim3(256), 0, s, a, gI, gP, grad);
    }
    return check_launch();
}

extern "C" int ptb_softmax_focal_fwd(const float* logits, const int64_t* labels, const float* class_weights, double* sums,
                                     float* pixel_out, int* error_flag, int B, int C, int64_t HW, int reduced, float gamma,
                                     float threshold, int64_t ignore_label, ptb_stream_t stream) {
    if (!logits || !labels || !sums || !error_flag || B < 0 || C < 1 || HW < 0) return PTB_EINVAL;
    if ((long long)B * HW == 0) return PTB_OK;
    SmfArgs a{logits, (const long long*)labels, class_weights, sums, pixel_out, error_flag, B, C, HW, reduced, gamma, threshold, ignore_label};
    hipLaunchKernelGGL(softmax_focal_fwd_kernel, dim3(grid_for((long long)B * HW, 256)), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch();
}

extern "C" int ptb_softmax_focal_bwd(const float* logits, const int64_t* labels, const float* class_weights, const float* coef,
                                     const float* grad_pix, float* grad, int B, int C, int64_t HW, int reduced, float gamma,
                                     float threshold, int64_t ignore_label, ptb_stream_t stream) {
    if (!logits || !labels || !coef || !grad || B < 0 || C < 1 || HW < 0) return PTB_EINVAL;
    if ((long long)B * HW == 0) return PTB_OK;
    SmfArgs a{logits, (const long long*)labels, class_weights, nullptr, nullptr, nullptr, B, C, HW, reduced, gamma, threshold, ignore_label};
    hipLaunchKernelGGL(softmax_focal_bwd_kernel, dim3(grid_for((long long)B * HW, 256)), dim3(256), 0, (hipStream_t)stream, a, coef, grad_pix, grad);
    return check_launch();
}
