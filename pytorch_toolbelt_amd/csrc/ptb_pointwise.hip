// ptb_pointwise.hip -- the remaining elementwise + reduce losses of the reference, one fused pass each:
//
//   SoftBCEWithLogitsLoss                      losses/soft_bce.py:9-48      (label smoothing, ignore_index, weight, pos_weight)
//   balanced_binary_cross_entropy_with_logits  losses/balanced_bce.py:10-49 (class counts and both log-sigmoid sums in ONE pass)
//   QualityFocalLoss                           losses/quality_focal_loss.py:5-46
//   wing_loss                                  losses/functional.py:250-277
//   log_cosh_loss                              losses/functional.py:326-342
//   BinarySoftF1Loss / soft_micro_f1 (one class)     losses/soft_f1.py:8-78 (soft TP / FP / FN counts in one pass)
//   SoftCrossEntropyLoss / label_smoothed_nll_loss   losses/soft_ce.py:9-33, losses/functional.py:280-323
//
// The reference evaluates each as a chain of 6-15 full-tensor torch ops; here the forward is one read of logits +
// targets with the math in registers and a hierarchical (wave -> workgroup -> 64 slotted fp64 atomics) sum, and the
// backward is one read + one write.  All HBM-bound streaming: 16 B per lane, no MFMA.  Scalar epilogues (means,
// normalisation, class-balance weights) stay on the device in the caller's autograd graph, so nothing synchronises.
#include "ptb_common.h"

namespace ptb {

namespace {

constexpr int PW_SLOTS = 64;  // == PTB_SUM_SLOTS
constexpr float kLog2e = 1.4426950408889634f, kLn2 = 0.6931471805599453f;

__device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float ex2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float lg2(float x) { return __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float fexp(float x) { return ex2(x * kLog2e); }
__device__ __forceinline__ float flog(float x) { return lg2(x) * kLn2; }

// sigmoid(x) and log(1 + exp(-|x|)) from one exponential
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

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// workgroup (256 threads) reduction of NS partial sums, then ONE atomic per sum into this workgroup's slot
template <int NS>
__device__ __forceinline__ void block_add(const float* part, double* slot) {
    __shared__ double red[NS][4];
    const int lane = threadIdx.x & 63, wave = wave_id();
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const double v = wave_sum((double)part[k]);
        if (lane == 0) red[k][wave] = v;
    }
    __syncthreads();
    if (threadIdx.x < NS) atomicAdd(&slot[threadIdx.x], red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3]);
}

}  // namespace

enum { PW_SOFT_BCE = 0, PW_BALANCED_BCE = 1, PW_QFL = 2, PW_WING = 3, PW_LOGCOSH = 4, PW_SOFT_F1 = 5, PW_KINDS = 6 };
enum { PWF_IGNORE = 1, PWF_SMOOTH = 2 };

struct PwArgs {
    const float* x;       // logits / predictions, n elements
    const float* t;       // targets, n elements
    const float* chan_w;  // [C] per-channel weight or null   (soft BCE)
    const float* chan_pw; // [C] per-channel pos_weight or null (soft BCE)
    double* sums;         // [PW_SLOTS][4], zeroed by the caller
    float* out;           // forward: optional per-element loss; apply kernel: gradient or per-element loss
    long long n, HW;
    int C;
    int flags;
    float p0, p1, p2;     // soft BCE: smooth factor; QFL: beta; wing: width, curvature, C constant
    float ignore_value;
};

// ---- per-element forward -------------------------------------------------------------------------------------------
template <int KIND>
__device__ __forceinline__ void pw_forward(float x, float t, float w, float pw, const PwArgs& a, float* s, float& loss) {
    if (KIND == PW_SOFT_BCE) {   // soft_bce.py:29-41; F.binary_cross_entropy_with_logits with weight / pos_weight
        const bool ig = (a.flags & PWF_IGNORE) && t == a.ignore_value;
        const float st = (a.flags & PWF_SMOOTH) ? (1.f - t) * a.p0 + t * (1.f - a.p0) : t;
        const Sig g = sigmoid_parts(x);
        const float lw = 1.f + (pw - 1.f) * st;
        loss = w * ((1.f - st) * x + lw * (g.log1pe + fmaxf(-x, 0.f)));
        loss = ig ? 0.f : loss;
        s[0] += loss;
    } else if (KIND == PW_BALANCED_BCE) {  // balanced_bce.py:27-40: sums of t*logsigmoid(x), (1-t)*logsigmoid(-x); class counts
        const bool ig = (a.flags & PWF_IGNORE) && t == a.ignore_value;
        const Sig g = sigmoid_parts(x);
        const float ls_pos = -(fmaxf(-x, 0.f) + g.log1pe), ls_neg = -(fmaxf(x, 0.f) + g.log1pe);
        s[0] += ig ? 0.f : t * ls_pos;
        s[1] += ig ? 0.f : (1.f - t) * ls_neg;
        s[2] += t == 1.f ? 1.f : 0.f;
        s[3] += t == 0.f ? 1.f : 0.f;
        loss = 0.f;
    } else if (KIND == PW_QFL) {  // quality_focal_loss.py:33-35
        const Sig g = sigmoid_parts(x);
        const float bce = fmaxf(x, 0.f) - x * t + g.log1pe;
        const float d = fabsf(g.p - t);
        float f = a.p0 == 2.f ? d * d : ex2(a.p0 * lg2(d));
        f = a.p0 == 0.f ? 1.f : f;
        loss = f * bce;
        s[0] += loss;
        s[1] += f;
    } else if (KIND == PW_SOFT_F1) {  // soft_f1.py:22-24, 63-78: p = clamp(sigmoid(x), eps, 1 - eps) (flag PWF_SMOOTH: x IS the
        // probability); soft counts sum p t, sum p, sum t -- TP = s0, FP = s1 - s0, FN = s2 - s0 -- and the number of kept elements
        const bool ig = (a.flags & PWF_IGNORE) && t == a.ignore_value;
        float p = (a.flags & PWF_SMOOTH) ? x : fminf(fmaxf(sigmoid_parts(x).p, a.p0), 1.f - a.p0);
        p = ig ? 0.f : p;
        const float tt = ig ? 0.f : t;
        s[0] += p * tt;
        s[1] += p;
        s[2] += tt;
        s[3] += ig ? 0.f : 1.f;
        loss = 0.f;
    } else if (KIND == PW_WING) {  // functional.py:260-269
        const float d = fabsf(t - x);
        loss = d < a.p0 ? a.p0 * logf(1.f + d / a.p1) : d - a.p2;
        s[0] += loss;
    } else {  // PW_LOGCOSH, functional.py:338-341: z + softplus(-2z) - log 2
        const float z = x - t, y = -2.f * z;
        loss = z + (fmaxf(y, 0.f) + flog(1.f + fexp(-fabsf(y)))) - kLn2;
        s[0] += loss;
    }
}

// ---- per-element backward (d loss_i / d x_i pieces combined with the device-side coefficients) ---------------------
// grad = k0 * g_i * dL_i (+ k1 * dF_i for QFL's normalised form); balanced BCE: k0 / k1 are the class weights times the
// upstream gradient.  LOSS = true emits the per-element loss of balanced BCE (reduction='none') instead.
template <int KIND, bool LOSS>
__device__ __forceinline__ float pw_backward(float x, float t, float w, float pw, float gi, float k0, float k1, const PwArgs& a) {
    if (KIND == PW_SOFT_BCE) {
        const bool ig = (a.flags & PWF_IGNORE) && t == a.ignore_value;
        const float st = (a.flags & PWF_SMOOTH) ? (1.f - t) * a.p0 + t * (1.f - a.p0) : t;
        const Sig g = sigmoid_parts(x);
        const float lw = 1.f + (pw - 1.f) * st;
        const float d = w * ((1.f - st) - lw * (1.f - g.p));
        return ig ? 0.f : k0 * gi * d;
    } else if (KIND == PW_BALANCED_BCE) {
        const bool ig = (a.flags & PWF_IGNORE) && t == a.ignore_value;
        const Sig g = sigmoid_parts(x);
        if (LOSS) {
            const float ls_pos = -(fmaxf(-x, 0.f) + g.log1pe), ls_neg = -(fmaxf(x, 0.f) + g.log1pe);
            return ig ? 0.f : -(k0 * t * ls_pos + k1 * (1.f - t) * ls_neg);
        }
        return ig ? 0.f : -gi * (k0 * t * (1.f - g.p) - k1 * (1.f - t) * g.p);
    } else if (KIND == PW_QFL) {
        const Sig g = sigmoid_parts(x);
        const float bce = fmaxf(x, 0.f) - x * t + g.log1pe;
        const float diff = g.p - t, d = fabsf(diff);
        float f = a.p0 == 2.f ? d * d : ex2(a.p0 * lg2(d));
        f = a.p0 == 0.f ? 1.f : f;
        float pwm1 = a.p0 == 2.f ? d : ex2((a.p0 - 1.f) * lg2(d));   // d^(beta-1)
        pwm1 = a.p0 == 1.f ? 1.f : pwm1;
        const float sgn = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);
        float df = a.p0 * pwm1 * sgn * g.p * (1.f - g.p);
        df = a.p0 == 0.f ? 0.f : df;
        return k0 * gi * (df * bce + f * diff) + k1 * df;
    } else if (KIND == PW_SOFT_F1) {   // d(k0 * sum p t + k1 * sum p) / dx; the clamp has zero slope outside (eps, 1 - eps)
        const bool ig = (a.flags & PWF_IGNORE) && t == a.ignore_value;
        if (a.flags & PWF_SMOOTH) return ig ? 0.f : gi * (k0 * t + k1);
        const float p = sigmoid_parts(x).p;
        const bool inside = p > a.p0 && p < 1.f - a.p0;
        return (ig || !inside) ? 0.f : gi * (k0 * t + k1) * p * (1.f - p);
    } else if (KIND == PW_WING) {
        const float diff = t - x, d = fabsf(diff);
        const float sgn = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);
        const float slope = d < a.p0 ? a.p0 / (a.p1 + d) : 1.f;
        return -k0 * gi * sgn * slope;
    } else {
        const float z = x - t;
        const Sig g = sigmoid_parts(2.f * z);
        return k0 * gi * (2.f * g.p - 1.f);   // tanh z
    }
}

typedef float pw_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 pw_ld16(const float* p) {     // 16-byte non-temporal load (streamed once)
    const pw_v4f v = __builtin_nontemporal_load(reinterpret_cast<const pw_v4f*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}

template <int KIND>
constexpr int pw_nsums() { return (KIND == PW_BALANCED_BCE || KIND == PW_SOFT_F1) ? 4 : (KIND == PW_QFL ? 2 : 1); }

template <int KIND, bool VEC>
__global__ __launch_bounds__(256) void pw_fwd_kernel(const PwArgs a) {
    constexpr int PIX = VEC ? 4 : 1;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    const long long groups = (a.n + PIX - 1) / PIX;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const bool per_chan = a.chan_w || a.chan_pw;
    // two groups per trip on the vector path: 4 x 16 B of logits and targets in flight per lane before the first transcendental
    constexpr int U = VEC ? 2 : 1;
    for (long long g0 = (long long)blockIdx.x * blockDim.x + threadIdx.x; g0 < groups; g0 += U * stride) {
        float x[U][PIX], t[U][PIX];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long g = g0 + u * stride;
#pragma unroll
            for (int k = 0; k < PIX; ++k) { x[u][k] = 0.f; t[u][k] = 0.f; }
            if (g >= groups) continue;
            const long long i = g * PIX;
            if (VEC) {
                const float4 xv = pw_ld16(a.x + i), tv = pw_ld16(a.t + i);
                x[u][0] = xv.x; t[u][0] = tv.x;
                if (PIX == 4) { x[u][1] = xv.y; x[u][2] = xv.z; x[u][3] = xv.w; t[u][1] = tv.y; t[u][2] = tv.z; t[u][3] = tv.w; }
            } else {
                x[u][0] = a.x[i]; t[u][0] = a.t[i];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long g = g0 + u * stride;
            if (g >= groups) continue;
            const long long i = g * PIX;
            float l[PIX];
            float w = 1.f, pw = 1.f;
            if (per_chan) {  // HW % 4 == 0 on the vector path: the 4 elements share a channel
                const int c = (int)((i / a.HW) % a.C);
                w = a.chan_w ? a.chan_w[c] : 1.f;
                pw = a.chan_pw ? a.chan_pw[c] : 1.f;
            }
#pragma unroll
            for (int k = 0; k < PIX; ++k) pw_forward<KIND>(x[u][k], t[u][k], w, pw, a, s, l[k]);
            if (a.out) {
                if (VEC) out_store4(a.out + i, make_float4(l[0], l[PIX > 1 ? 1 : 0], l[PIX > 2 ? 2 : 0], l[PIX > 3 ? 3 : 0]));
                else a.out[i] = l[0];
            }
        }
    }
    block_add<pw_nsums<KIND>()>(s, a.sums + (size_t)(blockIdx.x % PW_SLOTS) * 4);
}

template <int KIND, bool VEC, bool LOSS>
__global__ __launch_bounds__(256) void pw_apply_kernel(const PwArgs a, const float* __restrict__ coef, const float* __restrict__ grad_elem) {
    constexpr int PIX = VEC ? 4 : 1;
    const float k0 = coef[0], k1 = coef[1];
    const long long groups = (a.n + PIX - 1) / PIX;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const bool per_chan = a.chan_w || a.chan_pw;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += stride) {
        const long long i = g * PIX;
        float x[PIX], t[PIX], ge[PIX], o[PIX];
#pragma unroll
        for (int k = 0; k < PIX; ++k) ge[k] = 1.f;
        if (VEC) {
            const float4 xv = *reinterpret_cast<const float4*>(a.x + i), tv = *reinterpret_cast<const float4*>(a.t + i);
            x[0] = xv.x; t[0] = tv.x;
            if (PIX == 4) { x[1] = xv.y; x[2] = xv.z; x[3] = xv.w; t[1] = tv.y; t[2] = tv.z; t[3] = tv.w; }
            if (grad_elem) {
                const float4 gv = *reinterpret_cast<const float4*>(grad_elem + i);
                ge[0] = gv.x;
                if (PIX == 4) { ge[1] = gv.y; ge[2] = gv.z; ge[3] = gv.w; }
            }
        } else {
            x[0] = a.x[i]; t[0] = a.t[i];
            if (grad_elem) ge[0] = grad_elem[i];
        }
        float w = 1.f, pw = 1.f;
        if (per_chan) {
            const int c = (int)((i / a.HW) % a.C);
            w = a.chan_w ? a.chan_w[c] : 1.f;
            pw = a.chan_pw ? a.chan_pw[c] : 1.f;
        }
#pragma unroll
        for (int k = 0; k < PIX; ++k) o[k] = pw_backward<KIND, LOSS>(x[k], t[k], w, pw, ge[k], k0, k1, a);
        if (VEC) out_store4(a.out + i, make_float4(o[0], o[PIX > 1 ? 1 : 0], o[PIX > 2 ? 2 : 0], o[PIX > 3 ? 3 : 0]));
        else a.out[i] = o[0];
    }
}

// ------------------------------------------------------------------------------------------------ soft cross entropy
// label_smoothed_nll_loss over log_softmax (functional.py:280-323, soft_ce.py:24-33) for [B, C, HW] logits and int64
// labels [B, HW]: per pixel  nll = lse - x_t,  smooth = C * lse - sum_c x_c  (both 0 on ignored pixels).
// MODE 0: sums[slot][0..1] += (sum nll, sum smooth), optional per-pixel (1-eps) * nll + eps/C * smooth.
// MODE 1: grad[c] = k * g_px * ((1-eps) * (p_c - [c == t]) + eps/C * (C * p_c - 1)).
struct SceArgs {
    const float* x;
    const long long* labels;
    double* sums;
    float* pix_out;
    int* error_flag;
    int B, C;
    long long HW;
    float eps;
    int has_ignore;
    long long ignore_label;
};

template <int PIX, int CREG, int MODE>
__global__ __launch_bounds__(256) void soft_ce_kernel(const SceArgs a, const float* __restrict__ coef, const float* __restrict__ grad_pix,
                                                      float* __restrict__ grad) {
    float s[2] = {0.f, 0.f};
    const long long gpi = (a.HW + PIX - 1) / PIX;  // pixel groups per image
    const long long groups = gpi * a.B;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const float k0 = MODE == 1 ? coef[0] : 0.f;
    const float epsC = a.eps / (float)a.C;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += stride) {
        const long long b = g / gpi, p0 = (g - b * gpi) * PIX;
        const long long base = b * a.C * a.HW + p0;
        long long lab[PIX];
        bool ig[PIX];
#pragma unroll
        for (int k = 0; k < PIX; ++k) {
            lab[k] = a.labels[b * a.HW + p0 + k];
            ig[k] = a.has_ignore && lab[k] == a.ignore_label;
            if (!ig[k] && (lab[k] < 0 || lab[k] >= a.C)) { if (a.error_flag) *a.error_flag = 1; ig[k] = true; }
        }
        float x[CREG][PIX], m[PIX], z[PIX], sx[PIX], xt[PIX];
#pragma unroll
        for (int k = 0; k < PIX; ++k) { m[k] = -INFINITY; z[k] = 0.f; sx[k] = 0.f; xt[k] = 0.f; }
        // all class planes are requested first (only the loads sit behind the `c < C` guards), then reduced: with the
        // reduction inside the guarded block every load was waited for before the next one was issued (149 us at cfg4)
#pragma unroll
        for (int c = 0; c < CREG; ++c) {
#pragma unroll
            for (int k = 0; k < PIX; ++k) x[c][k] = 0.f;
            if (c < a.C) {
                if (PIX == 4) {
                    typedef float v4f __attribute__((ext_vector_type(4)));
                    const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(a.x + base + (long long)c * a.HW));
                    x[c][0] = v.x; x[c][PIX > 1 ? 1 : 0] = v.y; x[c][PIX > 2 ? 2 : 0] = v.z; x[c][PIX > 3 ? 3 : 0] = v.w;
                } else {
                    x[c][0] = a.x[base + (long long)c * a.HW];
                }
            }
        }
#pragma unroll
        for (int c = 0; c < CREG; ++c) {
            if (c < a.C) {
#pragma unroll
                for (int k = 0; k < PIX; ++k) {
                    m[k] = fmaxf(m[k], x[c][k]);
                    sx[k] += x[c][k];
                    xt[k] = lab[k] == c ? x[c][k] : xt[k];
                }
            }
        }
#pragma unroll
        for (int c = 0; c < CREG; ++c) {
            if (c < a.C) {
#pragma unroll
                for (int k = 0; k < PIX; ++k) {
                    const float e = fexp(x[c][k] - m[k]);
                    z[k] += e;
                    if (MODE == 1) x[c][k] = e;
                }
            }
        }
        if (MODE == 0) {
            float px[PIX];
#pragma unroll
            for (int k = 0; k < PIX; ++k) {
                const float lse = m[k] + flog(z[k]);
                const float nll = ig[k] ? 0.f : lse - xt[k];
                const float sm = ig[k] ? 0.f : (float)a.C * lse - sx[k];
                s[0] += nll;
                s[1] += sm;
                px[k] = (1.f - a.eps) * nll + epsC * sm;
            }
            if (a.pix_out) {
                if (PIX == 4) *reinterpret_cast<float4*>(a.pix_out + b * a.HW + p0) = make_float4(px[0], px[PIX > 1 ? 1 : 0], px[PIX > 2 ? 2 : 0], px[PIX > 3 ? 3 : 0]);
                else a.pix_out[b * a.HW + p0] = px[0];
            }
        } else {
            float gk[PIX], rz[PIX];
#pragma unroll
            for (int k = 0; k < PIX; ++k) {
                gk[k] = ig[k] ? 0.f : k0 * (grad_pix ? grad_pix[b * a.HW + p0 + k] : 1.f);
                rz[k] = rcp(z[k]);
            }
#pragma unroll
            for (int c = 0; c < CREG; ++c) {
                if (c < a.C) {
                    float o[PIX];
#pragma unroll
                    for (int k = 0; k < PIX; ++k) {
                        const float p = x[c][k] * rz[k];
                        o[k] = gk[k] * ((1.f - a.eps) * (p - (lab[k] == c ? 1.f : 0.f)) + epsC * ((float)a.C * p - 1.f));
                    }
                    if (PIX == 4) out_store4(grad + base + (long long)c * a.HW, make_float4(o[0], o[PIX > 1 ? 1 : 0], o[PIX > 2 ? 2 : 0], o[PIX > 3 ? 3 : 0]));
                    else grad[base + (long long)c * a.HW] = o[0];
                }
            }
        }
    }
    if (MODE == 0) block_add<2>(s, a.sums + (size_t)(blockIdx.x % PW_SLOTS) * 4);
}

// any number of classes: one pixel per thread, the class planes are walked three times (max, sum-exp, output)
template <int MODE>
__global__ __launch_bounds__(256) void soft_ce_generic_kernel(const SceArgs a, const float* __restrict__ coef, const float* __restrict__ grad_pix,
                                                              float* __restrict__ grad) {
    float s[2] = {0.f, 0.f};
    const long long total = (long long)a.B * a.HW;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const float k0 = MODE == 1 ? coef[0] : 0.f;
    const float epsC = a.eps / (float)a.C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const long long b = i / a.HW, p = i - b * a.HW;
        const long long base = b * a.C * a.HW + p;
        const long long lab = a.labels[i];
        bool ig = a.has_ignore && lab == a.ignore_label;
        if (!ig && (lab < 0 || lab >= a.C)) { if (a.error_flag) *a.error_flag = 1; ig = true; }
        float m = -INFINITY, sx = 0.f, xt = 0.f, z = 0.f;
        for (int c = 0; c < a.C; ++c) {
            const float v = a.x[base + (long long)c * a.HW];
            m = fmaxf(m, v); sx += v; xt = lab == c ? v : xt;
        }
        for (int c = 0; c < a.C; ++c) z += fexp(a.x[base + (long long)c * a.HW] - m);
        if (MODE == 0) {
            const float lse = m + flog(z);
            const float nll = ig ? 0.f : lse - xt, sm = ig ? 0.f : (float)a.C * lse - sx;
            s[0] += nll; s[1] += sm;
            if (a.pix_out) a.pix_out[i] = (1.f - a.eps) * nll + epsC * sm;
        } else {
            const float gk = ig ? 0.f : k0 * (grad_pix ? grad_pix[i] : 1.f), rz = rcp(z);
            for (int c = 0; c < a.C; ++c) {
                const float pr = fexp(a.x[base + (long long)c * a.HW] - m) * rz;
                grad[base + (long long)c * a.HW] = gk * ((1.f - a.eps) * (pr - (lab == c ? 1.f : 0.f)) + epsC * ((float)a.C * pr - 1.f));
            }
        }
    }
    if (MODE == 0) block_add<2>(s, a.sums + (size_t)(blockIdx.x % PW_SLOTS) * 4);
}

// ------------------------------------------------------------------------------------------------ binary bi-tempered
// BinaryBiTemperedLogisticLoss (losses/bitempered_loss.py:223-284 over bi_tempered_logistic_loss :135-180): per pixel the two
// activations (-x, x), the tempered softmax with its iterative normalisation (:25-75, num_iters = 5) and the loss terms, all
// in registers -- the reference runs ~60 full-tensor torch ops on a [B, H, W, 2] expansion.  Backward uses the closed form
// of the normalisation's derivative (the escort distribution, :94-104).
struct BtArgs {
    const float* x;
    const float* t;
    double* sums;
    float* out;       // forward: optional per-element loss; backward: gradient
    long long n;
    float t1, t2, smoothing, ignore_value;
    int iters, has_ignore;
};

__device__ __forceinline__ float bt_pow(float u, float e) { return ex2(e * lg2(u)); }          // u >= 0
__device__ __forceinline__ float bt_log_t(float u, float t) { return t == 1.0f ? flog(u) : (bt_pow(u, 1.0f - t) - 1.0f) / (1.0f - t); }
__device__ __forceinline__ float bt_exp_t(float u, float t) {
    return t == 1.0f ? fexp(u) : bt_pow(fmaxf(1.0f + (1.0f - t) * u, 0.0f), 1.0f / (1.0f - t));
}

// probabilities of the two classes for activations (a0, a1) under temperature t (tempered_softmax, :119-132)
__device__ __forceinline__ void bt_softmax2(float a0, float a1, float t, int iters, float& p0, float& p1) {
    const float mu = fmaxf(a0, a1);
    const float n0 = a0 - mu, n1 = a1 - mu;
    if (t == 1.0f) {
        const float e0 = fexp(n0), e1 = fexp(n1), r = rcp(e0 + e1);
        p0 = e0 * r; p1 = e1 * r;
        return;
    }
    float norm;
    if (t > 1.0f) {   // fixed point (:25-45)
        float c0 = n0, c1 = n1;
        for (int i = 0; i < iters; ++i) {
            const float z = bt_exp_t(c0, t) + bt_exp_t(c1, t);
            const float s = bt_pow(z, 1.0f - t);
            c0 = n0 * s; c1 = n1 * s;
        }
        const float z = bt_exp_t(c0, t) + bt_exp_t(c1, t);
        norm = -bt_log_t(1.0f / z, t);
    } else {          // bisection (:48-75)
        const float edge = -1.0f / (1.0f - t);
        const float dim = (n0 > edge ? 1.0f : 0.0f) + (n1 > edge ? 1.0f : 0.0f);
        float lo = 0.0f, hi = -bt_log_t(1.0f / dim, t);
        for (int i = 0; i < iters; ++i) {
            const float mid = (hi + lo) * 0.5f;
            const float mass = bt_exp_t(n0 - mid, t) + bt_exp_t(n1 - mid, t);
            if (mass < 1.0f) hi = mid; else lo = mid;
        }
        norm = (hi + lo) * 0.5f;
    }
    p0 = bt_exp_t(n0 - norm, t);
    p1 = bt_exp_t(n1 - norm, t);
}

template <bool BWD>
__device__ __forceinline__ float bt_element(float x, float tv, const BtArgs& a, float gi, float k0) {
    if (a.has_ignore && tv == a.ignore_value) return 0.0f;
    float y0 = 1.0f - tv, y1 = tv;                                   // onehot of the binary target (:265-266)
    if (a.smoothing > 0.0f) {                                        // :151-155 with 2 classes
        y0 = (1.0f - 2.0f * a.smoothing) * y0 + a.smoothing;
        y1 = (1.0f - 2.0f * a.smoothing) * y1 + a.smoothing;
    }
    float p0, p1;
    bt_softmax2(-x, x, a.t2, a.iters, p0, p1);
    const float e = 2.0f - a.t1;
    if (!BWD) {
        // :159-165 (y * log_t(y + 1e-10) is finite for y = 0: 0 * log_t(1e-10))
        const float l0 = y0 * bt_log_t(y0 + 1e-10f, a.t1) - y0 * bt_log_t(p0, a.t1) - bt_pow(y0, e) / e + bt_pow(p0, e) / e;
        const float l1 = y1 * bt_log_t(y1 + 1e-10f, a.t1) - y1 * bt_log_t(p1, a.t1) - bt_pow(y1, e) / e + bt_pow(p1, e) / e;
        return l0 + l1;
    }
    // dL/dp_c = -y_c p_c^(-t1) + p_c^(1 - t1);  dp_c/da_j = p_c^t2 (delta_cj - escort_j), escort = p^t2 / sum p^t2
    const float g0 = -y0 * bt_pow(p0, -a.t1) + bt_pow(p0, 1.0f - a.t1);
    const float g1 = -y1 * bt_pow(p1, -a.t1) + bt_pow(p1, 1.0f - a.t1);
    const float q0 = bt_pow(p0, a.t2), q1 = bt_pow(p1, a.t2);
    const float rs = rcp(q0 + q1);
    // a class outside the finite support (t2 < 1, p = 0) passes no gradient: the reference's relu backward zeroes it even
    // where d log_t / dp is infinite
    const float w0 = p0 > 0.0f ? g0 * q0 : 0.0f, w1 = p1 > 0.0f ? g1 * q1 : 0.0f;
    const float dot = w0 + w1;
    const float da0 = w0 - dot * q0 * rs, da1 = w1 - dot * q1 * rs;
    return k0 * gi * (da1 - da0);                                    // a0 = -x, a1 = x
}

template <bool BWD>
__global__ __launch_bounds__(256) void bitempered_binary_kernel(const BtArgs a, const float* __restrict__ coef,
                                                                const float* __restrict__ grad_elem) {
    float s[1] = {0.f};
    const float k0 = BWD ? coef[0] : 0.f;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += stride) {
        const float v = bt_element<BWD>(a.x[i], a.t[i], a, (BWD && grad_elem) ? grad_elem[i] : 1.0f, k0);
        if (BWD) a.out[i] = v;
        else { s[0] += v; if (a.out) a.out[i] = v; }
    }
    if (!BWD) block_add<1>(s, a.sums + (size_t)(blockIdx.x % PW_SLOTS) * 4);
}

// ------------------------------------------------------------------------------------------------ bi-tempered, rows of K classes
// bi_tempered_logistic_loss (losses/bitempered_loss.py:135-180) for activations [R, K] with dense (one-hot or soft) targets: one
// WAVE per row -- the row maximum, every iteration of the normalisation (fixed point for t2 > 1, :25-45; bisection for t2 < 1,
// :48-75; log-sum-exp for t2 = 1) and the loss terms are wave reductions over the K classes (lanes stride the row; rows of a
// classification head are a few KB and stay in L1/L2 between the passes).  Backward: closed form through the escort distribution
// (:94-104), like the binary kernel above.
struct BtRowArgs {
    const float* act;
    const float* onehot;
    const float* grad_loss;   // backward: [R]
    float* out;               // forward: loss [R]; backward: grad [R, K]
    long long R;
    int K;
    float t1, t2, smoothing;
    int iters;
};

__device__ __forceinline__ float bt_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float bt_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
// u^e for u >= 0 with the limits torch.pow gives at u = 0 (0 for e > 0, 1 for e = 0, inf for e < 0)
__device__ __forceinline__ float bt_pow0(float u, float e) { return u > 0.0f ? bt_pow(u, e) : (e > 0.0f ? 0.0f : (e == 0.0f ? 1.0f : INFINITY)); }

template <bool BWD>
__global__ __launch_bounds__(256) void bitempered_rows_kernel(const BtRowArgs a) {
    const int lane = threadIdx.x & 63, wave = wave_id();
    const int K = a.K;
    const float t1 = a.t1, t2 = a.t2;
    for (long long row = (long long)blockIdx.x * 4 + wave; row < a.R; row += (long long)gridDim.x * 4) {
        const float* x = a.act + row * K;
        const float* y = a.onehot + row * K;
        float mu = -INFINITY;
        for (int k = lane; k < K; k += 64) mu = fmaxf(mu, x[k]);
        mu = bt_wave_max(mu);
        float norm;   // normalisation of the SHIFTED activations: p_k = exp_t(x_k - mu - norm)
        if (t2 == 1.0f) {
            float z = 0.f;
            for (int k = lane; k < K; k += 64) z += fexp(x[k] - mu);
            norm = flog(bt_wave_sum(z));
        } else if (t2 > 1.0f) {
            float sc = 1.0f, z = 0.f;
            for (int it = 0; it <= a.iters; ++it) {
                z = 0.f;
                for (int k = lane; k < K; k += 64) z += bt_exp_t((x[k] - mu) * sc, t2);
                z = bt_wave_sum(z);
                if (it < a.iters) sc = bt_pow(z, 1.0f - t2);
            }
            norm = -bt_log_t(1.0f / z, t2);
        } else {
            const float edge = -1.0f / (1.0f - t2);
            float dim = 0.f;
            for (int k = lane; k < K; k += 64) dim += (x[k] - mu > edge) ? 1.0f : 0.0f;
            dim = bt_wave_sum(dim);
            float lo = 0.0f, hi = -bt_log_t(1.0f / dim, t2);
            for (int it = 0; it < a.iters; ++it) {
                const float mid = (hi + lo) * 0.5f;
                float mass = 0.f;
                for (int k = lane; k < K; k += 64) mass += bt_exp_t(x[k] - mu - mid, t2);
                mass = bt_wave_sum(mass);
                if (mass < 1.0f) hi = mid; else lo = mid;
            }
            norm = (hi + lo) * 0.5f;
        }
        const float sm_scale = a.smoothing > 0.0f ? 1.0f - a.smoothing * (float)K / (float)(K - 1) : 1.0f;
        const float sm_add = a.smoothing > 0.0f ? a.smoothing / (float)(K - 1) : 0.0f;
        const float e = 2.0f - t1;
        if (!BWD) {
            float l = 0.f;
            for (int k = lane; k < K; k += 64) {
                const float yk = sm_scale * y[k] + sm_add;
                const float p = t2 == 1.0f ? fexp(x[k] - mu - norm) : bt_exp_t(x[k] - mu - norm, t2);
                l += yk * bt_log_t(yk + 1e-10f, t1) - yk * bt_log_t(p, t1) - bt_pow0(yk, e) / e + bt_pow0(p, e) / e;   // :159-165
            }
            l = bt_wave_sum(l);
            if (lane == 0) a.out[row] = l;
        } else {
            // dL/dp_k = -y_k p_k^(-t1) + p_k^(1 - t1);  dp_k/dx_j = p_k^t2 (delta_kj - escort_j), escort = p^t2 / sum p^t2
            float dot = 0.f, S = 0.f;
            for (int k = lane; k < K; k += 64) {
                const float yk = sm_scale * y[k] + sm_add;
                const float p = t2 == 1.0f ? fexp(x[k] - mu - norm) : bt_exp_t(x[k] - mu - norm, t2);
                const float q = bt_pow0(p, t2);
                const float gk = -yk * bt_pow0(p, -t1) + bt_pow0(p, 1.0f - t1);
                dot += p > 0.0f ? gk * q : 0.0f;     // (a class outside the finite support passes no gradient, see the binary kernel)
                S += q;
            }
            dot = bt_wave_sum(dot); S = bt_wave_sum(S);
            const float gl = a.grad_loss[row], r = dot / S;
            for (int k = lane; k < K; k += 64) {
                const float yk = sm_scale * y[k] + sm_add;
                const float p = t2 == 1.0f ? fexp(x[k] - mu - norm) : bt_exp_t(x[k] - mu - norm, t2);
                const float q = bt_pow0(p, t2);
                const float gk = -yk * bt_pow0(p, -t1) + bt_pow0(p, 1.0f - t1);
                a.out[row * K + k] = gl * ((p > 0.0f ? gk * q : 0.0f) - r * q);
            }
        }
    }
}

static unsigned pw_grid(long long groups) {
    long long want = (groups + 255) / 256;
    const long long cap = g_loss_grid_cap > 0 ? g_loss_grid_cap : 8192;
    if (want > cap) want = cap;
    return (unsigned)(want > 0 ? want : 1);
}

static bool pw_vec(const PwArgs& a, const float* extra) {
    if (g_force_scalar || a.n % 4 != 0 || !aligned16(a.x) || !aligned16(a.t)) return false;
    if (a.out && !aligned16(a.out)) return false;
    if (extra && !aligned16(extra)) return false;
    if ((a.chan_w || a.chan_pw) && a.HW % 4 != 0) return false;
    return true;
}

static int fill_pw(PwArgs& a, int kind, const float* x, const float* t, const float* chan_w, const float* chan_pw, int64_t n, int C,
                   int64_t HW, int flags, float p0, float p1, float p2, float ignore_value) {
    if (!x || !t || n < 0 || kind < 0 || kind >= PW_KINDS) return PTB_EINVAL;
    if ((chan_w || chan_pw) && (C < 1 || HW < 1 || kind != PW_SOFT_BCE)) return PTB_EINVAL;
    a.x = x; a.t = t; a.chan_w = chan_w; a.chan_pw = chan_pw;
    a.n = n; a.HW = HW > 0 ? HW : 1; a.C = C > 0 ? C : 1;
    a.flags = flags; a.p0 = p0; a.p1 = p1; a.p2 = p2; a.ignore_value = ignore_value;
    return PTB_OK;
}

static int pw_zero(double* sums, int* error_flag, hipStream_t s) {   // slot sums (and the label flag) start from zero
    hipError_t e = hipMemsetAsync(sums, 0, (size_t)PW_SLOTS * 4 * sizeof(double), s);
    if (e == hipSuccess && error_flag) e = hipMemsetAsync(error_flag, 0, sizeof(int), s);
    if (e != hipSuccess) { set_hip_error(e); return PTB_ELAUNCH; }
    return PTB_OK;
}

}  // namespace ptb

using namespace ptb;

#define PTB_PW_DISPATCH(KERNEL, VECFLAG, ...)                                                       \
    switch (kind) {                                                                                 \
        case PW_SOFT_BCE: if (VECFLAG) KERNEL(PW_SOFT_BCE, true, __VA_ARGS__); else KERNEL(PW_SOFT_BCE, false, __VA_ARGS__); break;             \
        case PW_BALANCED_BCE: if (VECFLAG) KERNEL(PW_BALANCED_BCE, true, __VA_ARGS__); else KERNEL(PW_BALANCED_BCE, false, __VA_ARGS__); break; \
        case PW_QFL: if (VECFLAG) KERNEL(PW_QFL, true, __VA_ARGS__); else KERNEL(PW_QFL, false, __VA_ARGS__); break;                           \
        case PW_WING: if (VECFLAG) KERNEL(PW_WING, true, __VA_ARGS__); else KERNEL(PW_WING, false, __VA_ARGS__); break;                       \
        case PW_SOFT_F1: if (VECFLAG) KERNEL(PW_SOFT_F1, true, __VA_ARGS__); else KERNEL(PW_SOFT_F1, false, __VA_ARGS__); break;              \
        default: if (VECFLAG) KERNEL(PW_LOGCOSH, true, __VA_ARGS__); else KERNEL(PW_LOGCOSH, false, __VA_ARGS__); break;                      \
    }

extern "C" int ptb_pointwise_loss_fwd(int kind, const float* x, const float* t, const float* chan_w, const float* chan_pw, double* sums,
                                      float* elem_out, int64_t n, int C, int64_t HW, int flags, float p0, float p1, float p2,
                                      float ignore_value, ptb_stream_t stream) {
    PwArgs a{};
    if (int rc = fill_pw(a, kind, x, t, chan_w, chan_pw, n, C, HW, flags, p0, p1, p2, ignore_value)) return rc;
    if (!sums) return PTB_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (int rc = pw_zero(sums, nullptr, s)) return rc;
    if (n == 0) return PTB_OK;
    a.sums = sums; a.out = elem_out;
    const bool vec = pw_vec(a, nullptr);
    const dim3 grid(pw_grid(vec ? n / 4 : n)), block(256);
#define PTB_PW_FWD(K, V, dummy) hipLaunchKernelGGL((pw_fwd_kernel<K, V>), grid, block, 0, s, a)
    PTB_PW_DISPATCH(PTB_PW_FWD, vec, 0)
#undef PTB_PW_FWD
    return check_launch();
}

extern "C" int ptb_pointwise_loss_apply(int kind, int emit_loss, const float* x, const float* t, const float* chan_w, const float* chan_pw,
                                        const float* coef, const float* grad_elem, float* out, int64_t n, int C, int64_t HW, int flags,
                                        float p0, float p1, float p2, float ignore_value, ptb_stream_t stream) {
    PwArgs a{};
    if (int rc = fill_pw(a, kind, x, t, chan_w, chan_pw, n, C, HW, flags, p0, p1, p2, ignore_value)) return rc;
    if (!coef || !out) return PTB_EINVAL;
    if (emit_loss && kind != PW_BALANCED_BCE) return PTB_EINVAL;
    if (n == 0) return PTB_OK;
    a.out = out;
    hipStream_t s = (hipStream_t)stream;
    const bool vec = pw_vec(a, grad_elem);
    const dim3 grid(pw_grid(vec ? n / 4 : n)), block(256);
    if (emit_loss) {
        if (vec) hipLaunchKernelGGL((pw_apply_kernel<PW_BALANCED_BCE, true, true>), grid, block, 0, s, a, coef, grad_elem);
        else hipLaunchKernelGGL((pw_apply_kernel<PW_BALANCED_BCE, false, true>), grid, block, 0, s, a, coef, grad_elem);
        return check_launch();
    }
#define PTB_PW_BWD(K, V, dummy) hipLaunchKernelGGL((pw_apply_kernel<K, V, false>), grid, block, 0, s, a, coef, grad_elem)
    PTB_PW_DISPATCH(PTB_PW_BWD, vec, 0)
#undef PTB_PW_BWD
    return check_launch();
}

static int launch_sce(const SceArgs& a, int mode, const float* coef, const float* grad_pix, float* grad, hipStream_t s) {
    const bool vec = !g_force_scalar && a.HW % 4 == 0 && aligned16(a.x) && aligned16(a.labels) && (!a.pix_out || aligned16(a.pix_out)) &&
                     (!grad || aligned16(grad)) && (!grad_pix || aligned16(grad_pix));
#define PTB_SCE(PIX, CREG)                                                                                              \
    do {                                                                                                                \
        const dim3 grid(pw_grid((long long)a.B * ((a.HW + PIX - 1) / PIX))), block(256);                                \
        if (mode == 0) hipLaunchKernelGGL((soft_ce_kernel<PIX, CREG, 0>), grid, block, 0, s, a, coef, grad_pix, grad);  \
        else hipLaunchKernelGGL((soft_ce_kernel<PIX, CREG, 1>), grid, block, 0, s, a, coef, grad_pix, grad);            \
    } while (0)
    if (a.C <= 16) {
        if (vec) {
            if (a.C <= 4) PTB_SCE(4, 4);
            else if (a.C <= 8) PTB_SCE(4, 8);
            else PTB_SCE(4, 16);
        } else {
            if (a.C <= 4) PTB_SCE(1, 4);
            else if (a.C <= 8) PTB_SCE(1, 8);
            else PTB_SCE(1, 16);
        }
    } else {
        const dim3 grid(pw_grid((long long)a.B * a.HW)), block(256);
        if (mode == 0) hipLaunchKernelGGL(soft_ce_generic_kernel<0>, grid, block, 0, s, a, coef, grad_pix, grad);
        else hipLaunchKernelGGL(soft_ce_generic_kernel<1>, grid, block, 0, s, a, coef, grad_pix, grad);
    }
#undef PTB_SCE
    return check_launch();
}

extern "C" int ptb_soft_ce_fwd(const float* logits, const int64_t* labels, double* sums, float* pixel_out, int* error_flag, int B, int C,
                               int64_t HW, float eps, int has_ignore, int64_t ignore_label, ptb_stream_t stream) {
    if (!logits || !labels || !sums || B < 0 || C < 1 || HW < 0) return PTB_EINVAL;
    if (int rc = pw_zero(sums, error_flag, (hipStream_t)stream)) return rc;
    if ((long long)B * HW == 0) return PTB_OK;
    SceArgs a{logits, reinterpret_cast<const long long*>(labels), sums, pixel_out, error_flag, B, C, (long long)HW, eps, has_ignore,
              (long long)ignore_label};
    return launch_sce(a, 0, nullptr, nullptr, nullptr, (hipStream_t)stream);
}

extern "C" int ptb_soft_ce_bwd(const float* logits, const int64_t* labels, const float* coef, const float* grad_pix, float* grad, int B,
                               int C, int64_t HW, float eps, int has_ignore, int64_t ignore_label, ptb_stream_t stream) {
    if (!logits || !labels || !coef || !grad || B < 0 || C < 1 || HW < 0) return PTB_EINVAL;
    if ((long long)B * HW == 0) return PTB_OK;
    SceArgs a{logits, reinterpret_cast<const long long*>(labels), nullptr, nullptr, nullptr, B, C, (long long)HW, eps, has_ignore,
              (long long)ignore_label};
    return launch_sce(a, 1, coef, grad_pix, grad, (hipStream_t)stream);
}

extern "C" int ptb_bitempered_binary_fwd(const float* x, const float* t, double* sums, float* elem_out, int64_t n, float t1, float t2,
                                         float smoothing, int iters, int has_ignore, float ignore_value, ptb_stream_t stream) {
    if (!x || !t || !sums || n < 0 || iters < 0 || t1 == 2.0f) return PTB_EINVAL;
    if (int rc = pw_zero(sums, nullptr, (hipStream_t)stream)) return rc;
    if (n == 0) return PTB_OK;
    BtArgs a{x, t, sums, elem_out, (long long)n, t1, t2, smoothing, ignore_value, iters, has_ignore};
    hipLaunchKernelGGL(bitempered_binary_kernel<false>, dim3(pw_grid(n)), dim3(256), 0, (hipStream_t)stream, a, nullptr, nullptr);
    return check_launch();
}

extern "C" int ptb_bitempered_binary_bwd(const float* x, const float* t, const float* coef, const float* grad_elem, float* grad,
                                         int64_t n, float t1, float t2, float smoothing, int iters, int has_ignore, float ignore_value,
                                         ptb_stream_t stream) {
    if (!x || !t || !coef || !grad || n < 0 || iters < 0 || t1 == 2.0f) return PTB_EINVAL;
    if (n == 0) return PTB_OK;
    BtArgs a{x, t, nullptr, grad, (long long)n, t1, t2, smoothing, ignore_value, iters, has_ignore};
    hipLaunchKernelGGL(bitempered_binary_kernel<true>, dim3(pw_grid(n)), dim3(256), 0, (hipStream_t)stream, a, coef, grad_elem);
    return check_launch();
}

extern "C" int ptb_bitempered_rows(const float* activations, const float* onehot, const float* grad_loss, float* out, int64_t R, int K, float t1,
                                   float t2, float smoothing, int iters, int backward, ptb_stream_t stream) {
    if (!activations || !onehot || !out || R < 0 || K < 1 || iters < 0 || t1 == 2.0f || (backward && !grad_loss)) return PTB_EINVAL;
    if (smoothing > 0.0f && K < 2) return PTB_EINVAL;
    if (R == 0) return PTB_OK;
    BtRowArgs a{activations, onehot, grad_loss, out, (long long)R, K, t1, t2, smoothing, iters};
    const long long blocks = (R + 3) / 4;
    const dim3 grid((unsigned)(blocks < 256 * 8 ? blocks : 256 * 8)), block(256);
    if (backward) hipLaunchKernelGGL(bitempered_rows_kernel<true>, grid, block, 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(bitempered_rows_kernel<false>, grid, block, 0, (hipStream_t)stream, a);
    return check_launch();
}
