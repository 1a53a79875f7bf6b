// ptb_stack.hip -- reductions over a stack of predictions of ANY length with an explicit eps.
//
// Reference: inference/functional.py:247-331 (geometric_mean, harmonic_mean(eps), harmonic1p_mean, logodd_mean(eps), log1p_mean) and
// inference/tta.py:63-95 (`_deaugment_averaging` over dim 0 of [T, B, ...]): every reduction is post(mean_t(pre(x_t))).  The fused view
// kernels (ptb_views.hip) cover T <= 8 with the default eps = 1e-6 at compile-time unrolling; this file is the general case --
// tencrop TTA (T = 10), ensembles of more than 8 models, harmonic_mean / logodd_mean called with their eps argument -- as ONE pass
// over the T planes (HBM-bound: T reads + 1 write per element), plus its backward (T + 2 reads, T writes).
#include "ptb_common.h"

namespace ptb {
namespace {

__device__ __forceinline__ float s_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ float s_log(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }
__device__ __forceinline__ float s_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4f nt_load4(const float* p) { return __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p)); }

struct Eps { float lo, hi; };   // eps and (float)(1.0 - eps): the clamp bounds torch forms from the Python double

__device__ __forceinline__ float pre(float x, int op, Eps e) {
    switch (op) {
        case PTB_RED_GMEAN: return s_log(x);                                     // functional.py:261
        case PTB_RED_HMEAN: return s_rcp(x < e.lo ? e.lo : x);                   // functional.py:275
        case PTB_RED_HARMONIC1P: return s_rcp(x + 1.0f);                         // functional.py:292
        case PTB_RED_LOGODD: {                                                   // functional.py:311-312
            const float p = x < e.lo ? e.lo : (x > e.hi ? e.hi : x);
            return s_log(p * s_rcp(1.0f - p));
        }
        case PTB_RED_LOG1P: return s_log(1.0f + x);                              // functional.py:330
        default: return x;
    }
}
__device__ __forceinline__ float post(float s, int op, float T, Eps e) {
    if (op == PTB_RED_SUM) return s;
    const float m = s / T;
    switch (op) {
        case PTB_RED_GMEAN: return s_exp(m);
        case PTB_RED_HMEAN: return s_rcp(m < e.lo ? e.lo : m);
        case PTB_RED_HARMONIC1P: return s_rcp(m) - 1.0f;
        case PTB_RED_LOGODD: { const float ex = s_exp(m); return ex * s_rcp(1.0f + ex); }
        case PTB_RED_LOG1P: return s_exp(m) - 1.0f;
        default: return m;
    }
}
// d post / d mean through the forward output, d pre / d x
__device__ __forceinline__ float dpost(float out, int op, Eps e) {
    switch (op) {
        case PTB_RED_GMEAN: return out;
        case PTB_RED_HMEAN: return out >= 1.0f / e.lo ? 0.f : -out * out;        // out = 1 / max(m, eps): the clamp has zero slope
        case PTB_RED_HARMONIC1P: return -(out + 1.0f) * (out + 1.0f);
        case PTB_RED_LOGODD: return out * (1.0f - out);
        case PTB_RED_LOG1P: return out + 1.0f;
        default: return 1.0f;
    }
}
__device__ __forceinline__ float dpre(float x, int op, Eps e) {
    switch (op) {
        case PTB_RED_GMEAN: return 1.0f / x;
        case PTB_RED_HMEAN: return x < e.lo ? 0.f : -1.0f / (x * x);
        case PTB_RED_HARMONIC1P: return -1.0f / ((x + 1.0f) * (x + 1.0f));
        case PTB_RED_LOGODD: return (x < e.lo || x > e.hi) ? 0.f : 1.0f / (x * (1.0f - x));
        case PTB_RED_LOG1P: return 1.0f / (1.0f + x);
        default: return 1.0f;
    }
}

// one lane = 4 consecutive elements (VEC) or 1 (tail / unaligned); planes are walked 4 at a time so 4 x 16 B are in flight per lane
template <bool VEC>
__global__ __launch_bounds__(256) void stack_reduce_kernel(const float* __restrict__ src, int T, long long n, int op, Eps e, float* __restrict__ out) {
    constexpr int W = VEC ? 4 : 1;
    const long long items = (n + W - 1) / W;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < items; i += (long long)gridDim.x * 256) {
        const long long off = i * W;
        float s[W];
#pragma unroll
        for (int k = 0; k < W; ++k) s[k] = 0.f;
        int t = 0;
        for (; t + 4 <= T; t += 4) {
            float v[4][W];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float* p = src + (long long)(t + u) * n + off;
                if constexpr (VEC) { const v4f q = nt_load4(p); v[u][0] = q.x; v[u][1 % W] = q.y; v[u][2 % W] = q.z; v[u][3 % W] = q.w; }
                else v[u][0] = p[0];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int k = 0; k < W; ++k) s[k] += pre(v[u][k], op, e);      // sequential over t, like torch.mean / sum over dim 0
        }
        for (; t < T; ++t) {
            const float* p = src + (long long)t * n + off;
            float v[W];
            if constexpr (VEC) { const v4f q = nt_load4(p); v[0] = q.x; v[1 % W] = q.y; v[2 % W] = q.z; v[3 % W] = q.w; }
            else v[0] = p[0];
#pragma unroll
            for (int k = 0; k < W; ++k) s[k] += pre(v[k], op, e);
        }
        float r[W];
#pragma unroll
        for (int k = 0; k < W; ++k) r[k] = post(s[k], op, (float)T, e);
        if constexpr (VEC) out_store4(out + off, make_float4(r[0], r[1 % W], r[2 % W], r[3 % W]));
        else out[off] = r[0];
    }
}

// grad_t = g * dpost(out) * dpre(x_t) / T   (sum: g; mean: g / T)
template <bool VEC>
__global__ __launch_bounds__(256) void stack_reduce_bwd_kernel(const float* __restrict__ src, const float* __restrict__ out, const float* __restrict__ gout,
                                                               int T, long long n, int op, Eps e, float* __restrict__ grad) {
    constexpr int W = VEC ? 4 : 1;
    const long long items = (n + W - 1) / W;
    const float invT = op == PTB_RED_SUM ? 1.0f : 1.0f / (float)T;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < items; i += (long long)gridDim.x * 256) {
        const long long off = i * W;
        float c[W];
        if constexpr (VEC) {
            const float4 g = *reinterpret_cast<const float4*>(gout + off), o = *reinterpret_cast<const float4*>(out + off);
            c[0] = g.x * dpost(o.x, op, e) * invT; c[1 % W] = g.y * dpost(o.y, op, e) * invT;
            c[2 % W] = g.z * dpost(o.z, op, e) * invT; c[3 % W] = g.w * dpost(o.w, op, e) * invT;
        } else {
            c[0] = gout[off] * dpost(out[off], op, e) * invT;
        }
        for (int t = 0; t < T; ++t) {
            const long long q = (long long)t * n + off;
            if constexpr (VEC) {
                const v4f x = nt_load4(src + q);
                *reinterpret_cast<float4*>(grad + q) = make_float4(c[0] * dpre(x.x, op, e), c[1 % W] * dpre(x.y, op, e), c[2 % W] * dpre(x.z, op, e), c[3 % W] * dpre(x.w, op, e));
            } else {
                grad[q] = c[0] * dpre(src[q], op, e);
            }
        }
    }
}

inline int grid_for(long long items) {
    const long long blocks = (items + 255) / 256;
    return (int)(blocks < 256 * 16 ? (blocks < 1 ? 1 : blocks) : 256 * 16);
}

}  // namespace
}  // namespace ptb

using namespace ptb;

extern "C" int ptb_stack_reduce(const float* src, int T, int64_t n, int reduction, double eps, float* out, ptb_stream_t stream) {
    if (!src || !out || T < 1 || n < 0 || reduction < PTB_RED_SUM || reduction > PTB_RED_LOG1P || !(eps >= 0.0)) return PTB_EINVAL;
    if (n == 0) return PTB_OK;
    const Eps e{(float)eps, (float)(1.0 - eps)};
    hipStream_t s = (hipStream_t)stream;
    if (!g_force_scalar && n % 4 == 0 && aligned16(src) && aligned16(out))
        hipLaunchKernelGGL((stack_reduce_kernel<true>), dim3(grid_for(n / 4)), dim3(256), 0, s, src, T, (long long)n, reduction, e, out);
    else
        hipLaunchKernelGGL((stack_reduce_kernel<false>), dim3(grid_for(n)), dim3(256), 0, s, src, T, (long long)n, reduction, e, out);
    return check_launch();
}

extern "C" int ptb_stack_reduce_bwd(const float* src, const float* out, const float* grad_out, int T, int64_t n, int reduction, double eps,
                                    float* grad, ptb_stream_t stream) {
    if (!src || !out || !grad_out || !grad || T < 1 || n < 0 || reduction < PTB_RED_SUM || reduction > PTB_RED_LOG1P || !(eps >= 0.0)) return PTB_EINVAL;
    if (n == 0) return PTB_OK;
    const Eps e{(float)eps, (float)(1.0 - eps)};
    hipStream_t s = (hipStream_t)stream;
    if (!g_force_scalar && n % 4 == 0 && aligned16(src) && aligned16(out) && aligned16(grad_out) && aligned16(grad))
        hipLaunchKernelGGL((stack_reduce_bwd_kernel<true>), dim3(grid_for(n / 4)), dim3(256), 0, s, src, out, grad_out, T, (long long)n, reduction, e, grad);
    else
        hipLaunchKernelGGL((stack_reduce_bwd_kernel<false>), dim3(grid_for(n)), dim3(256), 0, s, src, out, grad_out, T, (long long)n, reduction, e, grad);
    return check_launch();
}
