// ptb_ensemble.hip -- model ensembling (reference inference/ensembling.py:12-123):
//
//   Ensembler.forward            = torch.stack([model_t(x)]) -> _deaugment_averaging(.., reduction)   (:89-123)
//   ApplySigmoidTo.forward       = output.mul(temperature).sigmoid_()                                  (:62-66)
//   ApplySoftmaxTo.forward       = output.mul(temperature).softmax(dim=1)                              (:38-42)
//
// One launch reads the T model outputs in place (no stack copy), applies the wrapper's activation in registers, applies
// the reduction's pre-transform, sums over the models in list order, post-transforms and writes the ensemble once:
// T reads + 1 write per element instead of the reference's (2 passes per activation) + stack (T reads + T writes) +
// reduce (T reads + temporaries).  HBM-bound streaming: 16 B per lane, non-temporal loads, no MFMA.
// The softmax variant keeps the C channel values of 4 pixels in registers (C <= 16) so every logit is read once.
#include "ptb_view_device.h"

namespace ptb {

constexpr int MAX_MODELS = 16;

struct EnsembleArgs {
    const float* in[MAX_MODELS];
    float* out;
    int T;
    int op;          // PTB_RED_*
    float divisor;   // 1 for sum, T otherwise
    int act;         // 0 none, 1 sigmoid(x * temperature), 2 softmax over C of (x * temperature)
    float temperature;
    long long n4;    // float4 per input (act 0 / 1)
    int B, C;        // softmax: inputs are [B, C, HW]
    long long hw4;   // softmax: float4 per channel plane
};

// v_exp_f32 / v_log_f32 / v_rcp_f32 (1 ulp each): probabilities and their logs stay within ~1e-7 absolute of the libm
// results, far inside the 1e-5 parity tolerance; the libm versions made these kernels ALU-bound (2-3 TB/s).
__device__ __forceinline__ float e_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ float e_log(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }
__device__ __forceinline__ float e_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

__device__ __forceinline__ float sigmoid_t(float x, float temperature) {
    // torch: x.mul(t).sigmoid_() -- 1 / (1 + exp(-z)) in fp32
    const float z = __fmul_rn(x, temperature);
    return e_rcp(1.0f + e_exp(-z));
}

// OPK: 0 = linear (sum / mean), 1 = any non-linear reduction (run-time op), 2 = gmean (the common one, branch-free)
template <int OPK>
__device__ __forceinline__ float e_pre(float x, int op) {
    if (OPK == 0) return x;
    if (OPK == 2 || op == PTB_RED_GMEAN) return e_log(x);
    if (op == PTB_RED_HMEAN) return e_rcp(x < kEps ? kEps : x);
    if (op == PTB_RED_HARMONIC1P) return e_rcp(x + 1.0f);
    return red_pre<1>(x, op);
}
template <int OPK>
__device__ __forceinline__ float e_post(float s, int op, float divisor) {
    if (OPK == 0) return red_post<0>(s, op, divisor);
    if (OPK == 2 || op == PTB_RED_GMEAN) return e_exp(s / divisor);
    return red_post<1>(s, op, divisor);
}

template <int OPK>
__device__ __forceinline__ float4 pre4(float4 v, int op) {
    return make_float4(e_pre<OPK>(v.x, op), e_pre<OPK>(v.y, op), e_pre<OPK>(v.z, op), e_pre<OPK>(v.w, op));
}
__device__ __forceinline__ float4 add4(float4 a, float4 b) {
    return make_float4(__fadd_rn(a.x, b.x), __fadd_rn(a.y, b.y), __fadd_rn(a.z, b.z), __fadd_rn(a.w, b.w));
}
template <int OPK>
__device__ __forceinline__ float4 post4(float4 s, int op, float divisor) {
    return make_float4(e_post<OPK>(s.x, op, divisor), e_post<OPK>(s.y, op, divisor), e_post<OPK>(s.z, op, divisor),
                       e_post<OPK>(s.w, op, divisor));
}

// act 0 / 1: pure elementwise over n4 float4 per input
template <int OPK, int ACT>
__global__ __launch_bounds__(256) void ensemble_kernel(const EnsembleArgs a) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < a.n4; i += stride) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int t = 0; t < a.T; ++t) {
            float4 v = ld16<true>(a.in[t] + 4 * i);
            if (ACT == 1) {
                v.x = sigmoid_t(v.x, a.temperature); v.y = sigmoid_t(v.y, a.temperature);
                v.z = sigmoid_t(v.z, a.temperature); v.w = sigmoid_t(v.w, a.temperature);
            }
            v = pre4<OPK>(v, a.op);
            s = t ? add4(s, v) : v;
        }
        out_store4(a.out + 4 * i, post4<OPK>(s, a.op, a.divisor));
    }
}

// scalar tail / unaligned variant of the above (n elements)
template <int ACT>
__global__ __launch_bounds__(256) void ensemble_scalar_kernel(const EnsembleArgs a, long long n) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    const bool nonlinear = a.op >= PTB_RED_GMEAN;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float s = 0.f;
        for (int t = 0; t < a.T; ++t) {
            float v = a.in[t][i];
            if (ACT == 1) v = sigmoid_t(v, a.temperature);
            v = nonlinear ? red_pre<1>(v, a.op) : v;
            s = t ? __fadd_rn(s, v) : v;
        }
        a.out[i] = nonlinear ? red_post<1>(s, a.op, a.divisor) : red_post<0>(s, a.op, a.divisor);
    }
}

// softmax over the channel dim of [B, C, HW] inputs: the C <= CREG channel values of PIX consecutive pixels stay in
// registers, so every logit is read exactly once.  PIX = 4 (16 B loads) up to 8 channels; PIX = 2 for 9..16 channels
// keeps the kernel at ~80 VGPRs (occupancy 6) instead of 167 (occupancy 3).
template <int PIX>
__device__ __forceinline__ void ld_pix(const float* p, float* o) {
    if (PIX == 4) {
        const float4 t = ld16<true>(p);
        o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w;
    } else {
        typedef float v2f __attribute__((ext_vector_type(2)));
        const v2f t = __builtin_nontemporal_load(reinterpret_cast<const v2f*>(p));
        o[0] = t.x; o[1] = t.y;
    }
}
template <int PIX>
__device__ __forceinline__ void st_pix(float* p, const float* v) {
    if (PIX == 4) out_store4(p, make_float4(v[0], v[1], v[2], v[3]));
    else *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1]);
}

template <int OPK, int CREG, int PIX>
__global__ __launch_bounds__(256) void ensemble_softmax_kernel(const EnsembleArgs a) {
    const long long hwp = a.hw4 * (4 / PIX);  // pixel groups per channel plane
    const long long total = (long long)a.B * hwp;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const long long b = i / hwp, p = i - b * hwp;
        const long long plane = hwp * PIX;
        const long long base = b * a.C * plane + p * PIX;  // element offset of channel 0
        float acc[CREG][PIX];
        for (int t = 0; t < a.T; ++t) {
            float x[CREG][PIX];
            float m[PIX], z[PIX];
#pragma unroll
            for (int j = 0; j < PIX; ++j) { m[j] = -INFINITY; z[j] = 0.f; }
            // request every class plane of this model first (only the loads sit behind the `c < C` guards), then reduce:
            // with the arithmetic inside the guarded block each load was waited for before the next one was issued
#pragma unroll
            for (int c = 0; c < CREG; ++c) {
#pragma unroll
                for (int j = 0; j < PIX; ++j) x[c][j] = 0.f;
                if (c < a.C) ld_pix<PIX>(a.in[t] + base + (long long)c * plane, x[c]);
            }
#pragma unroll
            for (int c = 0; c < CREG; ++c) {
                if (c < a.C) {
#pragma unroll
                    for (int j = 0; j < PIX; ++j) {
                        x[c][j] = __fmul_rn(x[c][j], a.temperature);
                        m[j] = fmaxf(m[j], x[c][j]);
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < CREG; ++c) {
                if (c < a.C) {
#pragma unroll
                    for (int j = 0; j < PIX; ++j) {
                        x[c][j] = e_exp(x[c][j] - m[j]);
                        z[j] = __fadd_rn(z[j], x[c][j]);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < PIX; ++j) z[j] = e_rcp(z[j]);
#pragma unroll
            for (int c = 0; c < CREG; ++c) {
                if (c < a.C) {
#pragma unroll
                    for (int j = 0; j < PIX; ++j) {
                        const float v = e_pre<OPK>(x[c][j] * z[j], a.op);
                        acc[c][j] = t ? __fadd_rn(acc[c][j], v) : v;
                    }
                }
            }
        }
#pragma unroll
        for (int c = 0; c < CREG; ++c) {
            if (c < a.C) {
                float o[PIX];
#pragma unroll
                for (int j = 0; j < PIX; ++j) o[j] = e_post<OPK>(acc[c][j], a.op, a.divisor);
                st_pix<PIX>(a.out + base + (long long)c * plane, o);
            }
        }
    }
}

// any C / any HW: one pixel per thread, the per-model softmax statistics (max, sum) are computed first (two passes
// over the channels), then every channel is normalised and reduced.  Reads each logit three times (mostly L2).
__global__ __launch_bounds__(256) void ensemble_softmax_generic_kernel(const EnsembleArgs a, long long HW) {
    const long long total = (long long)a.B * HW;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const bool nonlinear = a.op >= PTB_RED_GMEAN;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const long long b = i / HW, p = i - b * HW;
        const long long base = b * a.C * HW + p;
        float mx[MAX_MODELS], zs[MAX_MODELS];
        for (int t = 0; t < a.T; ++t) {
            float m = -INFINITY;
            for (int c = 0; c < a.C; ++c) m = fmaxf(m, __fmul_rn(a.in[t][base + c * HW], a.temperature));
            float z = 0.f;
            for (int c = 0; c < a.C; ++c) z = __fadd_rn(z, expf(__fmul_rn(a.in[t][base + c * HW], a.temperature) - m));
            mx[t] = m; zs[t] = z;
        }
        for (int c = 0; c < a.C; ++c) {
            float s = 0.f;
            for (int t = 0; t < a.T; ++t) {
                float v = expf(__fmul_rn(a.in[t][base + c * HW], a.temperature) - mx[t]) / zs[t];
                v = nonlinear ? red_pre<1>(v, a.op) : v;
                s = t ? __fadd_rn(s, v) : v;
            }
            a.out[base + c * HW] = nonlinear ? red_post<1>(s, a.op, a.divisor) : red_post<0>(s, a.op, a.divisor);
        }
    }
}

static unsigned grid_for(long long items) {
    const long long want = (items + 255) / 256;
    return (unsigned)(want < 16384 ? (want > 0 ? want : 1) : 16384);
}

}  // namespace ptb

using namespace ptb;

extern "C" int ptb_ensemble_reduce(const float* const* inputs, int T, int reduction, int activation, float temperature, int B,
                                   int C, int64_t HW, float* out, ptb_stream_t stream) {
    if (!inputs || !out || T < 1 || B < 0 || C < 1 || HW < 0) return PTB_EINVAL;
    if (T > MAX_MODELS) return PTB_EUNSUPPORTED;
    if (reduction < PTB_RED_SUM || reduction > PTB_RED_LOG1P) return PTB_EINVAL;
    if (activation < 0 || activation > 2) return PTB_EINVAL;
    const long long n = (long long)B * C * HW;
    if (n == 0) return PTB_OK;
    EnsembleArgs a{};
    bool aligned = aligned16(out);
    for (int t = 0; t < T; ++t) {
        if (!inputs[t]) return PTB_EINVAL;
        a.in[t] = inputs[t];
        aligned = aligned && aligned16(inputs[t]);
    }
    a.out = out; a.T = T; a.op = reduction;
    a.divisor = reduction == PTB_RED_SUM ? 1.0f : (float)T;
    a.act = activation; a.temperature = temperature;
    a.B = B; a.C = C;
    hipStream_t s = (hipStream_t)stream;
    const bool nonlinear = reduction >= PTB_RED_GMEAN;
    if (activation == 2) {
        if (!g_force_scalar && aligned && HW % 4 == 0 && C <= 16) {
            a.hw4 = HW / 4;
            const int pix = C <= 8 ? 4 : 2;
            const dim3 grid(grid_for((long long)B * a.hw4 * (4 / pix))), block(256);
#define PTB_SM(CREG, PIX)                                                                                     \
    do {                                                                                                      \
        if (reduction == PTB_RED_GMEAN) hipLaunchKernelGGL((ensemble_softmax_kernel<2, CREG, PIX>), grid, block, 0, s, a); \
        else if (nonlinear) hipLaunchKernelGGL((ensemble_softmax_kernel<1, CREG, PIX>), grid, block, 0, s, a); \
        else hipLaunchKernelGGL((ensemble_softmax_kernel<0, CREG, PIX>), grid, block, 0, s, a);               \
    } while (0)
            if (C <= 4) PTB_SM(4, 4);
            else if (C <= 8) PTB_SM(8, 4);
            else PTB_SM(16, 2);
#undef PTB_SM
        } else {
            hipLaunchKernelGGL(ensemble_softmax_generic_kernel, dim3(grid_for((long long)B * HW)), dim3(256), 0, s, a, (long long)HW);
        }
        return check_launch();
    }
    if (!g_force_scalar && aligned && n % 4 == 0) {
        a.n4 = n / 4;
        const dim3 grid(grid_for(a.n4)), block(256);
        if (activation == 1) {
            if (reduction == PTB_RED_GMEAN) hipLaunchKernelGGL((ensemble_kernel<2, 1>), grid, block, 0, s, a);
            else if (nonlinear) hipLaunchKernelGGL((ensemble_kernel<1, 1>), grid, block, 0, s, a);
            else hipLaunchKernelGGL((ensemble_kernel<0, 1>), grid, block, 0, s, a);
        } else {
            if (reduction == PTB_RED_GMEAN) hipLaunchKernelGGL((ensemble_kernel<2, 0>), grid, block, 0, s, a);
            else if (nonlinear) hipLaunchKernelGGL((ensemble_kernel<1, 0>), grid, block, 0, s, a);
            else hipLaunchKernelGGL((ensemble_kernel<0, 0>), grid, block, 0, s, a);
        }
    } else {
        const dim3 grid(grid_for(n)), block(256);
        if (activation == 1) hipLaunchKernelGGL(ensemble_scalar_kernel<1>, grid, block, 0, s, a, n);
        else hipLaunchKernelGGL(ensemble_scalar_kernel<0>, grid, block, 0, s, a, n);
    }
    return check_launch();
}
