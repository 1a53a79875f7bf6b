// ptb_resample.hip -- bilinear resize for multiscale TTA (reference inference/tta.py:599-621, 645-689, which call
// torch.nn.functional.interpolate(mode="bilinear")).  4-tap gather, HBM/L2-bound; index and weight arithmetic
// follows ATen's area_pixel_compute_source_index / compute_source_index_and_lambda in fp32 so results match the
// reference's torch kernels to rounding.
#include "ptb_common.h"

namespace ptb {

struct Taps { int i0, i1; float l0, l1; };

__device__ __forceinline__ Taps taps(int dst, float scale, int n_in, bool align_corners) {
    float src;
    if (align_corners) {
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
            *reinterpret_cast<float4*>(o) = make_float4(res[0], res[1], res[2], res[3]);
        } else {
#pragma unroll
            for (int m = 0; m < 4; ++m) if (4 * q + m < wout) o[m] = res[m];
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
