// ptb_core.hip -- library plumbing (version, error text, tunables) and TileMerger.merge (tiles.py:345-350).
#include <algorithm>
#include <string>

#include "ptb_common.h"

namespace ptb {

static thread_local std::string g_last_error;
int g_chunk_rows = 32;  // 32x64 chunks (512-thread workgroups) measured best on MI355X (profiles/)
int g_force_scalar = 0;
int g_nt_loads = 1;
int g_band_xcd = 0;   // ptb_set_tunable key 10
int g_band_rows = 64; // ptb_set_tunable key 11: 64-row work items (1024-thread workgroups, 256-byte segments for the transposing views too)
                      // measured 0.8-1 % faster than 32 rows on every memory region of the device (2.046 vs 2.058, 2.270 vs 2.290 ms)
int g_ms_tiled = 1;

void set_hip_error(hipError_t e) { g_last_error = hipGetErrorString(e); }
void set_error_text(const char* text) { g_last_error = text ? text : ""; }

// out[c][p] = (image[c][p] [+ extra[c][p] for p < extra_n]) / norm[p]; IEEE division, no eps clamp (uncovered pixels
// give NaN like the reference).  Streaming elementwise: 16 B/lane, grid-stride; the norm float4 is reused across the
// C channels in registers.  Channel strides let a rank merge a row band of a larger accumulator (parallel.py).
template <bool NT>
__global__ __launch_bounds__(256) void merge_div_kernel(const float* __restrict__ image, const float* __restrict__ norm,
                                                        float* __restrict__ out, int C, long long hw4, long long ics4,
                                                        long long ocs4, const float* __restrict__ extra, long long ecs4,
                                                        long long en4) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    const long long stride = (long long)gridDim.x * blockDim.x;
    const v4f* n4 = reinterpret_cast<const v4f*>(norm);
    const v4f* im4 = reinterpret_cast<const v4f*>(image);
    const v4f* ex4 = reinterpret_cast<const v4f*>(extra);
    v4f* o4 = reinterpret_cast<v4f*>(out);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < hw4; i += stride) {
        // everything here is touched exactly once: non-temporal loads (g_nt_loads) keep the streams out of L2 / MALL
        const v4f n = NT ? __builtin_nontemporal_load(&n4[i]) : n4[i];
        for (int c = 0; c < C; ++c) {
            v4f v = NT ? __builtin_nontemporal_load(&im4[c * ics4 + i]) : im4[c * ics4 + i];
            if (i < en4) {
                const v4f e = NT ? __builtin_nontemporal_load(&ex4[c * ecs4 + i]) : ex4[c * ecs4 + i];
                v.x = __fadd_rn(v.x, e.x); v.y = __fadd_rn(v.y, e.y); v.z = __fadd_rn(v.z, e.z); v.w = __fadd_rn(v.w, e.w);
            }
            v4f o;
            o.x = __fdiv_rn(v.x, n.x); o.y = __fdiv_rn(v.y, n.y); o.z = __fdiv_rn(v.z, n.z); o.w = __fdiv_rn(v.w, n.w);
#if PTB_NT_OUT
            __builtin_nontemporal_store(o, &o4[c * ocs4 + i]);
#else
            o4[c * ocs4 + i] = o;
#endif
        }
    }
}

__global__ __launch_bounds__(256) void merge_div_scalar_kernel(const float* __restrict__ image, const float* __restrict__ norm,
                                                               float* __restrict__ out, int C, long long hw, long long ics,
                                                               long long ocs, const float* __restrict__ extra, long long ecs,
                                                               long long en) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += stride) {
        const float n = norm[i];
        for (int c = 0; c < C; ++c) {
            float v = image[c * ics + i];
            if (i < en) v = __fadd_rn(v, extra[c * ecs + i]);
            out[c * ocs + i] = __fdiv_rn(v, n);
        }
    }
}

// merge restricted to the accumulator blocks (64 columns x `rows` rows) whose byte in `mask` is non-zero: the blocks a
// planned accumulation has not finalised itself.  One workgroup per block, 16 B per lane.
__global__ __launch_bounds__(256) void merge_div_masked_kernel(const float* __restrict__ image, const float* __restrict__ norm,
                                                               float* __restrict__ out, int C, int H, int W,
                                                               const uint8_t* __restrict__ mask, int rows) {
    const int nbx = (W + 63) / 64;
    const int bx = blockIdx.x % nbx, by = blockIdx.x / nbx;
    if (!mask[blockIdx.x]) return;
    const long long plane = (long long)H * W;
    const int x0 = bx * 64, y0 = by * rows;
    const int cw = min(64, W - x0), ch = min(rows, H - y0);
    for (int e = threadIdx.x; e < ch * cw; e += blockDim.x) {
        const int r = e / cw, cc = e - r * cw;
        const long long off = (long long)(y0 + r) * W + x0 + cc;
        const float n = norm[off];
        for (int c = 0; c < C; ++c) out[c * plane + off] = __fdiv_rn(image[c * plane + off], n);
    }
}

}  // namespace ptb

using namespace ptb;

extern "C" int ptb_merge_div_masked(const float* image, const float* norm, float* out, int C, int H, int W, const uint8_t* mask,
                                    int rows, ptb_stream_t stream) {
    if (!image || !norm || !out || !mask || C < 1 || H < 1 || W < 1 || rows < 1) return PTB_EINVAL;
    const long long blocks = (long long)((W + 63) / 64) * ((H + rows - 1) / rows);
    if (blocks > 0x7fffffffLL) return PTB_EUNSUPPORTED;
    hipLaunchKernelGGL(merge_div_masked_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, image, norm, out, C, H, W,
                       mask, rows);
    return check_launch();
}

// Diagnostic: what THIS GPU's memory system gives a pure read-only stream (16 B per lane, 8 independent non-temporal loads in
// flight per lane, 8192 workgroups -- the best configuration of tools/bw_probe.hip).  bench.py runs it over the very buffers
// the merge reads and reports the result next to the roofline fraction, because boxes of one pool differ by ~10 %.
__global__ __launch_bounds__(256) void read_probe_kernel(const float* __restrict__ in, float* __restrict__ sink, long long n4) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f* p = reinterpret_cast<const v4f*>(in);
    const long long stride = (long long)gridDim.x * blockDim.x;
    float acc = 0.f;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 7 * stride < n4; i += 8 * stride) {
        v4f v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(&p[i + u * stride]);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    for (; i < n4; i += stride) { const v4f v = __builtin_nontemporal_load(&p[i]); acc += v.x + v.y + v.z + v.w; }
    if (acc == 123.456f) sink[0] = acc;   // never true for real data: keeps the loads alive
}

extern "C" int ptb_read_probe(const void* buf, int64_t bytes, float* sink, ptb_stream_t stream) {
    if (!buf || !sink || bytes < 16 || !aligned16(buf)) return PTB_EINVAL;
    hipLaunchKernelGGL(read_probe_kernel, dim3(8192), dim3(256), 0, (hipStream_t)stream, static_cast<const float*>(buf), sink,
                       (long long)(bytes / 16));
    return check_launch();
}

// The same stream over SEVERAL buffers in ONE launch (bench.py: the per-batch model-output tensors of an image): the pointer
// table travels in the kernel arguments, a persistent grid walks 32 KiB chunks (256 lanes x 8 x 16 B) of the concatenation
// round-robin, so ramp-up and tail are paid once per pass -- like the band kernel's few launches per image, and unlike 46
// separate probe launches of ~45 us each, which under-read the box by 10-15 %.
constexpr int PROBE_BUFS = 64;
struct ProbeTable {
    const float* ptr[PROBE_BUFS];
    long long first_chunk[PROBE_BUFS + 1];   // prefix sums of the buffers' chunk counts
};

__global__ __launch_bounds__(256) void read_probe_multi_kernel(const ProbeTable t, int nbufs, float* __restrict__ sink) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    const long long total = t.first_chunk[nbufs];
    float acc = 0.f;
    int b = 0;
    for (long long chunk = blockIdx.x; chunk < total; chunk += gridDim.x) {
        while (chunk >= t.first_chunk[b + 1]) ++b;           // (chunks only grow: the scan never restarts)
        const v4f* p = reinterpret_cast<const v4f*>(t.ptr[b]) + (chunk - t.first_chunk[b]) * 2048 + threadIdx.x;
        v4f v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(&p[u * 256]);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    if (acc == 123.456f) sink[0] = acc;
}

extern "C" int64_t ptb_read_probe_multi(const void* const* bufs, const int64_t* bytes, int n, float* sink, int workgroups, ptb_stream_t stream) {
    if (!bufs || !bytes || !sink || n < 1 || n > PROBE_BUFS) return PTB_EINVAL;
    ProbeTable t{};
    long long chunks = 0;
    for (int i = 0; i < n; ++i) {
        if (!bufs[i] || bytes[i] < 0 || !aligned16(bufs[i])) return PTB_EINVAL;
        t.ptr[i] = static_cast<const float*>(bufs[i]);
        t.first_chunk[i] = chunks;
        chunks += bytes[i] / 32768;                           // whole chunks only: the tail of a buffer is not read
    }
    t.first_chunk[n] = chunks;
    if (!chunks) return PTB_EINVAL;
    const int grid = (int)std::min<long long>(chunks, workgroups > 0 ? workgroups : 8192);
    hipLaunchKernelGGL(read_probe_multi_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, t, n, sink);
    const int rc = check_launch();
    return rc != PTB_OK ? rc : (int64_t)chunks * 32768;       // bytes the launch reads
}

// out[i] = sum over the slots of slot_sums[s][i] (the loss kernels spread their fp64 atomics over PTB_SUM_SLOTS copies of the
// sums); a raised label-error flag turns every sum into NaN, so a label outside [0, C) can never train silently.
__global__ __launch_bounds__(256) void sums_finalize_kernel(const double* __restrict__ slots, int nslots, int n, double* __restrict__ out,
                                                            const int* __restrict__ flag) {
    const bool bad = flag && *flag;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        double t = 0.0;
        for (int s = 0; s < nslots; ++s) t += slots[(long long)s * n + i];
        out[i] = bad ? (double)__builtin_nanf("") : t;
    }
}

extern "C" int ptb_sums_finalize(const double* slot_sums, int nslots, int n, double* out, const int* error_flag, ptb_stream_t stream) {
    if (!slot_sums || !out || nslots < 1 || n < 1) return PTB_EINVAL;
    hipLaunchKernelGGL(sums_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, slot_sums, nslots, n, out, error_flag);
    return check_launch();
}

extern "C" int ptb_version(void) { return 102; }

extern "C" const char* ptb_last_hip_error(void) { return g_last_error.c_str(); }

extern "C" int ptb_set_tunable(int key, int value) {
    if (key == 0) {
        if (value != 16 && value != 32 && value != 64) return PTB_EINVAL;
        g_chunk_rows = value;
        return PTB_OK;
    }
    if (key == 1) {
        g_force_scalar = value ? 1 : 0;
        return PTB_OK;
    }
    if (key == 2) {
        g_nt_loads = value ? 1 : 0;
        return PTB_OK;
    }
    if (key == 3) {
        g_ms_tiled = value ? 1 : 0;
        return PTB_OK;
    }
    if (key == 4) {
        if (value < 0) return PTB_EINVAL;
        g_loss_grid_cap = value;
        return PTB_OK;
    }
    if (key == 5) {
        g_fused_pix2 = value ? 1 : 0;
        return PTB_OK;
    }
    if (key == 12) {
        if (value < 0 || value > 2) return PTB_EINVAL;
        g_focal_pk = value;
        return PTB_OK;
    }
    if (key == 15) {
        if (value != 64 && value != 128) return PTB_EINVAL;
        g_ms_tile_w = value;
        return PTB_OK;
    }
    if (key == 16) {
        g_nt_grad_stores = value ? 1 : 0;
        return PTB_OK;
    }
    if (key == 17) {
        g_rs_xcd_map = value & 7;       // bit 0: XCD-contiguous tile order; bit 1 (A/B only): histogram loads behind the ranking
        return PTB_OK;
    }
    if (key == 18) {
        g_rank_finish_fused = value ? 1 : 0;
        return PTB_OK;
    }
    if (key == 19) {
        g_lovasz_fused_dot = value ? 1 : 0;
        return PTB_OK;
    }
    if (key == 20) {
        g_stats_pk = value ? 1 : 0;
        return PTB_OK;
    }
    if (key == 21) {
        g_band_half_pf = value < 0 ? 0 : (value > 2 ? 2 : value);
        return PTB_OK;
    }
    if (key == 22) {
        g_band_rot_views = value ? 1 : 0;
        return PTB_OK;
    }
    if (key == 27) {
        g_band_chan_loop = value ? 1 : 0;
        return PTB_OK;
    }
    if (key == 25) {
        g_band_lds_db = value ? 1 : 0;
        return PTB_OK;
    }
    if (key == 23) {
        g_lovasz_rankdot = value ? 1 : 0;
        return PTB_OK;
    }
    if (key == 13) {
        if (value < 1) return PTB_EINVAL;
        g_focal_pk_grid = value;
        return PTB_OK;
    }
    if (key == 6) {
        if (value != 16 && value != 32 && value != 64) return PTB_EINVAL;
        g_ms_tile_rows = value;
        return PTB_OK;
    }
    if (key == 11) {
        if (value != 32 && value != 64) return PTB_EINVAL;
        g_band_rows = value;
        return PTB_OK;
    }
    if (key == 10) {
        if (value < 0 || value > 2) return PTB_EINVAL;
        g_band_xcd = value;
        return PTB_OK;
    }
    if (key == 9) {
        if (value < 0 || value > 64) return PTB_EINVAL;
        g_ms_strip = value;
        return PTB_OK;
    }
    if (key == 8) {
        g_loss_prefetch = value ? 1 : 0;
        return PTB_OK;
    }
    if (key == 7) {
        if (value != 0 && value != 2 && value != 4) return PTB_EINVAL;
        g_smf_bwd_stash = value;
        return PTB_OK;
    }
    return PTB_EINVAL;
}

extern "C" int ptb_merge_div_ex(const float* image, const float* norm, float* out, int C, int64_t HW, int64_t image_cs,
                                int64_t out_cs, const float* extra, int64_t extra_cs, int64_t extra_n, ptb_stream_t stream) {
    if (!image || !norm || !out || C < 1 || HW < 0 || image_cs < HW || out_cs < HW) return PTB_EINVAL;
    if (extra_n < 0 || extra_n > HW || (extra_n > 0 && (!extra || extra_cs < extra_n))) return PTB_EINVAL;
    if (HW == 0) return PTB_OK;
    hipStream_t s = (hipStream_t)stream;
    const bool vec = !g_force_scalar && HW % 4 == 0 && image_cs % 4 == 0 && out_cs % 4 == 0 && aligned16(image) &&
                     aligned16(norm) && aligned16(out) &&
                     (extra_n == 0 || (extra_n % 4 == 0 && extra_cs % 4 == 0 && aligned16(extra)));
    if (vec) {
        const long long hw4 = HW / 4;
        const long long want = (hw4 + 255) / 256;
        const int blocks = (int)(want < 256 * 16 ? want : 256 * 16);
        // (non-temporal loads were measured here too: no gain for this 1:1 read/write stream, so the default policy stays)
        hipLaunchKernelGGL(merge_div_kernel<false>, dim3(blocks), dim3(256), 0, s, image, norm, out, C, hw4, (long long)image_cs / 4,
                                (long long)out_cs / 4, extra, (long long)extra_cs / 4, (long long)extra_n / 4);
    } else {
        const long long want = (HW + 255) / 256;
        const int blocks = (int)(want < 256 * 16 ? want : 256 * 16);
        hipLaunchKernelGGL(merge_div_scalar_kernel, dim3(blocks), dim3(256), 0, s, image, norm, out, C, (long long)HW,
                           (long long)image_cs, (long long)out_cs, extra, (long long)extra_cs, (long long)extra_n);
    }
    return check_launch();
}

// dst[c][r][x] += src[c][r][x] over a [C, rows, cols] rectangle of a larger accumulator (halo rectangles of the multi-GPU
// merger).  One thread per 4 columns when everything is 16-byte aligned, else one per element; grid.y = channel * rows.
template <int VEC>
__global__ __launch_bounds__(256) void rect_add_kernel(float* __restrict__ dst, const float* __restrict__ src, int C, int rows, int cols,
                                                       long long dst_cs, long long dst_rs) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    const int x = (blockIdx.x * 256 + threadIdx.x) * VEC;
    if (x >= cols) return;
    for (long long cr = blockIdx.y; cr < (long long)C * rows; cr += gridDim.y) {
        const int c = (int)(cr / rows), r = (int)(cr - (long long)c * rows);
        float* d = dst + c * dst_cs + r * dst_rs + x;
        const float* s = src + cr * cols + x;
        if (VEC == 4) {
            const v4f b = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(s));
            *reinterpret_cast<v4f*>(d) = *reinterpret_cast<v4f*>(d) + b;
        } else {
            *d += *s;
        }
    }
}

extern "C" int ptb_rect_add(float* dst, const float* src, int C, int rows, int cols, int64_t dst_cs, int64_t dst_rs,
                            ptb_stream_t stream) {
    if (!dst || !src || C < 1 || rows < 0 || cols < 0 || dst_rs < cols || dst_cs < (int64_t)rows * dst_rs) return PTB_EINVAL;
    if (rows == 0 || cols == 0) return PTB_OK;
    const long long cr = (long long)C * rows;
    const int gy = (int)(cr < 65535 ? cr : 65535);
    hipStream_t s = (hipStream_t)stream;
    const bool vec = !g_force_scalar && cols % 4 == 0 && dst_cs % 4 == 0 && dst_rs % 4 == 0 && aligned16(dst) && aligned16(src);
    if (vec)
        hipLaunchKernelGGL(rect_add_kernel<4>, dim3((cols / 4 + 255) / 256, gy), dim3(256), 0, s, dst, src, C, rows, cols,
                           (long long)dst_cs, (long long)dst_rs);
    else
        hipLaunchKernelGGL(rect_add_kernel<1>, dim3((cols + 255) / 256, gy), dim3(256), 0, s, dst, src, C, rows, cols,
                           (long long)dst_cs, (long long)dst_rs);
    return check_launch();
}

extern "C" int ptb_merge_div(const float* image, const float* norm, float* out, int C, int64_t HW, ptb_stream_t stream) {
    return ptb_merge_div_ex(image, norm, out, C, HW, HW, HW, nullptr, 0, 0, stream);
}
