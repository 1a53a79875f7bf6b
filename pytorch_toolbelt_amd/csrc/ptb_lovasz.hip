// ptb_lovasz.hip -- Lovasz hinge / Lovasz-softmax losses for gfx950 (MI355X).
//
// Reference: losses/lovasz.py:23-34 (_lovasz_grad), :52-72 (_lovasz_hinge_flat), :110-140 (_lovasz_softmax_flat).
// Per class (and per image with per_image=True) the reference sorts the errors, gathers the ground truth, runs two
// cumsums, a division, a first difference and a dot product -- a Python loop over classes with a host sync each
// (`fg.sum() == 0`).  Here every (group, class) pair is one *segment* of a single pipeline:
//   1. error kernel:   key = error (ignored pixels get -inf so they sort last and contribute 0), value = index<<1 | fg
//   2. ONE rocPRIM radix sort, descending, over 64-bit composite keys (segment rank << 32 | order-preserving bits of
//      the error): all classes / images are sorted by a single multi-pass HBM-bound radix sort (rocPRIM's *segmented*
//      sort is built for many small segments and measured 24 ms for 16 segments of 1 M; the composite-key sort is
//      ~40x faster).  The ROCm primitive is used as-is for the sort, everything around it is hand-written
//   3. fused scan kernel: chunked prefix count of fg over the sorted order -> Jaccard gradient grad_k = J_k - J_{k-1}
//      -> sum_k relu(e_k) * grad_k per segment (fp64 atomics), and grad_k scattered back to pixel order for backward
//   4. backward kernel: d(loss)/d(pred) = coef[segment] * grad_at_pixel * d(error)/d(pred)
// No host synchronisation anywhere: class presence (G > 0) is returned as a device array and the mean over present
// classes is [segments]-sized scalar algebra on the caller's side.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "ptb_common.h"

namespace ptb {

enum { LOVASZ_SOFTMAX = 0, LOVASZ_HINGE = 1 };
constexpr int CHUNK = 2048;  // elements per workgroup in the scan kernels (256 threads x 8)

struct LovArgs {
    const float* pred;        // SOFTMAX: probabilities [B, C, HW]; HINGE: logits [B, HW] (C = 1)
    const long long* labels;  // [B, HW] int64 (SOFTMAX) -- or null when flabels is used
    const float* flabels;     // HINGE: [B, HW] float 0/1 labels
    int B, C;
    long long HW;
    int mode, per_image, has_ignore;
    long long ignore_label;
    float ignore_value;
    long long P;  // elements per segment
    int S;        // segments = groups * C
};

__device__ __forceinline__ void locate(const LovArgs& a, int s, long long i, long long& pred_off, long long& lab_off, int& c) {
    const int j = s / a.C;
    c = s % a.C;
    const long long b = a.per_image ? j : i / a.HW;
    const long long px = a.per_image ? i : i - b * a.HW;
    pred_off = (b * a.C + c) * a.HW + px;
    lab_off = b * a.HW + px;
}

// error, foreground bit, validity of element i of segment s
__device__ __forceinline__ void error_of(const LovArgs& a, int s, long long i, float& e, unsigned& fg, bool& valid) {
    long long po, lo;
    int c;
    locate(a, s, i, po, lo, c);
    const float p = a.pred[po];
    if (a.mode == LOVASZ_SOFTMAX) {
        const long long lab = a.labels[lo];
        valid = !(a.has_ignore && lab == a.ignore_label);
        fg = lab == c ? 1u : 0u;
        e = fabsf((float)fg - p);                         // lovasz.py:133
    } else {
        const float y = a.flabels[lo];
        valid = !(a.has_ignore && y == a.ignore_value);
        fg = y != 0.f ? 1u : 0u;
        e = 1.0f - p * (2.0f * y - 1.0f);                 // lovasz.py:65-66
    }
}

__device__ __forceinline__ unsigned ordered_bits(float e) {  // monotone float -> uint map (larger float => larger uint)
    const unsigned u = __float_as_uint(e);
    return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float from_ordered_bits(unsigned u) {
    return __uint_as_float(u ^ ((u >> 31) ? 0x80000000u : 0xFFFFFFFFu));
}

__global__ __launch_bounds__(256) void lovasz_error_kernel(const LovArgs a, unsigned long long* __restrict__ keys, unsigned* __restrict__ vals) {
    const long long n = a.P * a.S;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride) {
        const int s = (int)(t / a.P);
        const long long i = t - (long long)s * a.P;
        float e;
        unsigned fg;
        bool valid;
        error_of(a, s, i, e, fg, valid);
        // descending sort: segment 0 must come first, so it gets the largest segment rank
        keys[t] = ((unsigned long long)(unsigned)(a.S - 1 - s) << 32) | ordered_bits(valid ? e : -INFINITY);
        vals[t] = ((unsigned)i << 1) | (valid ? fg : 0u);
    }
}

// phase a: foreground count of every CHUNK of the sorted order
__global__ __launch_bounds__(256) void lovasz_count_kernel(const unsigned* __restrict__ vals, long long P, int chunks_per_seg,
                                                           unsigned* __restrict__ chunk_count) {
    const int s = blockIdx.x / chunks_per_seg, k = blockIdx.x % chunks_per_seg;
    const long long base = (long long)s * P, i0 = (long long)k * CHUNK;
    unsigned cnt = 0;
    for (int u = threadIdx.x; u < CHUNK; u += 256) {
        const long long i = i0 + u;
        if (i < P) cnt += vals[base + i] & 1u;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    __shared__ unsigned w[4];
    if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) chunk_count[blockIdx.x] = w[0] + w[1] + w[2] + w[3];
}

// phase b: exclusive scan of the chunk counts of each segment (one workgroup per segment), total -> fg_total[s]
__global__ __launch_bounds__(256) void lovasz_chunk_scan_kernel(unsigned* __restrict__ chunk_count, int chunks_per_seg,
                                                                unsigned* __restrict__ fg_total) {
    __shared__ unsigned sh[256];
    __shared__ unsigned carry;
    unsigned* cc = chunk_count + (long long)blockIdx.x * chunks_per_seg;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int k0 = 0; k0 < chunks_per_seg; k0 += 256) {
        const int k = k0 + threadIdx.x;
        const unsigned v = k < chunks_per_seg ? cc[k] : 0u;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (unsigned o = 1; o < 256; o <<= 1) {  // Hillis-Steele inclusive scan in LDS
            const unsigned add = threadIdx.x >= o ? sh[threadIdx.x - o] : 0u;
            __syncthreads();
            sh[threadIdx.x] += add;
            __syncthreads();
        }
        if (k < chunks_per_seg) cc[k] = carry + sh[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 255) carry += sh[255];
        __syncthreads();
    }
    if (threadIdx.x == 0) fg_total[blockIdx.x] = carry;
}

__device__ __forceinline__ float jaccard_at(float G, float k1, float cum) {  // lovasz.py:29-31 at sorted position k (k1 = k+1)
    const float inter = G - cum;
    const float uni = G + (k1 - cum);
    return 1.0f - inter / uni;
}

// phase c: per element of the sorted order: cum fg -> grad_k = J_k - J_{k-1}; accumulate relu(e_k) * grad_k; scatter grad_k
__global__ __launch_bounds__(256) void lovasz_dot_kernel(const unsigned long long* __restrict__ keys, const unsigned* __restrict__ vals, long long P,
                                                         int chunks_per_seg, const unsigned* __restrict__ chunk_off,
                                                         const unsigned* __restrict__ fg_total, double* __restrict__ seg_loss,
                                                         float* __restrict__ grad_at_pixel) {
    const int s = blockIdx.x / chunks_per_seg, k = blockIdx.x % chunks_per_seg;
    const long long base = (long long)s * P, i0 = (long long)k * CHUNK;
    const float G = (float)fg_total[s];
    // each thread owns 8 consecutive sorted positions
    const long long first = i0 + (long long)threadIdx.x * 8;
    unsigned v[8];
    float e[8];
    unsigned local = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const long long i = first + u;
        v[u] = i < P ? vals[base + i] : 0u;
        e[u] = i < P ? from_ordered_bits((unsigned)keys[base + i]) : -INFINITY;
        local += v[u] & 1u;
    }
    // exclusive prefix of `local` across the 256 threads
    __shared__ unsigned wsum[4];
    unsigned incl = local;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    unsigned wave_off = 0;
    for (int w = 0; w < wave; ++w) wave_off += wsum[w];
    unsigned cum = chunk_off[blockIdx.x] + wave_off + incl - local;  // fg count strictly before this thread's first element
    double acc = 0.0;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const long long i = first + u;
        if (i < P) {
            const unsigned fg = v[u] & 1u;
            const float before = (float)cum;
            cum += fg;
            const float kf = (float)(i + 1);
            const float jk = jaccard_at(G, kf, (float)cum);
            const float jprev = i == 0 ? 0.0f : jaccard_at(G, kf - 1.0f, before);
            const float g = jk - jprev;                       // lovasz.py:32-33
            acc += (double)(fmaxf(e[u], 0.0f) * g);           // dot(relu(errors_sorted), grad), lovasz.py:71 / :139
            grad_at_pixel[base + (v[u] >> 1)] = g;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0 && acc != 0.0) atomicAdd(&seg_loss[s], acc);
}

__global__ __launch_bounds__(256) void lovasz_bwd_kernel(const LovArgs a, const float* __restrict__ coef,
                                                         const float* __restrict__ grad_at_pixel, float* __restrict__ grad) {
    const long long n = a.P * a.S;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride) {
        const int s = (int)(t / a.P);
        const long long i = t - (long long)s * a.P;
        long long po, lo;
        int c;
        locate(a, s, i, po, lo, c);
        float e;
        unsigned fg;
        bool valid;
        error_of(a, s, i, e, fg, valid);
        float gx = 0.f;
        if (valid && e > 0.f) {
            const float g = coef[s] * grad_at_pixel[t];
            if (a.mode == LOVASZ_SOFTMAX) {
                const float d = a.pred[po] - (float)fg;      // d|fg - p|/dp = sign(p - fg)
                gx = d > 0.f ? g : (d < 0.f ? -g : 0.f);
            } else {
                gx = -g * (2.0f * a.flabels[lo] - 1.0f);     // d(1 - x*sign)/dx
            }
        }
        grad[po] = gx;
    }
}

static int fill(LovArgs& a, const float* pred, const int64_t* labels, const float* flabels, int B, int C, int64_t HW, int mode,
                int per_image, int has_ignore, int64_t ignore_label, float ignore_value) {
    if (!pred || B < 0 || C < 1 || HW < 0) return PTB_EINVAL;
    if (mode == LOVASZ_SOFTMAX ? !labels : (!flabels || C != 1)) return PTB_EINVAL;
    a.pred = pred; a.labels = (const long long*)labels; a.flabels = flabels;
    a.B = B; a.C = C; a.HW = HW; a.mode = mode; a.per_image = per_image; a.has_ignore = has_ignore;
    a.ignore_label = ignore_label; a.ignore_value = ignore_value;
    a.P = per_image ? HW : (long long)B * HW;
    a.S = (per_image ? B : 1) * C;
    if (a.P >= (1LL << 31)) return PTB_EUNSUPPORTED;
    return PTB_OK;
}

static int blocks_for(long long n) {
    const long long want = (n + 255) / 256;
    return (int)(want < 1 ? 1 : (want < 256 * 16 ? want : 256 * 16));
}

}  // namespace ptb

using namespace ptb;

static int key_bits(int segments) {
    int b = 0;
    while ((1 << b) < segments) ++b;
    return 32 + b;
}

// bytes of rocPRIM temporary storage for sorting `segments` segments of `per_segment` elements
extern "C" int64_t ptb_lovasz_temp_bytes(int64_t per_segment, int segments) {
    size_t bytes = 0;
    const size_t n = (size_t)(per_segment * segments);
    hipError_t e = rocprim::radix_sort_pairs_desc(nullptr, bytes, (unsigned long long*)nullptr, (unsigned long long*)nullptr,
                                                  (unsigned*)nullptr, (unsigned*)nullptr, n, 0, key_bits(segments), (hipStream_t)0);
    if (e != hipSuccess) return -1;
    return (int64_t)bytes;
}

// Workspaces (all device, provided by the caller, n = P*S elements): keys_a, keys_b u64[n]; vals_a, vals_b u32[n];
// chunk u32[S*ceil(P/2048)]; fg_total u32[S]; seg_loss double[S] (zeroed by the caller);
// grad_at_pixel float[n] (kept for backward); temp = ptb_lovasz_temp_bytes bytes.
extern "C" int ptb_lovasz_fwd(const float* pred, const int64_t* labels, const float* flabels, int B, int C, int64_t HW, int mode,
                              int per_image, int has_ignore, int64_t ignore_label, float ignore_value, uint64_t* keys_a, uint64_t* keys_b,
                              unsigned* vals_a, unsigned* vals_b, unsigned* chunk, unsigned* fg_total,
                              double* seg_loss, float* grad_at_pixel, void* temp, int64_t temp_bytes, ptb_stream_t stream) {
    LovArgs a{};
    if (int rc = fill(a, pred, labels, flabels, B, C, HW, mode, per_image, has_ignore, ignore_label, ignore_value)) return rc;
    if (!keys_a || !keys_b || !vals_a || !vals_b || !chunk || !fg_total || !seg_loss || !grad_at_pixel) return PTB_EINVAL;
    const long long n = a.P * a.S;
    if (n == 0) return PTB_OK;
    if (n >= (1LL << 31)) return PTB_EUNSUPPORTED;  // offsets / packed indices are 32-bit
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(lovasz_error_kernel, dim3(blocks_for(n)), dim3(256), 0, s, a, (unsigned long long*)keys_a, vals_a);
    if (int rc = check_launch()) return rc;
    size_t tb = (size_t)temp_bytes;
    hipError_t e = rocprim::radix_sort_pairs_desc(temp, tb, (unsigned long long*)keys_a, (unsigned long long*)keys_b, vals_a, vals_b,
                                                  (size_t)n, 0, key_bits(a.S), s);
    if (e != hipSuccess) { set_hip_error(e); return PTB_ELAUNCH; }
    const int cps = (int)((a.P + CHUNK - 1) / CHUNK);
    const long long total_chunks = (long long)cps * a.S;
    if (total_chunks > 0x7fffffffLL) return PTB_EUNSUPPORTED;
    hipLaunchKernelGGL(lovasz_count_kernel, dim3((unsigned)total_chunks), dim3(256), 0, s, vals_b, a.P, cps, chunk);
    hipLaunchKernelGGL(lovasz_chunk_scan_kernel, dim3(a.S), dim3(256), 0, s, chunk, cps, fg_total);
    hipLaunchKernelGGL(lovasz_dot_kernel, dim3((unsigned)total_chunks), dim3(256), 0, s, (const unsigned long long*)keys_b, vals_b, a.P, cps, chunk, fg_total, seg_loss,
                       grad_at_pixel);
    return check_launch();
}

extern "C" int ptb_lovasz_bwd(const float* pred, const int64_t* labels, const float* flabels, const float* coef,
                              const float* grad_at_pixel, float* grad, int B, int C, int64_t HW, int mode, int per_image, int has_ignore,
                              int64_t ignore_label, float ignore_value, ptb_stream_t stream) {
    LovArgs a{};
    if (int rc = fill(a, pred, labels, flabels, B, C, HW, mode, per_image, has_ignore, ignore_label, ignore_value)) return rc;
    if (!coef || !grad_at_pixel || !grad) return PTB_EINVAL;
    const long long n = a.P * a.S;
    if (n == 0) return PTB_OK;
    hipLaunchKernelGGL(lovasz_bwd_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, a, coef, grad_at_pixel, grad);
    return check_launch();
}
