// ptb_edges.hip -- the two ends of the tiled-inference loop, on the device:
//
//   * ptb_split_tiles_u8: ImageSlicer.split (tiles.py:177-204) + image_to_tensor (utils/torch_utils.py:204-231,
//     HWC -> CHW) + .float() [+ per-channel affine] [+ *_image_augment, tta.py:257-284,319-341,385-422,470-484] from a
//     device-resident uint8 HWC image straight into the chunk-major fp32 batch [V*B, C, th, tw] the model consumes.
//     The reference pads the whole image on the host, materialises 361 tile views, converts each to CHW, stacks, casts
//     and uploads 1.14 GB of fp32; here 75 MB of uint8 go up once and each output element is written exactly once.
//   * ptb_merge_crop: TileMerger.merge (tiles.py:345-346) + CHW -> HWC (np.moveaxis) + .astype(uint8) (truncating,
//     README.md:225) or argmax over channels + ImageSlicer.crop_to_orignal_size (tiles.py:271-280) in one pass that
//     writes only the cropped window (25-100 MB to download instead of the 419 MB padded fp32 map).
//
// Both are HBM-bound streaming kernels (no MFMA).  The split kernel is write-bound (V*4 output bytes per input byte)
// and reuses the augment scatter of the view kernels: one 64 x CH pixel chunk per workgroup, 16 B stores per lane,
// transposing views through the XOR-swizzled LDS tile.
#include "ptb_view_device.h"

namespace ptb {

constexpr int MAX_SPLIT_C = 16;

struct SplitArgs {
    const uint8_t* img;  // [IH, IW, IC] uint8, contiguous
    int IH, IW, IC;
    int b0;              // batch index of the first tile of this launch group
    float pad;           // border value (already a uint8 value, as float)
    int affine;          // 0: out = float(u8); 1: out = float(u8) * scale[c] + bias[c] (two roundings, like torch)
    float scale[MAX_SPLIT_C], bias[MAX_SPLIT_C];
    int tx[MAX_GROUP], ty[MAX_GROUP];  // tile origins in image coordinates; tiles may hang over any border
};

__device__ __forceinline__ float split_pixel(const SplitArgs& g, int gy, int gx, int c) {
    float f = g.pad;
    if (gy >= 0 && gy < g.IH && gx >= 0 && gx < g.IW) f = (float)g.img[((long long)gy * g.IW + gx) * g.IC + c];
    if (g.affine) f = __fadd_rn(__fmul_rn(f, g.scale[c]), g.bias[c]);
    return f;
}

template <int CH>
__global__ __launch_bounds__(CH * 16) void edge_split_kernel(const ViewArgs a, const SplitArgs g, int B) {
    __shared__ __attribute__((aligned(16))) float st[CW * CH];
    const int tid = threadIdx.x;
    const int cpt = a.chunks_x * a.chunks_y;
    int bid = blockIdx.x;
    const int chunk = bid % cpt;
    bid /= cpt;
    const int c = bid % a.C;
    const int lb = bid / a.C;  // tile within this launch group
    const int x0 = (chunk % a.chunks_x) * CW, y0 = (chunk / a.chunks_x) * CH;
    const int cw = min(CW, a.W - x0), ch = min(CH, a.H - y0);
    const int q = tid & 15, r = tid >> 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < ch && 4 * q < cw) {
        const int gy = g.ty[lb] + y0 + r, gx = g.tx[lb] + x0 + 4 * q;
        v.x = split_pixel(g, gy, gx, c);
        v.y = split_pixel(g, gy, gx + 1, c);
        v.z = split_pixel(g, gy, gx + 2, c);
        v.w = split_pixel(g, gy, gx + 3, c);
    }
    scatter_chunk<CH, false>(a, B, g.b0 + lb, c, x0, y0, cw, ch, v, st, tid);
}

// any tile shape: one output element per thread (grid-stride over this group's V * n * C * th * tw elements)
__global__ __launch_bounds__(256) void edge_split_scalar_kernel(const ViewArgs a, const SplitArgs g, int B, int n) {
    const long long plane = (long long)a.H * a.W;
    const long long per_view = (long long)n * a.C * plane;
    const long long total = per_view * a.nviews;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        const int k = (int)(t / per_view);
        long long rem = t - (long long)k * per_view;
        const int lb = (int)(rem / (a.C * plane));
        rem -= (long long)lb * a.C * plane;
        const int c = (int)(rem / plane);
        const long long px = rem - (long long)c * plane;
        const int i = (int)(px / a.W), j = (int)(px - (long long)i * a.W);
        const int code = (a.codes >> (3 * k)) & 7;
        int R, Cc;  // source (tile-local) pixel of output (i, j) of view k
        if (code & 1) { R = (code & 2) ? a.H - 1 - j : j; Cc = (code & 4) ? a.W - 1 - i : i; }
        else { R = (code & 2) ? a.H - 1 - i : i; Cc = (code & 4) ? a.W - 1 - j : j; }
        const float f = split_pixel(g, g.ty[lb] + R, g.tx[lb] + Cc, c);
        a.dst[(((long long)k * B + g.b0 + lb) * a.C + c) * plane + px] = f;
    }
}

// ------------------------------------------------------------------------------------------------ merge + crop
enum { OUT_F32 = 0, OUT_U8 = 1, OUT_ARGMAX_U8 = 2, OUT_ARGMAX_I64 = 3 };

struct CropArgs {
    const float* image;  // [C, H, W] accumulator
    const float* norm;   // [H, W]
    void* out;
    int C, H, W;
    int top, left, OH, OW;
    int layout;  // 0: [C, OH, OW]   1: [OH, OW, C]
    int kind;    // OUT_*
};

// numpy / torch float -> uint8 cast as x86-64 performs it: truncate toward zero to int32, keep the low byte
// (values outside the int32 range, NaN and infinities give 0).  In [0, 256) this is the plain truncation of
// ImageSlicer.merge / README.md:225 (quirk Q6).
__device__ __forceinline__ uint8_t cast_u8(float v) {
    if (!(fabsf(v) < 2147483648.0f)) return 0;
    return (uint8_t)((int)v & 255);
}

// 4 consecutive source floats of one row; `vec` (uniform): the window is 16 B aligned in the accumulator
__device__ __forceinline__ void load_px4(const float* p, int nv, bool vec, float* o) {
    if (!p) { o[0] = o[1] = o[2] = o[3] = 1.0f; return; }  // norm == NULL: the image is already normalised
    if (vec && nv == 4) {
        const float4 t = ld16<true>(p);
        o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w;
    } else {
        for (int m = 0; m < 4; ++m) o[m] = m < nv ? p[m] : 1.0f;
    }
}

__device__ __forceinline__ void store_f32x4(float* p, const float* v, int nv) {
    if (nv == 4 && (reinterpret_cast<uintptr_t>(p) & 15u) == 0) out_store4(p, make_float4(v[0], v[1], v[2], v[3]));
    else for (int m = 0; m < nv; ++m) p[m] = v[m];
}
__device__ __forceinline__ void store_u8x4(uint8_t* p, const uint8_t* v, int nv) {
    if (nv == 4 && (reinterpret_cast<uintptr_t>(p) & 3u) == 0)
        *reinterpret_cast<uint32_t*>(p) = (uint32_t)v[0] | ((uint32_t)v[1] << 8) | ((uint32_t)v[2] << 16) | ((uint32_t)v[3] << 24);
    else for (int m = 0; m < nv; ++m) p[m] = v[m];
}

// Channel-planar outputs and argmax: one pass over the channels with running state, any C.
__global__ __launch_bounds__(256) void merge_crop_planar_kernel(const CropArgs a, bool vec) {
    const int groups_x = (a.OW + 3) / 4;
    const long long total = (long long)a.OH * groups_x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long iplane = (long long)a.H * a.W, oplane = (long long)a.OH * a.OW;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        const int y = (int)(t / groups_x), x = (int)(t - (long long)y * groups_x) * 4;
        const int nv = min(4, a.OW - x);
        const long long src = (long long)(y + a.top) * a.W + a.left + x;
        const long long dpx = (long long)y * a.OW + x;
        float n[4], v[4];
        float best[4] = {0.f, 0.f, 0.f, 0.f};
        int arg[4] = {0, 0, 0, 0};
        load_px4(a.norm ? a.norm + src : nullptr, nv, vec, n);
        for (int c = 0; c < a.C; ++c) {
            load_px4(a.image + c * iplane + src, nv, vec, v);
            if (a.norm) {
#pragma unroll
                for (int m = 0; m < 4; ++m) v[m] = __fdiv_rn(v[m], n[m]);  // tiles.py:346: no eps clamp
            }
            if (a.kind >= OUT_ARGMAX_U8) {
#pragma unroll
                for (int m = 0; m < 4; ++m) {  // first maximum wins; NaN counts as the maximum (numpy / torch argmax)
                    const bool take = c == 0 ? true : (v[m] > best[m] || (v[m] != v[m] && best[m] == best[m]));
                    best[m] = take ? v[m] : best[m];
                    arg[m] = take ? c : arg[m];
                }
            } else if (a.kind == OUT_U8) {
                uint8_t b[4];
#pragma unroll
                for (int m = 0; m < 4; ++m) b[m] = cast_u8(v[m]);
                store_u8x4(static_cast<uint8_t*>(a.out) + c * oplane + dpx, b, nv);
            } else {
                store_f32x4(static_cast<float*>(a.out) + c * oplane + dpx, v, nv);
            }
        }
        if (a.kind == OUT_ARGMAX_U8) {
            uint8_t b[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) b[m] = (uint8_t)arg[m];
            store_u8x4(static_cast<uint8_t*>(a.out) + dpx, b, nv);
        } else if (a.kind == OUT_ARGMAX_I64) {
            long long* o = static_cast<long long*>(a.out) + dpx;
            for (int m = 0; m < nv; ++m) o[m] = arg[m];
        }
    }
}

// Channel-last outputs with the CT <= 4 channels of 4 pixels held in registers: the thread's 4*CT output elements are
// contiguous, so they leave as CT 16-byte (fp32) or CT 4-byte (uint8) stores.
// fp32 output with OW % 4 == 0 (`repack`): thread t's run starts at element 4 * CT * t of the output, i.e. the 256 threads of a
// workgroup own 256 * CT consecutive float4 -- but a lane's own CT float4 are adjacent, so storing them directly makes every
// store instruction hit 64 lanes x 16 B at a stride of 16 * CT B (measured 3.2 TB/s).  The runs are exchanged through LDS
// instead (lane writes float4 CT * tid + g, reads float4 256 * g + tid), so each store instruction covers 1 KiB contiguous.
template <int CT>
__global__ __launch_bounds__(256) void merge_crop_hwc_kernel(const CropArgs a, bool vec, bool repack) {
    __shared__ float4 xchg[256 * CT];
    const int groups_x = (a.OW + 3) / 4;
    const long long total = (long long)a.OH * groups_x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long iplane = (long long)a.H * a.W;
    if (repack) {
        for (long long t0 = (long long)blockIdx.x * blockDim.x; t0 < total; t0 += stride) {   // workgroup-uniform trip count
            const long long t = t0 + threadIdx.x;
            if (t < total) {
                const int y = (int)(t / groups_x), x = (int)(t - (long long)y * groups_x) * 4;
                const long long src = (long long)(y + a.top) * a.W + a.left + x;
                float n[4], v[CT][4];
                load_px4(a.norm ? a.norm + src : nullptr, 4, vec, n);
#pragma unroll
                for (int c = 0; c < CT; ++c) {
                    load_px4(a.image + c * iplane + src, 4, vec, v[c]);
                    if (a.norm) {
#pragma unroll
                        for (int m = 0; m < 4; ++m) v[c][m] = __fdiv_rn(v[c][m], n[m]);
                    }
                }
#pragma unroll
                for (int g = 0; g < CT; ++g)
                    xchg[CT * threadIdx.x + g] = make_float4(v[(4 * g) % CT][(4 * g) / CT], v[(4 * g + 1) % CT][(4 * g + 1) / CT],
                                                             v[(4 * g + 2) % CT][(4 * g + 2) / CT], v[(4 * g + 3) % CT][(4 * g + 3) / CT]);
            }
            __syncthreads();
            const long long live = (total - t0 < 256 ? total - t0 : 256) * CT;   // float4 this workgroup produced
            float4* o = reinterpret_cast<float4*>(static_cast<float*>(a.out)) + t0 * CT;
#pragma unroll
            for (int g = 0; g < CT; ++g) {
                const int i = 256 * g + threadIdx.x;
                if (i < live) o[i] = xchg[i];
            }
            __syncthreads();
        }
        return;
    }
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        const int y = (int)(t / groups_x), x = (int)(t - (long long)y * groups_x) * 4;
        const int nv = min(4, a.OW - x);
        const long long src = (long long)(y + a.top) * a.W + a.left + x;
        const long long dpx = (long long)y * a.OW + x;
        float n[4], v[CT][4];
        load_px4(a.norm ? a.norm + src : nullptr, nv, vec, n);
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            load_px4(a.image + c * iplane + src, nv, vec, v[c]);
            if (a.norm) {
#pragma unroll
                for (int m = 0; m < 4; ++m) v[c][m] = __fdiv_rn(v[c][m], n[m]);
            }
        }
        // element e = m * CT + c of the thread's contiguous run; group g = elements 4g .. 4g+3
        if (a.kind == OUT_U8) {
            uint8_t* o = static_cast<uint8_t*>(a.out) + dpx * CT;
#pragma unroll
            for (int g = 0; g < CT; ++g) {
                uint8_t b[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) b[j] = cast_u8(v[(4 * g + j) % CT][(4 * g + j) / CT]);
                const int left = nv * CT - 4 * g;
                if (left > 0) store_u8x4(o + 4 * g, b, left < 4 ? left : 4);
            }
        } else {
            float* o = static_cast<float*>(a.out) + dpx * CT;
#pragma unroll
            for (int g = 0; g < CT; ++g) {
                float b[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) b[j] = v[(4 * g + j) % CT][(4 * g + j) / CT];
                const int left = nv * CT - 4 * g;
                if (left > 0) store_f32x4(o + 4 * g, b, left < 4 ? left : 4);
            }
        }
    }
}

// channel-last with more than 4 channels: element stores (correct for any C; not a tuned path)
__global__ __launch_bounds__(256) void merge_crop_hwc_generic_kernel(const CropArgs a) {
    const int groups_x = (a.OW + 3) / 4;
    const long long total = (long long)a.OH * groups_x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long iplane = (long long)a.H * a.W;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        const int y = (int)(t / groups_x), x = (int)(t - (long long)y * groups_x) * 4;
        const int nv = min(4, a.OW - x);
        const long long src = (long long)(y + a.top) * a.W + a.left + x;
        const long long dpx = (long long)y * a.OW + x;
        float n[4];
        for (int m = 0; m < nv; ++m) n[m] = a.norm ? a.norm[src + m] : 1.0f;
        for (int c = 0; c < a.C; ++c) {
            for (int m = 0; m < nv; ++m) {
                const float v = a.norm ? __fdiv_rn(a.image[c * iplane + src + m], n[m]) : a.image[c * iplane + src + m];
                const long long o = (dpx + m) * a.C + c;
                if (a.kind == OUT_U8) static_cast<uint8_t*>(a.out)[o] = cast_u8(v);
                else static_cast<float*>(a.out)[o] = v;
            }
        }
    }
}

}  // namespace ptb

using namespace ptb;

extern "C" int ptb_split_tiles_u8(const uint8_t* image, int IH, int IW, int IC, const int64_t* xs, const int64_t* ys, int B,
                                  int th, int tw, int V, const int* views, const float* scale, const float* bias, int pad_value,
                                  float* out, ptb_stream_t stream) {
    if (!image || !out || !xs || !ys || IH < 1 || IW < 1 || IC < 1 || B < 0 || th < 1 || tw < 1) return PTB_EINVAL;
    if (IC > MAX_SPLIT_C) return PTB_EUNSUPPORTED;
    if (pad_value < 0 || pad_value > 255) return PTB_EINVAL;
    if ((scale == nullptr) != (bias == nullptr)) return PTB_EINVAL;
    if (V < 1 || V > 8 || !views) return PTB_EINVAL;
    int codes = 0, nt = 0;
    for (int k = 0; k < V; ++k) {
        if (views[k] < 0 || views[k] > 7) return PTB_EINVAL;
        codes |= views[k] << (3 * k);
        nt += views[k] & 1;
    }
    if (nt && th != tw) return PTB_EINVAL;
    for (int b = 0; b < B; ++b) {  // a tile may hang over the border but must be addressable with 32-bit coordinates
        if (xs[b] < -(1 << 30) || xs[b] > (1 << 30) || ys[b] < -(1 << 30) || ys[b] > (1 << 30)) return PTB_EBOUNDS;
    }
    if (B == 0) return PTB_OK;
    hipStream_t s = (hipStream_t)stream;
    ViewArgs a{};
    a.dst = out;
    a.H = th; a.W = tw; a.C = IC;
    a.nviews = V;
    a.codes = codes;
    a.scale = 1.0f;
    SplitArgs g{};
    g.img = image; g.IH = IH; g.IW = IW; g.IC = IC;
    g.pad = (float)pad_value;
    g.affine = scale ? 1 : 0;
    for (int c = 0; c < IC; ++c) { g.scale[c] = scale ? scale[c] : 1.0f; g.bias[c] = bias ? bias[c] : 0.0f; }
    const bool fast = !g_force_scalar && tw % 4 == 0 && (nt == 0 || th % 4 == 0) && aligned16(out);
    const int ch = g_chunk_rows;
    a.chunks_x = (tw + CW - 1) / CW;
    a.chunks_y = (th + ch - 1) / ch;
    for (int b0 = 0; b0 < B; b0 += MAX_GROUP) {
        const int n = B - b0 < MAX_GROUP ? B - b0 : MAX_GROUP;
        g.b0 = b0;
        for (int t = 0; t < n; ++t) { g.tx[t] = (int)xs[b0 + t]; g.ty[t] = (int)ys[b0 + t]; }
        if (fast) {
            const long long blocks = (long long)n * IC * a.chunks_x * a.chunks_y;
            if (blocks > 0x7fffffffLL) return PTB_EUNSUPPORTED;
            if (ch == 64) hipLaunchKernelGGL(edge_split_kernel<64>, dim3((unsigned)blocks), dim3(1024), 0, s, a, g, B);
            else if (ch == 32) hipLaunchKernelGGL(edge_split_kernel<32>, dim3((unsigned)blocks), dim3(512), 0, s, a, g, B);
            else hipLaunchKernelGGL(edge_split_kernel<16>, dim3((unsigned)blocks), dim3(256), 0, s, a, g, B);
        } else {
            const long long total = (long long)V * n * IC * th * tw;
            const long long want = (total + 255) / 256;
            hipLaunchKernelGGL(edge_split_scalar_kernel, dim3((unsigned)(want < 16384 ? want : 16384)), dim3(256), 0, s, a, g, B, n);
        }
        if (int rc = check_launch()) return rc;
    }
    return PTB_OK;
}

extern "C" int ptb_merge_crop(const float* image, const float* norm, int C, int H, int W, int top, int left, int OH, int OW,
                              int layout, int kind, void* out, ptb_stream_t stream) {
    if (!image || !out || C < 1 || H < 1 || W < 1 || OH < 0 || OW < 0) return PTB_EINVAL;
    if (top < 0 || left < 0 || (long long)top + OH > H || (long long)left + OW > W) return PTB_EBOUNDS;
    if (layout < 0 || layout > 1 || kind < OUT_F32 || kind > OUT_ARGMAX_I64) return PTB_EINVAL;
    if (kind == OUT_ARGMAX_U8 && C > 256) return PTB_EUNSUPPORTED;
    if (OH == 0 || OW == 0) return PTB_OK;
    CropArgs a{image, norm, out, C, H, W, top, left, OH, OW, layout, kind};
    const long long total = (long long)OH * ((OW + 3) / 4);
    const long long want = (total + 255) / 256;
    const dim3 grid((unsigned)(want < 16384 ? want : 16384)), block(256);
    hipStream_t s = (hipStream_t)stream;
    const bool vec = !g_force_scalar && W % 4 == 0 && left % 4 == 0 && aligned16(image) && aligned16(norm);
    // fp32 channel-last: exchange the lanes' runs through LDS so that the stores are lane-contiguous (C == 1 already is)
    const bool repack = !g_force_scalar && kind == OUT_F32 && C > 1 && OW % 4 == 0 && aligned16(out);
    if (layout == 0 || kind >= OUT_ARGMAX_U8) hipLaunchKernelGGL(merge_crop_planar_kernel, grid, block, 0, s, a, vec);
    else if (C == 1) hipLaunchKernelGGL(merge_crop_hwc_kernel<1>, grid, block, 0, s, a, vec, false);
    else if (C == 2) hipLaunchKernelGGL(merge_crop_hwc_kernel<2>, grid, block, 0, s, a, vec, repack);
    else if (C == 3) hipLaunchKernelGGL(merge_crop_hwc_kernel<3>, grid, block, 0, s, a, vec, repack);
    else if (C == 4) hipLaunchKernelGGL(merge_crop_hwc_kernel<4>, grid, block, 0, s, a, vec, repack);
    else hipLaunchKernelGGL(merge_crop_hwc_generic_kernel, grid, block, 0, s, a);
    return check_launch();
}

// ------------------------------------------------------------------------------------------------ 3-D tiles
// VolumeMerger.integrate_batch (reference inference/tiles_3d.py:195-208): volume[:, z:z+d, y:y+h, x:x+w] += tile * weight,
// norm_mask[...] += weight, tile after tile.  One launch per tile: a tile never overlaps itself, so every launch owns
// its accumulator region exclusively (race-free without atomics) and the stream order reproduces the reference's
// sequential fp32 order bit for bit.  A 3-D tile is megabytes, so a launch per tile is not launch-bound.
namespace ptb {

struct VolArgs {
    float* volume;        // [C, D, H, W]
    float* norm;          // [D, H, W]
    const float* weight;  // [d, h, w]
    const float* tile;    // [C, d, h, w]
    int C, d, h, w, D, H, W;
    int z0, y0, x0;
};

template <bool VEC>
__global__ __launch_bounds__(256) void volume_accumulate_kernel(const VolArgs a) {
    constexpr int PIX = VEC ? 4 : 1;
    const int wq = (a.w + PIX - 1) / PIX;
    const long long per_chan = (long long)a.d * a.h * wq;
    const long long total = per_chan * a.C;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long tplane = (long long)a.d * a.h * a.w, vplane = (long long)a.D * a.H * a.W;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int c = (int)(i / per_chan);
        long long r = i - (long long)c * per_chan;
        const int q = (int)(r % wq);
        r /= wq;
        const int y = (int)(r % a.h), z = (int)(r / a.h);
        const long long toff = ((long long)z * a.h + y) * a.w + (long long)q * PIX;
        const long long voff = ((long long)(a.z0 + z) * a.H + (a.y0 + y)) * a.W + a.x0 + (long long)q * PIX;
        if (VEC) {
            const float4 t = *reinterpret_cast<const float4*>(a.tile + c * tplane + toff);
            const float4 w4 = *reinterpret_cast<const float4*>(a.weight + toff);
            float4* vp = reinterpret_cast<float4*>(a.volume + c * vplane + voff);
            float4 v = *vp;
            v.x = __fadd_rn(v.x, __fmul_rn(t.x, w4.x)); v.y = __fadd_rn(v.y, __fmul_rn(t.y, w4.y));
            v.z = __fadd_rn(v.z, __fmul_rn(t.z, w4.z)); v.w = __fadd_rn(v.w, __fmul_rn(t.w, w4.w));
            *vp = v;
            if (c == 0) {
                float4* np = reinterpret_cast<float4*>(a.norm + voff);
                float4 n = *np;
                n.x = __fadd_rn(n.x, w4.x); n.y = __fadd_rn(n.y, w4.y); n.z = __fadd_rn(n.z, w4.z); n.w = __fadd_rn(n.w, w4.w);
                *np = n;
            }
        } else {
            const float wv = a.weight[toff];
            a.volume[c * vplane + voff] = __fadd_rn(a.volume[c * vplane + voff], __fmul_rn(a.tile[c * tplane + toff], wv));
            if (c == 0) a.norm[voff] = __fadd_rn(a.norm[voff], wv);
        }
    }
}

}  // namespace ptb

extern "C" int ptb_volume_accumulate(float* volume, float* norm, const float* weight, const float* tiles, const int64_t* zs,
                                     const int64_t* ys, const int64_t* xs, int B, int C, int d, int h, int w, int D, int H, int W,
                                     ptb_stream_t stream) {
    if (!volume || !norm || !weight || !tiles || !zs || !ys || !xs) return PTB_EINVAL;
    if (B < 0 || C < 1 || d < 1 || h < 1 || w < 1 || D < 1 || H < 1 || W < 1) return PTB_EINVAL;
    for (int b = 0; b < B; ++b)
        if (zs[b] < 0 || ys[b] < 0 || xs[b] < 0 || zs[b] + d > D || ys[b] + h > H || xs[b] + w > W) return PTB_EBOUNDS;
    VolArgs a{volume, norm, weight, nullptr, C, d, h, w, D, H, W, 0, 0, 0};
    const long long tile_elems = (long long)C * d * h * w;
    const bool base_vec = !g_force_scalar && w % 4 == 0 && W % 4 == 0 && aligned16(volume) && aligned16(norm) && aligned16(weight) &&
                          aligned16(tiles) && tile_elems % 4 == 0;
    for (int b = 0; b < B; ++b) {
        a.tile = tiles + (long long)b * tile_elems;
        a.z0 = (int)zs[b]; a.y0 = (int)ys[b]; a.x0 = (int)xs[b];
        const bool vec = base_vec && a.x0 % 4 == 0;
        const long long items = (long long)C * d * h * (vec ? w / 4 : w);
        const long long want = (items + 255) / 256;
        const dim3 grid((unsigned)(want < 16384 ? want : 16384)), block(256);
        if (vec) hipLaunchKernelGGL(ptb::volume_accumulate_kernel<true>, grid, block, 0, (hipStream_t)stream, a);
        else hipLaunchKernelGGL(ptb::volume_accumulate_kernel<false>, grid, block, 0, (hipStream_t)stream, a);
        if (int rc = ptb::check_launch()) return rc;
    }
    return PTB_OK;
}
