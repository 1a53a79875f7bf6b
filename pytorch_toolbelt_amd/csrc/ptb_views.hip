// ptb_views.hip -- view-gather kernels for gfx950 (MI355X):
//   * TTA de-augment + reduce            (reference inference/tta.py:287-316,344-365,442-467,503-524)
//   * TTA augment / per-view transform   (reference inference/tta.py:257-284,319-341,385-422,470-484)
//   * TileMerger.integrate_batch         (reference inference/tiles.py:321-339)
//   * the fusion of de-augment+reduce with integrate_batch (the reduced tile never reaches HBM)
//
// One kernel family serves all four: a 64-column x CH-row *chunk* of the output is produced by one
// workgroup of CH*16 threads, each thread owning one float4 (16 B/lane: 64-lane waves issue 1 KiB
// global_load_dwordx4).  A D4 view is (T, fr, fc): row-preserving views (T=0) are read straight from
// HBM with mirrored addressing (reversed rows are still one contiguous 256 B segment per 16 lanes);
// transposing views (T=1) read the *source* block coalesced along source rows, scatter it transposed
// into an XOR-swizzled LDS tile (ds_write_b32, <=2-way = free on CDNA4) and read it back as one
// conflict-free ds_read_b128 per lane.  These kernels are HBM-bound streaming gathers: no MFMA.
//
// Accumulation into the TileMerger image is race-free without atomics: the host splits the batch's
// tile rectangles into disjoint *cells* (the arrangement of their edges); every cell is owned by
// exactly one set of workgroups, which walk the (<=4) covering tiles in batch order and read-modify-
// write each accumulator element exactly once per launch.  Per-XCD L2s are not coherent, so
// exclusive ownership within a launch (plus the kernel-boundary release/acquire between launches) is
// what makes this correct on MI355X; it also halves the RMW traffic of 50%-overlap batches and
// reproduces the reference's sequential fp32 order bit-for-bit (mul and add are not contracted).
#include <algorithm>
#include <vector>

#include "ptb_view_device.h"

namespace ptb {

// ------------------------------------------------------------------------------------------------ fast kernels
template <int CH, int NV, int CODES, int OPK, int MODE, int LD>
__global__ __launch_bounds__(CH * 16) void view_plain_kernel(const ViewArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[lds_tiles(NV, CODES) ? lds_tiles(NV, CODES) * CW * CH : 4];
    const int tid = threadIdx.x;
    const int cpt = a.chunks_x * a.chunks_y;
    int bid = blockIdx.x;
    const int chunk = bid % cpt;
    bid /= cpt;
    const int c = bid % a.C;
    const int t = bid / a.C;
    const int cx0 = (chunk % a.chunks_x) * CW, cy0 = (chunk / a.chunks_x) * CH;
    const int cw = min(CW, a.W - cx0), ch = min(CH, a.H - cy0);

    int codes = a.codes, nv = a.nviews;
    long long src_tile = t;
    if (MODE == MODE_PERVIEW) {
        codes = (a.codes >> (3 * (t / a.tiles_per_view))) & 7;
        nv = 1;
        src_tile = t % a.src_tile_mod;
    }
    const long long plane = src_tile * a.src_tile_stride + (long long)c * a.H * a.W;
    float4 val = gather_reduce<CH, NV, CODES, OPK, LD>(a.src, plane, a.src_view_stride, nv, codes, a.H, a.W, cx0, cy0, cw, ch, a.op,
                                                  a.divisor, lds, tid, false);
    if (MODE == MODE_PERVIEW && a.scale != 1.0f) {
        val.x *= a.scale; val.y *= a.scale; val.z *= a.scale; val.w *= a.scale;
    }
    const int q = tid & 15, r = tid >> 4;
    if (r < ch && 4 * q < cw) {
        float* o = a.dst + (long long)t * a.dst_tile_stride + (long long)c * a.dst_chan_stride +
                   (long long)(cy0 + r) * a.dst_row_stride + cx0 + 4 * q;
        out_store4(o, val);
    }
}

template <int CH, int NV, int CODES, int OPK, int LD>
__global__ __launch_bounds__(CH * 16) void view_accum_kernel(const ViewArgs a, const CellArgs g) {
    __shared__ __attribute__((aligned(16))) float lds[lds_tiles(NV, CODES) ? lds_tiles(NV, CODES) * CW * CH : 4];
    const int tid = threadIdx.x;
    const int c = blockIdx.x % a.C;
    const int chunk = blockIdx.x / a.C;
    int ci = 0;
    while (ci < a.ncells - 1 && chunk >= g.cells[ci].chunk_end) ++ci;
    const Cell& cell = g.cells[ci];
    const int first = ci ? g.cells[ci - 1].chunk_end : 0;
    const int ncx = (cell.w + CW - 1) / CW;
    const int lc = chunk - first;
    const int cx0 = (lc % ncx) * CW, cy0 = (lc / ncx) * CH;
    const int cw = min(CW, cell.w - cx0), ch = min(CH, cell.h - cy0);
    const int q = tid & 15, r = tid >> 4;
    const bool act = (r < ch) && (4 * q < cw);
    const int ax = cell.ox + cx0, ay = cell.oy + cy0;  // chunk origin in the accumulator

    float* ip = a.dst + (long long)c * a.dst_chan_stride + (long long)(ay + r) * a.dst_row_stride + ax + 4 * q;
    float* np = a.norm + (long long)(ay + r) * a.dst_row_stride + ax + 4 * q;
    const bool do_norm = c == 0 && a.norm != nullptr;  // norm == NULL: the caller keeps the (data independent) normaliser itself
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f), nacc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (act && !cell.fresh) {  // first touch: the accumulator is logically zero, nothing to read (and it was never memset)
        acc = *reinterpret_cast<const float4*>(ip);
        if (do_norm) nacc = *reinterpret_cast<const float4*>(np);
    }
    float4 nfull = make_float4(1.f, 1.f, 1.f, 1.f);
    if (act && cell.final_)  // issued with the first loads: at the end it would add one exposed memory latency per workgroup
        nfull = *reinterpret_cast<const float4*>(a.norm_full + (long long)(ay + r) * a.dst_row_stride + ax + 4 * q);
    const int nt = cell.ntiles;
    for (int e = 0; e < nt; ++e) {
        const int gt = cell.tile[e];
        const int lx = ax - g.tile_x[gt], ly = ay - g.tile_y[gt];
        const long long plane = (long long)g.tile_id[gt] * a.src_tile_stride + (long long)c * a.H * a.W;
        const float4 val = round_src4<LD>(gather_reduce<CH, NV, CODES, OPK, LD>(a.src, plane, a.src_view_stride, a.nviews, a.codes, a.H, a.W, lx, ly, cw,
                                                        ch, a.op, a.divisor, lds, tid, e + 1 < nt), a.round_src);
        if (act) {
            // (requesting the window value before the gather was A/B-tested on one box: no gain, and 6 more VGPRs cost a wave
            // of occupancy -- the other resident workgroups already hide this L2 latency)
            const float4 w4 = *reinterpret_cast<const float4*>(a.weight + (long long)(ly + r) * a.W + lx + 4 * q);
            // tile*weight rounded, then added: the reference's two torch ops (tiles.py:338), no FMA contraction
            acc.x = __fadd_rn(acc.x, __fmul_rn(val.x, w4.x));
            acc.y = __fadd_rn(acc.y, __fmul_rn(val.y, w4.y));
            acc.z = __fadd_rn(acc.z, __fmul_rn(val.z, w4.z));
            acc.w = __fadd_rn(acc.w, __fmul_rn(val.w, w4.w));
            if (do_norm) {
                nacc.x = __fadd_rn(nacc.x, w4.x); nacc.y = __fadd_rn(nacc.y, w4.y);
                nacc.z = __fadd_rn(nacc.z, w4.z); nacc.w = __fadd_rn(nacc.w, w4.w);
            }
        }
    }
    if (act) {
        if (cell.final_) {
            // last touch of a planned cell: the weighted sum is complete, so the merged value (tiles.py:346) is written
            // right away and the accumulator is never stored -- the separate merge pass over this cell disappears
            const long long off = (long long)c * a.dst_chan_stride + (long long)(ay + r) * a.dst_row_stride + ax + 4 * q;
            out_store4(a.merged + off,
                       make_float4(__fdiv_rn(acc.x, nfull.x), __fdiv_rn(acc.y, nfull.y), __fdiv_rn(acc.z, nfull.z), __fdiv_rn(acc.w, nfull.w)));
            if (a.keep_acc) *reinterpret_cast<float4*>(ip) = acc;     // (self-planned mergers: `image` stays readable, exactly)
        } else {
            *reinterpret_cast<float4*>(ip) = acc;
            if (do_norm) *reinterpret_cast<float4*>(np) = nacc;
        }
    }
}

// norm_mask only (a.dst == NULL): the same cell walk without any tile data -- used to materialise the normaliser that
// the accumulate calls skipped (it depends on the crop list and the window only, never on the predictions).
template <int CH>
__global__ __launch_bounds__(CH * 16) void norm_accum_kernel(const ViewArgs a, const CellArgs g) {
    const int tid = threadIdx.x;
    const int chunk = blockIdx.x;
    int ci = 0;
    while (ci < a.ncells - 1 && chunk >= g.cells[ci].chunk_end) ++ci;
    const Cell& cell = g.cells[ci];
    const int first = ci ? g.cells[ci - 1].chunk_end : 0;
    const int ncx = (cell.w + CW - 1) / CW;
    const int lc = chunk - first;
    const int cx0 = (lc % ncx) * CW, cy0 = (lc / ncx) * CH;
    const int cw = min(CW, cell.w - cx0), ch = min(CH, cell.h - cy0);
    const int q = tid & 15, r = tid >> 4;
    if (!((r < ch) && (4 * q < cw))) return;
    const int ax = cell.ox + cx0, ay = cell.oy + cy0;
    float* np = a.norm + (long long)(ay + r) * a.dst_row_stride + ax + 4 * q;
    float4 nacc = cell.fresh ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4*>(np);
    for (int e = 0; e < cell.ntiles; ++e) {
        const int gt = cell.tile[e];
        const int lx = ax - g.tile_x[gt], ly = ay - g.tile_y[gt];
        const float4 w4 = *reinterpret_cast<const float4*>(a.weight + (long long)(ly + r) * a.W + lx + 4 * q);
        nacc.x = __fadd_rn(nacc.x, w4.x); nacc.y = __fadd_rn(nacc.y, w4.y);
        nacc.z = __fadd_rn(nacc.z, w4.z); nacc.w = __fadd_rn(nacc.w, w4.w);
    }
    *reinterpret_cast<float4*>(np) = nacc;
}

// ------------------------------------------------------------------------------------------------ augment (scatter)
// *_image_augment: every source chunk is read ONCE and written to all V views (the gather formulation above would read
// it V times).  Row-preserving views are stored straight from the registers at mirrored addresses; the transposing
// views share one transposed, XOR-swizzled LDS copy of the chunk (ST[c][r] = S[r][c]) read back with ds_read_b128.
// NONLIN = backward of a non-linear reduction: the scattered value is u = g * post'(out) / V (a.norm = forward output,
// a.divisor = V) and every destination is multiplied by pre'(x_k) read at the destination (a.weight = forward input).
// ------------------------------------------------------------------------------------------------ deferred band merge
// TileMerger(crops=..., defer=True): the merger keeps references to the model outputs instead of accumulating them, and when
// the last tile covering a horizontal BAND of the image (the rows between two consecutive tile edges) has arrived, one launch
// reads every covering tile of the band -- they may live in different batch tensors, hence per-tile source pointers --
// applies the inverse views, reduces, blends in tile order and writes `sum / norm_full` straight to the merged map.  The
// accumulator image never exists in HBM: per 8 tiles the incremental path moves 312 MB for 268 MB of model outputs, this one
// only the outputs and the result.  Same chunk / cover walk and the same gather_reduce as view_accum_kernel (CH = 32), so
// the fp32 operation order per pixel -- and therefore every bit of the result -- is that of the incremental path.
constexpr int BAND_CELLS = 40, BAND_TILES = 48;
struct BandCell {
    int ox, oy, w, h;
    int chunk_end;
    int ntiles;
    int tile[MAX_COVER];
};
struct BandArgs {
    BandCell cells[BAND_CELLS];
    int tile_x[BAND_TILES], tile_y[BAND_TILES];
    const void* tile_src[BAND_TILES];   // view 0, channel 0 of the tile
    long long tile_vs[BAND_TILES];      // elements between consecutive views of this tile (its batch size * C * H * W)
};

template <int NV, int CODES, int OPK, int LD>
__global__ __launch_bounds__(512) void band_merge_kernel(const ViewArgs a, const BandArgs g) {
    constexpr int CH = 32;
    __shared__ __attribute__((aligned(16))) float lds[lds_tiles(NV, CODES) ? lds_tiles(NV, CODES) * CW * CH : 4];
    const int tid = threadIdx.x;
    const int c = blockIdx.x % a.C;
    const int chunk = blockIdx.x / a.C;
    int ci = 0;
    while (ci < a.ncells - 1 && chunk >= g.cells[ci].chunk_end) ++ci;
    const BandCell& cell = g.cells[ci];
    const int first = ci ? g.cells[ci - 1].chunk_end : 0;
    const int ncx = (cell.w + CW - 1) / CW;
    const int lc = chunk - first;
    const int cx0 = (lc % ncx) * CW, cy0 = (lc / ncx) * CH;
    const int cw = min(CW, cell.w - cx0), ch = min(CH, cell.h - cy0);
    const int q = tid & 15, r = tid >> 4;
    const bool act = (r < ch) && (4 * q < cw);
    const int ax = cell.ox + cx0, ay = cell.oy + cy0;
    const long long pix = (long long)(ay + r) * a.dst_row_stride + ax + 4 * q;
    float4 nfull = make_float4(1.f, 1.f, 1.f, 1.f);
    if (act) nfull = *reinterpret_cast<const float4*>(a.norm_full + pix);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const int nt = cell.ntiles;
    for (int e = 0; e < nt; ++e) {
        const int gt = cell.tile[e];
        const int lx = ax - g.tile_x[gt], ly = ay - g.tile_y[gt];
        const float4 val = round_src4<LD>(gather_reduce<CH, NV, CODES, OPK, LD>(static_cast<const float*>(g.tile_src[gt]), (long long)c * a.H * a.W,
                                                                g.tile_vs[gt], a.nviews, a.codes, a.H, a.W, lx, ly, cw, ch, a.op,
                                                                a.divisor, lds, tid, e + 1 < nt), a.round_src);
        if (act) {
            const float4 w4 = *reinterpret_cast<const float4*>(a.weight + (long long)(ly + r) * a.W + lx + 4 * q);
            acc.x = __fadd_rn(acc.x, __fmul_rn(val.x, w4.x));   // tiles.py:338, no FMA contraction
            acc.y = __fadd_rn(acc.y, __fmul_rn(val.y, w4.y));
            acc.z = __fadd_rn(acc.z, __fmul_rn(val.z, w4.z));
            acc.w = __fadd_rn(acc.w, __fmul_rn(val.w, w4.w));
        }
    }
    if (act)
        *reinterpret_cast<float4*>(a.merged + (long long)c * a.dst_chan_stride + pix) =
            make_float4(__fdiv_rn(acc.x, nfull.x), __fdiv_rn(acc.y, nfull.y), __fdiv_rn(acc.z, nfull.z), __fdiv_rn(acc.w, nfull.w));
}

template <int CH, bool NONLIN>
__global__ __launch_bounds__(CH * 16) void view_scatter_kernel(const ViewArgs a, int B) {
    __shared__ __attribute__((aligned(16))) float st[CW * CH];
    const int tid = threadIdx.x;
    const int cpt = a.chunks_x * a.chunks_y;
    int bid = blockIdx.x;
    const int chunk = bid % cpt;
    bid /= cpt;
    const int c = bid % a.C;
    const int b = bid / a.C;
    const int x0 = (chunk % a.chunks_x) * CW, y0 = (chunk / a.chunks_x) * CH;
    const int cw = min(CW, a.W - x0), ch = min(CH, a.H - y0);
    const int q = tid & 15, r = tid >> 4;
    const bool act = r < ch && 4 * q < cw;
    const long long plane = (long long)a.H * a.W;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    const long long src_off = ((long long)b * a.C + c) * plane + (long long)(y0 + r) * a.W + x0 + 4 * q;
    if (act) v = ld16<true>(a.src + src_off);
    if (a.scale != 1.0f) { v.x *= a.scale; v.y *= a.scale; v.z *= a.scale; v.w *= a.scale; }
    if (NONLIN && act) {
        const float4 o4 = ld16<true>(a.norm + src_off);
        v.x *= red_dpost(o4.x, a.op) / a.divisor; v.y *= red_dpost(o4.y, a.op) / a.divisor;
        v.z *= red_dpost(o4.z, a.op) / a.divisor; v.w *= red_dpost(o4.w, a.op) / a.divisor;
    }
    scatter_chunk<CH, NONLIN>(a, B, b, c, x0, y0, cw, ch, v, st, tid);
}

// scalar fallback of the non-linear backward (any shape): one thread per element of the reduced tensor
__global__ __launch_bounds__(256) void deaug_bwd_scalar_kernel(const ViewArgs a, int B) {
    const long long plane = (long long)a.H * a.W;
    const long long n = (long long)B * a.C * plane;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride) {
        const long long bc = t / plane, px = t - bc * plane;
        const int R = (int)(px / a.W), Cc = (int)(px - (long long)R * a.W);
        const float u = a.src[t] * red_dpost(a.norm[t], a.op) / a.divisor;
        for (int k = 0; k < a.nviews; ++k) {
            const int code = (a.codes >> (3 * k)) & 7;  // scatter view (inverse of the forward de-augment view)
            int i, j;
            if (code & 1) { i = (code & 4) ? a.W - 1 - Cc : Cc; j = (code & 2) ? a.H - 1 - R : R; }
            else { i = (code & 2) ? a.H - 1 - R : R; j = (code & 4) ? a.W - 1 - Cc : Cc; }
            const long long b = bc / a.C, c = bc - b * a.C;
            const long long off = (((long long)k * B + b) * a.C + c) * plane + (long long)i * a.W + j;
            a.dst[off] = u * red_dpre(a.weight[off], a.op);
        }
    }
}

// ------------------------------------------------------------------------------------------------ scalar kernels
// Any shape / alignment (odd tile sizes such as the reference's 51/26 test): one element per thread, 64x4 threads
// sweeping a 64x64 chunk.  Same cell ownership and the same arithmetic order as the fast kernels.
__device__ __forceinline__ float scalar_reduce(const float* __restrict__ plane, long long view_stride, int nv, int codes,
                                               int H, int W, int i, int j, int op, float divisor, bool nonlinear) {
    float s = 0.f;
    for (int k = 0; k < nv; ++k) {
        const int code = (codes >> (3 * k)) & 7;
        int rr = (code & 1) ? j : i, cc = (code & 1) ? i : j;
        const int rows = (code & 1) ? W : H, cols = (code & 1) ? H : W;
        if (code & 2) rr = rows - 1 - rr;
        if (code & 4) cc = cols - 1 - cc;
        float x = plane[(long long)k * view_stride + (long long)rr * cols + cc];
        x = nonlinear ? red_pre<1>(x, op) : x;
        s = k ? __fadd_rn(s, x) : x;
    }
    return nonlinear ? red_post<1>(s, op, divisor) : red_post<0>(s, op, divisor);
}

template <int MODE>
__global__ __launch_bounds__(256) void view_plain_scalar_kernel(const ViewArgs a) {
    const int cpt = a.chunks_x * a.chunks_y;
    int bid = blockIdx.x;
    const int chunk = bid % cpt;
    bid /= cpt;
    const int c = bid % a.C;
    const int t = bid / a.C;
    const int cx0 = (chunk % a.chunks_x) * CW, cy0 = (chunk / a.chunks_x) * 64;
    const int cw = min(CW, a.W - cx0), ch = min(64, a.H - cy0);
    int codes = a.codes, nv = a.nviews;
    long long src_tile = t;
    if (MODE == MODE_PERVIEW) {
        codes = (a.codes >> (3 * (t / a.tiles_per_view))) & 7;
        nv = 1;
        src_tile = t % a.src_tile_mod;
    }
    const float* plane = a.src + src_tile * a.src_tile_stride + (long long)c * a.H * a.W;
    const bool nonlinear = a.op >= PTB_RED_GMEAN;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    if (tx >= cw) return;
    for (int r = ty; r < ch; r += 4) {
        float val = scalar_reduce(plane, a.src_view_stride, nv, codes, a.H, a.W, cy0 + r, cx0 + tx, a.op, a.divisor, nonlinear);
        if (MODE == MODE_PERVIEW && a.scale != 1.0f) val *= a.scale;
        a.dst[(long long)t * a.dst_tile_stride + (long long)c * a.dst_chan_stride + (long long)(cy0 + r) * a.dst_row_stride + cx0 + tx] = val;
    }
}

__global__ __launch_bounds__(256) void view_accum_scalar_kernel(const ViewArgs a, const CellArgs g) {
    const int c = blockIdx.x % a.C;
    const int chunk = blockIdx.x / a.C;
    int ci = 0;
    while (ci < a.ncells - 1 && chunk >= g.cells[ci].chunk_end) ++ci;
    const Cell& cell = g.cells[ci];
    const int first = ci ? g.cells[ci - 1].chunk_end : 0;
    const int ncx = (cell.w + CW - 1) / CW;
    const int lc = chunk - first;
    const int cx0 = (lc % ncx) * CW, cy0 = (lc / ncx) * 64;
    const int cw = min(CW, cell.w - cx0), ch = min(64, cell.h - cy0);
    const int ax = cell.ox + cx0, ay = cell.oy + cy0;
    const bool nonlinear = a.op >= PTB_RED_GMEAN;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    if (tx >= cw) return;
    for (int r = ty; r < ch; r += 4) {
        const long long off = (long long)(ay + r) * a.dst_row_stride + ax + tx;
        float acc = cell.fresh ? 0.f : a.dst[(long long)c * a.dst_chan_stride + off];
        const bool do_norm = c == 0 && a.norm != nullptr;
        float nacc = (do_norm && !cell.fresh) ? a.norm[off] : 0.f;
        for (int e = 0; e < cell.ntiles; ++e) {
            const int gt = cell.tile[e];
            const int lx = ax - g.tile_x[gt] + tx, ly = ay - g.tile_y[gt] + r;
            const float* plane = a.src + (long long)g.tile_id[gt] * a.src_tile_stride + (long long)c * a.H * a.W;
            const float val = scalar_reduce(plane, a.src_view_stride, a.nviews, a.codes, a.H, a.W, ly, lx, a.op, a.divisor, nonlinear);
            const float w = a.weight[(long long)ly * a.W + lx];
            acc = __fadd_rn(acc, __fmul_rn(val, w));
            nacc = __fadd_rn(nacc, w);
        }
        if (cell.final_) {
            a.merged[(long long)c * a.dst_chan_stride + off] = __fdiv_rn(acc, a.norm_full[off]);
            if (a.keep_acc) a.dst[(long long)c * a.dst_chan_stride + off] = acc;
        } else {
            a.dst[(long long)c * a.dst_chan_stride + off] = acc;
            if (do_norm) a.norm[off] = nacc;
        }
    }
}

__global__ __launch_bounds__(256) void norm_accum_scalar_kernel(const ViewArgs a, const CellArgs g) {
    const int chunk = blockIdx.x;
    int ci = 0;
    while (ci < a.ncells - 1 && chunk >= g.cells[ci].chunk_end) ++ci;
    const Cell& cell = g.cells[ci];
    const int first = ci ? g.cells[ci - 1].chunk_end : 0;
    const int ncx = (cell.w + CW - 1) / CW;
    const int lc = chunk - first;
    const int cx0 = (lc % ncx) * CW, cy0 = (lc / ncx) * 64;
    const int cw = min(CW, cell.w - cx0), ch = min(64, cell.h - cy0);
    const int ax = cell.ox + cx0, ay = cell.oy + cy0;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    if (tx >= cw) return;
    for (int r = ty; r < ch; r += 4) {
        const long long off = (long long)(ay + r) * a.dst_row_stride + ax + tx;
        float nacc = cell.fresh ? 0.f : a.norm[off];
        for (int e = 0; e < cell.ntiles; ++e) {
            const int gt = cell.tile[e];
            nacc = __fadd_rn(nacc, a.weight[(long long)(ay - g.tile_y[gt] + r) * a.W + ax - g.tile_x[gt] + tx]);
        }
        a.norm[off] = nacc;
    }
}

// ------------------------------------------------------------------------------------------------ host: cells
// Split the tile rectangles of one launch group into disjoint cells (arrangement of their edges), each with the
// ascending list of covering tiles.  Returns false when the group needs more than MAX_CELLS / MAX_COVER.
// Host-side freshness bitmap of the accumulator (owned by the caller): one byte per block of 64 columns x `rows` rows,
// 1 = never written since the last reset.  A cell whose blocks are all fresh is written with plain stores.
struct Fresh {
    uint8_t* map;  // may be null: everything is read-modify-write
    int rows;      // block rows (must equal the chunk rows of the launch)
    int H, W;      // accumulator size
    int nbx() const { return (W + CW - 1) / CW; }
    int nby() const { return (H + rows - 1) / rows; }
};

enum { DECOMP_OK = 0, DECOMP_SPLIT = 1, DECOMP_NEEDS_ZERO = 2 };

// Cut a block-aligned cell into rectangles on which a per-block 0/1 state is uniform: column strips wherever two
// neighbouring block columns differ in some row, then row runs inside each strip.  Any sub-rectangle of a cell is a cell
// with the same cover list, so this never changes the result -- only which stores are first-touch / final.
template <typename Emit>
static void split_by_state(const Cell& c, const std::vector<uint8_t>& state, int nx, int ny, int bx0, int by0, int rows, Emit emit) {
    int sx0 = 0;
    for (int sx = 1; sx <= nx; ++sx) {
        bool cut = sx == nx;
        for (int y = 0; y < ny && !cut; ++y) cut = state[(size_t)y * nx + sx] != state[(size_t)y * nx + sx - 1];
        if (!cut) continue;
        int run_start = 0;
        for (int y = 1; y <= ny; ++y) {
            if (y < ny && state[(size_t)y * nx + sx0] == state[(size_t)run_start * nx + sx0]) continue;
            Cell piece = c;
            piece.ox = (bx0 + sx0) * CW;
            piece.w = std::min((bx0 + sx) * CW, c.ox + c.w) - piece.ox;
            piece.oy = (by0 + run_start) * rows;
            piece.h = std::min((by0 + y) * rows, c.oy + c.h) - piece.oy;
            emit(piece, state[(size_t)run_start * nx + sx0]);
            run_start = y;
        }
        sx0 = sx;
    }
}

// Tag cells with their freshness, splitting a cell into rectangles of uniform freshness.  Cells that are not aligned to
// the block grid cannot use first-touch stores: if they touch any fresh block the caller has to zero-fill first
// (DECOMP_NEEDS_ZERO).
static int apply_freshness(std::vector<Cell>& cells, const Fresh& fr) {
    if (!fr.map) return DECOMP_OK;
    std::vector<Cell> out;
    std::vector<uint8_t> state;
    const int nbx = fr.nbx();
    for (const Cell& c : cells) {
        const bool aligned = c.ox % CW == 0 && c.oy % fr.rows == 0 && (c.w % CW == 0 || c.ox + c.w == fr.W) &&
                             (c.h % fr.rows == 0 || c.oy + c.h == fr.H);
        const int bx0 = c.ox / CW, bx1 = (c.ox + c.w - 1) / CW, by0 = c.oy / fr.rows, by1 = (c.oy + c.h - 1) / fr.rows;
        if (!aligned) {
            for (int by = by0; by <= by1; ++by)
                for (int bx = bx0; bx <= bx1; ++bx)
                    if (fr.map[by * nbx + bx]) return DECOMP_NEEDS_ZERO;
            out.push_back(c);
            continue;
        }
        const int nx = bx1 - bx0 + 1, ny = by1 - by0 + 1;
        state.assign((size_t)nx * ny, 0);
        for (int by = by0; by <= by1; ++by)
            for (int bx = bx0; bx <= bx1; ++bx) state[(size_t)(by - by0) * nx + (bx - bx0)] = fr.map[by * nbx + bx] ? 1 : 0;
        split_by_state(c, state, nx, ny, bx0, by0, fr.rows, [&](Cell piece, uint8_t st) { piece.fresh = st; out.push_back(piece); });
    }
    cells.swap(out);
    return DECOMP_OK;
}

static void mark_written(const std::vector<Cell>& cells, const Fresh& fr) {
    if (!fr.map) return;
    const int nbx = fr.nbx();
    for (const Cell& c : cells)
        for (int by = c.oy / fr.rows; by <= (c.oy + c.h - 1) / fr.rows; ++by)
            for (int bx = c.ox / CW; bx <= (c.ox + c.w - 1) / CW; ++bx) fr.map[by * nbx + bx] = 0;
}

// Planned accumulation (the caller knows every tile the image will receive): `rem` counts, per accumulator block, the
// planned tiles that have not touched it yet; `done` marks blocks whose merged value has been written.  Same block grid
// as Fresh.  A cell whose block rows all drop to zero in this launch is final.  The planned geometry is block aligned
// (checked by the caller), so every block lies inside exactly one cell of a launch.
struct Plan {
    uint8_t* rem;   // null: not planned
    uint8_t* done;
    int rows, H, W;
    int nbx() const { return (W + CW - 1) / CW; }
};

enum { DECOMP_PLAN_MISMATCH = 3 };

static int apply_plan(std::vector<Cell>& cells, const Plan& pl) {
    if (!pl.rem) return DECOMP_OK;
    std::vector<Cell> out;
    std::vector<uint8_t> fin;
    const int nbx = pl.nbx();
    for (const Cell& c : cells) {
        const bool aligned = c.ox % CW == 0 && c.oy % pl.rows == 0 && (c.w % CW == 0 || c.ox + c.w == pl.W) &&
                             (c.h % pl.rows == 0 || c.oy + c.h == pl.H);
        if (!aligned) return DECOMP_PLAN_MISMATCH;
        const int bx0 = c.ox / CW, bx1 = (c.ox + c.w - 1) / CW, by0 = c.oy / pl.rows, by1 = (c.oy + c.h - 1) / pl.rows;
        const int nx = bx1 - bx0 + 1, ny = by1 - by0 + 1;
        fin.assign((size_t)nx * ny, 0);
        for (int by = by0; by <= by1; ++by)
            for (int bx = bx0; bx <= bx1; ++bx) {
                const int left = (int)pl.rem[by * nbx + bx] - c.ntiles;
                if (left < 0 || pl.done[by * nbx + bx]) return DECOMP_PLAN_MISMATCH;  // more tiles than planned here
                fin[(size_t)(by - by0) * nx + (bx - bx0)] = left == 0;
            }
        split_by_state(c, fin, nx, ny, bx0, by0, pl.rows, [&](Cell piece, uint8_t st) { piece.final_ = st; out.push_back(piece); });
    }
    cells.swap(out);
    return DECOMP_OK;
}

static void commit_plan(const std::vector<Cell>& cells, const Plan& pl) {
    if (!pl.rem) return;
    const int nbx = pl.nbx();
    for (const Cell& c : cells)
        for (int by = c.oy / pl.rows; by <= (c.oy + c.h - 1) / pl.rows; ++by)
            for (int bx = c.ox / CW; bx <= (c.ox + c.w - 1) / CW; ++bx) {
                pl.rem[by * nbx + bx] = (uint8_t)(pl.rem[by * nbx + bx] - c.ntiles);
                if (c.final_) pl.done[by * nbx + bx] = 1;
            }
}

static int decompose(const int* xs, const int* ys, const int* ids, int n, int tw, int th, int chunk_rows, const Fresh& fr,
                     CellArgs& out, int& ncells, int& total_chunks, std::vector<Cell>& cells, const Plan* plan = nullptr) {
    if (n > MAX_GROUP) return DECOMP_SPLIT;
    std::vector<int> ye;
    ye.reserve(2 * n);
    for (int t = 0; t < n; ++t) { ye.push_back(ys[t]); ye.push_back(ys[t] + th); }
    std::sort(ye.begin(), ye.end());
    ye.erase(std::unique(ye.begin(), ye.end()), ye.end());
    cells.clear();
    std::vector<int> active, xe;
    for (size_t yi = 0; yi + 1 < ye.size(); ++yi) {
        const int y0 = ye[yi], y1 = ye[yi + 1];
        active.clear();
        for (int t = 0; t < n; ++t) if (ys[t] <= y0 && y0 < ys[t] + th) active.push_back(t);
        if (active.empty()) continue;
        xe.clear();
        for (int t : active) { xe.push_back(xs[t]); xe.push_back(xs[t] + tw); }
        std::sort(xe.begin(), xe.end());
        xe.erase(std::unique(xe.begin(), xe.end()), xe.end());
        for (size_t xi = 0; xi + 1 < xe.size(); ++xi) {
            const int x0 = xe[xi], x1 = xe[xi + 1];
            Cell c{};
            for (int t : active) {
                if (xs[t] <= x0 && x0 < xs[t] + tw) {
                    if (c.ntiles == MAX_COVER) return DECOMP_SPLIT;
                    c.tile[c.ntiles++] = t;
                }
            }
            if (!c.ntiles) continue;
            c.ox = x0; c.oy = y0; c.w = x1 - x0; c.h = y1 - y0;
            // merge with the cell directly above when it has the same x-range and cover (fewer, taller cells)
            bool merged = false;
            for (auto it = cells.rbegin(); it != cells.rend(); ++it) {
                if (it->oy + it->h == y0 && it->ox == x0 && it->w == c.w && it->ntiles == c.ntiles &&
                    std::equal(c.tile, c.tile + c.ntiles, it->tile)) {
                    it->h += c.h;
                    merged = true;
                    break;
                }
            }
            if (!merged) cells.push_back(c);
        }
    }
    if (int rc = apply_freshness(cells, fr)) return rc;
    if (plan) if (int rc = apply_plan(cells, *plan)) return rc;
    if ((int)cells.size() > MAX_CELLS) return DECOMP_SPLIT;
    std::stable_sort(cells.begin(), cells.end(), [](const Cell& a, const Cell& b) { return a.ntiles > b.ntiles; });
    int run = 0;
    for (size_t i = 0; i < cells.size(); ++i) {
        run += ((cells[i].w + CW - 1) / CW) * ((cells[i].h + chunk_rows - 1) / chunk_rows);
        cells[i].chunk_end = run;
        out.cells[i] = cells[i];
    }
    for (int t = 0; t < n; ++t) { out.tile_x[t] = xs[t]; out.tile_y[t] = ys[t]; out.tile_id[t] = ids[t]; }
    ncells = (int)cells.size();
    total_chunks = run;
    return DECOMP_OK;
}

// ------------------------------------------------------------------------------------------------ dispatch
static int pack_runtime(int V, const int* views) {
    int v = 0;
    for (int k = 0; k < V; ++k) v |= (views[k] & 7) << (3 * k);
    return v;
}

template <int CH, int MODE>
static void launch_plain_ch(const ViewArgs& a, int blocks, hipStream_t s, bool nonlinear) {
    const dim3 grid(blocks), block(CH * 16);
    // half-precision sources (in_dtype != PTB_F32) are compiled for the default chunk rows only (callers check)
#define PTB_PLAIN_LD(NV, CODES, LD)                                                                               \
    do {                                                                                                          \
        if (nonlinear) hipLaunchKernelGGL((view_plain_kernel<CH, NV, CODES, 1, MODE, LD>), grid, block, 0, s, a); \
        else hipLaunchKernelGGL((view_plain_kernel<CH, NV, CODES, 0, MODE, LD>), grid, block, 0, s, a);           \
    } while (0)
#define PTB_PLAIN(NV, CODES)                                                                                      \
    do {                                                                                                          \
        if constexpr (CH == 32) {                                                                                 \
            if (a.in_dtype == PTB_F16) { PTB_PLAIN_LD(NV, CODES, 2); break; }                                     \
            if (a.in_dtype == PTB_BF16) { PTB_PLAIN_LD(NV, CODES, 3); break; }                                    \
        }                                                                                                         \
        if (nonlinear) hipLaunchKernelGGL((view_plain_kernel<CH, NV, CODES, 1, MODE, 1>), grid, block, 0, s, a);  \
        else if (g_nt_loads) hipLaunchKernelGGL((view_plain_kernel<CH, NV, CODES, 0, MODE, 1>), grid, block, 0, s, a); \
        else hipLaunchKernelGGL((view_plain_kernel<CH, NV, CODES, 0, MODE, 0>), grid, block, 0, s, a);            \
    } while (0)
    if constexpr (MODE == MODE_PERVIEW) {
        hipLaunchKernelGGL((view_plain_kernel<CH, 1, -1, 0, MODE_PERVIEW, 1>), grid, block, 0, s, a);
    } else {
        if (a.nviews == 2 && a.codes == CODES_FLIPLR) PTB_PLAIN(2, CODES_FLIPLR);
        else if (a.nviews == 2 && a.codes == CODES_FLIPUD) PTB_PLAIN(2, CODES_FLIPUD);
        else if (a.nviews == 3 && a.codes == CODES_FLIPS) PTB_PLAIN(3, CODES_FLIPS);
        else if (a.nviews == 4 && a.codes == CODES_D2) PTB_PLAIN(4, CODES_D2);
        else if (a.nviews == 8 && a.codes == CODES_D4) PTB_PLAIN(8, CODES_D4);
        else PTB_PLAIN(8, -1);
    }
#undef PTB_PLAIN
#undef PTB_PLAIN_LD
}

template <int CH>
static void launch_accum_ch(const ViewArgs& a, const CellArgs& g, int blocks, hipStream_t s, bool nonlinear) {
    const dim3 grid(blocks), block(CH * 16);
#define PTB_ACCUM_LD(NV, CODES, LD)                                                                               \
    do {                                                                                                          \
        if (nonlinear) hipLaunchKernelGGL((view_accum_kernel<CH, NV, CODES, 1, LD>), grid, block, 0, s, a, g);    \
        else hipLaunchKernelGGL((view_accum_kernel<CH, NV, CODES, 0, LD>), grid, block, 0, s, a, g);              \
    } while (0)
#define PTB_ACCUM(NV, CODES)                                                                                      \
    do {                                                                                                          \
        if constexpr (CH == 32) {                                                                                 \
            if (a.in_dtype == PTB_F16) { PTB_ACCUM_LD(NV, CODES, 2); break; }                                     \
            if (a.in_dtype == PTB_BF16) { PTB_ACCUM_LD(NV, CODES, 3); break; }                                    \
        }                                                                                                         \
        if (nonlinear) hipLaunchKernelGGL((view_accum_kernel<CH, NV, CODES, 1, 1>), grid, block, 0, s, a, g);     \
        else if (g_nt_loads) hipLaunchKernelGGL((view_accum_kernel<CH, NV, CODES, 0, 1>), grid, block, 0, s, a, g); \
        else hipLaunchKernelGGL((view_accum_kernel<CH, NV, CODES, 0, 0>), grid, block, 0, s, a, g);               \
    } while (0)
    if (a.nviews == 1 && a.codes == CODES_ID) PTB_ACCUM(1, CODES_ID);
    else if (a.nviews == 2 && a.codes == CODES_FLIPLR) PTB_ACCUM(2, CODES_FLIPLR);
    else if (a.nviews == 2 && a.codes == CODES_FLIPUD) PTB_ACCUM(2, CODES_FLIPUD);
    else if (a.nviews == 3 && a.codes == CODES_FLIPS) PTB_ACCUM(3, CODES_FLIPS);
    else if (a.nviews == 4 && a.codes == CODES_D2) PTB_ACCUM(4, CODES_D2);
    else if (a.nviews == 8 && a.codes == CODES_D4) PTB_ACCUM(8, CODES_D4);
    else PTB_ACCUM(8, -1);
#undef PTB_ACCUM
#undef PTB_ACCUM_LD
}

static bool has_transpose(int V, int codes) {
    for (int k = 0; k < V; ++k) if ((codes >> (3 * k)) & 1) return true;
    return false;
}
static int count_transpose(int V, int codes) {
    int n = 0;
    for (int k = 0; k < V; ++k) n += (codes >> (3 * k)) & 1;
    return n;
}

static int validate_views(int V, const int* views, int H, int W) {
    if (V < 1 || V > MAX_VIEWS || !views) return PTB_EINVAL;
    int nt = 0;
    for (int k = 0; k < V; ++k) {
        if (views[k] < 0 || views[k] > 7) return PTB_EINVAL;
        nt += views[k] & 1;
    }
    if (nt && H != W) return PTB_EINVAL;
    return PTB_OK;
}

static void fill_reduction(ViewArgs& a, int reduction, int V) {
    a.op = reduction;
    a.divisor = reduction == PTB_RED_SUM ? 1.0f : (float)V;
}

static bool aligned_elems(const void* p, int dtype) {  // 4 source elements per lane: 16 B of fp32, 8 B of fp16 / bf16
    return (reinterpret_cast<uintptr_t>(p) & (dtype == PTB_F32 ? 15u : 7u)) == 0;
}

static int run_plain(ViewArgs& a, int ntiles_out, int mode, hipStream_t s) {
    const bool nonlinear = a.op >= PTB_RED_GMEAN;
    const int nT = mode == MODE_PERVIEW ? 1 : count_transpose(a.nviews, a.codes);
    const bool tr = mode == MODE_PERVIEW ? has_transpose(a.nviews, a.codes) : nT > 0;
    bool fast = !g_force_scalar && (a.W % 4 == 0) && (a.dst_row_stride % 4 == 0) && (a.dst_chan_stride % 4 == 0) &&
                (a.dst_tile_stride % 4 == 0) && aligned_elems(a.src, a.in_dtype) && aligned16(a.dst) && nT <= MAX_T;
    if (tr && a.H % 4 != 0) fast = false;
    if ((long long)a.H * a.W % 4 != 0) fast = false;
    const int ch = fast ? g_chunk_rows : 64;
    if (a.in_dtype != PTB_F32 && (!fast || ch != 32 || mode != MODE_REDUCE)) return PTB_EUNSUPPORTED;
    a.chunks_x = (a.W + CW - 1) / CW;
    a.chunks_y = (a.H + ch - 1) / ch;
    const long long blocks = (long long)ntiles_out * a.C * a.chunks_x * a.chunks_y;
    if (blocks <= 0) return PTB_OK;
    if (blocks > 0x7fffffffLL) return PTB_EUNSUPPORTED;
    if (!fast) {
        if (mode == MODE_PERVIEW) hipLaunchKernelGGL(view_plain_scalar_kernel<MODE_PERVIEW>, dim3((unsigned)blocks), dim3(256), 0, s, a);
        else hipLaunchKernelGGL(view_plain_scalar_kernel<MODE_REDUCE>, dim3((unsigned)blocks), dim3(256), 0, s, a);
        return check_launch();
    }
    if (mode == MODE_PERVIEW) {
        if (ch == 64) launch_plain_ch<64, MODE_PERVIEW>(a, (int)blocks, s, false);
        else if (ch == 32) launch_plain_ch<32, MODE_PERVIEW>(a, (int)blocks, s, false);
        else launch_plain_ch<16, MODE_PERVIEW>(a, (int)blocks, s, false);
    } else {
        if (ch == 64) launch_plain_ch<64, MODE_REDUCE>(a, (int)blocks, s, nonlinear);
        else if (ch == 32) launch_plain_ch<32, MODE_REDUCE>(a, (int)blocks, s, nonlinear);
        else launch_plain_ch<16, MODE_REDUCE>(a, (int)blocks, s, nonlinear);
    }
    return check_launch();
}

static int launch_group(const ViewArgs& a, const CellArgs& g, const std::vector<Cell>& cells, const Fresh& fr, bool fast, int ch,
                        hipStream_t s) {
    const long long blocks = (long long)a.total_chunks * a.C;
    if (blocks <= 0) return PTB_OK;
    if (blocks > 0x7fffffffLL) return PTB_EUNSUPPORTED;
    const bool nonlinear = a.op >= PTB_RED_GMEAN;
    if (!a.dst) {  // norm only (C == 1)
        if (!fast) hipLaunchKernelGGL(norm_accum_scalar_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a, g);
        else if (ch == 64) hipLaunchKernelGGL(norm_accum_kernel<64>, dim3((unsigned)blocks), dim3(1024), 0, s, a, g);
        else if (ch == 32) hipLaunchKernelGGL(norm_accum_kernel<32>, dim3((unsigned)blocks), dim3(512), 0, s, a, g);
        else hipLaunchKernelGGL(norm_accum_kernel<16>, dim3((unsigned)blocks), dim3(256), 0, s, a, g);
    } else if (!fast) {
        hipLaunchKernelGGL(view_accum_scalar_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a, g);
    } else if (ch == 64) {
        launch_accum_ch<64>(a, g, (int)blocks, s, nonlinear);
    } else if (ch == 32) {
        launch_accum_ch<32>(a, g, (int)blocks, s, nonlinear);
    } else {
        launch_accum_ch<16>(a, g, (int)blocks, s, nonlinear);
    }
    const int rc = check_launch();
    if (rc == PTB_OK) mark_written(cells, fr);
    return rc;
}

// Accumulate a run of tiles [lo, hi) of the batch; splits recursively until each launch group decomposes.
static int run_accum(ViewArgs& a, const int* xs, const int* ys, int lo, int hi, bool fast, int ch, const Fresh& fr, hipStream_t s) {
    if (lo >= hi) return PTB_OK;
    CellArgs g;
    int ids[MAX_GROUP];
    std::vector<Cell> cells;
    const int n = hi - lo;
    int st = n <= MAX_GROUP ? DECOMP_OK : DECOMP_SPLIT;
    if (st == DECOMP_OK) {
        for (int t = 0; t < n; ++t) ids[t] = lo + t;
        st = decompose(xs + lo, ys + lo, ids, n, a.W, a.H, ch, fr, g, a.ncells, a.total_chunks, cells);
    }
    if (st == DECOMP_NEEDS_ZERO) return PTB_EFRESH;
    if (st == DECOMP_SPLIT) {
        if (n == 1) return PTB_EUNSUPPORTED;  // cannot happen: one tile is one cell
        const int mid = lo + n / 2;
        const int rc = run_accum(a, xs, ys, lo, mid, fast, ch, fr, s);
        return rc ? rc : run_accum(a, xs, ys, mid, hi, fast, ch, fr, s);
    }
    return launch_group(a, g, cells, fr, fast, ch, s);
}

// Dry run of run_accum's grouping on a scratch bitmap: PTB_EFRESH if any launch group would need a zero-fill.
static int probe_accum(const int* xs, const int* ys, int lo, int hi, int tw, int th, int ch, Fresh& fr,
                       std::vector<Cell>* record = nullptr, std::vector<int>* record_group = nullptr, int* group_counter = nullptr) {
    if (lo >= hi) return PTB_OK;
    CellArgs g;
    int ids[MAX_GROUP];
    std::vector<Cell> cells;
    const int n = hi - lo;
    int nc = 0, tc = 0;
    int st = n <= MAX_GROUP ? DECOMP_OK : DECOMP_SPLIT;
    if (st == DECOMP_OK) {
        for (int t = 0; t < n; ++t) ids[t] = lo + t;
        st = decompose(xs + lo, ys + lo, ids, n, tw, th, ch, fr, g, nc, tc, cells);
    }
    if (st == DECOMP_NEEDS_ZERO) return PTB_EFRESH;
    if (st == DECOMP_SPLIT) {
        if (n == 1) return PTB_EUNSUPPORTED;
        const int mid = lo + n / 2;
        const int rc = probe_accum(xs, ys, lo, mid, tw, th, ch, fr, record, record_group, group_counter);
        return rc ? rc : probe_accum(xs, ys, mid, hi, tw, th, ch, fr, record, record_group, group_counter);
    }
    if (record) {
        for (int i = 0; i < nc; ++i) {
            Cell c = g.cells[i];
            for (int e = 0; e < c.ntiles; ++e) c.tile[e] = g.tile_id[c.tile[e]];  // batch indices
            record->push_back(c);
            record_group->push_back(*group_counter);
        }
        ++*group_counter;
    }
    mark_written(cells, fr);
    return PTB_OK;
}

static int accumulate_impl(float* image, float* norm, const float* weight, const float* in, int V, const int* views,
                           int reduction, const int64_t* xs64, const int64_t* ys64, int B, int C, int th, int tw, int H, int W,
                           uint8_t* fresh, int fresh_rows, hipStream_t s, int in_dtype = PTB_F32) {
    const int round_src = (in_dtype & PTB_ROUND_SRC) ? 1 : 0;
    in_dtype &= ~PTB_ROUND_SRC;
    if (in_dtype < PTB_F32 || in_dtype > PTB_BF16) return PTB_EINVAL;
    const bool norm_only = !image && !in;  // ptb_norm_accumulate
    if ((!norm_only && (!image || !in)) || (norm_only && !norm) || !weight || !xs64 || !ys64) return PTB_EINVAL;
    if (B < 0 || C < 1 || th < 1 || tw < 1 || H < 1 || W < 1) return PTB_EINVAL;
    if (reduction < PTB_RED_SUM || reduction > PTB_RED_LOG1P) return PTB_EINVAL;
    if (int rc = validate_views(V, views, th, tw)) return rc;
    if (B == 0) return PTB_OK;
    std::vector<int> xs(B), ys(B);
    bool aligned = true;
    for (int b = 0; b < B; ++b) {
        if (xs64[b] < 0 || ys64[b] < 0 || xs64[b] + tw > W || ys64[b] + th > H) return PTB_EBOUNDS;
        xs[b] = (int)xs64[b];
        ys[b] = (int)ys64[b];
        if (xs[b] % 4) aligned = false;
    }
    ViewArgs a{};
    a.src = in; a.dst = image; a.norm = norm; a.weight = weight;
    a.in_dtype = in_dtype;
    a.round_src = round_src;
    a.H = th; a.W = tw; a.C = C;
    a.src_view_stride = (long long)B * C * th * tw;
    a.src_tile_stride = (long long)C * th * tw;
    a.dst_tile_stride = 0;
    a.dst_chan_stride = (long long)H * W;
    a.dst_row_stride = W;
    a.nviews = V;
    a.codes = pack_runtime(V, views);
    a.scale = 1.0f;
    fill_reduction(a, reduction, V);
    const int nT = count_transpose(V, a.codes);
    bool fast = !g_force_scalar && aligned && (tw % 4 == 0) && (W % 4 == 0) && ((long long)H * W % 4 == 0) &&
                ((long long)th * tw % 4 == 0) && (norm_only || (aligned_elems(in, in_dtype) && aligned16(image))) && aligned16(norm) &&
                aligned16(weight) && nT <= MAX_T;
    if (nT) {  // transposed source blocks are addressed by tile-local rows: need 4-aligned row offsets too
        if (th % 4) fast = false;
        for (int b = 0; b < B && fast; ++b) if ((ys[b] - ys[0]) % 4) fast = false;
    }
    const int ch = fast ? g_chunk_rows : 64;
    if (in_dtype != PTB_F32 && (!fast || ch != 32)) return PTB_EUNSUPPORTED;  // half sources: default vector kernels only
    Fresh fr{fresh, fresh_rows, H, W};
    if (fresh) {
        if (fresh_rows < 1) return PTB_EINVAL;
        if (fresh_rows != ch) return PTB_EFRESH;  // bitmap granularity differs from this launch's chunk rows
        // a batch that is split into several launches may have been partly applied when a later part needs zeroing:
        // decide on a scratch copy of the bitmap first (skipped in the common case of a batch that is one launch group,
        // where run_accum itself reports PTB_EFRESH before anything is launched)
        if (B > MAX_GROUP || B > 16) {
            std::vector<uint8_t> probe(fresh, fresh + (size_t)fr.nbx() * fr.nby());
            Fresh pf{probe.data(), fresh_rows, H, W};
            const int rc = probe_accum(xs.data(), ys.data(), 0, B, tw, th, ch, pf);
            if (rc) return rc;
        } else {
            CellArgs g;
            int ids[MAX_GROUP], nc = 0, tc = 0;
            std::vector<Cell> cells;
            for (int t = 0; t < B; ++t) ids[t] = t;
            const int st = decompose(xs.data(), ys.data(), ids, B, tw, th, ch, fr, g, nc, tc, cells);
            if (st == DECOMP_NEEDS_ZERO) return PTB_EFRESH;
            if (st == DECOMP_SPLIT) {  // rare: several groups after all -> full dry run
                std::vector<uint8_t> probe(fresh, fresh + (size_t)fr.nbx() * fr.nby());
                Fresh pf{probe.data(), fresh_rows, H, W};
                const int rc = probe_accum(xs.data(), ys.data(), 0, B, tw, th, ch, pf);
                if (rc) return rc;
            } else {  // one group: launch it right away with the plan we already have
                a.ncells = nc; a.total_chunks = tc;
                return launch_group(a, g, cells, fr, fast, ch, s);
            }
        }
    }
    return run_accum(a, xs.data(), ys.data(), 0, B, fast, ch, fr, s);
}

}  // namespace ptb

using namespace ptb;

extern "C" int ptb_tile_accumulate(float* image, float* norm, const float* weight, const float* tiles, const int64_t* xs,
                                   const int64_t* ys, int B, int C, int th, int tw, int H, int W, uint8_t* fresh, int fresh_rows,
                                   ptb_stream_t stream) {
    const int ident = PTB_VIEW_IDENT;
    return accumulate_impl(image, norm, weight, tiles, 1, &ident, PTB_RED_SUM, xs, ys, B, C, th, tw, H, W, fresh, fresh_rows,
                           (hipStream_t)stream);
}

// Planned variant of ptb_tile_accumulate / ptb_deaug_accumulate: one launch group per call, see the header.
extern "C" int ptb_accumulate_planned(float* image, const float* norm_full, float* merged, const float* weight, const void* in,
                                      int in_dtype, int V, const int* views, int reduction, const int64_t* xs64, const int64_t* ys64, int B, int C, int th,
                                      int tw, int H, int W, uint8_t* fresh, int fresh_rows, uint8_t* remaining, uint8_t* done,
                                      ptb_stream_t stream) {
    return ptb_accumulate_planned2(image, norm_full, merged, weight, in, in_dtype, V, views, reduction, xs64, ys64, B, C, th, tw, H, W, fresh, fresh_rows,
                                   remaining, done, 0, stream);
}

// flags bit 0 (PTB_PLANNED_KEEP_SUMS): a block that is finalised (merged = sum / norm written) ALSO stores its weighted sum in
// `image`, so the accumulator stays complete and exact at any time (+ one store of the image per image): what a merger that planned
// ITSELF from the previous image's crop sequence uses -- its caller never asked for a plan and may read `.image` whenever it likes.
extern "C" int ptb_accumulate_planned2(float* image, const float* norm_full, float* merged, const float* weight, const void* in,
                                       int in_dtype, int V, const int* views, int reduction, const int64_t* xs64, const int64_t* ys64, int B, int C, int th,
                                       int tw, int H, int W, uint8_t* fresh, int fresh_rows, uint8_t* remaining, uint8_t* done, int flags,
                                       ptb_stream_t stream) {
    if (flags & ~1) return PTB_EINVAL;
    const int round_src = (in_dtype & PTB_ROUND_SRC) ? 1 : 0;
    in_dtype &= ~PTB_ROUND_SRC;
    if (!image || !norm_full || !merged || !weight || !in || !xs64 || !ys64 || !remaining || !done) return PTB_EINVAL;
    if (B < 0 || C < 1 || th < 1 || tw < 1 || H < 1 || W < 1 || fresh_rows < 1) return PTB_EINVAL;
    if (reduction < PTB_RED_SUM || reduction > PTB_RED_LOG1P || in_dtype < PTB_F32 || in_dtype > PTB_BF16) return PTB_EINVAL;
    if (int rc = validate_views(V, views, th, tw)) return rc;
    if (B == 0) return PTB_OK;
    if (B > 16) return PTB_EUNSUPPORTED;
    int xs[16], ys[16], ids[16];
    for (int b = 0; b < B; ++b) {
        if (xs64[b] < 0 || ys64[b] < 0 || xs64[b] + tw > W || ys64[b] + th > H) return PTB_EBOUNDS;
        xs[b] = (int)xs64[b]; ys[b] = (int)ys64[b]; ids[b] = b;
        if (xs[b] % CW || ys[b] % fresh_rows) return PTB_EUNSUPPORTED;
    }
    ViewArgs a{};
    a.src = static_cast<const float*>(in); a.dst = image; a.norm = nullptr; a.weight = weight; a.merged = merged; a.norm_full = norm_full;
    a.in_dtype = in_dtype;
    a.round_src = round_src;
    a.keep_acc = flags & 1;
    a.H = th; a.W = tw; a.C = C;
    a.src_view_stride = (long long)B * C * th * tw;
    a.src_tile_stride = (long long)C * th * tw;
    a.dst_chan_stride = (long long)H * W;
    a.dst_row_stride = W;
    a.nviews = V;
    a.codes = pack_runtime(V, views);
    a.scale = 1.0f;
    fill_reduction(a, reduction, V);
    const int nT = count_transpose(V, a.codes);
    const bool fast = !g_force_scalar && (tw % 4 == 0) && (W % 4 == 0) && ((long long)H * W % 4 == 0) && ((long long)th * tw % 4 == 0) &&
                      aligned_elems(in, in_dtype) && aligned16(image) && aligned16(merged) && aligned16(norm_full) && aligned16(weight) &&
                      nT <= MAX_T && (!nT || th % 4 == 0);
    const int ch = fast ? g_chunk_rows : 64;
    if (in_dtype != PTB_F32 && (!fast || ch != 32)) return PTB_EUNSUPPORTED;
    if (fresh_rows != ch) return PTB_EUNSUPPORTED;  // plan / first-touch block rows must equal this launch's chunk rows
    Fresh fr{fresh, fresh_rows, H, W};
    Plan pl{remaining, done, fresh_rows, H, W};
    CellArgs g;
    std::vector<Cell> cells;
    const int st = decompose(xs, ys, ids, B, tw, th, ch, fr, g, a.ncells, a.total_chunks, cells, &pl);
    if (st == DECOMP_NEEDS_ZERO) return PTB_EFRESH;
    if (st != DECOMP_OK) return PTB_EUNSUPPORTED;   // several launch groups, or the batch does not fit the plan
    const int rc = launch_group(a, g, cells, fr, fast, ch, (hipStream_t)stream);
    if (rc == PTB_OK) commit_plan(cells, pl);
    return rc;
}

static void launch_band(const ViewArgs& a, const BandArgs& g, int blocks, hipStream_t s) {
    const dim3 grid(blocks), block(512);
    const bool nonlinear = a.op >= PTB_RED_GMEAN;
#define PTB_BAND_LD(NV, CODES, LD)                                                                             \
    do {                                                                                                       \
        if (nonlinear) hipLaunchKernelGGL((band_merge_kernel<NV, CODES, 1, LD>), grid, block, 0, s, a, g);     \
        else hipLaunchKernelGGL((band_merge_kernel<NV, CODES, 0, LD>), grid, block, 0, s, a, g);               \
    } while (0)
#define PTB_BAND(NV, CODES)                                                                                    \
    do {                                                                                                       \
        if (a.in_dtype == PTB_F16) PTB_BAND_LD(NV, CODES, 2);                                                  \
        else if (a.in_dtype == PTB_BF16) PTB_BAND_LD(NV, CODES, 3);                                            \
        else PTB_BAND_LD(NV, CODES, 1);                                                                        \
    } while (0)
    if (a.nviews == 1 && a.codes == CODES_ID) PTB_BAND(1, CODES_ID);
    else if (a.nviews == 2 && a.codes == CODES_FLIPLR) PTB_BAND(2, CODES_FLIPLR);
    else if (a.nviews == 2 && a.codes == CODES_FLIPUD) PTB_BAND(2, CODES_FLIPUD);
    else if (a.nviews == 3 && a.codes == CODES_FLIPS) PTB_BAND(3, CODES_FLIPS);
    else if (a.nviews == 4 && a.codes == CODES_D2) PTB_BAND(4, CODES_D2);
    else if (a.nviews == 8 && a.codes == CODES_D4) PTB_BAND(8, CODES_D4);
    else PTB_BAND(8, -1);
#undef PTB_BAND
#undef PTB_BAND_LD
}

extern "C" int ptb_merge_band(float* merged, const float* norm_full, const float* weight, const void* const* tile_src,
                              const int64_t* tile_view_stride, int in_dtype, int V, const int* views, int reduction,
                              const int64_t* xs64, const int64_t* ys64, int n, int C, int th, int tw, int H, int W, int y0, int y1,
                              ptb_stream_t stream) {
    const int round_src = (in_dtype & PTB_ROUND_SRC) ? 1 : 0;
    in_dtype &= ~PTB_ROUND_SRC;
    if (!merged || !norm_full || !weight || !tile_src || !tile_view_stride || !xs64 || !ys64) return PTB_EINVAL;
    if (n < 1 || C < 1 || th < 1 || tw < 1 || H < 1 || W < 1 || y0 < 0 || y1 <= y0 || y1 > H) return PTB_EINVAL;
    if (reduction < PTB_RED_SUM || reduction > PTB_RED_LOG1P || in_dtype < PTB_F32 || in_dtype > PTB_BF16) return PTB_EINVAL;
    if (int rc = validate_views(V, views, th, tw)) return rc;
    if (n > BAND_TILES) return PTB_EUNSUPPORTED;
    ViewArgs a{};
    a.weight = weight; a.merged = merged; a.norm_full = norm_full;
    a.in_dtype = in_dtype;
    a.round_src = round_src;
    a.H = th; a.W = tw; a.C = C;
    a.dst_chan_stride = (long long)H * W;
    a.dst_row_stride = W;
    a.nviews = V;
    a.codes = pack_runtime(V, views);
    a.scale = 1.0f;
    fill_reduction(a, reduction, V);
    const int nT = count_transpose(V, a.codes);
    bool fast = !g_force_scalar && (tw % 4 == 0) && (th % 4 == 0) && (W % 4 == 0) && (y0 % 4 == 0) && (y1 % 4 == 0) &&
                ((long long)H * W % 4 == 0) && aligned16(merged) && aligned16(norm_full) && aligned16(weight) && nT <= MAX_T;
    BandArgs g{};
    for (int t = 0; t < n; ++t) {
        if (xs64[t] < 0 || ys64[t] < 0 || xs64[t] + tw > W || ys64[t] + th > H) return PTB_EBOUNDS;
        if (ys64[t] > y0 || ys64[t] + th < y1) return PTB_EINVAL;   // every tile of a band covers all of its rows
        if (!tile_src[t] || tile_view_stride[t] < (long long)C * th * tw) return PTB_EINVAL;
        fast = fast && xs64[t] % 4 == 0 && ys64[t] % 4 == 0 && aligned_elems(tile_src[t], in_dtype) && tile_view_stride[t] % 4 == 0;
        g.tile_x[t] = (int)xs64[t]; g.tile_y[t] = (int)ys64[t];
        g.tile_src[t] = tile_src[t];
        g.tile_vs[t] = tile_view_stride[t];
    }
    if (!fast) return PTB_EUNSUPPORTED;   // the caller falls back to the incremental path
    std::vector<int> xe;
    for (int t = 0; t < n; ++t) { xe.push_back(g.tile_x[t]); xe.push_back(g.tile_x[t] + tw); }
    std::sort(xe.begin(), xe.end());
    xe.erase(std::unique(xe.begin(), xe.end()), xe.end());
    std::vector<BandCell> cells;
    for (size_t xi = 0; xi + 1 < xe.size(); ++xi) {
        BandCell c{};
        c.ox = xe[xi]; c.oy = y0; c.w = xe[xi + 1] - xe[xi]; c.h = y1 - y0;
        for (int t = 0; t < n; ++t) {
            if (g.tile_x[t] <= c.ox && c.ox < g.tile_x[t] + tw) {
                if (c.ntiles == MAX_COVER) return PTB_EUNSUPPORTED;
                c.tile[c.ntiles++] = t;     // ascending t = the order the tiles were integrated in
            }
        }
        if (!c.ntiles) continue;
        if (!cells.empty() && cells.back().ox + cells.back().w == c.ox && cells.back().ntiles == c.ntiles &&
            std::equal(c.tile, c.tile + c.ntiles, cells.back().tile)) {
            cells.back().w += c.w;   // same cover as the strip to the left: one wider cell
            continue;
        }
        cells.push_back(c);
    }
    if (cells.empty() || (int)cells.size() > BAND_CELLS) return PTB_EUNSUPPORTED;
    std::stable_sort(cells.begin(), cells.end(), [](const BandCell& l, const BandCell& r) { return l.ntiles > r.ntiles; });
    int run = 0;
    for (size_t i = 0; i < cells.size(); ++i) {
        run += ((cells[i].w + CW - 1) / CW) * ((cells[i].h + 31) / 32);
        cells[i].chunk_end = run;
        g.cells[i] = cells[i];
    }
    a.ncells = (int)cells.size();
    a.total_chunks = run;
    const long long blocks = (long long)run * C;
    if (blocks > 0x7fffffffLL) return PTB_EUNSUPPORTED;
    launch_band(a, g, (int)blocks, (hipStream_t)stream);
    return check_launch();
}

extern "C" int ptb_norm_accumulate(float* norm, const float* weight, const int64_t* xs, const int64_t* ys, int B, int th, int tw,
                                   int H, int W, uint8_t* fresh, int fresh_rows, ptb_stream_t stream) {
    const int ident = PTB_VIEW_IDENT;
    return accumulate_impl(nullptr, norm, weight, nullptr, 1, &ident, PTB_RED_SUM, xs, ys, B, 1, th, tw, H, W, fresh, fresh_rows,
                           (hipStream_t)stream);
}

extern "C" int ptb_deaug_accumulate(float* image, float* norm, const float* weight, const float* in, int V, const int* views,
                                    int reduction, const int64_t* xs, const int64_t* ys, int B, int C, int th, int tw, int H,
                                    int W, uint8_t* fresh, int fresh_rows, ptb_stream_t stream) {
    return accumulate_impl(image, norm, weight, in, V, views, reduction, xs, ys, B, C, th, tw, H, W, fresh, fresh_rows,
                           (hipStream_t)stream);
}

extern "C" int ptb_deaug_accumulate_t(float* image, float* norm, const float* weight, const void* in, int in_dtype, int V,
                                      const int* views, int reduction, const int64_t* xs, const int64_t* ys, int B, int C, int th,
                                      int tw, int H, int W, uint8_t* fresh, int fresh_rows, ptb_stream_t stream) {
    return accumulate_impl(image, norm, weight, static_cast<const float*>(in), V, views, reduction, xs, ys, B, C, th, tw, H, W, fresh,
                           fresh_rows, (hipStream_t)stream, in_dtype);
}

static int deaug_reduce_impl(const float* in, int in_dtype, float* out, int V, const int* views, int reduction, int B, int C, int H,
                             int W, ptb_stream_t stream);

extern "C" int ptb_deaug_reduce(const float* in, float* out, int V, const int* views, int reduction, int B, int C, int H, int W,
                                ptb_stream_t stream) {
    return deaug_reduce_impl(in, PTB_F32, out, V, views, reduction, B, C, H, W, stream);
}

extern "C" int ptb_deaug_reduce_t(const void* in, int in_dtype, float* out, int V, const int* views, int reduction, int B, int C,
                                  int H, int W, ptb_stream_t stream) {
    if (in_dtype < PTB_F32 || in_dtype > PTB_BF16) return PTB_EINVAL;
    return deaug_reduce_impl(static_cast<const float*>(in), in_dtype, out, V, views, reduction, B, C, H, W, stream);
}

static int deaug_reduce_impl(const float* in, int in_dtype, float* out, int V, const int* views, int reduction, int B, int C, int H,
                             int W, ptb_stream_t stream) {
    if (!in || !out || B < 0 || C < 1 || H < 1 || W < 1) return PTB_EINVAL;
    if (reduction < PTB_RED_SUM || reduction > PTB_RED_LOG1P) return PTB_EINVAL;
    if (int rc = validate_views(V, views, H, W)) return rc;
    if (B == 0) return PTB_OK;
    ViewArgs a{};
    a.src = in; a.dst = out;
    a.in_dtype = in_dtype;
    a.H = H; a.W = W; a.C = C;
    a.src_view_stride = (long long)B * C * H * W;
    a.src_tile_stride = (long long)C * H * W;
    a.dst_tile_stride = (long long)C * H * W;
    a.dst_chan_stride = (long long)H * W;
    a.dst_row_stride = W;
    a.nviews = V;
    a.codes = pack_runtime(V, views);
    a.scale = 1.0f;
    fill_reduction(a, reduction, V);
    return run_plain(a, B, MODE_REDUCE, (hipStream_t)stream);
}

extern "C" int ptb_view_transform(const float* in, float* out, int V, const int* views, int in_is_batch, float scale, int B,
                                  int C, int H, int W, ptb_stream_t stream) {
    if (!in || !out || B < 0 || C < 1 || H < 1 || W < 1) return PTB_EINVAL;
    if (int rc = validate_views(V, views, H, W)) return rc;
    if (B == 0) return PTB_OK;
    ViewArgs a{};
    a.src = in; a.dst = out;
    a.H = H; a.W = W; a.C = C;
    a.src_view_stride = 0;
    a.src_tile_stride = (long long)C * H * W;
    a.dst_tile_stride = (long long)C * H * W;
    a.dst_chan_stride = (long long)H * W;
    a.dst_row_stride = W;
    a.nviews = V;
    a.codes = pack_runtime(V, views);
    a.tiles_per_view = B;
    a.src_tile_mod = in_is_batch ? B : V * B;
    a.scale = scale;
    a.op = PTB_RED_SUM;
    a.divisor = 1.0f;
    const bool tr = has_transpose(V, a.codes);
    if (in_is_batch && V > 1 && !g_force_scalar && W % 4 == 0 && (!tr || H % 4 == 0) && aligned16(in) && aligned16(out)) {
        const int ch = g_chunk_rows;
        a.chunks_x = (W + CW - 1) / CW;
        a.chunks_y = (H + ch - 1) / ch;
        const long long blocks = (long long)B * C * a.chunks_x * a.chunks_y;
        if (blocks > 0x7fffffffLL) return PTB_EUNSUPPORTED;
        hipStream_t s = (hipStream_t)stream;
        if (ch == 64) hipLaunchKernelGGL((view_scatter_kernel<64, false>), dim3((unsigned)blocks), dim3(1024), 0, s, a, B);
        else if (ch == 32) hipLaunchKernelGGL((view_scatter_kernel<32, false>), dim3((unsigned)blocks), dim3(512), 0, s, a, B);
        else hipLaunchKernelGGL((view_scatter_kernel<16, false>), dim3((unsigned)blocks), dim3(256), 0, s, a, B);
        return check_launch();
    }
    return run_plain(a, V * B, MODE_PERVIEW, (hipStream_t)stream);
}

// Test hook (no device work): the launch plan ptb_tile_accumulate / ptb_deaug_accumulate would use for this batch.
// out receives up to `cap` records of 12 ints: group, ox, oy, w, h, fresh, chunk_end, ntiles, tile[4] (batch indices).
// Returns the number of cells (>= 0) or a PTB_E* code; `fresh` (optional) is updated exactly as a real call would.
extern "C" int ptb_debug_plan(const int64_t* xs64, const int64_t* ys64, int B, int th, int tw, int H, int W, int chunk_rows,
                              uint8_t* fresh, int fresh_rows, int* out, int cap) {
    if (!xs64 || !ys64 || B < 0 || th < 1 || tw < 1 || H < 1 || W < 1 || chunk_rows < 1 || !out) return PTB_EINVAL;
    std::vector<int> xs(B), ys(B);
    for (int b = 0; b < B; ++b) {
        if (xs64[b] < 0 || ys64[b] < 0 || xs64[b] + tw > W || ys64[b] + th > H) return PTB_EBOUNDS;
        xs[b] = (int)xs64[b];
        ys[b] = (int)ys64[b];
    }
    Fresh fr{fresh, fresh_rows, H, W};
    if (fresh && fresh_rows != chunk_rows) return PTB_EFRESH;
    std::vector<uint8_t> probe;
    Fresh pf = fr;
    if (fresh) {  // decide on a scratch copy first, like accumulate_impl
        probe.assign(fresh, fresh + (size_t)fr.nbx() * fr.nby());
        pf.map = probe.data();
        if (int rc = probe_accum(xs.data(), ys.data(), 0, B, tw, th, chunk_rows, pf)) return rc;
    }
    std::vector<Cell> cells;
    std::vector<int> groups;
    int counter = 0;
    if (int rc = probe_accum(xs.data(), ys.data(), 0, B, tw, th, chunk_rows, fr, &cells, &groups, &counter)) return rc;
    const int n = (int)cells.size();
    for (int i = 0; i < n && i < cap; ++i) {
        int* o = out + 12 * i;
        const Cell& c = cells[i];
        o[0] = groups[i]; o[1] = c.ox; o[2] = c.oy; o[3] = c.w; o[4] = c.h; o[5] = c.fresh; o[6] = c.chunk_end; o[7] = c.ntiles;
        for (int e = 0; e < MAX_COVER; ++e) o[8 + e] = e < c.ntiles ? c.tile[e] : -1;
    }
    return n;
}

// Backward of ptb_deaug_reduce for the non-linear reductions (gmean, hmean, harmonic1p, logodd, log1p):
//   grad_in[k*B+b] = view_k^-1( grad_out * post'(m) / V ) * pre'(in[k*B+b])
extern "C" int ptb_deaug_reduce_bwd(const float* in, const float* out, const float* grad_out, float* grad_in, int V, const int* views,
                                    int reduction, int B, int C, int H, int W, ptb_stream_t stream) {
    if (!in || !out || !grad_out || !grad_in || B < 0 || C < 1 || H < 1 || W < 1) return PTB_EINVAL;
    if (reduction < PTB_RED_GMEAN || reduction > PTB_RED_LOG1P) return PTB_EINVAL;
    if (int rc = validate_views(V, views, H, W)) return rc;
    if (B == 0) return PTB_OK;
    static const int inverse[8] = {0, 1, 2, 5, 4, 3, 6, 7};  // the two quarter turns swap, the rest are involutions
    int inv[MAX_VIEWS];
    for (int k = 0; k < V; ++k) inv[k] = inverse[views[k]];
    ViewArgs a{};
    a.src = grad_out; a.dst = grad_in; a.norm = const_cast<float*>(out); a.weight = in;
    a.H = H; a.W = W; a.C = C;
    a.nviews = V;
    a.codes = pack_runtime(V, inv);
    a.scale = 1.0f;
    a.op = reduction;
    a.divisor = (float)V;
    hipStream_t s = (hipStream_t)stream;
    const bool tr = has_transpose(V, a.codes);
    const bool fast = !g_force_scalar && W % 4 == 0 && (!tr || H % 4 == 0) && aligned16(in) && aligned16(out) &&
                      aligned16(grad_out) && aligned16(grad_in);
    if (fast) {
        const int ch = g_chunk_rows;
        a.chunks_x = (W + CW - 1) / CW;
        a.chunks_y = (H + ch - 1) / ch;
        const long long blocks = (long long)B * C * a.chunks_x * a.chunks_y;
        if (blocks > 0x7fffffffLL) return PTB_EUNSUPPORTED;
        if (ch == 64) hipLaunchKernelGGL((view_scatter_kernel<64, true>), dim3((unsigned)blocks), dim3(1024), 0, s, a, B);
        else if (ch == 32) hipLaunchKernelGGL((view_scatter_kernel<32, true>), dim3((unsigned)blocks), dim3(512), 0, s, a, B);
        else hipLaunchKernelGGL((view_scatter_kernel<16, true>), dim3((unsigned)blocks), dim3(256), 0, s, a, B);
    } else {
        const long long n = (long long)B * C * H * W;
        const long long want = (n + 255) / 256;
        hipLaunchKernelGGL(deaug_bwd_scalar_kernel, dim3((unsigned)(want < 8192 ? want : 8192)), dim3(256), 0, s, a, B);
    }
    return check_launch();
}
