// ptb_bandplan.hip -- deferred planned merging of a whole image, planned once, driven from C (gfx950 / MI355X).
//
// TileMerger(crops=tiler.crops, defer=True) (reference data flow: inference/tta.py:442-467 feeding inference/tiles.py:321-346).
// The crop list of an image is known before its first batch, so everything that depends on geometry only is computed ONCE per
// merger: the image is cut into *bands* (the rows between two consecutive tile edges; every tile that touches a band covers all
// its rows), consecutive bands are grouped into *launch groups* of about `rows_per_launch` rows, and every group is expanded into
// a table of 64 x 32 work items -- origin, extent, the (<= 4) covering tiles in integration order and the item's offset inside
// each of them.  The table lives in HBM (uploaded once); per image only the tile source pointers change, and those travel in the
// kernel arguments of the group's launch (<= 224 tiles x 16 B).  `ptb_band_plan_submit` takes one batch of model outputs into the
// plan and launches every group whose last tile has arrived: one launch reads all covering tiles of its rows once, applies the
// inverse TTA views, reduces, blends in integration order and writes sum / norm to the merged map.  No accumulator image ever
// exists in HBM, the per-batch host work is a few table look-ups in C (no Python planning), and the fp32 operation order per
// pixel is the incremental path's, so the result is bit-identical to it -- and to the reference's sequential loop.
//
// Why groups of several bands: a launch of one 256-row band at the headline geometry is 2560 workgroups = 3.3 rounds of the
// 768 that fit the chip, so ramp-up, the partial last round and the kernel boundary cost ~10 % of its 100 us; a group of 4 bands
// pays them once per 400 us.  Items are ordered heavy-first (4-tile items before 2- and 1-tile ones) so the tail is short work.
#include <algorithm>
#include <vector>

#include "ptb_view_device.h"

namespace ptb {

constexpr int PLAN_TILES = 224;   // tiles of one launch group (kernarg: 224 x 16 B + ViewArgs < 4 KiB)
constexpr int PLAN_CH = 32;       // item rows of the 512-thread instance; plans are created with 32 or 64 rows per item (ptb_set_tunable key 11)

struct BandItem {                 // 64 B = one cache line per workgroup, read with scalar loads only (never copied to registers as a
    int ax, ay;                   // whole: run-time indexing of a by-value copy would put it in scratch memory); (ax, ay) = origin
    int cwch;                     // extent: columns | rows << 16  (<= 64 x 64)
    int ntiles;                   // covering tiles (0: uncovered pixels -> 0 / 0 = NaN like the reference's merge)
    int partial;                  // 1: write the un-normalised weighted sum (multi-GPU boundary rows) instead of sum / norm
    int pad[3];
    unsigned long long cover[MAX_COVER];   // ascending integration order: tile slot | lx << 16 | ly << 32 (item origin in the tile)
};
static_assert(sizeof(BandItem) == 64, "BandItem layout");

struct GroupTiles {
    const void* src[PLAN_TILES];  // view 0, channel 0 of the tile
    long long vs[PLAN_TILES];     // elements between consecutive views of this tile (its batch size * C * th * tw)
};

// PF (round 5): the loads of covering tile e + 1 are requested -- as they lie in memory: 16 registers for the eight d4 views of a half /
// bf16 source, 32 for fp32 -- before tile e is transposed, reduced and blended.  Without it a workgroup has nothing in flight while it
// works through its two barriers; with it the instances need 73-96 registers and one 1024-thread workgroup fits a CU instead of two,
// and that still wins: 5000 x 5000, d4, C = 4 per image 1.49 -> 1.34 ms for half / bf16 model outputs (54 -> 60 % of 8 TB/s for their
// 6.5 GB) and 2.14 -> 2.05 / 2.20 -> 2.16 ms for fp32 (two boxes); two tiles ahead loses (half 1.34 -> 1.70 ms, fp32 128 registers:
// no gain) -- profiles/HISTORY.md section 9.5.  Instances whose view codes are read at run time keep the plain loop (they would spill).
template <int NV, int CODES, int OPK, int LD, int CH = PLAN_CH, bool PF = false>
__global__ __launch_bounds__(16 * CH) void band_plan_kernel(const ViewArgs a, const BandItem* __restrict__ items, const GroupTiles t) {
    // DB (round 6): the prefetching instances run ONE 1024-thread workgroup per CU, so the transposing views' LDS tiles can be doubled
    // (2 x 64 KiB of the CU's 160): covering tile e + 1 is scattered into the other set while slow waves still read set e, and the
    // second barrier of every covering tile -- "the tiles are reused" -- goes away.  PMC on bf16 sources had the waves of this kernel
    // 20 % of their time in barriers (SQ_WAIT_ANY - SQ_WAIT_INST_ANY, profiles/r06_band_half_pmc_raw.txt).
    constexpr bool DB = PF && CH == 64 && lds_tiles(NV, CODES) > 0;
    constexpr int LDS_SET = lds_tiles(NV, CODES) * CW * CH;
    __shared__ __attribute__((aligned(16))) float lds[LDS_SET ? LDS_SET * (DB ? 2 : 1) : 4];
    const int tid = threadIdx.x;
    unsigned bid = blockIdx.x;
    if (a.ncells == 1) {   // XCD-aware order (A/B, ptb_set_tunable key 10): every XCD walks a contiguous eighth of the (item, channel) list
        const unsigned per_xcd = ((unsigned)a.total_chunks + 7u) / 8u;
        bid = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
        if (bid >= (unsigned)a.total_chunks) return;
    }
    int c = bid % a.C;
    unsigned item = bid / a.C;
    if (a.chan_loop) { c = 0; item = bid; }      // (one workgroup per item, all channels: the identity-view instances)
    else if (a.ncells == 2) {   // channel-major: consecutive workgroups = neighbouring chunks of ONE channel plane
        const unsigned n_items = (unsigned)a.total_chunks / (unsigned)a.C;
        c = bid / n_items;
        item = bid - c * n_items;
    }
    const BandItem* __restrict__ it = items + item;
    const int cwch = it->cwch, partial = it->partial;
    const int cw = cwch & 0xffff, ch = cwch >> 16;
    const int q = tid & 15, r = tid >> 4;
    const bool act = (r < ch) && (4 * q < cw);
    const long long pix = (long long)(it->ay + r) * a.dst_row_stride + it->ax + 4 * q;
    float4 nfull = make_float4(1.f, 1.f, 1.f, 1.f);
    if (act && !partial) nfull = *reinterpret_cast<const float4*>(a.norm_full + pix);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const int nt = it->ntiles;
    const bool rot = PF && a.rot_views && (item & 1u);      // (wave-uniform; A/B of the view issue order, ptb_set_tunable key 22)
    // ALL (round 6): one or two row-preserving views -- the plain loop without TTA (`integrate_batch(pred, crops)`, tiles.py:321-339), the
    // flips of one axis -- bring only 16-32 bytes per lane and covering tile, so one tile ahead leaves a CU with ~32 KiB in flight and a
    // workgroup's <= 4 covering tiles a chain of exposed latencies (0.45 ms per 5000 x 5000 image: 54 % of its bytes' time).  These
    // instances request EVERY covering tile of the item up front (<= 4 x NV raw values: <= 32 registers), no LDS, no barrier.
    constexpr bool ALL = PF && lds_tiles(NV, CODES) == 0 && NV <= 2 && CODES >= 0;
    if constexpr (ALL) {
        // a.chan_loop (identity view): the workgroup walks ALL channels of its item -- the window values of the covering tiles and the
        // normaliser are per PIXEL, and with one 16-byte load per lane, channel and covering tile they were as many bytes through the
        // L1 as the model outputs themselves (and the normaliser, 105 MB, was re-read per channel).  They are loaded once, the next
        // channel's tiles are requested while the current one is blended.
        const int c_end = a.chan_loop ? a.C : c + 1;
        if (a.chan_loop) { c = 0; }
        float4 wv[MAX_COVER];
#pragma unroll
        for (int e = 0; e < MAX_COVER; ++e) {
            wv[e] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < nt && act) {
                const unsigned long long cv = it->cover[e];
                const int lx = (int)((cv >> 16) & 0xffff), ly = (int)((cv >> 32) & 0xffff);
                wv[e] = *reinterpret_cast<const float4*>(a.weight + (long long)(ly + r) * a.W + lx + 4 * q);
            }
        }
        typename RawOf<LD>::type all[MAX_COVER][NV];
        auto request = [&](int cc) {
#pragma unroll
            for (int e = 0; e < MAX_COVER; ++e) {
                if (e < nt) {
                    const unsigned long long cv = it->cover[e];
                    const int slot = (int)(cv & 0xffff), lx = (int)((cv >> 16) & 0xffff), ly = (int)((cv >> 32) & 0xffff);
                    gather_load_raw<CH, NV, CODES, LD>(static_cast<const float*>(t.src[slot]), (long long)cc * a.H * a.W, t.vs[slot], a.nviews, a.codes, a.H,
                                                       a.W, lx, ly, cw, ch, tid, all[e]);
                }
            }
        };
        request(c);
        for (; c < c_end; ++c) {
            acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int e = 0; e < MAX_COVER; ++e) {
                if (e < nt) {
                    float4 v[NV];
                    gather_widen<CH, NV, CODES, LD>(all[e], a.nviews, a.codes, cw, ch, tid, v);
                    float4 val = gather_tail<CH, NV, CODES, OPK>(v, a.nviews, a.codes, cw, ch, a.op, a.divisor, lds, tid, false);
                    val = round_src4<LD>(val, a.round_src);
                    acc.x = __fadd_rn(acc.x, __fmul_rn(val.x, wv[e].x));   // tiles.py:338
                    acc.y = __fadd_rn(acc.y, __fmul_rn(val.y, wv[e].y));
                    acc.z = __fadd_rn(acc.z, __fmul_rn(val.z, wv[e].z));
                    acc.w = __fadd_rn(acc.w, __fmul_rn(val.w, wv[e].w));
                }
            }
            if (c + 1 < c_end) request(c + 1);      // (the raw registers are free again: the next channel's tiles travel while this one is divided and stored)
            if (act) {
                float* o = a.merged + (long long)c * a.dst_chan_stride + pix;
                if (partial) *reinterpret_cast<float4*>(o) = acc;
                else out_store4(o, make_float4(__fdiv_rn(acc.x, nfull.x), __fdiv_rn(acc.y, nfull.y), __fdiv_rn(acc.z, nfull.z), __fdiv_rn(acc.w, nfull.w)));
            }
        }
        return;
    }
    typename RawOf<LD>::type nxt[PF ? NV : 1];
    // The window values of a covering tile travel WITH its view loads (round 6): loaded after the prefetch of tile e + 1 had been issued --
    // where tile e needs them -- they put an s_waitcnt vmcnt(0) at the end of every iteration (the loads sit behind divergent guards, so
    // the compiler cannot count them), i.e. the workgroup waited for the whole prefetch plus one exposed L2 round trip per covering tile.
    float4 wnxt = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (PF) {
        if (nt > 0) {
            const unsigned long long cv = it->cover[0];
            const int slot = (int)(cv & 0xffff), lx = (int)((cv >> 16) & 0xffff), ly = (int)((cv >> 32) & 0xffff);
            if (act) wnxt = *reinterpret_cast<const float4*>(a.weight + (long long)(ly + r) * a.W + lx + 4 * q);
            if (NV >= 2 && rot)
                gather_load_raw<CH, NV, CODES, LD, NV / 2>(static_cast<const float*>(t.src[slot]), (long long)c * a.H * a.W, t.vs[slot], a.nviews, a.codes, a.H,
                                                           a.W, lx, ly, cw, ch, tid, nxt);
            else
                gather_load_raw<CH, NV, CODES, LD>(static_cast<const float*>(t.src[slot]), (long long)c * a.H * a.W, t.vs[slot], a.nviews, a.codes, a.H, a.W, lx,
                                                   ly, cw, ch, tid, nxt);
        }
    }
    for (int e = 0; e < nt; ++e) {
        const unsigned long long cv = it->cover[e];
        const int slot = (int)(cv & 0xffff), lx = (int)((cv >> 16) & 0xffff), ly = (int)((cv >> 32) & 0xffff);
        float4 val;
        float4 w4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (PF) {
            float4 v[NV];
            gather_widen<CH, NV, CODES, LD>(nxt, a.nviews, a.codes, cw, ch, tid, v);   // (the buffer is free again: tile e + 1 is requested into it)
            w4 = wnxt;
            if (e + 1 < nt) {
                const unsigned long long cn = it->cover[e + 1];
                const int sn = (int)(cn & 0xffff), nlx = (int)((cn >> 16) & 0xffff), nly = (int)((cn >> 32) & 0xffff);
                if (act) wnxt = *reinterpret_cast<const float4*>(a.weight + (long long)(nly + r) * a.W + nlx + 4 * q);
                if (NV >= 2 && rot)
                    gather_load_raw<CH, NV, CODES, LD, NV / 2>(static_cast<const float*>(t.src[sn]), (long long)c * a.H * a.W, t.vs[sn], a.nviews, a.codes, a.H,
                                                               a.W, nlx, nly, cw, ch, tid, nxt);
                else
                    gather_load_raw<CH, NV, CODES, LD>(static_cast<const float*>(t.src[sn]), (long long)c * a.H * a.W, t.vs[sn], a.nviews, a.codes, a.H, a.W,
                                                       nlx, nly, cw, ch, tid, nxt);
            }
            const bool db = DB && a.lds_db;      // (wave-uniform; ptb_set_tunable key 25 for same-process A/B runs)
            val = gather_tail<CH, NV, CODES, OPK>(v, a.nviews, a.codes, cw, ch, a.op, a.divisor, db ? lds + (e & 1) * LDS_SET : lds, tid, db ? false : e + 1 < nt);
        } else {
            val = gather_reduce<CH, NV, CODES, OPK, LD>(static_cast<const float*>(t.src[slot]), (long long)c * a.H * a.W, t.vs[slot],
                                                        a.nviews, a.codes, a.H, a.W, lx, ly, cw, ch, a.op, a.divisor, lds, tid,
                                                        e + 1 < nt);
            if (act) w4 = *reinterpret_cast<const float4*>(a.weight + (long long)(ly + r) * a.W + lx + 4 * q);      // (nothing else is in flight here; held across the views the run-time-code instances would spill)
        }
        val = round_src4<LD>(val, a.round_src);
        if (act) {
            acc.x = __fadd_rn(acc.x, __fmul_rn(val.x, w4.x));   // tiles.py:338: tile * weight rounded, then added (no FMA contraction)
            acc.y = __fadd_rn(acc.y, __fmul_rn(val.y, w4.y));
            acc.z = __fadd_rn(acc.z, __fmul_rn(val.z, w4.z));
            acc.w = __fadd_rn(acc.w, __fmul_rn(val.w, w4.w));
        }
    }
    if (act) {
        float* o = a.merged + (long long)c * a.dst_chan_stride + pix;
        if (partial) *reinterpret_cast<float4*>(o) = acc;          // (partial sums are read back soon: plain store)
        else out_store4(o, make_float4(__fdiv_rn(acc.x, nfull.x), __fdiv_rn(acc.y, nfull.y), __fdiv_rn(acc.z, nfull.z),
                                       __fdiv_rn(acc.w, nfull.w)));   // tiles.py:346
    }
}

static void launch_plan(const ViewArgs& a, const BandItem* items, const GroupTiles& t, int blocks, int ch, hipStream_t s) {
    const dim3 grid(blocks), block(16 * ch);
    const bool nonlinear = a.op >= PTB_RED_GMEAN;
#define PTB_PLAN_PF(NV, CODES, LD)                                                                                  \
    do {                                                                                                            \
        if (ch == 64) {                                                                                             \
            if (nonlinear) hipLaunchKernelGGL((band_plan_kernel<NV, CODES, 1, LD, 64, true>), grid, block, 0, s, a, items, t); \
            else hipLaunchKernelGGL((band_plan_kernel<NV, CODES, 0, LD, 64, true>), grid, block, 0, s, a, items, t);       \
        } else if (nonlinear) hipLaunchKernelGGL((band_plan_kernel<NV, CODES, 1, LD, PLAN_CH, true>), grid, block, 0, s, a, items, t);    \
        else hipLaunchKernelGGL((band_plan_kernel<NV, CODES, 0, LD, PLAN_CH, true>), grid, block, 0, s, a, items, t);              \
    } while (0)
#define PTB_PLAN_LD(NV, CODES, LD)                                                                                  \
    do {                                                                                                            \
        if (ch == 64) {                                                                                             \
            if (nonlinear) hipLaunchKernelGGL((band_plan_kernel<NV, CODES, 1, LD, 64>), grid, block, 0, s, a, items, t); \
            else hipLaunchKernelGGL((band_plan_kernel<NV, CODES, 0, LD, 64>), grid, block, 0, s, a, items, t);       \
        } else if (nonlinear) hipLaunchKernelGGL((band_plan_kernel<NV, CODES, 1, LD>), grid, block, 0, s, a, items, t);    \
        else hipLaunchKernelGGL((band_plan_kernel<NV, CODES, 0, LD>), grid, block, 0, s, a, items, t);              \
    } while (0)
#define PTB_PLAN(NV, CODES)                                                                                         \
    do {                                                                                                            \
        if (a.in_dtype == PTB_F16) { if (g_band_half_pf) PTB_PLAN_PF(NV, CODES, 2); else PTB_PLAN_LD(NV, CODES, 2); }      \
        else if (a.in_dtype == PTB_BF16) { if (g_band_half_pf) PTB_PLAN_PF(NV, CODES, 3); else PTB_PLAN_LD(NV, CODES, 3); } \
        else if (g_band_half_pf >= 2) PTB_PLAN_PF(NV, CODES, 1);                                                    \
        else PTB_PLAN_LD(NV, CODES, 1);                                                                             \
    } while (0)
#define PTB_PLAN_RT(NV, CODES) /* view codes read at run time: no prefetching instance (it would spill) */          \
    do {                                                                                                            \
        if (a.in_dtype == PTB_F16) PTB_PLAN_LD(NV, CODES, 2);                                                       \
        else if (a.in_dtype == PTB_BF16) PTB_PLAN_LD(NV, CODES, 3);                                                 \
        else PTB_PLAN_LD(NV, CODES, 1);                                                                             \
    } while (0)
    if (a.nviews == 1 && a.codes == CODES_ID) PTB_PLAN(1, CODES_ID);
    else if (a.nviews == 2 && a.codes == CODES_FLIPLR) PTB_PLAN(2, CODES_FLIPLR);
    else if (a.nviews == 2 && a.codes == CODES_FLIPUD) PTB_PLAN(2, CODES_FLIPUD);
    else if (a.nviews == 3 && a.codes == CODES_FLIPS) PTB_PLAN(3, CODES_FLIPS);
    else if (a.nviews == 4 && a.codes == CODES_D2) PTB_PLAN(4, CODES_D2);
    else if (a.nviews == 8 && a.codes == CODES_D4) PTB_PLAN(8, CODES_D4);
    else PTB_PLAN_RT(8, -1);
#undef PTB_PLAN
#undef PTB_PLAN_RT
#undef PTB_PLAN_LD
#undef PTB_PLAN_PF
}

struct Group {
    int y0, y1;                 // rows of the merged map this launch writes
    std::vector<int> tiles;     // plan indices, ascending (slot = position)
    int last_tile;              // the group is complete once this plan index is in (-1: no tile at all)
    long long item_off;         // first item in the table
    int item_cnt;
};

}  // namespace ptb

struct ptb_band_plan {
    int n, C, th, tw, H, W;
    int ch = ptb::PLAN_CH;                // rows per work item (ptb_set_tunable key 11 at creation: 32 | 64)
    std::vector<int> xs, ys;
    std::vector<ptb::BandItem> items;
    std::vector<ptb::Group> groups;
    std::vector<int> last_group;          // per tile: the last group that reads it
    std::vector<std::vector<int>> ready;  // per tile: groups complete once it is in
    std::vector<int> band_y0, band_y1, band_group;   // every band's rows and the launch group that writes them
    std::vector<char> group_launched;     // per image
    const ptb::BandItem* dev_items = nullptr;
    int n_bands = 0;
    // per image
    int pos = 0, launched = 0;
    std::vector<const void*> src;
    std::vector<long long> vs;
    int cfg_set = 0, cfg_dtype = 0, cfg_V = 0, cfg_codes = 0, cfg_red = 0;
    int cfg_views[ptb::MAX_VIEWS] = {0};
    const void *cfg_merged = nullptr, *cfg_norm = nullptr, *cfg_weight = nullptr;
    // custody (round 6): the byte ranges of the batches of this image that a later launch group still reads, in integration order --
    // what ptb_band_plan_submit_next checks a new batch against (a model writing into a reused output buffer)
    struct Held { uintptr_t p0, p1; int last_group; };
    std::vector<Held> held;
    bool monotone = true;                 // groups complete in index order (row-major crops): custody is released from the front
};

using namespace ptb;

// x-cells of one band (tiles = plan indices covering it), split into 64 x 32 items appended to `out`
static int band_items(const ptb_band_plan& p, const std::vector<int>& cover, const std::vector<int>& slot_of, int y0, int y1, int partial,
                      std::vector<BandItem>& out) {
    std::vector<int> xe{0, p.W};
    for (int t : cover) { xe.push_back(p.xs[t]); xe.push_back(p.xs[t] + p.tw); }
    std::sort(xe.begin(), xe.end());
    xe.erase(std::unique(xe.begin(), xe.end()), xe.end());
    struct XCell { int ox, w, n; int tile[MAX_COVER]; };
    std::vector<XCell> cells;
    for (size_t xi = 0; xi + 1 < xe.size(); ++xi) {
        XCell c{};
        c.ox = xe[xi]; c.w = xe[xi + 1] - xe[xi];
        for (int t : cover) {   // ascending plan index = the order the tiles are integrated in
            if (p.xs[t] <= c.ox && c.ox < p.xs[t] + p.tw) {
                if (c.n == MAX_COVER) return PTB_EUNSUPPORTED;
                c.tile[c.n++] = t;
            }
        }
        if (!cells.empty() && cells.back().ox + cells.back().w == c.ox && cells.back().n == c.n &&
            std::equal(c.tile, c.tile + c.n, cells.back().tile)) {
            cells.back().w += c.w;   // same cover as the strip to the left: one wider cell
            continue;
        }
        cells.push_back(c);
    }
    for (const XCell& c : cells) {
        for (int cy = y0; cy < y1; cy += p.ch) {
            for (int cx = 0; cx < c.w; cx += CW) {
                BandItem it{};
                it.ax = c.ox + cx; it.ay = cy;
                it.cwch = std::min(CW, c.w - cx) | (std::min(p.ch, y1 - cy) << 16);
                it.ntiles = c.n;
                it.partial = partial;
                for (int e = 0; e < c.n; ++e) {
                    const int t = c.tile[e];
                    it.cover[e] = (unsigned long long)slot_of[t] | ((unsigned long long)(it.ax - p.xs[t]) << 16) |
                                  ((unsigned long long)(it.ay - p.ys[t]) << 32);
                }
                out.push_back(it);
            }
        }
    }
    return PTB_OK;
}

extern "C" int64_t ptb_band_plan_create(const int64_t* xs64, const int64_t* ys64, int n, int C, int th, int tw, int H, int W,
                                        int rows_per_launch, int final_lo, int final_hi, const int64_t* cuts, int ncuts,
                                        ptb_band_plan** out) {
    return ptb_band_plan_create2(xs64, ys64, n, C, th, tw, H, W, rows_per_launch, final_lo, final_hi, cuts, ncuts, nullptr, 0, out);
}

// `early` = n_early row ranges [lo, hi) (their ends must be among the cuts): the rows a neighbouring rank waits for.  With them the
// launch groups are formed by CLASS instead of by position: the bands of every early range together (one launch per range, as soon as
// the tiles feeding it are in -- the caller issues those first), and the remaining bands in groups of ~rows_per_launch
// rows that do not break at the cuts.  A rank of an 8-way sharded 5000 x 5000 image gets 2-3 launches instead of 6.
extern "C" int64_t ptb_band_plan_create2(const int64_t* xs64, const int64_t* ys64, int n, int C, int th, int tw, int H, int W,
                                         int rows_per_launch, int final_lo, int final_hi, const int64_t* cuts, int ncuts,
                                         const int64_t* early, int n_early, ptb_band_plan** out) {
    return ptb_band_plan_create3(xs64, ys64, n, C, th, tw, H, W, rows_per_launch, final_lo, final_hi, cuts, ncuts, early, n_early, 0, out);
}

// flags bit 0 (PTB_PLAN_CLIP_ROWS): the plan's rows [0, H) are a WINDOW of the image -- tiles may hang over its top / bottom edge
// (ys < 0, ys + th > H, even entirely outside) and only the rows inside are merged: the rank of a communication-free sharded merge
// (parallel.band_plan(partition="pixel_rows")) reads, of every tile that touches the pixel rows it owns, exactly the part that lies
// on them, and writes nothing else.
extern "C" int64_t ptb_band_plan_create3(const int64_t* xs64, const int64_t* ys64, int n, int C, int th, int tw, int H, int W,
                                         int rows_per_launch, int final_lo, int final_hi, const int64_t* cuts, int ncuts,
                                         const int64_t* early, int n_early, int flags, ptb_band_plan** out) {
    const bool clip = (flags & 1) != 0;
    if (flags & ~1) return PTB_EINVAL;
    if (!xs64 || !ys64 || !out || n < 1 || C < 1 || th < 1 || tw < 1 || H < 1 || W < 1 || ncuts < 0 || (ncuts && !cuts)) return PTB_EINVAL;
    if (n_early < 0 || (n_early && !early)) return PTB_EINVAL;
    *out = nullptr;
    if (g_force_scalar || tw % 4 || th % 4 || W % 4 || tw > 32767 || th > 32767) return PTB_EUNSUPPORTED;
    ptb_band_plan* p = new ptb_band_plan();
    p->n = n; p->C = C; p->th = th; p->tw = tw; p->H = H; p->W = W;
    p->ch = g_band_rows;
    p->xs.resize(n); p->ys.resize(n);
    std::vector<int> edges{0, H};
    for (int t = 0; t < n; ++t) {
        if (xs64[t] < 0 || xs64[t] + tw > W) { delete p; return PTB_EBOUNDS; }
        if (!clip && (ys64[t] < 0 || ys64[t] + th > H)) { delete p; return PTB_EBOUNDS; }
        if (ys64[t] < -(int64_t)0x3fffffff || ys64[t] > (int64_t)0x3fffffff) { delete p; return PTB_EBOUNDS; }
        if (xs64[t] % 4 || ys64[t] % 4) { delete p; return PTB_EUNSUPPORTED; }
        p->xs[t] = (int)xs64[t]; p->ys[t] = (int)ys64[t];
        edges.push_back(std::min(std::max(p->ys[t], 0), H)); edges.push_back(std::min(std::max(p->ys[t] + th, 0), H));
    }
    // caller-given row cuts (multi-GPU: ownership boundaries, rows other ranks also cover): band edges AND launch-group breaks
    std::vector<int> breaks;
    for (int k = 0; k < ncuts; ++k) {
        if (cuts[k] % 4) { delete p; return PTB_EUNSUPPORTED; }
        if (cuts[k] > 0 && cuts[k] < H) { edges.push_back((int)cuts[k]); breaks.push_back((int)cuts[k]); }
    }
    std::sort(edges.begin(), edges.end());
    edges.erase(std::unique(edges.begin(), edges.end()), edges.end());
    // bands: consecutive edge intervals with the tiles that cover them
    struct Band { int y0, y1; std::vector<int> cover; };
    std::vector<Band> bands;
    for (size_t i = 0; i + 1 < edges.size(); ++i) {
        Band b{edges[i], edges[i + 1], {}};
        for (int t = 0; t < n; ++t)
            if (p->ys[t] <= b.y0 && p->ys[t] + th >= b.y1) b.cover.push_back(t);
        if ((int)b.cover.size() > PLAN_TILES) { delete p; return PTB_EUNSUPPORTED; }
        bands.push_back(std::move(b));
    }
    p->n_bands = (int)bands.size();
    // launch groups: consecutive bands up to ~rows_per_launch rows and PLAN_TILES tiles; a band without tiles (uncovered rows)
    // rides with its neighbour
    const int target = std::max(1, rows_per_launch);
    p->last_group.assign(n, -1);
    p->ready.assign(n, {});
    p->band_y0.resize(bands.size()); p->band_y1.resize(bands.size()); p->band_group.assign(bands.size(), -1);
    for (size_t b = 0; b < bands.size(); ++b) { p->band_y0[b] = bands[b].y0; p->band_y1[b] = bands[b].y1; }
    // emit one launch group from a list of band indices (any rows, ascending)
    auto emit_group = [&](const std::vector<size_t>& members, const std::vector<int>& tiles) -> int {
        Group g{};
        g.y0 = bands[members.front()].y0;
        g.y1 = bands[members.back()].y1;      // (the hull when the members are not adjacent: ptb_band_plan_rows_launched has the exact rows)
        g.tiles = tiles;
        g.last_tile = tiles.empty() ? -1 : tiles.back();
        std::vector<int> slot_of(n, 0);
        for (size_t s = 0; s < tiles.size(); ++s) slot_of[tiles[s]] = (int)s;
        g.item_off = (long long)p->items.size();
        std::vector<BandItem> its;
        const int gi = (int)p->groups.size();
        for (size_t b : members) {
            const int partial = (bands[b].y0 >= final_lo && bands[b].y1 <= final_hi) ? 0 : 1;
            const int rc = band_items(*p, bands[b].cover, slot_of, bands[b].y0, bands[b].y1, partial, its);
            if (rc != PTB_OK) return rc;
            p->band_group[b] = gi;
        }
        std::stable_sort(its.begin(), its.end(), [](const BandItem& l, const BandItem& r) { return l.ntiles > r.ntiles; });
        g.item_cnt = (int)its.size();
        p->items.insert(p->items.end(), its.begin(), its.end());
        for (int t : tiles) p->last_group[t] = std::max(p->last_group[t], gi);
        p->groups.push_back(std::move(g));
        return PTB_OK;
    };
    auto with_band = [&](const std::vector<int>& tiles, size_t b) {
        std::vector<int> merged_tiles = tiles;
        merged_tiles.insert(merged_tiles.end(), bands[b].cover.begin(), bands[b].cover.end());
        std::sort(merged_tiles.begin(), merged_tiles.end());
        merged_tiles.erase(std::unique(merged_tiles.begin(), merged_tiles.end()), merged_tiles.end());
        return merged_tiles;
    };
    if (n_early > 0) {
        for (int k = 0; k < n_early; ++k)
            if (early[2 * k] % 4 || early[2 * k + 1] % 4) { delete p; return PTB_EUNSUPPORTED; }
        // the early range a band lies in (-1: none).  Round 6: ONE launch group per early range instead of one for all of them -- a rank
        // that ships rows to BOTH neighbours (the balanced cuts of parallel.band_plan) would otherwise hold the rows for the upper
        // neighbour back until the tiles feeding the lower one are in, i.e. until its last tile.
        auto early_of = [&](const Band& b) {
            for (int k = 0; k < n_early; ++k) if (b.y0 >= early[2 * k] && b.y1 <= early[2 * k + 1]) return k;
            return -1;
        };
        for (int cls = 0; cls < 2; ++cls) {             // early bands first: their group indices come first too
            std::vector<size_t> members;
            std::vector<int> tiles;
            int rows = 0, range = -2;
            for (size_t b = 0; b < bands.size(); ++b) {
                const int er = early_of(bands[b]);
                if ((er >= 0 ? 0 : 1) != cls) continue;
                std::vector<int> merged_tiles = with_band(tiles, b);
                const int h = bands[b].y1 - bands[b].y0;
                const bool new_range = cls == 0 && range != -2 && er != range;
                range = er;
                if (!members.empty() && ((int)merged_tiles.size() > PLAN_TILES || new_range || (cls == 1 && rows + h > target && !tiles.empty()))) {
                    const int rc = emit_group(members, tiles);
                    if (rc != PTB_OK) { delete p; return rc; }
                    members.clear(); rows = 0;
                    merged_tiles = with_band({}, b);
                }
                tiles.swap(merged_tiles);
                members.push_back(b);
                rows += h;
            }
            if (!members.empty()) {
                const int rc = emit_group(members, tiles);
                if (rc != PTB_OK) { delete p; return rc; }
            }
        }
    } else {
        size_t bi = 0;
        while (bi < bands.size()) {
            std::vector<int> tiles;
            std::vector<size_t> members;
            size_t bj = bi;
            while (bj < bands.size()) {
                std::vector<int> merged_tiles = with_band(tiles, bj);
                const bool first = bj == bi;
                if (!first && ((int)merged_tiles.size() > PLAN_TILES || (bands[bj].y1 - bands[bi].y0 > target && !tiles.empty()) ||
                               std::find(breaks.begin(), breaks.end(), bands[bj].y0) != breaks.end())) break;
                tiles.swap(merged_tiles);
                members.push_back(bj);
                ++bj;
            }
            const int rc = emit_group(members, tiles);
            if (rc != PTB_OK) { delete p; return rc; }
            bi = bj;
        }
    }
    // a group without any tile (an image whose first / last rows nobody covers) completes with the first tile of the image
    for (size_t gi = 0; gi < p->groups.size(); ++gi) {
        const int lt = p->groups[gi].last_tile < 0 ? 0 : p->groups[gi].last_tile;
        p->ready[lt].push_back((int)gi);
    }
    p->src.assign(n, nullptr);
    p->vs.assign(n, 0);
    p->group_launched.assign(p->groups.size(), 0);
    for (size_t gi = 0; gi + 1 < p->groups.size(); ++gi)
        if (p->groups[gi].last_tile > p->groups[gi + 1].last_tile) p->monotone = false;
    *out = p;
    return (int64_t)(p->items.size() * sizeof(BandItem));
}

extern "C" int ptb_band_plan_upload(ptb_band_plan* p, void* dev_table, ptb_stream_t stream) {
    if (!p || !dev_table || (reinterpret_cast<uintptr_t>(dev_table) & 63u)) return PTB_EINVAL;
    const hipError_t e = hipMemcpyAsync(dev_table, p->items.data(), p->items.size() * sizeof(BandItem), hipMemcpyHostToDevice, (hipStream_t)stream);
    if (e != hipSuccess) { set_hip_error(e); return PTB_ELAUNCH; }
    p->dev_items = static_cast<const BandItem*>(dev_table);
    return PTB_OK;
}

extern "C" int ptb_band_plan_info(const ptb_band_plan* p, int* n_groups, int* n_bands, int64_t* n_items, int64_t* last_group_of_tile,
                                  int64_t* group_rows /* [3 * n_groups]: y0, y1, last tile */) {
    if (!p) return PTB_EINVAL;
    if (n_groups) *n_groups = (int)p->groups.size();
    if (n_bands) *n_bands = p->n_bands;
    if (n_items) *n_items = (int64_t)p->items.size();
    if (last_group_of_tile) for (int t = 0; t < p->n; ++t) last_group_of_tile[t] = p->last_group[t];
    if (group_rows)
        for (size_t g = 0; g < p->groups.size(); ++g) {
            group_rows[3 * g] = p->groups[g].y0; group_rows[3 * g + 1] = p->groups[g].y1; group_rows[3 * g + 2] = p->groups[g].last_tile;
        }
    return PTB_OK;
}

extern "C" int ptb_band_plan_reset(ptb_band_plan* p) {
    if (!p) return PTB_EINVAL;
    p->pos = 0; p->launched = 0; p->cfg_set = 0;
    p->held.clear();
    std::fill(p->group_launched.begin(), p->group_launched.end(), 0);
    return PTB_OK;
}

extern "C" int ptb_band_plan_state(const ptb_band_plan* p, int* pos, int* launched) {
    if (!p) return PTB_EINVAL;
    if (pos) *pos = p->pos;
    if (launched) *launched = p->launched;
    return PTB_OK;
}

extern "C" void ptb_band_plan_destroy(ptb_band_plan* p) { delete p; }

extern "C" int ptb_band_plan_submit(ptb_band_plan* p, int pos, int B, const void* batch, int64_t tile_stride, int64_t view_stride,
                                    int in_dtype, int V, const int* views, int reduction, float* merged, const float* norm_full,
                                    const float* weight, ptb_stream_t stream) {
    if (!p || !batch || !merged || !norm_full || !weight || B < 1) return PTB_EINVAL;
    if (!p->dev_items) return PTB_EINVAL;
    const int dtype_arg = in_dtype;      // (with PTB_ROUND_SRC: part of the image's configuration)
    in_dtype &= ~PTB_ROUND_SRC;
    if (reduction < PTB_RED_SUM || reduction > PTB_RED_LOG1P || in_dtype < PTB_F32 || in_dtype > PTB_BF16) return PTB_EINVAL;
    if (V < 1 || V > MAX_VIEWS || !views) return PTB_EINVAL;
    int nT = 0;
    for (int k = 0; k < V; ++k) {
        if (views[k] < 0 || views[k] > 7) return PTB_EINVAL;
        nT += views[k] & 1;
    }
    if (nT && p->th != p->tw) return PTB_EINVAL;
    // not the next planned tiles, or a different configuration than the image started with: the caller leaves deferred mode
    if (pos != p->pos || pos + B > p->n) return PTB_EUNSUPPORTED;
    const int codes = [&] { int v = 0; for (int k = 0; k < V; ++k) v |= (views[k] & 7) << (3 * k); return v; }();
    if (p->cfg_set) {
        if (p->cfg_dtype != dtype_arg || p->cfg_V != V || p->cfg_codes != codes || p->cfg_red != reduction || p->cfg_merged != merged ||
            p->cfg_norm != norm_full || p->cfg_weight != weight) return PTB_EUNSUPPORTED;
    }
    const long long per_tile = (long long)p->C * p->th * p->tw;
    const unsigned mask = in_dtype == PTB_F32 ? 15u : 7u;
    const size_t esz = in_dtype == PTB_F32 ? 4 : 2;
    if (nT > MAX_T || tile_stride < per_tile || view_stride < per_tile || tile_stride % 4 || view_stride % 4 ||
        (reinterpret_cast<uintptr_t>(batch) & mask) || !aligned16(merged) || !aligned16(norm_full) || !aligned16(weight))
        return PTB_EUNSUPPORTED;
    p->cfg_set = 1; p->cfg_dtype = dtype_arg; p->cfg_V = V; p->cfg_codes = codes; p->cfg_red = reduction;
    for (int k = 0; k < V; ++k) p->cfg_views[k] = views[k];
    p->cfg_merged = merged; p->cfg_norm = norm_full; p->cfg_weight = weight;
    {
        int lg = 0;
        for (int b = 0; b < B; ++b) lg = std::max(lg, p->last_group[pos + b]);
        const uintptr_t p0 = reinterpret_cast<uintptr_t>(batch);
        p->held.push_back({p0, p0 + (size_t)((long long)(V - 1) * view_stride + (long long)(B - 1) * tile_stride + per_tile) * esz, lg});
    }
    for (int b = 0; b < B; ++b) {
        p->src[pos + b] = static_cast<const char*>(batch) + (size_t)b * (size_t)tile_stride * esz;
        p->vs[pos + b] = view_stride;
    }
    p->pos = pos + B;
    ViewArgs a{};
    a.weight = weight; a.merged = merged; a.norm_full = norm_full;
    a.in_dtype = in_dtype;
    a.round_src = (dtype_arg & PTB_ROUND_SRC) ? 1 : 0;
    a.rot_views = g_band_rot_views;
    a.lds_db = g_band_lds_db;
    a.H = p->th; a.W = p->tw; a.C = p->C;
    a.dst_chan_stride = (long long)p->H * p->W;
    a.dst_row_stride = p->W;
    a.nviews = V;
    a.codes = codes;
    a.scale = 1.0f;
    a.op = reduction;
    a.divisor = reduction == PTB_RED_SUM ? 1.0f : (float)V;
    int launched = 0;
    for (int t = pos; t < pos + B; ++t) {
        for (int gi : p->ready[t]) {
            const Group& g = p->groups[gi];
            p->group_launched[gi] = 1;
            if (!g.item_cnt) { ++p->launched; continue; }
            GroupTiles gt;
            for (size_t s = 0; s < g.tiles.size(); ++s) { gt.src[s] = p->src[g.tiles[s]]; gt.vs[s] = p->vs[g.tiles[s]]; }
            for (size_t s = g.tiles.size(); s < (size_t)PLAN_TILES; ++s) { gt.src[s] = nullptr; gt.vs[s] = 0; }
            // identity view on the prefetching instances: one workgroup per item walks the channels (see band_plan_kernel, ALL)
            a.chan_loop = (V == 1 && codes == CODES_ID && g_band_chan_loop && (in_dtype != PTB_F32 ? g_band_half_pf >= 1 : g_band_half_pf >= 2)) ? 1 : 0;
            const long long blocks = (long long)g.item_cnt * (a.chan_loop ? 1 : p->C);
            if (blocks > 0x7fffffffLL) return PTB_EUNSUPPORTED;
            a.ncells = a.chan_loop ? 0 : g_band_xcd; a.total_chunks = (int)blocks;
            launch_plan(a, p->dev_items + g.item_off, gt, a.ncells == 1 ? (int)(8 * ((blocks + 7) / 8)) : (int)blocks, p->ch, (hipStream_t)stream);   // (rounded up only where the kernel's XCD order guards the surplus)
            const int rc = check_launch();
            if (rc != PTB_OK) return rc;
            ++p->launched;
            ++launched;
        }
    }
    if (launched || p->launched == (int)p->groups.size()) {
        if (p->launched == (int)p->groups.size()) p->held.clear();
        else if (p->monotone) {      // groups 0 .. launched-1 are out: batches no later group reads leave custody
            size_t k = 0;
            while (k < p->held.size() && p->held[k].last_group < p->launched) ++k;
            p->held.erase(p->held.begin(), p->held.begin() + (long)k);
        }
    }
    return launched;
}

// The next planned batch of an image as the shortest possible host call (round 6: one integrate_batch of the reference's loop, inference/
// tiles.py:321-339, costs the interpreter ~7 us through the 14-argument form above -- more than the kernels of the plain, TTA-free loop).
// Everything but the batch itself is what the image's first ptb_band_plan_submit set: dtype (+ PTB_ROUND_SRC), views, reduction, merged /
// norm / weight; `batch` is a contiguous [V * B, C, th, tw] model output for the next B planned tiles.  Custody is checked here: a batch
// whose bytes overlap one that a later launch group still reads is refused with PTB_EHELD before anything is recorded (the model writes
// into a reused output buffer: the earlier predictions are gone).  Returns the launches issued (>= 0), PTB_EUNSUPPORTED when the image
// has no configuration yet / the tiles run out, or a negative code of ptb_band_plan_submit.
extern "C" int ptb_band_plan_submit_next(ptb_band_plan* p, const void* batch, int B, ptb_stream_t stream) {
    if (!p || !batch || B < 1) return PTB_EINVAL;
    if (!p->cfg_set || p->pos + B > p->n) return PTB_EUNSUPPORTED;
    const int dt = p->cfg_dtype & ~PTB_ROUND_SRC;
    const size_t esz = dt == PTB_F32 ? 4 : 2;
    const long long per_tile = (long long)p->C * p->th * p->tw;
    const uintptr_t p0 = reinterpret_cast<uintptr_t>(batch), p1 = p0 + (size_t)((long long)p->cfg_V * B * per_tile) * esz;
    for (const ptb_band_plan::Held& h : p->held)
        if (h.p0 < p1 && p0 < h.p1) return PTB_EHELD;
    return ptb_band_plan_submit(p, p->pos, B, batch, per_tile, (long long)B * per_tile, p->cfg_dtype, p->cfg_V, p->cfg_views, p->cfg_red,
                                static_cast<float*>(const_cast<void*>(p->cfg_merged)), static_cast<const float*>(p->cfg_norm),
                                static_cast<const float*>(p->cfg_weight), stream);
}

// 1 when every launch group that writes rows r0 .. r1-1 of the merged map has been issued for the current image (a band the rows
// cut through counts as a whole), 0 otherwise.
extern "C" int ptb_band_plan_rows_launched(const ptb_band_plan* p, int r0, int r1) {
    if (!p || r1 < r0) return PTB_EINVAL;
    for (size_t b = 0; b < p->band_y0.size(); ++b)
        if (p->band_y0[b] < r1 && p->band_y1[b] > r0 && !p->group_launched[p->band_group[b]]) return 0;
    return 1;
}

// Strided rectangle -> contiguous buffer: dst[c][r][x] = src[c * chan_stride + r * row_stride + x] (the send side of the halo
// exchange; ptb_rect_add is the receive side).  16-byte accesses when the rectangle allows.
template <int V>
__global__ __launch_bounds__(256) void halo_pack_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, int rows, int cols,
                                                        long long cs, long long rs) {
    const long long per_row = cols / V;
    const long long total = (long long)C * rows * per_row;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long x = i % per_row, cr = i / per_row;
        const long long r = cr % rows, c = cr / rows;
        const float* s = src + c * cs + r * rs + x * V;
        float* d = dst + i * V;
        if (V == 4) *reinterpret_cast<float4*>(d) = *reinterpret_cast<const float4*>(s);
        else d[0] = s[0];
    }
}

extern "C" int ptb_halo_pack(const float* src, int64_t chan_stride, int64_t row_stride, int C, int rows, int cols, float* dst,
                             ptb_stream_t stream) {
    if (!src || !dst || C < 1 || rows < 0 || cols < 0 || row_stride < cols || chan_stride < (int64_t)rows * row_stride) return PTB_EINVAL;
    if (rows == 0 || cols == 0) return PTB_OK;
    const bool vec = !g_force_scalar && cols % 4 == 0 && chan_stride % 4 == 0 && row_stride % 4 == 0 && aligned16(src) && aligned16(dst);
    const long long total = (long long)C * rows * (vec ? cols / 4 : cols);
    const int grid = (int)std::min<long long>((total + 255) / 256, 256 * 16);
    if (vec) hipLaunchKernelGGL(halo_pack_kernel<4>, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, dst, C, rows, cols, (long long)chan_stride, (long long)row_stride);
    else hipLaunchKernelGGL(halo_pack_kernel<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, dst, C, rows, cols, (long long)chan_stride, (long long)row_stride);
    return check_launch();
}

// One rank's step of a sharded merge as ONE host call: ptb_band_plan_submit, then every outgoing halo rectangle whose rows have all
// been written by their launches is packed into its send buffer (rects: n_sends x {r0, r1, c0, c1} in the plan's local rows;
// `packed` [n_sends] in/out, 0 at the start of an image), and -- when the last of them has just been packed -- `ready_event`
// (a hipEvent_t; NULL only without sends) is recorded on the stream: the communication stream waits for it and posts the sends.
// Returns the number of band launches (>= 0) or a negative code; *all_packed (may be NULL) = every rectangle is in its buffer.
extern "C" int ptb_band_plan_submit_rank(ptb_band_plan* p, int pos, int B, const void* batch, int64_t tile_stride, int64_t view_stride,
                                         int in_dtype, int V, const int* views, int reduction, float* merged, const float* norm_full,
                                         const float* weight, int n_sends, const int64_t* rects, float* const* send_bufs, int* packed,
                                         void* ready_event, int* all_packed, ptb_stream_t stream) {
    if (n_sends < 0 || (n_sends && (!rects || !send_bufs || !packed || !ready_event))) return PTB_EINVAL;
    const int rc = ptb_band_plan_submit(p, pos, B, batch, tile_stride, view_stride, in_dtype, V, views, reduction, merged, norm_full, weight, stream);
    if (rc < 0) return rc;
    int done = 0, fresh = 0;
    for (int k = 0; k < n_sends; ++k) {
        if (!packed[k] && rc > 0) {
            const int r0 = (int)rects[4 * k], r1 = (int)rects[4 * k + 1], c0 = (int)rects[4 * k + 2], c1 = (int)rects[4 * k + 3];
            if (r0 < 0 || r1 > p->H || c0 < 0 || c1 > p->W || r1 < r0 || c1 < c0 || !send_bufs[k]) return PTB_EINVAL;
            if (ptb_band_plan_rows_launched(p, r0, r1) == 1) {
                const int prc = ptb_halo_pack(merged + (long long)r0 * p->W + c0, (int64_t)p->H * p->W, p->W, p->C, r1 - r0, c1 - c0, send_bufs[k], stream);
                if (prc != PTB_OK) return prc;
                packed[k] = 1;
                ++fresh;
            }
        }
        done += packed[k] ? 1 : 0;
    }
    const bool all = done == n_sends;
    if (all && fresh && ready_event) {
        const hipError_t e = hipEventRecord((hipEvent_t)ready_event, (hipStream_t)stream);
        if (e != hipSuccess) { set_hip_error(e); return PTB_ELAUNCH; }
    }
    if (all_packed) *all_packed = all ? 1 : 0;
    return rc;
}

int ptb::g_band_chan_loop = 1;     // ptb_set_tunable key 27: identity-view band launches run one workgroup per item over all channels (window / normaliser loaded once per pixel)
int ptb::g_band_lds_db = 1;        // ptb_set_tunable key 25: the prefetching band instances alternate between two sets of LDS tiles (one barrier per covering tile instead of two)
int ptb::g_band_rot_views = 0;     // ptb_set_tunable key 22 (A/B): odd work items of the band plan kernel issue their view loads starting at view NV / 2
int ptb::g_band_half_pf = 2;       // ptb_set_tunable key 21: the band plan kernel requests covering tile e + 1 before it finishes tile e -- 0: never (round 4's
                                   // instances), 1: for half / bf16 sources, 2: for fp32 sources as well
int ptb::g_rank_finish_fused = 1;   // ptb_set_tunable key 18: 0 = ptb_rect_add + ptb_merge_div_ex launches (A/B, bit-identity test)

// The end of a rank's image in one launch: over <= 2 row ranges of `merged` (full width, the rows that held partial sums),
// value = (merged + recv_0 + recv_1 + ...) / norm with the received rectangles added in their order -- the arithmetic of ptb_rect_add
// per rectangle followed by ptb_merge_div_ex per range, without storing and re-reading the sums in between.
struct FinishArgs {
    float* merged;
    const float* norm;
    int C, W;
    long long plane;
    int n_recvs, n_ranges;
    int r0[4], r1[4], c0[4], c1[4];
    const float* buf[4];
    int q0[2], q1[2];
};

__global__ __launch_bounds__(256) void band_finish_kernel(const FinishArgs a) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    const int w4 = a.W >> 2;
    const int rows0 = a.q1[0] - a.q0[0];
    const long long total = (long long)(rows0 + (a.n_ranges > 1 ? a.q1[1] - a.q0[1] : 0)) * w4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int t = (int)(i / w4), x = (int)(i - (long long)t * w4) * 4;
        const int r = t < rows0 ? a.q0[0] + t : a.q0[1] + (t - rows0);
        const v4f n = *reinterpret_cast<const v4f*>(a.norm + (long long)r * a.W + x);
        for (int c = 0; c < a.C; ++c) {
            float* mp = a.merged + c * a.plane + (long long)r * a.W + x;
            v4f v = *reinterpret_cast<const v4f*>(mp);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (k < a.n_recvs && r >= a.r0[k] && r < a.r1[k] && x >= a.c0[k] && x < a.c1[k]) {       // (c0, c1 are multiples of 4 on this path)
                    const int cols = a.c1[k] - a.c0[k];
                    const v4f e = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(
                        a.buf[k] + ((long long)c * (a.r1[k] - a.r0[k]) + (r - a.r0[k])) * cols + (x - a.c0[k])));
                    v.x = __fadd_rn(v.x, e.x); v.y = __fadd_rn(v.y, e.y); v.z = __fadd_rn(v.z, e.z); v.w = __fadd_rn(v.w, e.w);
                }
            }
            v4f o;
            o.x = __fdiv_rn(v.x, n.x); o.y = __fdiv_rn(v.y, n.y); o.z = __fdiv_rn(v.z, n.z); o.w = __fdiv_rn(v.w, n.w);
            __builtin_nontemporal_store(o, reinterpret_cast<v4f*>(mp));
        }
    }
}

// The end of a rank's image as ONE host call: the partial sums received from the neighbours (n_recvs rectangles {r0, r1, c0, c1} in
// the plan's rows, packed [C][r1 - r0][c1 - c0] buffers) are added to `merged`, then the row ranges that held partial sums
// (n_ranges x {r0, r1}) are divided by `norm` in place (tiles.py:346).  One launch (band_finish_kernel) for the shapes a row-band
// partition produces, else ptb_rect_add + ptb_merge_div_ex per item; the same additions in the same order either way.
extern "C" int ptb_band_plan_finish_rank(const ptb_band_plan* p, float* merged, const float* norm, int n_recvs, const int64_t* rects,
                                         const float* const* recv_bufs, int n_ranges, const int64_t* ranges, ptb_stream_t stream) {
    if (!p || !merged || !norm || n_recvs < 0 || n_ranges < 0 || (n_recvs && (!rects || !recv_bufs)) || (n_ranges && !ranges)) return PTB_EINVAL;
    const int64_t plane = (int64_t)p->H * p->W;
    for (int k = 0; k < n_recvs; ++k) {
        const int64_t r0 = rects[4 * k], r1 = rects[4 * k + 1], c0 = rects[4 * k + 2], c1 = rects[4 * k + 3];
        if (r0 < 0 || r1 > p->H || c0 < 0 || c1 > p->W || r1 < r0 || c1 < c0 || !recv_bufs[k]) return PTB_EINVAL;
    }
    for (int k = 0; k < n_ranges; ++k)
        if (ranges[2 * k] < 0 || ranges[2 * k + 1] > p->H || ranges[2 * k + 1] < ranges[2 * k]) return PTB_EINVAL;
    // one launch when the shapes allow it: <= 4 rectangles, <= 2 ranges, 16-byte columns, every rectangle inside one of the ranges
    // (a rectangle outside them would only be added, not divided: the launches below do that)
    bool fused = g_rank_finish_fused && !g_force_scalar && n_recvs <= 4 && n_ranges >= 1 && n_ranges <= 2 && p->W % 4 == 0 && aligned16(merged) && aligned16(norm);
    for (int k = 0; fused && k < n_recvs; ++k) {
        const int64_t r0 = rects[4 * k], r1 = rects[4 * k + 1], c0 = rects[4 * k + 2], c1 = rects[4 * k + 3];
        bool inside = r1 == r0;
        for (int q = 0; q < n_ranges; ++q) inside = inside || (r0 >= ranges[2 * q] && r1 <= ranges[2 * q + 1]);
        fused = inside && c0 % 4 == 0 && c1 % 4 == 0 && aligned16(recv_bufs[k]);
    }
    if (fused && n_ranges == 2 && ranges[0] < ranges[3] && ranges[2] < ranges[1]) fused = false;      // overlapping ranges: divided twice by the launches below
    if (fused) {
        FinishArgs a{};
        a.merged = merged; a.norm = norm; a.C = p->C; a.W = p->W; a.plane = plane; a.n_recvs = n_recvs; a.n_ranges = n_ranges;
        for (int k = 0; k < n_recvs; ++k) {
            a.r0[k] = (int)rects[4 * k]; a.r1[k] = (int)rects[4 * k + 1]; a.c0[k] = (int)rects[4 * k + 2]; a.c1[k] = (int)rects[4 * k + 3];
            a.buf[k] = recv_bufs[k];
        }
        long long rows = 0;
        for (int q = 0; q < n_ranges; ++q) { a.q0[q] = (int)ranges[2 * q]; a.q1[q] = (int)ranges[2 * q + 1]; rows += ranges[2 * q + 1] - ranges[2 * q]; }
        if (rows == 0) return PTB_OK;
        const long long want = (rows * (p->W / 4) + 255) / 256;
        hipLaunchKernelGGL(band_finish_kernel, dim3((unsigned)(want < 256 * 16 ? want : 256 * 16)), dim3(256), 0, (hipStream_t)stream, a);
        return check_launch();
    }
    for (int k = 0; k < n_recvs; ++k) {
        const int64_t r0 = rects[4 * k], r1 = rects[4 * k + 1], c0 = rects[4 * k + 2], c1 = rects[4 * k + 3];
        const int rc = ptb_rect_add(merged + r0 * p->W + c0, recv_bufs[k], p->C, (int)(r1 - r0), (int)(c1 - c0), plane, p->W, stream);
        if (rc != PTB_OK) return rc;
    }
    for (int k = 0; k < n_ranges; ++k) {
        const int64_t r0 = ranges[2 * k], r1 = ranges[2 * k + 1];
        if (r0 < 0 || r1 > p->H || r1 < r0) return PTB_EINVAL;
        if (r1 == r0) continue;
        float* rows = merged + r0 * p->W;
        const int rc = ptb_merge_div_ex(rows, norm + r0 * p->W, rows, p->C, (r1 - r0) * p->W, plane, plane, nullptr, 0, 0, stream);
        if (rc != PTB_OK) return rc;
    }
    return PTB_OK;
}
