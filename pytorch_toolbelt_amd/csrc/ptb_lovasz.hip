// ptb_lovasz.hip -- Lovasz hinge / Lovasz-softmax losses for gfx950 (MI355X).
//
// Reference: losses/lovasz.py:23-34 (_lovasz_grad), :52-72 (_lovasz_hinge_flat), :110-140 (_lovasz_softmax_flat).
// Per class (and per image with per_image=True) the reference sorts the errors, gathers the ground truth, runs two
// cumsums, a division, a first difference and a dot product -- a Python loop over classes with a host sync each
// (`fg.sum() == 0`).  Here every (group, class) pair is one *segment* of a single pipeline:
//   1. error kernel:   key = ~kappa, kappa = bits(max(error, +0)) << 1 | fg (ignored pixels: 0 -- they sort last and contribute 0);
//      it is also the histogram step of the first sort pass (the keys are in registers) and reads a
//      pixel's int64 label once for all the classes it handles.  No value is written: the first scatter pass makes index << 1 | fg
//      from the element's position and its key
//   2. a hand-written segmented LSD radix sort for gfx950 (below): every segment is sorted by its 32-bit keys in four 8-bit
//      passes; a pass = per-tile digit histograms (wave-private LDS counters), a scan over the tiles of every segment, and a
//      stable scatter that ranks the 4096 keys of a tile with ballot-matched lane groups (no atomics where order matters; the
//      match folds each ballot into the lane's mask with gfx950's three-input v_bitop3_b32), stages them by digit in LDS and
//      writes every digit run coalesced (tile-major histograms, scanned in spans of 32 tiles, so every histogram access is a
//      coalesced 1 KiB row).  No inter-workgroup communication, so it is deterministic and needs no forward-progress assumptions.  (Round 1 used rocPRIM's radix sort over 64-bit
//      composite keys here: 0.72 ms of the 1.29 ms for 16 segments of 1 M.)
//   3. foreground count per 2048-element chunk of the sorted order (the tile-shaped histogram kernel in count form) -> chunk scan ->
//      dot kernel: Jaccard gradient grad_k = J_k - J_{k-1} from the prefix count, sum_k relu(e_k) * grad_k as one partial sum per
//      workgroup, added per segment in a fixed order; ptb_lovasz_reduce: the class / image means as one kernel
//   4. for the backward the gradient at every element's rank is BINNED: one more pass of the sort's three kernels, keyed by the
//      pixel block (index >> 12..14), whose scatter kernel computes grad_k -- and the segment's loss -- on the way; the backward kernel puts a block's pairs in
//      pixel order in LDS: d(loss)/d(pred) = coef[segment] * grad_at_pixel * d(error)/d(pred).  (ptb_lovasz_fwd / ptb_lovasz_bwd keep
//      the older form -- the dot kernel scatters grad_k to pixel order word by word -- for segments of more than 4 M elements.)
// No host synchronisation anywhere: class presence (G > 0) is a device array.
#include <cstring>

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

// ------------------------------------------------------------------------------------------------ segmented radix sort
// S segments of P (key, value) pairs each, contiguous; ascending by the 32-bit key, stable, four passes of 8 bits.
// Tile = 4096 elements per 256-thread workgroup; wave w of a tile owns elements w*1024 .. w*1024+1023 and its item j covers the
// 64 consecutive elements w*1024 + j*64 + lane (every load instruction of a wave reads 256 contiguous bytes).
#ifndef PTB_RS_ITEMS
#define PTB_RS_ITEMS 16
#endif
constexpr int RS_ITEMS = PTB_RS_ITEMS, RS_TILE = 256 * RS_ITEMS, RS_WAVE_SPAN = 64 * RS_ITEMS;

// Lanes of this wave whose digit equals mine: per bit one ballot (a scalar pair) folded into the lane's mask halves with
// XNOR against the sign-extended bit (a per-lane select between the ballot and its complement costs three times that: two scalar
// operands in one VOP3 are not encodable).  Returns the group size; `below` = members in lower lanes.
__device__ __forceinline__ unsigned match_digit(unsigned key, int shift, unsigned& below) {      // digit = bits shift .. shift+7 of key
    unsigned m_lo = 0xFFFFFFFFu, m_hi = 0xFFFFFFFFu;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const int sb = __builtin_amdgcn_sbfe((int)key, (unsigned)(shift + b), 1u);      // 0 or ~0: v_bfe_i32 straight from the key
        const unsigned long long bal = __ballot(sb < 0);
        // m &= ~(ballot ^ sb) as ONE gfx950 three-input bit operation per mask half (truth table 0x90 = a & ~(b ^ c)): 4 vector
        // instructions per bit instead of the 6-7 the two-input forms compile to
        m_lo = __builtin_amdgcn_bitop3_b32(m_lo, (unsigned)bal, (unsigned)sb, 0x90);
        m_hi = __builtin_amdgcn_bitop3_b32(m_hi, (unsigned)(bal >> 32), (unsigned)sb, 0x90);
    }
    below = __builtin_amdgcn_mbcnt_hi(m_hi, __builtin_amdgcn_mbcnt_lo(m_lo, 0u));      // set bits of the mask below this lane (v_mbcnt: no lane masks needed)
    return __popc(m_lo) + __popc(m_hi);
}

// Workgroup -> linear tile.  Workgroups go to the 8 XCDs round-robin by blockIdx, and every XCD has its own L2: with the identity
// mapping the tiles t, t+1, ... whose digit runs are neighbours in the output are written through 8 different L2s and every 64-byte
// run reaches HBM as its own partial line.  xcd_map gives XCD x a contiguous range of tiles, so neighbouring runs meet in one
// write-back L2 and leave as whole lines.
__device__ __forceinline__ unsigned tile_of_block(int xcd_map) {
    const unsigned bid = blockIdx.x, nb = gridDim.x;
    if (!xcd_map || nb < 16) return bid;
    const unsigned per = nb >> 3, rem = nb & 7u, x = bid & 7u, slot = bid >> 3;
    return x * per + (x < rem ? x : rem) + slot;
}

// pass step 1: digit histogram of every tile -> hist[(seg * T + tile) * 256 + digit]  (one coalesced 1 KiB row per tile)
// COUNT (the binning pass of the gradient, keys = the sorted index << 1 | fg values): also the foreground count of the tile's two
// CHUNKs -> chunk_count (what lovasz_count_kernel computes from one more read of the same array).
#ifndef PTB_RS_HIST_COPIES
#define PTB_RS_HIST_COPIES 4
#endif
constexpr int RS_HIST_COPIES = PTB_RS_HIST_COPIES;   // wave-private counter copies (power of two)
// FGH (the LAST level of the key-only forward, lovasz_rankdot_kernel): next to the digit histogram a second one of the same shape that
// counts only the foreground keys of every digit -> hist_fg (keys are ~kappa: fg = ~key & 1).
template <bool COUNT, bool HIST = true, bool INV = false, bool FGH = false>
__global__ __launch_bounds__(256) void rs_hist_kernel(const unsigned* __restrict__ keys, long long P, int T, int shift, unsigned* __restrict__ hist,
                                                      unsigned* __restrict__ chunk_count, int chunks_per_seg, unsigned* __restrict__ hist_fg = nullptr) {
    static_assert(RS_TILE == 2 * CHUNK && RS_WAVE_SPAN * 2 == CHUNK, "a tile is two chunks, a chunk two waves");
    // counts only, no order: LDS atomics (ds_add_u32 without return).  Four copies per wave (lane & 3) keep the same-address
    // serialisation short when the digit is nearly constant (the exponent byte of probabilities); a wave whose 64 keys share
    // one digit adds 64 from a single lane.
    __shared__ unsigned h[4][RS_HIST_COPIES][256];
    __shared__ unsigned hf[FGH ? 4 : 1][FGH ? 2 : 1][FGH ? 256 : 1];
    const int seg = blockIdx.x / T, tile = blockIdx.x % T;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int w = 0; w < 4; ++w)
#pragma unroll
        for (int c = 0; c < RS_HIST_COPIES; ++c) h[w][c][threadIdx.x] = 0;
    if constexpr (FGH) {
#pragma unroll
        for (int w = 0; w < 4; ++w) { hf[w][0][threadIdx.x] = 0; hf[w][1][threadIdx.x] = 0; }
    }
    __syncthreads();
    const long long t0 = (long long)tile * RS_TILE;
    const unsigned* kp = keys + (long long)seg * P + t0;
    const long long left = P - t0;
    unsigned k[RS_ITEMS];
    // counting is order-free: a full, 16-byte aligned tile is read with 4 x 16-byte loads per thread (item j = elements j/4 * 256 +
    // lane * 4 + j%4 of the wave's span) instead of 16 x 4-byte ones -- this kernel is all load issue: 18 -> 13 us for 67 MB
    const bool vec = left >= RS_TILE && ((((long long)seg * P + t0) & 3) == 0) && ((reinterpret_cast<uintptr_t>(keys) & 15u) == 0);
    if (vec) {
#pragma unroll
        for (int q = 0; q < RS_ITEMS / 4; ++q) {
            const uint4 u = *reinterpret_cast<const uint4*>(kp + wave * RS_WAVE_SPAN + q * 256 + lane * 4);
            k[4 * q] = u.x; k[4 * q + 1] = u.y; k[4 * q + 2] = u.z; k[4 * q + 3] = u.w;
        }
    } else {
#pragma unroll
        for (int j = 0; j < RS_ITEMS; ++j) {
            const int idx = wave * RS_WAVE_SPAN + j * 64 + lane;
            k[j] = idx < left ? kp[idx] : 0u;
        }
    }
#pragma unroll
    for (int j = 0; j < RS_ITEMS; ++j) {
        const int idx = wave * RS_WAVE_SPAN + j * 64 + lane;
        const bool valid = vec || idx < left;
        if constexpr (HIST) {
            const unsigned d = (k[j] >> shift) & 255u;
            const unsigned d0 = __builtin_amdgcn_readfirstlane(d);
            const bool full = vec || wave * RS_WAVE_SPAN + j * 64 + 63 < left;      // (wave-uniform)
            if (full && __all(d == d0)) {
                if (lane == 0) atomicAdd(&h[wave][0][d0], 64u);
                if constexpr (FGH) {
                    const unsigned nfg = (unsigned)__popcll(__ballot((~k[j] & 1u) != 0u));
                    if (lane == 0 && nfg) atomicAdd(&hf[wave][0][d0], nfg);
                }
            } else if (valid) {
                atomicAdd(&h[wave][lane & (RS_HIST_COPIES - 1)][d], 1u);
                if constexpr (FGH) {
                    if (~k[j] & 1u) atomicAdd(&hf[wave][lane & 1][d], 1u);
                }
            }
        }
    }
    __shared__ unsigned wfg[4];
    if constexpr (COUNT) {
        unsigned fgs = 0;
#pragma unroll
        for (int j = 0; j < RS_ITEMS; ++j)
            fgs += (unsigned)__popcll(__ballot((vec || wave * RS_WAVE_SPAN + j * 64 + lane < left) && ((INV ? ~k[j] : k[j]) & 1u)));
        if (lane == 0) wfg[wave] = fgs;
    }
    __syncthreads();
    unsigned tot = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w)
#pragma unroll
        for (int c = 0; c < RS_HIST_COPIES; ++c) tot += h[w][c][threadIdx.x];
    if constexpr (HIST) hist[((long long)seg * T + tile) * 256 + threadIdx.x] = tot;
    if constexpr (FGH) {
        unsigned tf = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) tf += hf[w][0][threadIdx.x] + hf[w][1][threadIdx.x];
        hist_fg[((long long)seg * T + tile) * 256 + threadIdx.x] = tf;
    }
    if constexpr (COUNT) {
        if (threadIdx.x < 2 && tile * 2 + (int)threadIdx.x < chunks_per_seg)
            chunk_count[(long long)seg * chunks_per_seg + tile * 2 + threadIdx.x] = wfg[2 * threadIdx.x] + wfg[2 * threadIdx.x + 1];
    }
}

// The error kernel and the first pass's histogram kernel in one: a workgroup owns the 4096 pixels of one tile position of a group
// (image, or the whole batch) and a chunk of classes.  The pixels' labels are read once (int64: read per class they are twice the
// bytes of the probabilities) and every class of the chunk is a pass over the same pixels: error -> key written, digit 0
// counted in LDS, the tile's histogram row written.  Element i of a segment is pixel (b = i / HW, i % HW) of the batch, or pixel i of
// image j when per_image; n < 2^31, so offsets into pred fit 32 bits.
// The key ("kappa"): an error that is not positive contributes relu(e) * grad = 0 and has a zero gradient wherever it sorts among the
// non-positive ones, so all of them (and the ignored pixels) share the key of +0; a positive float has a clear sign bit, so its 31
// significant bits move up by one and the foreground flag takes the freed bit: kappa = bits(e) << 1 | fg, sorted in decreasing
// order (key = ~kappa, ascending).  That is the order of the errors exactly, ties broken by fg and then (the sort is stable) by
// index -- the reference's torch.sort leaves the order of ties open and the loss does not depend on it.  Everything downstream reads
// the error AND the foreground flag out of the key: the forward without a gradient sorts keys alone (4 bytes per element and pass
// instead of 8), the training path makes its (index << 1 | fg) values in the first scatter pass instead of loading them (IOTA).
__device__ __forceinline__ unsigned keyonly_key(float e, unsigned fg, bool valid) {
    const unsigned m = (valid && !(e <= 0.0f)) ? (__float_as_uint(e) & 0x7FFFFFFFu) : 0u;      // (NaN stays NaN: it sorts first and poisons the sum)
    return ~((m << 1) | (valid ? fg : 0u));
}

template <int MODE>
__global__ __launch_bounds__(256) void lovasz_error_hist_kernel(const LovArgs a, int T, int cchunk, unsigned* __restrict__ keys,
                                                                unsigned* __restrict__ hist) {
    __shared__ unsigned h[4][4][256];
    const int j = blockIdx.x / T, tile = blockIdx.x % T;
    const int c0 = blockIdx.y * cchunk, c1 = min(a.C, c0 + cchunk);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int w = 0; w < 4; ++w)
#pragma unroll
        for (int c = 0; c < 4; ++c) h[w][c][threadIdx.x] = 0;
    const long long t0 = (long long)tile * RS_TILE;
    const long long left = a.P - t0;
    const unsigned HW = (unsigned)a.HW;
    unsigned off0[RS_ITEMS];                 // offset of the pixel in class 0's plane of its image
    int lab[MODE == LOVASZ_SOFTMAX ? RS_ITEMS : 1];      // class label, -1 = no class of this call, -2 = ignored
    float yv[MODE == LOVASZ_HINGE ? RS_ITEMS : 1];
    unsigned okmask = 0;
#pragma unroll
    for (int jj = 0; jj < RS_ITEMS; ++jj) {
        const int idx = wave * RS_WAVE_SPAN + jj * 64 + lane;
        const bool ok = idx < left;
        const unsigned i = (unsigned)(t0 + idx);
        const unsigned b = a.per_image ? (unsigned)j : (ok ? i / HW : 0u);
        const unsigned px = a.per_image ? i : i - b * HW;
        off0[jj] = b * (unsigned)a.C * HW + px;
        okmask |= ok ? 1u << jj : 0u;
        const long long lo = (long long)b * a.HW + px;
        if constexpr (MODE == LOVASZ_SOFTMAX) {
            const long long L = ok ? a.labels[lo] : 0;
            lab[jj] = (a.has_ignore && L == a.ignore_label) ? -2 : ((L >= 0 && L < a.C) ? (int)L : -1);
        } else {
            yv[jj] = ok ? a.flabels[lo] : 0.f;
        }
    }
    __syncthreads();
    for (int c = c0; c < c1; ++c) {
        const int s = j * a.C + c;
        const long long sbase = (long long)s * a.P + t0;
        float p[RS_ITEMS];
#pragma unroll
        for (int jj = 0; jj < RS_ITEMS; ++jj) p[jj] = (okmask >> jj & 1u) ? a.pred[off0[jj] + (unsigned)c * HW] : 0.f;
#pragma unroll
        for (int jj = 0; jj < RS_ITEMS; ++jj) {
            const int idx = wave * RS_WAVE_SPAN + jj * 64 + lane;
            const bool ok = okmask >> jj & 1u;
            float e;
            unsigned fg;
            bool valid;
            if constexpr (MODE == LOVASZ_SOFTMAX) {
                valid = lab[jj] != -2;
                fg = lab[jj] == c ? 1u : 0u;
                e = fabsf((float)fg - p[jj]);                         // lovasz.py:133
            } else {
                const float y = yv[jj];
                valid = !(a.has_ignore && y == a.ignore_value);
                fg = y != 0.f ? 1u : 0u;
                e = 1.0f - p[jj] * (2.0f * y - 1.0f);                 // lovasz.py:65-66
            }
            const unsigned key = keyonly_key(e, fg, valid);      // ascending sort of the complement = descending errors
            if (ok) keys[sbase + idx] = key;
            const unsigned d = key & 255u;
            const unsigned d0 = __builtin_amdgcn_readfirstlane(d);
            const bool full = wave * RS_WAVE_SPAN + jj * 64 + 63 < left;      // (wave-uniform)
            if (full && __all(d == d0)) {
                if (lane == 0) atomicAdd(&h[wave][0][d0], 64u);
            } else if (ok) {
                atomicAdd(&h[wave][lane & 3][d], 1u);
            }
        }
        __syncthreads();
        unsigned tot = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w)
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) { tot += h[w][cc][threadIdx.x]; h[w][cc][threadIdx.x] = 0; }
        hist[((long long)s * T + tile) * 256 + threadIdx.x] = tot;
        __syncthreads();
    }
}

// (Measured and dropped, round 6: the tile scan done by the LAST histogram workgroup of a span instead of by this launch.  What crosses
// workgroups inside a launch has to get past the XCDs' non-coherent L2s: with a device-wide release fence per workgroup (__threadfence) a
// Lovasz call took +1.25 ms; with the rows published by agent-scope atomic stores and read back with atomic loads, +35-40 us -- against
// the 5 us the launch costs.  tools/ab_lovasz.py on the experimental build; the launches stay.)
// pass step 2: one workgroup per (segment, span of RS_SPAN tiles); thread = digit: running count over the span's tiles in place
// (every access is a coalesced 1 KiB row, the loads of a span are independent of each other), span total -> span_tot
constexpr int RS_SPAN = 32;
__device__ __forceinline__ void tilescan_block(unsigned block, unsigned* __restrict__ hist, int T, int spans, unsigned* __restrict__ span_tot) {
    const int seg = block / spans, span = block % spans;
    const int t0 = span * RS_SPAN, t1 = min(T, t0 + RS_SPAN);
    unsigned* row = hist + ((long long)seg * T + t0) * 256 + threadIdx.x;
    unsigned v[RS_SPAN];
#pragma unroll
    for (int t = 0; t < RS_SPAN; ++t) v[t] = t0 + t < t1 ? row[(long long)t * 256] : 0u;
    unsigned run = 0;
#pragma unroll
    for (int t = 0; t < RS_SPAN; ++t) {
        if (t0 + t < t1) row[(long long)t * 256] = run;
        run += v[t];
    }
    span_tot[((long long)seg * spans + span) * 256 + threadIdx.x] = run;
}
__global__ __launch_bounds__(256) void rs_tilescan_kernel(unsigned* __restrict__ hist, int T, int spans, unsigned* __restrict__ span_tot) {
    tilescan_block(blockIdx.x, hist, T, spans, span_tot);
}
// the digit histogram and the foreground histogram of the last level (same shape) in one launch: blocks n .. 2n-1 take the second pair
__global__ __launch_bounds__(256) void rs_tilescan2_kernel(unsigned* __restrict__ hist, unsigned* __restrict__ hist_fg, int T, int spans,
                                                           unsigned* __restrict__ span_tot, unsigned* __restrict__ span_tot_fg, unsigned n) {
    if (blockIdx.x < n) tilescan_block(blockIdx.x, hist, T, spans, span_tot);
    else tilescan_block(blockIdx.x - n, hist_fg, T, spans, span_tot_fg);
}

__device__ __forceinline__ float jaccard_at(float G, float k1, float cum) {  // lovasz.py:29-31 at sorted position k (k1 = k+1)
    const float inter = G - cum;
    const float uni = G + (k1 - cum);
    return 1.0f - inter / uni;
}

// pass step 3: stable scatter of one tile.  NW waves per workgroup share the tile's 4096 elements: wave w owns elements
// w * 4096/NW ..., its item j the 64 consecutive ones behind j * 64 (rank order = (wave, item, lane) = index order).
template <int NW>
__device__ __forceinline__ unsigned block_inclusive_scan_n(unsigned v, unsigned* wave_tot /* [NW] LDS */) {   // over the first 256 threads' values
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    __syncthreads();                       // (wave_tot may still be read from a previous call)
    if (lane == 63 && wave < 4) wave_tot[wave] = incl;
    __syncthreads();
    unsigned off = 0;
    for (int w = 0; w < wave && w < 4; ++w) off += wave_tot[w];
    return incl + off;
}

// (Measured and dropped: a first-pass instance that COMPUTES its pairs from pred / labels instead of loading what the error kernel wrote --
// 134 MB less written and read, but the int64 labels per class and the index arithmetic inside this already VALU-heavy kernel made the
// forward 30 us slower.)
// GRAD (the binning pass of the gradient: keys_in = the sorted index << 1 | fg values): the value carried with a key is not loaded
// but computed -- the Lovasz gradient at the element's sorted position, from the foreground count before it (chunk_off + the count
// inside the tile) exactly as lovasz_dot_kernel computes it.
// IOTA (the first pass of the training path): the keys are the key-only kappa keys (error bits << 1 | fg, see keyonly_key) and the value
// of element i is not loaded but made here: i << 1 | fg, the fg taken from its key -- the error kernel writes no values and this
// pass reads none (134 MB of the training path's traffic at [4,16,512,512]).
// ANYORDER (the FIRST pass of the key-only sort; the gradient BINNING pass, whose backward kernel places a block's pairs by their pixel
// index and never looks at their order): keys are all there is to an element, so the order in which equal digits of one wave
// leave this pass cannot be told from any other once the remaining passes have run -- the rank inside the wave's digit group comes from a
// returning LDS add (ds_add_rtn_u32: one instruction per key) instead of eight ballots and sixteen three-input bit operations.
template <int NW, bool GRAD, bool KEYONLY = false, bool IOTA = false, bool ANYORDER = false>
__global__ __launch_bounds__(NW * 64) void rs_scatter_kernel(const unsigned* __restrict__ keys_in, const unsigned* __restrict__ vals_in,
                                                             unsigned* __restrict__ keys_out, unsigned* __restrict__ vals_out, long long P, int T,
                                                             int shift, const unsigned* __restrict__ hist, int spans,
                                                             const unsigned* __restrict__ span_tot, int xcd_map,
                                                             const unsigned* __restrict__ chunk_off, const unsigned* __restrict__ fg_total,
                                                             int chunks_per_seg, const unsigned* __restrict__ err_keys = nullptr,
                                                             double* __restrict__ tile_partial = nullptr) {
    static_assert(!GRAD || NW == 4, "the gradient variant counts foreground per pair of waves (= one CHUNK)");
    constexpr int ITEMS = RS_TILE / (NW * 64), SPAN = 64 * ITEMS, NT = NW * 64;
    __shared__ unsigned wave_hist[NW][256];  // per wave: running digit counts, then the wave's start inside the tile's digit run
    __shared__ unsigned digit_base[256];     // global position of slot i of digit d = digit_base[d] + i
    __shared__ unsigned wave_tot[4];
    static_assert(!(GRAD && KEYONLY), "the gradient variant carries a value");
    static_assert(!IOTA || (!GRAD && !KEYONLY), "IOTA is the pair sort's first pass");
    __shared__ unsigned skey[RS_TILE], sval[KEYONLY ? 1 : RS_TILE];
    const unsigned lin = tile_of_block(xcd_map & 1);
    const bool late = (xcd_map & 2) != 0;     // (A/B: ptb_set_tunable(17, 3) puts the histogram loads back behind the ranking)
    const int seg = lin / T, tile = lin % T;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long t0 = (long long)tile * RS_TILE;
    const long long base = (long long)seg * P;
    const long long left = P - t0;
    const int count = left < RS_TILE ? (int)left : RS_TILE;
    if (threadIdx.x < 256) {
#pragma unroll
        for (int w = 0; w < NW; ++w) wave_hist[w][threadIdx.x] = 0;
    }
    // what the digit threads need from the scanned histograms does not depend on the keys: requested first, consumed after the
    // ranking (issued behind the barrier these loads were ~2 us of exposed latency per tile)
    unsigned rs = 0, before = 0, tile_before = 0;
    if (threadIdx.x < 256 && !late) {
        // this digit's elements in earlier spans of the segment, and in the whole segment
        const int my_span = tile / RS_SPAN;
        for (int sp = 0; sp < spans; ++sp) {
            const unsigned c = span_tot[((long long)seg * spans + sp) * 256 + threadIdx.x];
            before += sp < my_span ? c : 0u;
            rs += c;
        }
        tile_before = hist[((long long)seg * T + tile) * 256 + threadIdx.x];
    }
    unsigned k[ITEMS], v[ITEMS], rank[ITEMS];
    const bool full_tile = count == RS_TILE;       // (all but a segment's last tile: no per-element guards, the loads issue back to back)
    if (full_tile) {
        const unsigned* kp = keys_in + base + t0 + wave * SPAN + lane;
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) k[j] = kp[j * 64];
        if constexpr (!GRAD && !KEYONLY && !IOTA) {
            const unsigned* vp = vals_in + base + t0 + wave * SPAN + lane;
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) v[j] = vp[j * 64];
        }
    } else {
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const int idx = wave * SPAN + j * 64 + lane;
            // elements beyond the segment end are padded with the largest key: they rank behind every real element of the tile
            // (they are the last in tile order and the sort is stable) and are never written
            k[j] = idx < count ? keys_in[base + t0 + idx] : 0xFFFFFFFFu;
            if constexpr (!GRAD && !KEYONLY && !IOTA) v[j] = idx < count ? vals_in[base + t0 + idx] : 0u;
        }
    }
    if constexpr (IOTA) {
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) v[j] = ((unsigned)(t0 + wave * SPAN + j * 64 + lane) << 1) | (~k[j] & 1u);
    }
    __shared__ unsigned wfg[NW];
    if constexpr (GRAD) {
        unsigned fgs = 0;
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) fgs += (unsigned)__popcll(__ballot((wave * SPAN + j * 64 + lane < count) && (k[j] & 1u)));
        if (lane == 0) wfg[wave] = fgs;
    }
    __syncthreads();
    if constexpr (GRAD) {
        const int chunk = tile * 2 + (wave >> 1);
        unsigned cum = (chunk < chunks_per_seg ? chunk_off[(long long)seg * chunks_per_seg + chunk] : 0u) + ((wave & 1) ? wfg[wave - 1] : 0u);
        const float G = (float)fg_total[seg];
        // J at the position before the wave's first element (wave-uniform); afterwards an element's J_{k-1} is its left neighbour's
        // J_k: the lane below, or lane 63 of the previous item -- one division per element instead of two.  This telescoping form IS the
        // reference's jaccard[1:] - jaccard[:-1] (lovasz.py:32-33); up to position 2^24 it also has the bits of evaluating J_{k-1}
        // afresh from (float)(i + 1) - 1.0f, beyond that (float)(i + 1) rounds and only the telescoping form matches the reference
        const long long w0 = t0 + (long long)wave * SPAN;
        float carry = w0 == 0 ? 0.0f : jaccard_at(G, (float)w0, (float)cum);
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const int idx = wave * SPAN + j * 64 + lane;
            const bool ok = idx < count;
            const unsigned fg = ok ? (k[j] & 1u) : 0u;
            const unsigned long long bal = __ballot(fg != 0u);
            const unsigned before_u = cum + __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
            cum += (unsigned)__popcll(bal);
            const long long i = t0 + idx;
            const float kf = (float)(unsigned)(i + 1);           // (positions are < 2^31: one v_cvt_f32_u32 instead of the int64 conversion sequence)
            const float jk = jaccard_at(G, kf, (float)(before_u + fg));
            // the left neighbour's J_k: one DPP move (wave_shr:1) and a v_readlane for the carry -- as ds_bpermute shuffles these were two
            // more trips through the LDS pipeline per item, next to the ranking's own
            float left_j, last_j;
            if (xcd_map & 4) {       // (A/B: ptb_set_tunable(17, 5) = the ds_bpermute shuffles)
                left_j = __shfl_up(jk, 1);
                last_j = __shfl(jk, 63);
            } else {
                left_j = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(jk), __float_as_int(jk), 0x138, 0xF, 0xF, false));
                last_j = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(jk), 63));
            }
            const float jprev = lane == 0 ? carry : left_j;
            carry = last_j;
            v[j] = __float_as_uint(jk - jprev);                  // lovasz.py:32-33
        }
        if (err_keys) {
            // the loss itself, from the gradient just computed: sum_k relu(e_k) * grad_k over the tile (lovasz.py:71 / :139) -- what
            // lovasz_dot_kernel would read the sorted pairs once more for.  One partial sum per tile, added per segment in tile order.
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) {
                const int idx = wave * SPAN + j * 64 + lane;
                if (idx < count) {
                    const float e = __uint_as_float((~err_keys[base + t0 + idx]) >> 1);      // kappa key: error bits << 1 | fg (never negative)
                    acc += (double)(e * __uint_as_float(v[j]));
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
            __shared__ double wacc[NW];
            if (lane == 0) wacc[wave] = acc;
            __syncthreads();
            if (threadIdx.x == 0) {
                double t = 0.0;
#pragma unroll
                for (int w = 0; w < NW; ++w) t += wacc[w];
                tile_partial[(long long)seg * T + tile] = t;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const unsigned d = (k[j] >> shift) & 255u;
        if constexpr (ANYORDER) {
            static_assert(!ANYORDER || KEYONLY || GRAD, "only a key-only pass or the gradient binning (whose bins are unordered sets) may reorder equal digits");
            // some order among the wave's keys of this digit.  Padding keys (a segment's last tile) stay out: the stable ranking puts them
            // behind every real key by construction, an arbitrary one would hand them slots of real keys
            rank[j] = (full_tile || wave * SPAN + j * 64 + lane < count) ? atomicAdd(&wave_hist[wave][d], 1u) : 0xFFFFFFFFu;
        } else {
            unsigned below;
            const unsigned group = match_digit(k[j], shift, below);
            const unsigned old = wave_hist[wave][d];                    // every lane of the group reads the counter ...
            __builtin_amdgcn_wave_barrier();
            if (below == 0) wave_hist[wave][d] = old + group;           // ... before its first lane advances it
            __builtin_amdgcn_wave_barrier();
            rank[j] = old + below;                                      // rank among the wave's elements with this digit, in order
        }
    }
    __syncthreads();
    // per digit (thread = digit, the first 256 threads): waves' starts inside the run, the tile's count, the run's start inside the tile
    // and in the output
    unsigned tot = 0;
    if (threadIdx.x < 256) {
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const unsigned c = wave_hist[w][threadIdx.x];
            wave_hist[w][threadIdx.x] = tot;
            tot += c;
        }
        if (late) {
            const int my_span = tile / RS_SPAN;
            for (int sp = 0; sp < spans; ++sp) {
                const unsigned c = span_tot[((long long)seg * spans + sp) * 256 + threadIdx.x];
                before += sp < my_span ? c : 0u;
                rs += c;
            }
            tile_before = hist[((long long)seg * T + tile) * 256 + threadIdx.x];
        }
    }
    const unsigned excl = block_inclusive_scan_n<NW>(tot, wave_tot) - tot;
    const unsigned dstart = block_inclusive_scan_n<NW>(rs, wave_tot) - rs;   // elements of the segment with a smaller digit
    if (threadIdx.x < 256) {
        // (the run's start inside the staged tile goes into every wave's start: the staging below reads ONE table per element)
#pragma unroll
        for (int w = 0; w < NW; ++w) wave_hist[w][threadIdx.x] += excl;
        digit_base[threadIdx.x] = dstart + before + tile_before - excl;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const unsigned d = (k[j] >> shift) & 255u;
        const unsigned slot = wave_hist[wave][d] + rank[j];
        if constexpr (ANYORDER) {
            if (rank[j] != 0xFFFFFFFFu) {
                skey[slot] = k[j];
                if constexpr (!KEYONLY) sval[slot] = v[j];
            }
        } else {
            skey[slot] = k[j];
            if constexpr (!KEYONLY) sval[slot] = v[j];
        }
    }
    __syncthreads();
    if (full_tile) {
#pragma unroll
        for (int u = 0; u < ITEMS; ++u) {
            const int i = threadIdx.x + u * NT;
            const unsigned kk = skey[i];
            const long long pos = base + digit_base[(kk >> shift) & 255u] + i;
            keys_out[pos] = kk;
            if constexpr (!KEYONLY) vals_out[pos] = sval[i];
        }
        return;
    }
    for (int i = threadIdx.x; i < count; i += NT) {
        const unsigned kk = skey[i];
        const long long pos = base + digit_base[(kk >> shift) & 255u] + i;
        keys_out[pos] = kk;
        if constexpr (!KEYONLY) vals_out[pos] = sval[i];
    }
}

// The LAST level of the key-only forward without a scatter (round 6).  After three passes a segment is ordered by the low 24 bits of its
// keys; the fourth pass would only move every key to its final position -- and all the loss needs from that position is the position
// itself, k, and the number of foreground keys in front of it: sum_k relu(e_k) * (J_k - J_{k-1}) with J from (k, cum fg)
// (lovasz.py:29-33, :71 / :139).  Both follow from the scanned histograms exactly as the scatter computes its destinations -- k = (keys
// of the segment with a smaller top digit) + (same digit, earlier tiles) + (same digit, earlier in this tile: the ballot-matched stable
// rank) -- and the same three terms counted over foreground keys only (hist_fg).  So the keys are read once more and NOT written: no
// fourth scatter's stores (67 MB at [4,16,512,512]), no foreground count over the sorted order, no chunk scan, no dot kernel re-reading
// them (another 2 x 67 MB and three launches).  One partial sum per tile, added per segment in tile order by lovasz_segsum_kernel.
// G (the segment's foreground total) = the sum of the foreground histogram; fg_total[seg] is written by tile 0 for the class reduction.
template <int NW>
__global__ __launch_bounds__(NW * 64) void lovasz_rankdot_kernel(const unsigned* __restrict__ keys_in, long long P, int T, int shift,
                                                                const unsigned* __restrict__ hist, const unsigned* __restrict__ hist_fg, int spans,
                                                                const unsigned* __restrict__ span_tot, const unsigned* __restrict__ span_tot_fg,
                                                                int xcd_map, unsigned* __restrict__ fg_total, double* __restrict__ tile_partial) {
    constexpr int ITEMS = RS_TILE / (NW * 64), SPAN = 64 * ITEMS;
    static_assert(NW * 64 == 256, "thread = digit");
    __shared__ unsigned wave_hist[NW][256], wave_fg[NW][256];   // per wave: running (all / foreground) counts per digit, then the wave's start inside the tile's digit run
    __shared__ unsigned base_all[256], base_fg[256];            // position / foreground count in front of the tile's first key of digit d
    __shared__ unsigned wave_tot[4];
    __shared__ double wacc[NW];
    const unsigned lin = tile_of_block(xcd_map & 1);
    const int seg = lin / T, tile = lin % T;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long t0 = (long long)tile * RS_TILE;
    const long long base = (long long)seg * P;
    const long long left = P - t0;
    const int count = left < RS_TILE ? (int)left : RS_TILE;
#pragma unroll
    for (int w = 0; w < NW; ++w) { wave_hist[w][threadIdx.x] = 0; wave_fg[w][threadIdx.x] = 0; }
    // thread = digit: this digit's keys (all / foreground) in earlier spans, in the whole segment, and in earlier tiles of this span
    unsigned rs = 0, before = 0, rs_f = 0, before_f = 0;
    {
        const int my_span = tile / RS_SPAN;
        for (int sp = 0; sp < spans; ++sp) {
            const unsigned c = span_tot[((long long)seg * spans + sp) * 256 + threadIdx.x];
            const unsigned f = span_tot_fg[((long long)seg * spans + sp) * 256 + threadIdx.x];
            before += sp < my_span ? c : 0u; rs += c;
            before_f += sp < my_span ? f : 0u; rs_f += f;
        }
    }
    const unsigned tile_before = hist[((long long)seg * T + tile) * 256 + threadIdx.x];
    const unsigned tile_before_f = hist_fg[((long long)seg * T + tile) * 256 + threadIdx.x];
    unsigned k[ITEMS], rank[ITEMS];      // rank: position among the wave's keys of the digit | the same over foreground keys << 16 (both < 4096)
    if (count == RS_TILE) {
        const unsigned* kp = keys_in + base + t0 + wave * SPAN + lane;
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) k[j] = kp[j * 64];
    } else {
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const int idx = wave * SPAN + j * 64 + lane;
            k[j] = idx < count ? keys_in[base + t0 + idx] : 0xFFFFFFFFu;      // (padding: kappa = 0 -- error 0, background, ranks behind every real key of the tile)
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const unsigned d = (k[j] >> shift) & 255u;
        // lanes of the wave with my digit: mask halves as in match_digit, kept here because the foreground rank needs them as well
        unsigned m_lo = 0xFFFFFFFFu, m_hi = 0xFFFFFFFFu;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const int sb = __builtin_amdgcn_sbfe((int)k[j], (unsigned)(shift + b), 1u);
            const unsigned long long bal = __ballot(sb < 0);
            m_lo = __builtin_amdgcn_bitop3_b32(m_lo, (unsigned)bal, (unsigned)sb, 0x90);
            m_hi = __builtin_amdgcn_bitop3_b32(m_hi, (unsigned)(bal >> 32), (unsigned)sb, 0x90);
        }
        const unsigned long long fgb = __ballot((~k[j] & 1u) != 0u);
        const unsigned f_lo = m_lo & (unsigned)fgb, f_hi = m_hi & (unsigned)(fgb >> 32);
        const unsigned below = __builtin_amdgcn_mbcnt_hi(m_hi, __builtin_amdgcn_mbcnt_lo(m_lo, 0u));
        const unsigned below_f = __builtin_amdgcn_mbcnt_hi(f_hi, __builtin_amdgcn_mbcnt_lo(f_lo, 0u));
        const unsigned group = __popc(m_lo) + __popc(m_hi), group_f = __popc(f_lo) + __popc(f_hi);
        const unsigned old = wave_hist[wave][d], old_f = wave_fg[wave][d];
        __builtin_amdgcn_wave_barrier();
        if (below == 0) { wave_hist[wave][d] = old + group; wave_fg[wave][d] = old_f + group_f; }
        __builtin_amdgcn_wave_barrier();
        rank[j] = (old + below) | ((old_f + below_f) << 16);
    }
    __syncthreads();
    unsigned tot = 0, tot_f = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const unsigned c = wave_hist[w][threadIdx.x], f = wave_fg[w][threadIdx.x];
        wave_hist[w][threadIdx.x] = tot; wave_fg[w][threadIdx.x] = tot_f;
        tot += c; tot_f += f;
    }
    const unsigned incl = block_inclusive_scan_n<NW>(rs, wave_tot);        // keys of the segment with a digit <= mine
    const unsigned incl_f = block_inclusive_scan_n<NW>(rs_f, wave_tot);
    base_all[threadIdx.x] = incl - rs + before + tile_before;
    base_fg[threadIdx.x] = incl_f - rs_f + before_f + tile_before_f;
    __shared__ unsigned g_total;
    if (threadIdx.x == 255) g_total = incl_f;                              // the segment's foreground count
    __syncthreads();
    const float G = (float)g_total;
    if (tile == 0 && threadIdx.x == 0) fg_total[seg] = g_total;
    double acc = 0.0;
#pragma unroll 4
    for (int j = 0; j < ITEMS; ++j) {
        const int idx = wave * SPAN + j * 64 + lane;
        if (idx < count) {
            const unsigned d = (k[j] >> shift) & 255u;
            const unsigned kap = ~k[j];
            const unsigned fg = kap & 1u;
            const float e = __uint_as_float(kap >> 1);                     // kappa key: never negative; NaN stays NaN and poisons the sum like the reference's
            const unsigned pos = base_all[d] + wave_hist[wave][d] + (rank[j] & 0xffffu);   // final 0-based position in the segment's sorted order
            const unsigned cum = base_fg[d] + wave_fg[wave][d] + (rank[j] >> 16);          // foreground keys in front of it
            // J_k and J_{k-1} (lovasz.py:29-33) with v_rcp_f32 instead of two IEEE divisions per key: both are functions of (G, k, cum)
            // alone, so a key's J_k IS its successor's J_{k-1} bit for bit and the sum telescopes exactly as with the divisions; the
            // 1-ulp reciprocal moves every J by <= 1.2e-7, i.e. the loss by <= 1.2e-7 x the total variation of the sorted errors (<= max e)
            const float cf = (float)cum, pf = (float)pos;
            const float ik = G - (cf + (float)fg), uk = G + ((pf + 1.0f) - (cf + (float)fg));
            const float ip = G - cf, up = G + (pf - cf);
            const float jk = 1.0f - ik * __builtin_amdgcn_rcpf(uk);
            const float jp = pos == 0u ? 0.0f : 1.0f - ip * __builtin_amdgcn_rcpf(up);
            acc += (double)(e * (jk - jp));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) wacc[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) t += wacc[w];
        tile_partial[(long long)seg * T + tile] = t;
    }
}

// (Measured and dropped: the same pass with one 16 KiB staging buffer used twice and packed 12-bit slots -- 22 KB of LDS, 80 VGPRs, 6
// workgroups per CU instead of 4 -- is 1 % slower: the pass is not waiting for occupancy.  2048-element tiles: +8 %; 8192: no change.)
// phase a (foreground count of every CHUNK of the sorted order) is rs_hist_kernel<COUNT>.  (Counting inside the last scatter pass instead
// -- one atomic per wave of staged slots, 262 k device-scope atomics on 8 192 counters -- made that pass 68 -> 176 us.)
// phase b: exclusive scan of the chunk counts of each segment (one workgroup per segment), total -> fg_total[s]
__device__ __forceinline__ void chunk_scan_block(unsigned seg, unsigned* __restrict__ chunk_count, int chunks_per_seg,
                                                 unsigned* __restrict__ fg_total) {
    __shared__ unsigned sh[256];
    __shared__ unsigned carry;
    unsigned* cc = chunk_count + (long long)seg * chunks_per_seg;
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
    if (threadIdx.x == 0) fg_total[seg] = carry;
}
__global__ __launch_bounds__(256) void lovasz_chunk_scan_kernel(unsigned* __restrict__ chunk_count, int chunks_per_seg,
                                                                unsigned* __restrict__ fg_total) {
    chunk_scan_block(blockIdx.x, chunk_count, chunks_per_seg, fg_total);
}
// The two scans between the binning pass's histogram kernel and its scatter -- the tiles of every span, the chunks of every segment --
// are independent of each other: one launch (a launch in this chain costs ~5 us whatever it does).
__global__ __launch_bounds__(256) void lovasz_scans_kernel(unsigned* __restrict__ hist, int T, int spans, unsigned* __restrict__ span_tot,
                                                           unsigned n_tilescan, unsigned* __restrict__ chunk_count, int chunks_per_seg,
                                                           unsigned* __restrict__ fg_total) {
    if (blockIdx.x < n_tilescan) tilescan_block(blockIdx.x, hist, T, spans, span_tot);
    else chunk_scan_block(blockIdx.x - n_tilescan, chunk_count, chunks_per_seg, fg_total);
}


// phase c: per element of the sorted order: cum fg -> grad_k = J_k - J_{k-1}; accumulate relu(e_k) * grad_k; scatter grad_k back
// to pixel order.  The scatter is 16.7 M four-byte writes to random addresses at [4,16,512,512].  Measured alternatives to the
// direct scatter, both dropped: keeping a segment's scatter on one XCD (b % 8 placement) so that its lines
// fill up in one L2: +6 %; returning through 16384-pixel bins (append (pixel, grad) runs to <= 1024 sequential streams, then
// order each bin in LDS and write it coalesced): 205 + 36 us against 198 us for the direct scatter.
template <bool KEYONLY = false>
__global__ __launch_bounds__(256) void lovasz_dot_kernel(const unsigned* __restrict__ keys, const unsigned* __restrict__ vals, long long P,
                                                         int chunks_per_seg, const unsigned* __restrict__ chunk_off,
                                                         const unsigned* __restrict__ fg_total, double* __restrict__ partial,
                                                         float* __restrict__ grad_at_pixel) {
    const int s = blockIdx.x / chunks_per_seg, k = blockIdx.x % chunks_per_seg;
    const long long base = (long long)s * P, i0 = (long long)k * CHUNK;
    const float G = (float)fg_total[s];
    // each thread owns 8 consecutive sorted positions
    const long long first = i0 + (long long)threadIdx.x * 8;
    unsigned v[8];
    float e[8];
    unsigned local = 0;
    if constexpr (KEYONLY) {       // key = ~(bits(e) << 1 | fg): error and foreground flag come out of the key itself
        unsigned kk[8];
        if (first + 8 <= P && ((base + first) & 3) == 0) {
            const uint4 ka = *reinterpret_cast<const uint4*>(keys + base + first), kb = *reinterpret_cast<const uint4*>(keys + base + first + 4);
            kk[0] = ka.x; kk[1] = ka.y; kk[2] = ka.z; kk[3] = ka.w; kk[4] = kb.x; kk[5] = kb.y; kk[6] = kb.z; kk[7] = kb.w;
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) kk[u] = first + u < P ? keys[base + first + u] : 0xFFFFFFFFu;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const unsigned kap = ~kk[u];
            v[u] = kap & 1u;
            e[u] = __uint_as_float(kap >> 1);
            local += v[u];
        }
    } else if (first + 8 <= P && ((base + first) & 3) == 0) {
        // 2 x 16-byte loads per array: with eight 4-byte loads a wave instruction touches 16 cache lines and uses an eighth of each
        const uint4 va = *reinterpret_cast<const uint4*>(vals + base + first), vb = *reinterpret_cast<const uint4*>(vals + base + first + 4);
        const uint4 ka = *reinterpret_cast<const uint4*>(keys + base + first), kb = *reinterpret_cast<const uint4*>(keys + base + first + 4);
        v[0] = va.x; v[1] = va.y; v[2] = va.z; v[3] = va.w; v[4] = vb.x; v[5] = vb.y; v[6] = vb.z; v[7] = vb.w;
        const unsigned kk[8] = {ka.x, ka.y, ka.z, ka.w, kb.x, kb.y, kb.z, kb.w};
#pragma unroll
        for (int u = 0; u < 8; ++u) { e[u] = __uint_as_float((~kk[u]) >> 1); local += v[u] & 1u; }
    } else {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long long i = first + u;
            v[u] = i < P ? vals[base + i] : 0u;
            e[u] = i < P ? __uint_as_float((~keys[base + i]) >> 1) : 0.0f;
            local += v[u] & 1u;
        }
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
    // J at the position before this thread's first one; from then on every J_k is the next element's J_{k-1} (one division per
    // element instead of two).  The telescoping difference is the reference's jaccard[1:] - jaccard[:-1]; re-evaluating J_{k-1} from
    // (float)(i + 1) - 1.0f has the same bits only while positions stay below 2^24 (segments above 16.7 M elements: the A/B
    // settings of tunables 17 / 19 may then differ in the last bits, the telescoping one being the reference's)
    float jprev = (first == 0 || first >= P) ? 0.0f : jaccard_at(G, (float)first, (float)cum);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const long long i = first + u;
        if (i < P) {
            const unsigned fg = v[u] & 1u;
            cum += fg;
            const float kf = (float)(unsigned)(i + 1);
            const float jk = jaccard_at(G, kf, (float)cum);
            const float g = jk - jprev;                       // lovasz.py:32-33
            jprev = jk;
            acc += (double)((e[u] <= 0.0f ? 0.0f : e[u]) * g);   // dot(relu(errors_sorted), grad), lovasz.py:71 / :139; a NaN error stays NaN (F.relu keeps it) and poisons the loss like the reference's
            if constexpr (!KEYONLY) {
                if (grad_at_pixel) grad_at_pixel[base + (v[u] >> 1)] = g;      // (NULL: forward only -- the 16.7 M random writes are a quarter of the call)
            }
        }
    }
    // One partial sum per workgroup, added up per segment by lovasz_segsum_kernel.  (Consecutive workgroups belong to the same
    // segment, so an fp64 atomic per wave meant ~2 000 atomics in a row on ONE address per segment: they serialise, and they -- not
    // the gradient scatter -- were 150 of this kernel's 200 us.)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    __shared__ double wacc[4];
    if (lane == 0) wacc[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = wacc[0] + wacc[1] + wacc[2] + wacc[3];
}

// seg_loss[s] = sum of the segment's per-chunk partial sums (fixed order: the result does not depend on scheduling)
__global__ __launch_bounds__(256) void lovasz_segsum_kernel(const double* __restrict__ partial, int chunks_per_seg, double* __restrict__ seg_loss) {
    const int s = blockIdx.x;
    double acc = 0.0;
    for (int k = threadIdx.x; k < chunks_per_seg; k += 256) acc += partial[(long long)s * chunks_per_seg + k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    __shared__ double wacc[4];
    if ((threadIdx.x & 63) == 0) wacc[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) seg_loss[s] = wacc[0] + wacc[1] + wacc[2] + wacc[3];
}

// backward: one thread per pixel (b, px), all classes: the label is read once (per class it is twice the bytes of a probability),
// every class is one coalesced read of pred and of the gradient at the pixel's rank and one streamed write.
template <int MODE>
__global__ __launch_bounds__(256) void lovasz_bwd_kernel(const LovArgs a, const float* __restrict__ coef,
                                                         const float* __restrict__ grad_at_pixel, float* __restrict__ grad) {
    const long long npx = (long long)a.B * a.HW;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < npx; q += stride) {
        const long long b = q / a.HW, px = q - b * a.HW;
        const long long i = a.per_image ? px : q;                 // element of its segment
        const long long j = a.per_image ? b : 0;                  // group
        bool valid;
        long long lab = 0;
        float y = 0.f;
        if constexpr (MODE == LOVASZ_SOFTMAX) {
            lab = a.labels[q];
            valid = !(a.has_ignore && lab == a.ignore_label);
        } else {
            y = a.flabels[q];
            valid = !(a.has_ignore && y == a.ignore_value);
        }
#pragma unroll 4
        for (int c = 0; c < a.C; ++c) {
            const long long po = (b * a.C + c) * a.HW + px;
            const long long s = j * a.C + c;
            const float p = a.pred[po];
            const float gp = grad_at_pixel[s * a.P + i];
            float gx = 0.f;
            if constexpr (MODE == LOVASZ_SOFTMAX) {
                const float fg = lab == c ? 1.f : 0.f;
                const float e = fabsf(fg - p);
                if (valid && e > 0.f) {
                    const float g = coef[s] * gp;
                    const float d = p - fg;                       // d|fg - p|/dp = sign(p - fg)
                    gx = d > 0.f ? g : (d < 0.f ? -g : 0.f);
                }
            } else {
                const float sg = 2.0f * y - 1.0f;
                const float e = 1.0f - p * sg;
                if (valid && e > 0.f) gx = -(coef[s] * gp) * sg;  // d(1 - x*sign)/dx
            }
            __builtin_nontemporal_store(gx, &grad[po]);
        }
    }
}

// backward from the BINNED gradient (ptb_lovasz_fwd_binned): block b of segment s -- its pixels b * bs .. b * bs + bs - 1 -- holds the
// (index << 1 | fg, gradient at the pixel's rank) pairs of exactly these pixels in arbitrary order; a workgroup puts them in pixel order
// in LDS and streams pred / grad like lovasz_bwd_kernel.  grid.x = group * blocks, grid.y = chunks of classes.
template <int MODE>
__global__ __launch_bounds__(256) void lovasz_bwd_binned_kernel(const LovArgs a, const float* __restrict__ coef, const float* __restrict__ gscale,
                                                                const unsigned* __restrict__ bvals,
                                                                const float* __restrict__ bgrad, float* __restrict__ grad, int bs_log2,
                                                                int nblocks, int cchunk) {
    extern __shared__ float gl[];
    const int j = blockIdx.x / nblocks, blk = blockIdx.x % nblocks;
    const int c0 = blockIdx.y * cchunk, c1 = min(a.C, c0 + cchunk);
    const unsigned bs = 1u << bs_log2;
    const long long i0 = (long long)blk << bs_log2;
    const int cnt = (int)min((long long)bs, a.P - i0);
    const unsigned HW = (unsigned)a.HW;
    if (bs_log2 == 12 && cnt == 4096) {
        // the common block (4096 pixels, 16 per thread): everything a class needs is requested before it is used, and the pixel's
        // label / offset is computed once for all the classes of the chunk
        unsigned off0[16];
        int labv[MODE == LOVASZ_SOFTMAX ? 16 : 1];
        float yv[MODE == LOVASZ_HINGE ? 16 : 1];
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            const unsigned i = (unsigned)i0 + threadIdx.x + 256u * m;
            const unsigned b = a.per_image ? (unsigned)j : i / HW;
            const unsigned px = a.per_image ? i : i - b * HW;
            off0[m] = b * (unsigned)a.C * HW + px;
            const long long lo = (long long)b * a.HW + px;
            if constexpr (MODE == LOVASZ_SOFTMAX) {
                const long long L = a.labels[lo];
                labv[m] = (a.has_ignore && L == a.ignore_label) ? -2 : ((L >= 0 && L < a.C) ? (int)L : -1);
            } else {
                yv[m] = a.flabels[lo];
            }
        }
        for (int c = c0; c < c1; ++c) {
            const long long s = (long long)j * a.C + c;
            const long long sb = s * a.P + i0;
            unsigned bv[16];
            float bg[16], pv[16];
#pragma unroll
            for (int m = 0; m < 16; ++m) { bv[m] = bvals[sb + threadIdx.x + 256 * m]; bg[m] = bgrad[sb + threadIdx.x + 256 * m]; }
#pragma unroll
            for (int m = 0; m < 16; ++m) pv[m] = a.pred[off0[m] + (unsigned)c * HW];
#pragma unroll
            for (int m = 0; m < 16; ++m) gl[(bv[m] >> 1) & 4095u] = bg[m];
            __syncthreads();
            const float cf = gscale ? gscale[0] * coef[s] : coef[s];      // (= the [S]-sized torch product, same rounding)
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                const float p = pv[m], gp = gl[threadIdx.x + 256 * m];
                float gx = 0.f;
                if constexpr (MODE == LOVASZ_SOFTMAX) {
                    const float fg = labv[m] == c ? 1.f : 0.f;
                    const float e = fabsf(fg - p);
                    if (labv[m] != -2 && e > 0.f) {
                        const float g = cf * gp;
                        const float d = p - fg;
                        gx = d > 0.f ? g : (d < 0.f ? -g : 0.f);
                    }
                } else {
                    const float y = yv[m];
                    const float sg = 2.0f * y - 1.0f;
                    const float e = 1.0f - p * sg;
                    if (!(a.has_ignore && y == a.ignore_value) && e > 0.f) gx = -(cf * gp) * sg;
                }
                __builtin_nontemporal_store(gx, &grad[off0[m] + (unsigned)c * HW]);
            }
            __syncthreads();
        }
        return;
    }
    for (int c = c0; c < c1; ++c) {
        const long long s = (long long)j * a.C + c;
        const long long sb = s * a.P + i0;
#pragma unroll 4
        for (int u = threadIdx.x; u < cnt; u += 256) gl[(bvals[sb + u] >> 1) & (bs - 1u)] = bgrad[sb + u];
        __syncthreads();
        const float cf = gscale ? gscale[0] * coef[s] : coef[s];      // (= the [S]-sized torch product, same rounding)
#pragma unroll 4
        for (int u = threadIdx.x; u < cnt; u += 256) {
            const unsigned i = (unsigned)(i0 + u);
            const unsigned b = a.per_image ? (unsigned)j : i / HW;
            const unsigned px = a.per_image ? i : i - b * HW;
            const long long po = ((long long)b * a.C + c) * a.HW + px, lo = (long long)b * a.HW + px;
            const float p = a.pred[po];
            const float gp = gl[u];
            float gx = 0.f;
            if constexpr (MODE == LOVASZ_SOFTMAX) {
                const long long lab = a.labels[lo];
                const bool valid = !(a.has_ignore && lab == a.ignore_label);
                const float fg = lab == c ? 1.f : 0.f;
                const float e = fabsf(fg - p);
                if (valid && e > 0.f) {
                    const float g = cf * gp;
                    const float d = p - fg;
                    gx = d > 0.f ? g : (d < 0.f ? -g : 0.f);
                }
            } else {
                const float y = a.flabels[lo];
                const bool valid = !(a.has_ignore && y == a.ignore_value);
                const float sg = 2.0f * y - 1.0f;
                const float e = 1.0f - p * sg;
                if (valid && e > 0.f) gx = -(cf * gp) * sg;
            }
            __builtin_nontemporal_store(gx, &grad[po]);
        }
        __syncthreads();
    }
}

// loss = mean over the groups of (sum over the selected classes of seg_loss / number of selected classes), and its derivative with
// respect to every seg_loss (losses/lovasz.py:92-108, :110-140: classes = "present" | "all"; the hinge loss is C = 1, "all").  One
// workgroup; thread = group, fixed summation order.
__device__ __forceinline__ void reduce_block(const double* seg_loss, const unsigned* __restrict__ fg_total, int groups,
                                             int C, int present_only, float* __restrict__ loss_out, float* __restrict__ coef_out) {
    __shared__ double part[256];
    double acc = 0.0;
    for (int g = threadIdx.x; g < groups; g += 256) {
        double sum = 0.0, cnt = 0.0;
        for (int c = 0; c < C; ++c) {
            const bool use = !present_only || fg_total[g * C + c] > 0u;
            if (use) { sum += seg_loss[g * C + c]; cnt += 1.0; }
        }
        const double den = cnt < 1.0 ? 1.0 : cnt;
        acc += sum / den;
        for (int c = 0; c < C; ++c) {
            const bool use = !present_only || fg_total[g * C + c] > 0u;
            coef_out[g * C + c] = use ? (float)(1.0 / (den * groups)) : 0.f;
        }
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) *loss_out = (float)(part[0] / groups);
}
__global__ __launch_bounds__(256) void lovasz_reduce_kernel(const double* __restrict__ seg_loss, const unsigned* __restrict__ fg_total, int groups,
                                                            int C, int present_only, float* __restrict__ loss_out, float* __restrict__ coef_out) {
    reduce_block(seg_loss, fg_total, groups, C, present_only, loss_out, coef_out);
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

int g_rs_xcd_map = 1;   // ptb_set_tunable key 17: XCD-contiguous tile order in the radix scatter
int g_lovasz_rankdot = 1;   // ptb_set_tunable key 23: the last level of the key-only Lovasz forward evaluates the loss from ranks instead of scattering (0: four scatters + count + dot)
int g_lovasz_fused_dot = 1;   // ptb_set_tunable key 19: the binning scatter of the training path also evaluates the loss (0: lovasz_dot_kernel)
static void launch_scatter(unsigned tiles, hipStream_t s, const unsigned* kin, const unsigned* vin, unsigned* kout, unsigned* vout, long long P, int T,
                           int shift, const unsigned* hist, int spans, const unsigned* span_tot, bool iota = false) {
    if (iota) {
        hipLaunchKernelGGL((rs_scatter_kernel<4, false, false, true>), dim3(tiles), dim3(256), 0, s, kin, (const unsigned*)nullptr, kout, vout, P, T, shift, hist,
                           spans, span_tot, g_rs_xcd_map, (const unsigned*)nullptr, (const unsigned*)nullptr, 0);
        return;
    }
    // (8 waves per workgroup -- 8 items per thread, 64 VGPRs, 24 waves per CU instead of 16 -- measured 4 % slower: more waves do not help this pass)
    hipLaunchKernelGGL((rs_scatter_kernel<4, false>), dim3(tiles), dim3(256), 0, s, kin, vin, kout, vout, P, T, shift, hist, spans, span_tot, g_rs_xcd_map,
                       (const unsigned*)nullptr, (const unsigned*)nullptr, 0);
}

static int blocks_for(long long n) {
    const long long want = (n + 255) / 256;
    return (int)(want < 1 ? 1 : (want < 256 * 16 ? want : 256 * 16));
}

}  // namespace ptb

using namespace ptb;

// bytes of sort workspace for `segments` segments of `per_segment` elements: the per-tile digit histograms
// u32[segments][tiles][256] followed by the span totals u32[segments][ceil(tiles / 32)][256] and the dot kernel's partial sums
// double[segments][ceil(per_segment / 2048)]
extern "C" int64_t ptb_lovasz_temp_bytes(int64_t per_segment, int segments) {
    if (per_segment < 0 || segments < 0) return -1;
    const int64_t tiles = (per_segment + RS_TILE - 1) / RS_TILE;
    const int64_t spans = (tiles + RS_SPAN - 1) / RS_SPAN;
    const int64_t chunks = (per_segment + CHUNK - 1) / CHUNK;
    // (round 6: twice the histogram space -- the last level of the key-only forward keeps a foreground histogram of the same shape)
    return 2 * ((int64_t)segments * 256 * tiles + (int64_t)segments * 256 * spans) * (int64_t)sizeof(unsigned) + (int64_t)segments * chunks * (int64_t)sizeof(double);
}

// Workspaces (all device, provided by the caller, n = P*S elements): keys_a, keys_b u32[n]; vals_a, vals_b u32[n];
// chunk u32[S*ceil(P/2048)]; fg_total u32[S]; seg_loss double[S] (written here);
// grad_at_pixel float[n] (kept for backward; NULL when no gradient will be asked for); temp = ptb_lovasz_temp_bytes bytes.
// pixels per block of the binned gradient: >= 4096 and <= 256 blocks per segment; -1 when a block would not fit the LDS budget
static int binned_block_log2(long long P) {
    int l = 12;
    while (((P + (1LL << l) - 1) >> l) > 256) ++l;
    return l <= 14 ? l : -1;
}

static int lovasz_fwd_impl(const float* pred, const int64_t* labels, const float* flabels, int B, int C, int64_t HW, int mode,
                           int per_image, int has_ignore, int64_t ignore_label, float ignore_value, uint32_t* keys_a, uint32_t* keys_b,
                           unsigned* vals_a, unsigned* vals_b, unsigned* chunk, unsigned* fg_total,
                           double* seg_loss, float* grad_at_pixel, void* temp, int64_t temp_bytes, ptb_stream_t stream, bool binned) {
    LovArgs a{};
    if (int rc = fill(a, pred, labels, flabels, B, C, HW, mode, per_image, has_ignore, ignore_label, ignore_value)) return rc;
    if (!keys_a || !keys_b || !vals_a || !vals_b || !chunk || !fg_total || !seg_loss) return PTB_EINVAL;
    const long long n = a.P * a.S;
    if (n == 0) return PTB_OK;
    if (n >= (1LL << 31)) return PTB_EUNSUPPORTED;  // offsets / packed indices are 32-bit
    if (!temp || temp_bytes < ptb_lovasz_temp_bytes(a.P, a.S)) return PTB_EINVAL;
    const int bl = binned ? binned_block_log2(a.P) : -1;
    if (binned && bl < 0) return PTB_EUNSUPPORTED;      // (before anything is launched: the caller falls back to the scattered gradient)
    hipStream_t s = (hipStream_t)stream;
    const int T = (int)((a.P + RS_TILE - 1) / RS_TILE);
    const long long tiles = (long long)T * a.S;
    if (tiles > 0x7fffffffLL) return PTB_EUNSUPPORTED;
    unsigned* hist = static_cast<unsigned*>(temp);
    unsigned* span_tot = hist + (long long)a.S * 256 * T;
    const int spans = (T + RS_SPAN - 1) / RS_SPAN;
    const int cps = (int)((a.P + CHUNK - 1) / CHUNK);
    const long long total_chunks = (long long)cps * a.S;
    if (total_chunks > 0x7fffffffLL) return PTB_EUNSUPPORTED;
    // errors -> (key, value) pairs + the first pass's tile histograms; classes are split over blockIdx.y until ~2048 workgroups exist
    {
        const long long gt = (long long)(a.S / a.C) * T;
        int nch = (int)((2048 + gt - 1) / gt);
        nch = nch < 1 ? 1 : (nch > a.C ? a.C : nch);
        const int cchunk = (a.C + nch - 1) / nch;
        nch = (a.C + cchunk - 1) / cchunk;
        if (nch > 65535) return PTB_EUNSUPPORTED;
        const dim3 grid((unsigned)gt, (unsigned)nch);
        // kappa keys (keyonly_key), no values written: the first scatter pass makes them (IOTA)
        if (a.mode == LOVASZ_SOFTMAX)
            hipLaunchKernelGGL(lovasz_error_hist_kernel<LOVASZ_SOFTMAX>, grid, dim3(256), 0, s, a, T, cchunk, keys_a, hist);
        else
            hipLaunchKernelGGL(lovasz_error_hist_kernel<LOVASZ_HINGE>, grid, dim3(256), 0, s, a, T, cchunk, keys_a, hist);
        if (int rc = check_launch()) return rc;
    }
    // four stable 8-bit passes, ping-ponging a -> b -> a -> b -> a
    unsigned *kin = keys_a, *kout = keys_b, *vin = vals_a, *vout = vals_b;
    for (int shift = 0; shift < 32; shift += 8) {
        if (shift) hipLaunchKernelGGL(rs_hist_kernel<false>, dim3((unsigned)tiles), dim3(256), 0, s, kin, a.P, T, shift, hist, (unsigned*)nullptr, 0);
        hipLaunchKernelGGL(rs_tilescan_kernel, dim3(a.S * spans), dim3(256), 0, s, hist, T, spans, span_tot);
        launch_scatter((unsigned)tiles, s, kin, vin, kout, vout, a.P, T, shift, hist, spans, span_tot, shift == 0);
        if (int rc = check_launch()) return rc;
        unsigned* tk = kin; kin = kout; kout = tk;
        unsigned* tv = vin; vin = vout; vout = tv;
    }
    // (an even number of passes: the sorted pairs are back in keys_a / vals_a)
    double* partial = reinterpret_cast<double*>(span_tot + (long long)a.S * 256 * spans);      // (behind the histograms: the binning pass needs them while the partial sums exist)
    if (binned) {
        // The gradient is BINNED, not scattered: one more pass of the sort's own kernels, keyed by the pixel block, groups the
        // (index << 1 | fg, gradient) pairs by block of 2^bl pixels (positions s * P + block * 2^bl ...: every pixel occurs once).
        // 16.7 M random 4-byte writes (177 us at [4,16,512,512]) become 64-byte runs and an LDS placement in the backward kernel.
        // Its histogram kernel also counts the foreground per chunk (no lovasz_count_kernel), its scatter kernel computes the
        // gradient it carries (the dot kernel writes none).
        const int shift = bl + 1;
        hipLaunchKernelGGL(rs_hist_kernel<true>, dim3((unsigned)tiles), dim3(256), 0, s, vin, a.P, T, shift, hist, chunk, cps);
        hipLaunchKernelGGL(lovasz_scans_kernel, dim3(a.S * spans + a.S), dim3(256), 0, s, hist, T, spans, span_tot, (unsigned)(a.S * spans), chunk, cps, fg_total);
        if (g_lovasz_fused_dot) {
            // the binning scatter also evaluates the loss (it computes every element's gradient anyway and reads its error key):
            // no lovasz_dot_kernel, the sorted values are read once instead of twice
            if (g_lovasz_rankdot)      // (the any-order ranking: tunable 23, with the key-only forward's)
                hipLaunchKernelGGL((rs_scatter_kernel<4, true, false, false, true>), dim3((unsigned)tiles), dim3(256), 0, s, vin, (const unsigned*)nullptr, keys_b, vals_b, a.P, T,
                                   shift, hist, spans, span_tot, g_rs_xcd_map, chunk, fg_total, cps, kin, partial);
            else
                hipLaunchKernelGGL((rs_scatter_kernel<4, true>), dim3((unsigned)tiles), dim3(256), 0, s, vin, (const unsigned*)nullptr, keys_b, vals_b, a.P, T, shift,
                                   hist, spans, span_tot, g_rs_xcd_map, chunk, fg_total, cps, kin, partial);
            hipLaunchKernelGGL(lovasz_segsum_kernel, dim3(a.S), dim3(256), 0, s, partial, T, seg_loss);
        } else {
            hipLaunchKernelGGL(lovasz_dot_kernel<true>, dim3((unsigned)total_chunks), dim3(256), 0, s, kin, (const unsigned*)nullptr, a.P, cps, chunk, fg_total, partial, (float*)nullptr);
            hipLaunchKernelGGL(lovasz_segsum_kernel, dim3(a.S), dim3(256), 0, s, partial, cps, seg_loss);
            hipLaunchKernelGGL((rs_scatter_kernel<4, true>), dim3((unsigned)tiles), dim3(256), 0, s, vin, (const unsigned*)nullptr, keys_b, vals_b, a.P, T, shift,
                               hist, spans, span_tot, g_rs_xcd_map, chunk, fg_total, cps);
        }
        if (int rc = check_launch()) return rc;
        return bl;
    }
    // (foreground per chunk: the tile-shaped kernel in its count-only form, 16 loads in flight per thread -- the chunk-shaped
    // lovasz_count_kernel it replaces took 21 us for the same 67 MB)
    hipLaunchKernelGGL((rs_hist_kernel<true, false>), dim3((unsigned)tiles), dim3(256), 0, s, vin, a.P, T, 0, hist, chunk, cps);
    hipLaunchKernelGGL(lovasz_chunk_scan_kernel, dim3(a.S), dim3(256), 0, s, chunk, cps, fg_total);
    hipLaunchKernelGGL(lovasz_dot_kernel<false>, dim3((unsigned)total_chunks), dim3(256), 0, s, kin, vin, a.P, cps, chunk, fg_total, partial, grad_at_pixel);
    hipLaunchKernelGGL(lovasz_segsum_kernel, dim3(a.S), dim3(256), 0, s, partial, cps, seg_loss);
    return check_launch();
}

extern "C" int ptb_lovasz_fwd(const float* pred, const int64_t* labels, const float* flabels, int B, int C, int64_t HW, int mode,
                              int per_image, int has_ignore, int64_t ignore_label, float ignore_value, uint32_t* keys_a, uint32_t* keys_b,
                              unsigned* vals_a, unsigned* vals_b, unsigned* chunk, unsigned* fg_total,
                              double* seg_loss, float* grad_at_pixel, void* temp, int64_t temp_bytes, ptb_stream_t stream) {
    return lovasz_fwd_impl(pred, labels, flabels, B, C, HW, mode, per_image, has_ignore, ignore_label, ignore_value, keys_a, keys_b, vals_a, vals_b,
                           chunk, fg_total, seg_loss, grad_at_pixel, temp, temp_bytes, stream, false);
}

// The forward WITHOUT a gradient (evaluation, torch.no_grad()): a key-only sort.  Nothing but the key travels through the four passes
// (see keyonly_key): 4 bytes per element and pass instead of 8, half the LDS staging, no value arrays at all.  seg_loss / fg_total as
// in ptb_lovasz_fwd; keys_a / keys_b u32[n], chunk u32[S * ceil(P / 2048)], temp = ptb_lovasz_temp_bytes bytes.
extern "C" int ptb_lovasz_fwd_keys(const float* pred, const int64_t* labels, const float* flabels, int B, int C, int64_t HW, int mode,
                                   int per_image, int has_ignore, int64_t ignore_label, float ignore_value, uint32_t* keys_a, uint32_t* keys_b,
                                   unsigned* chunk, unsigned* fg_total, double* seg_loss, void* temp, int64_t temp_bytes, ptb_stream_t stream) {
    LovArgs a{};
    if (int rc = fill(a, pred, labels, flabels, B, C, HW, mode, per_image, has_ignore, ignore_label, ignore_value)) return rc;
    if (!keys_a || !keys_b || !chunk || !fg_total || !seg_loss) return PTB_EINVAL;
    const long long n = a.P * a.S;
    if (n == 0) return PTB_OK;
    if (n >= (1LL << 31)) return PTB_EUNSUPPORTED;
    if (!temp || temp_bytes < ptb_lovasz_temp_bytes(a.P, a.S)) return PTB_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int T = (int)((a.P + RS_TILE - 1) / RS_TILE);
    const long long tiles = (long long)T * a.S;
    const int spans = (T + RS_SPAN - 1) / RS_SPAN;
    const int cps = (int)((a.P + CHUNK - 1) / CHUNK);
    const long long total_chunks = (long long)cps * a.S;
    if (tiles > 0x7fffffffLL || total_chunks > 0x7fffffffLL) return PTB_EUNSUPPORTED;
    unsigned* hist = static_cast<unsigned*>(temp);
    unsigned* span_tot = hist + (long long)a.S * 256 * T;
    double* partial = reinterpret_cast<double*>(span_tot + (long long)a.S * 256 * spans);
    {
        const long long gt = (long long)(a.S / a.C) * T;
        int nch = (int)((2048 + gt - 1) / gt);
        nch = nch < 1 ? 1 : (nch > a.C ? a.C : nch);
        const int cchunk = (a.C + nch - 1) / nch;
        nch = (a.C + cchunk - 1) / cchunk;
        if (nch > 65535) return PTB_EUNSUPPORTED;
        const dim3 grid((unsigned)gt, (unsigned)nch);
        if (a.mode == LOVASZ_SOFTMAX)
            hipLaunchKernelGGL(lovasz_error_hist_kernel<LOVASZ_SOFTMAX>, grid, dim3(256), 0, s, a, T, cchunk, keys_a, hist);
        else
            hipLaunchKernelGGL(lovasz_error_hist_kernel<LOVASZ_HINGE>, grid, dim3(256), 0, s, a, T, cchunk, keys_a, hist);
        if (int rc = check_launch()) return rc;
    }
    unsigned *kin = keys_a, *kout = keys_b;
    const int scatter_passes = g_lovasz_rankdot ? 3 : 4;
    for (int pass = 0; pass < scatter_passes; ++pass) {
        const int shift = 8 * pass;
        if (shift) hipLaunchKernelGGL(rs_hist_kernel<false>, dim3((unsigned)tiles), dim3(256), 0, s, kin, a.P, T, shift, hist, (unsigned*)nullptr, 0);
        hipLaunchKernelGGL(rs_tilescan_kernel, dim3(a.S * spans), dim3(256), 0, s, hist, T, spans, span_tot);
        if (pass == 0 && g_lovasz_rankdot)      // (tunable 23 = 0 keeps the round-5 pipeline as it was, for A/B runs)
            hipLaunchKernelGGL((rs_scatter_kernel<4, false, true, false, true>), dim3((unsigned)tiles), dim3(256), 0, s, kin, (const unsigned*)nullptr, kout, (unsigned*)nullptr,
                               a.P, T, shift, hist, spans, span_tot, g_rs_xcd_map, (const unsigned*)nullptr, (const unsigned*)nullptr, 0);
        else
            hipLaunchKernelGGL((rs_scatter_kernel<4, false, true>), dim3((unsigned)tiles), dim3(256), 0, s, kin, (const unsigned*)nullptr, kout, (unsigned*)nullptr, a.P, T,
                               shift, hist, spans, span_tot, g_rs_xcd_map, (const unsigned*)nullptr, (const unsigned*)nullptr, 0);
        if (int rc = check_launch()) return rc;
        unsigned* tk = kin; kin = kout; kout = tk;
    }
    if (g_lovasz_rankdot) {
        // the last level: positions and foreground counts from the histograms, no scatter (lovasz_rankdot_kernel)
        unsigned* hist_fg = reinterpret_cast<unsigned*>(partial + total_chunks);
        unsigned* span_tot_fg = hist_fg + (long long)a.S * 256 * T;
        hipLaunchKernelGGL((rs_hist_kernel<false, true, false, true>), dim3((unsigned)tiles), dim3(256), 0, s, kin, a.P, T, 24, hist, (unsigned*)nullptr, 0, hist_fg);
        hipLaunchKernelGGL(rs_tilescan2_kernel, dim3(2 * a.S * spans), dim3(256), 0, s, hist, hist_fg, T, spans, span_tot, span_tot_fg, (unsigned)(a.S * spans));
        hipLaunchKernelGGL(lovasz_rankdot_kernel<4>, dim3((unsigned)tiles), dim3(256), 0, s, kin, a.P, T, 24, hist, hist_fg, spans, span_tot, span_tot_fg,
                           g_rs_xcd_map, fg_total, partial);
        hipLaunchKernelGGL(lovasz_segsum_kernel, dim3(a.S), dim3(256), 0, s, partial, T, seg_loss);
        return check_launch();
    }
    hipLaunchKernelGGL((rs_hist_kernel<true, false, true>), dim3((unsigned)tiles), dim3(256), 0, s, kin, a.P, T, 0, hist, chunk, cps);
    hipLaunchKernelGGL(lovasz_chunk_scan_kernel, dim3(a.S), dim3(256), 0, s, chunk, cps, fg_total);
    hipLaunchKernelGGL(lovasz_dot_kernel<true>, dim3((unsigned)total_chunks), dim3(256), 0, s, kin, (const unsigned*)nullptr, a.P, cps, chunk, fg_total, partial, (float*)nullptr);
    hipLaunchKernelGGL(lovasz_segsum_kernel, dim3(a.S), dim3(256), 0, s, partial, cps, seg_loss);
    return check_launch();
}


extern "C" int ptb_lovasz_fwd_binned(const float* pred, const int64_t* labels, const float* flabels, int B, int C, int64_t HW, int mode,
                                     int per_image, int has_ignore, int64_t ignore_label, float ignore_value, uint32_t* keys_a, uint32_t* keys_b,
                                     unsigned* vals_a, unsigned* vals_b, unsigned* chunk, unsigned* fg_total,
                                     double* seg_loss, float* scratch, void* temp, int64_t temp_bytes, ptb_stream_t stream) {
    return lovasz_fwd_impl(pred, labels, flabels, B, C, HW, mode, per_image, has_ignore, ignore_label, ignore_value, keys_a, keys_b, vals_a, vals_b,
                           chunk, fg_total, seg_loss, scratch, temp, temp_bytes, stream, true);
}

static int lovasz_bwd_binned_impl(const float* pred, const int64_t* labels, const float* flabels, const float* coef, const float* gscale,
                                  const uint32_t* binned_vals, const float* binned_grad, float* grad, int B, int C, int64_t HW, int mode,
                                  int per_image, int has_ignore, int64_t ignore_label, float ignore_value, int block_log2, ptb_stream_t stream) {
    LovArgs a{};
    if (int rc = fill(a, pred, labels, flabels, B, C, HW, mode, per_image, has_ignore, ignore_label, ignore_value)) return rc;
    if (!coef || !binned_vals || !binned_grad || !grad || block_log2 < 12 || block_log2 > 14) return PTB_EINVAL;
    const long long n = a.P * a.S;
    if (n == 0) return PTB_OK;
    if (n >= (1LL << 31)) return PTB_EUNSUPPORTED;
    const long long nblocks = (a.P + (1LL << block_log2) - 1) >> block_log2;
    if (nblocks > 256) return PTB_EINVAL;
    const long long gx = (long long)(a.S / a.C) * nblocks;
    int nch = (int)((2048 + gx - 1) / gx);
    nch = nch < 1 ? 1 : (nch > a.C ? a.C : nch);
    const int cchunk = (a.C + nch - 1) / nch;
    nch = (a.C + cchunk - 1) / cchunk;
    if (gx > 0x7fffffffLL || nch > 65535) return PTB_EUNSUPPORTED;
    const size_t lds = sizeof(float) << block_log2;
    const dim3 grid((unsigned)gx, (unsigned)nch);
    if (a.mode == LOVASZ_SOFTMAX)
        hipLaunchKernelGGL(lovasz_bwd_binned_kernel<LOVASZ_SOFTMAX>, grid, dim3(256), lds, (hipStream_t)stream, a, coef, gscale, binned_vals, binned_grad, grad,
                           block_log2, (int)nblocks, cchunk);
    else
        hipLaunchKernelGGL(lovasz_bwd_binned_kernel<LOVASZ_HINGE>, grid, dim3(256), lds, (hipStream_t)stream, a, coef, gscale, binned_vals, binned_grad, grad,
                           block_log2, (int)nblocks, cchunk);
    return check_launch();
}

extern "C" int ptb_lovasz_bwd_binned(const float* pred, const int64_t* labels, const float* flabels, const float* coef, const uint32_t* binned_vals,
                                     const float* binned_grad, float* grad, int B, int C, int64_t HW, int mode, int per_image, int has_ignore,
                                     int64_t ignore_label, float ignore_value, int block_log2, ptb_stream_t stream) {
    return lovasz_bwd_binned_impl(pred, labels, flabels, coef, nullptr, binned_vals, binned_grad, grad, B, C, HW, mode, per_image, has_ignore, ignore_label,
                                  ignore_value, block_log2, stream);
}

// ptb_lovasz_bwd_binned with the incoming gradient of the scalar loss as a DEVICE scalar: coefficient of segment s = gscale[0] * coef_unit[s]
// (coef_unit = ptb_lovasz_reduce's coef_out), multiplied inside the kernel instead of by an [S]-sized launch in front of it
extern "C" int ptb_lovasz_bwd_binned2(const float* pred, const int64_t* labels, const float* flabels, const float* coef_unit, const float* gscale,
                                      const uint32_t* binned_vals, const float* binned_grad, float* grad, int B, int C, int64_t HW, int mode,
                                      int per_image, int has_ignore, int64_t ignore_label, float ignore_value, int block_log2, ptb_stream_t stream) {
    if (!gscale) return PTB_EINVAL;
    return lovasz_bwd_binned_impl(pred, labels, flabels, coef_unit, gscale, binned_vals, binned_grad, grad, B, C, HW, mode, per_image, has_ignore,
                                  ignore_label, ignore_value, block_log2, stream);
}

extern "C" int ptb_lovasz_bwd(const float* pred, const int64_t* labels, const float* flabels, const float* coef,
                              const float* grad_at_pixel, float* grad, int B, int C, int64_t HW, int mode, int per_image, int has_ignore,
                              int64_t ignore_label, float ignore_value, ptb_stream_t stream) {
    LovArgs a{};
    if (int rc = fill(a, pred, labels, flabels, B, C, HW, mode, per_image, has_ignore, ignore_label, ignore_value)) return rc;
    if (!coef || !grad_at_pixel || !grad) return PTB_EINVAL;
    const long long n = a.P * a.S;
    if (n == 0) return PTB_OK;
    const int blocks = blocks_for((long long)a.B * a.HW);
    if (a.mode == LOVASZ_SOFTMAX)
        hipLaunchKernelGGL(lovasz_bwd_kernel<LOVASZ_SOFTMAX>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, coef, grad_at_pixel, grad);
    else
        hipLaunchKernelGGL(lovasz_bwd_kernel<LOVASZ_HINGE>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, coef, grad_at_pixel, grad);
    return check_launch();
}

extern "C" int ptb_lovasz_reduce(const double* seg_loss, const unsigned* fg_total, int groups, int C, int present_only, float* loss_out,
                                 float* coef_out, ptb_stream_t stream) {
    if (!seg_loss || !fg_total || !loss_out || !coef_out || groups < 1 || C < 1) return PTB_EINVAL;
    hipLaunchKernelGGL(lovasz_reduce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, seg_loss, fg_total, groups, C, present_only, loss_out, coef_out);
    return check_launch();
}
