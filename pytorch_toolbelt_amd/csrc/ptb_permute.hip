// ptb_permute.hip -- the eight D4 image views as pure data movement for ANY element width (gfx950 / MI355X).
//
// Reference: inference/functional.py:47-132 -- torch_fliplr / torch_flipud / torch_rot90_cw / ... are x.flip(3), x.flip(2),
// x.rot90(k, dims=(2, 3)), x.transpose(2, 3): index permutations of dims 2 and 3 of a tensor of any dtype and of any rank >= 4
// (dims beyond the fourth ride along with their pixel).  The fp32 view kernels of ptb_views.hip multiply by a scale and reduce; a
// label mask (int64), a boolean mask, a float64 map or a 5-D tensor must come out with exactly the bits it went in with, so this
// kernel moves opaque elements of 1 / 2 / 4 / 8 / 16 bytes:
//
//     out[k*B*C + p][i][j][e] = in[src plane][si][sj][e],   (a, b) = transpose_k ? (j, i) : (i, j),
//                               si = flip_rows_k ? H-1-a : a,   sj = flip_cols_k ? W-1-b : b,   e < run
//
// with the library's 3-bit view code (bit 0 transpose, bit 1 flip source rows, bit 2 flip source cols, include/ptb_hip.h).  The
// source is [planes or V*planes][H][W][run], the output [V*planes][Ho][Wo][run] with (Ho, Wo) = (W, H) for transposing views -- non-square
// planes are fine, like x.rot90 on a non-square tensor.
//
// One workgroup moves one TS x TS tile of a SOURCE plane through LDS to its place in every view of the call (see view_permute_kernel):
// an augmented batch is one read and V writes of the image, 16 bytes per lane on both global sides whenever the rows allow.
#include "ptb_common.h"

namespace ptb {

typedef unsigned int E16 __attribute__((ext_vector_type(4)));     // an opaque 16-byte element (complex128, or 2 x 8 bytes of trailing dims)

// tile edge in elements: rows of at least 128 bytes, tiles of at most 32 KB
template <typename T> struct TileSize { static constexpr int value = sizeof(T) <= 2 ? 128 : (sizeof(T) <= 8 ? 64 : 32); };

struct PermArgs {
    const void* in;
    void* out;
    int V, codes;              // 3 bits per view
    int in_is_batch;           // 1: every view reads plane p (ONE read of the tile feeds all V views); 0: view k reads plane k * planes + p
    long long planes;          // B * C
    int H, W;                  // source plane
    int tiles_y, tiles_x;      // of the SOURCE plane
    int vec;                   // 1: both planes' rows are 16-byte multiples at 16-byte aligned bases (16-byte global accesses)
};

// One workgroup = one TS x TS tile of a SOURCE plane.  The tile is read once -- 16 bytes per lane along the source rows where the
// rows allow -- into a padded LDS tile, then written to its place in every view of the call (augment: up to 8 views out of one read;
// de-augment without a reduction: the view of that chunk): 16 bytes per lane along the OUTPUT rows, each lane collecting its
// 16 / sizeof(T) elements from the tile at their source positions.  Both global sides are coalesced for all eight views; the
// permutation happens on the LDS side, an element at a time.
// 16 bytes of 1- or 2-byte elements with the element order reversed (a mirrored row segment)
template <typename T>
__device__ __forceinline__ E16 reverse16(const E16 v) {
    if constexpr (sizeof(T) == 1) return E16{__builtin_bswap32(v.w), __builtin_bswap32(v.z), __builtin_bswap32(v.y), __builtin_bswap32(v.x)};
    else return E16{(v.w >> 16) | (v.w << 16), (v.z >> 16) | (v.z << 16), (v.y >> 16) | (v.y << 16), (v.x >> 16) | (v.x << 16)};
}

// Label masks and half-precision maps (1- and 2-byte elements), FULL tiles with 16-byte rows (round 6).  The generic kernel below moves
// these an element at a time through LDS: 16 (8) byte-sized LDS operations per 16 bytes read and again per 16 bytes written to every view
// -- it ran at 35 % / 62 % of the HBM rate the 4-byte elements reach.  Here the tile is kept TWICE, as it lies in the source (one
// ds_write_b128 per lane) and transposed (the lane's 16 / 8 elements scattered one by one: the only element-sized LDS traffic left), so
// EVERY view is one ds_read_b128 per 16 output bytes -- from the straight copy for the row-preserving views, from the transposed one for
// the transposing views -- plus, for a mirrored axis along the output row, a reversal of the 16 bytes in registers.  Rows are whole
// multiples of 128 bytes (every row starts at bank 0), so the 16-byte groups of row R are stored at group ^ (R / N): the transposing
// scatter of a wave (8 source rows x 8 or 4 x 16 chunks: LDS rows N apart) then spreads over the banks instead of piling onto one.
template <typename T>
__device__ __forceinline__ void permute_full_tile_small(const PermArgs& a, const T* __restrict__ src, long long p, int k0, int r0, int c0, unsigned char* lds) {
    constexpr int TS = TileSize<T>::value, SZ = (int)sizeof(T);
    constexpr int N = 16 / SZ;                   // elements per 16-byte group
    constexpr int PB = TS * SZ;                  // bytes per LDS row
    constexpr int CPR = PB / 16;                 // 16-byte groups per row
    static_assert(SZ <= 2 && PB % 128 == 0, "1- and 2-byte elements");
    unsigned char* S = lds;                      // S[r][c]   = source (r, c)
    unsigned char* Tt = lds + TS * PB;           // Tt[c][r]  = source (r, c)
    const int tid = threadIdx.x;
    auto grp = [](int row, int g) { return (g ^ (row / N)) & (CPR - 1); };
    for (int e = tid; e < TS * CPR; e += 256) {
        const int r = e / CPR, g = e - r * CPR;
        union { E16 v; T t[N]; } u;
        u.v = __builtin_nontemporal_load(reinterpret_cast<const E16*>(src + (long long)r * a.W + g * N));
        *reinterpret_cast<E16*>(S + r * PB + grp(r, g) * 16) = u.v;
        const int tg = (r * SZ) / 16, to = (r * SZ) % 16;            // where element (., r) sits in a row of Tt
#pragma unroll
        for (int m = 0; m < N; ++m) {
            const int R = g * N + m;
            *reinterpret_cast<T*>(Tt + R * PB + grp(R, tg) * 16 + to) = u.t[m];
        }
    }
    __syncthreads();
    const int nk = a.in_is_batch ? a.V : 1;
    for (int kk = 0; kk < nk; ++kk) {
        const int k = k0 + kk;
        const int code = (a.codes >> (3 * k)) & 7;
        const bool tr = code & 1, fr = code & 2, fc = code & 4;
        const int Ho = tr ? a.W : a.H, Wo = tr ? a.H : a.W;
        const int a0 = fr ? a.H - r0 - TS : r0, b0 = fc ? a.W - c0 - TS : c0;
        const int i0 = tr ? b0 : a0, j0 = tr ? a0 : b0;
        T* __restrict__ dst = static_cast<T*>(a.out) + (((long long)k * a.planes + p) * Ho + i0) * (long long)Wo + j0;
        for (int e = tid; e < TS * CPR; e += 256) {
            const int i = e / CPR, g = e - i * CPR;                  // output-tile row, 16-byte group along it
            // output (i, j): (aa, bb) = tr ? (j, i) : (i, j); source row = fr ? TS-1-aa : aa, source col = fc ? TS-1-bb : bb
            E16 v;
            if (!tr) {
                const int row = fr ? TS - 1 - i : i;
                const int sg = fc ? CPR - 1 - g : g;                 // mirrored columns: the group from the other end, reversed below
                v = *reinterpret_cast<const E16*>(S + row * PB + grp(row, sg) * 16);
                if (fc) v = reverse16<T>(v);
            } else {
                const int row = fc ? TS - 1 - i : i;                 // a row of Tt = a source COLUMN
                const int sg = fr ? CPR - 1 - g : g;                 // along it run the source ROWS
                v = *reinterpret_cast<const E16*>(Tt + row * PB + grp(row, sg) * 16);
                if (fr) v = reverse16<T>(v);
            }
            __builtin_nontemporal_store(v, reinterpret_cast<E16*>(dst + (long long)i * Wo + g * N));
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void view_permute_kernel(const PermArgs a) {
    constexpr int TS = TileSize<T>::value;
    constexpr int N = 16 / (int)sizeof(T);        // elements per 16-byte access
    constexpr bool SMALL = sizeof(T) <= 2;
    // (1- / 2-byte elements: the two copies of permute_full_tile_small; a partial tile lays its padded tile over the same bytes)
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[SMALL ? 2 * TS * TS * sizeof(T) : sizeof(T) * TS * (TS + 1)];
    static_assert(!SMALL || 2 * TS * TS >= TS * (TS + 1), "the padded tile fits");
    T (*tile)[TS + 1] = reinterpret_cast<T (*)[TS + 1]>(lds_raw);
    const int tid = threadIdx.x;
    long long bid = blockIdx.x;
    const int tx = (int)(bid % a.tiles_x); bid /= a.tiles_x;
    const int ty = (int)(bid % a.tiles_y); bid /= a.tiles_y;
    const long long p = bid % a.planes;
    const int k0 = a.in_is_batch ? 0 : (int)(bid / a.planes);
    const int r0 = ty * TS, c0 = tx * TS;                          // the tile's origin in the source plane
    const int sh = min(TS, a.H - r0), sw = min(TS, a.W - c0);      // ... and extent
    const T* __restrict__ src = static_cast<const T*>(a.in) + ((a.in_is_batch ? p : (long long)k0 * a.planes + p) * a.H + r0) * (long long)a.W + c0;
    if constexpr (SMALL) {
        if (a.vec && sw == TS && sh == TS) {      // (wave-uniform: the whole workgroup takes one path)
            permute_full_tile_small<T>(a, src, p, k0, r0, c0, lds_raw);
            return;
        }
    }
    if (a.vec && sw == TS) {
        constexpr int CPR = TS / N;                                // 16-byte chunks per tile row
        for (int e = tid; e < sh * CPR; e += 256) {
            const int r = e / CPR, c = (e - r * CPR) * N;
            union { E16 v; T t[N]; } u;
            u.v = __builtin_nontemporal_load(reinterpret_cast<const E16*>(src + (long long)r * a.W + c));
#pragma unroll
            for (int m = 0; m < N; ++m) tile[r][c + m] = u.t[m];
        }
    } else {
        for (int e = tid; e < sh * TS; e += 256) {
            const int r = e / TS, c = e - r * TS;
            if (c < sw) tile[r][c] = src[(long long)r * a.W + c];
        }
    }
    __syncthreads();
    const int nk = a.in_is_batch ? a.V : 1;
    for (int kk = 0; kk < nk; ++kk) {
        const int k = k0 + kk;
        const int code = (a.codes >> (3 * k)) & 7;
        const bool tr = code & 1, fr = code & 2, fc = code & 4;
        const int Ho = tr ? a.W : a.H, Wo = tr ? a.H : a.W;
        // where the tile lands: source (r, c) -> (a, b) = (fr ? H-1-r : r, fc ? W-1-c : c) -> output (i, j) = tr ? (b, a) : (a, b)
        const int a0 = fr ? a.H - r0 - sh : r0, b0 = fc ? a.W - c0 - sw : c0;      // origin of the (flipped) rectangle
        const int i0 = tr ? b0 : a0, j0 = tr ? a0 : b0;
        const int oh = tr ? sw : sh, ow = tr ? sh : sw;                            // extent of the output tile
        T* __restrict__ dst = static_cast<T*>(a.out) + (((long long)k * a.planes + p) * Ho + i0) * (long long)Wo + j0;
        auto at = [&](int i, int j) -> T {                                          // output-tile (i, j) -> the element in the LDS tile
            const int aa = tr ? j : i, bb = tr ? i : j;
            return tile[fr ? sh - 1 - aa : aa][fc ? sw - 1 - bb : bb];
        };
        if (a.vec && ow == TS && (j0 % N) == 0) {
            constexpr int CPR = TS / N;
            for (int e = tid; e < oh * CPR; e += 256) {
                const int i = e / CPR, j = (e - i * CPR) * N;
                union { E16 v; T t[N]; } u;
#pragma unroll
                for (int m = 0; m < N; ++m) u.t[m] = at(i, j + m);
                __builtin_nontemporal_store(u.v, reinterpret_cast<E16*>(dst + (long long)i * Wo + j));
            }
        } else {
            for (int e = tid; e < oh * TS; e += 256) {
                const int i = e / TS, j = e - i * TS;
                if (j < ow) dst[(long long)i * Wo + j] = at(i, j);
            }
        }
    }
}

// run > 1 (dims beyond the fourth): one lane per output element, the pixel's `run` elements stay together
template <typename T>
__global__ __launch_bounds__(256) void view_permute_run_kernel(const PermArgs a, long long run, long long total) {
    for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long long)gridDim.x * blockDim.x) {
        long long rest = o;
        const long long e = rest % run; rest /= run;
        const int code0 = a.codes & 7;
        const int Wo = (code0 & 1) ? a.H : a.W, Ho = (code0 & 1) ? a.W : a.H;      // (all views of one call share the output shape)
        const int j = (int)(rest % Wo); rest /= Wo;
        const int i = (int)(rest % Ho); rest /= Ho;
        const long long p = rest % a.planes;
        const int k = (int)(rest / a.planes);
        const int code = (a.codes >> (3 * k)) & 7;
        const int aa = (code & 1) ? j : i, bb = (code & 1) ? i : j;
        const int si = (code & 2) ? a.H - 1 - aa : aa, sj = (code & 4) ? a.W - 1 - bb : bb;
        const long long sp = a.in_is_batch ? p : (long long)k * a.planes + p;
        static_cast<T*>(a.out)[o] = static_cast<const T*>(a.in)[((sp * a.H + si) * (long long)a.W + sj) * run + e];
    }
}

template <typename T>
static int launch_permute(const PermArgs& a, long long run, hipStream_t s) {
    const int tr0 = a.codes & 1;
    const int Ho = tr0 ? a.W : a.H, Wo = tr0 ? a.H : a.W;
    if (run == 1) {
        constexpr int TS = TileSize<T>::value;
        constexpr int N = 16 / (int)sizeof(T);
        PermArgs b = a;
        b.tiles_y = (a.H + TS - 1) / TS;
        b.tiles_x = (a.W + TS - 1) / TS;
        // 16-byte accesses: rows of both planes are whole 16-byte chunks from 16-byte aligned bases (then every row starts aligned)
        b.vec = !g_force_scalar && a.W % N == 0 && Wo % N == 0 && aligned16(a.in) && aligned16(a.out) ? 1 : 0;
        const long long blocks = (long long)(a.in_is_batch ? 1 : a.V) * a.planes * b.tiles_y * b.tiles_x;
        if (blocks > 0x7fffffffLL) return PTB_EUNSUPPORTED;
        hipLaunchKernelGGL(view_permute_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, s, b);
    } else {
        const long long total = (long long)a.V * a.planes * Ho * Wo * run;
        const long long want = (total + 255) / 256;
        hipLaunchKernelGGL(view_permute_run_kernel<T>, dim3((unsigned)(want < 65536 ? want : 65536)), dim3(256), 0, s, a, run, total);
    }
    return check_launch();
}

}  // namespace ptb

using namespace ptb;

extern "C" int ptb_view_permute(const void* in, void* out, int V, const int* views, int in_is_batch, int64_t planes, int H, int W,
                                int elem_bytes, int64_t run, ptb_stream_t stream) {
    if (!in || !out || !views || V < 1 || V > 8 || planes < 0 || H < 0 || W < 0 || run < 1) return PTB_EINVAL;
    int codes = 0, nT = 0;
    for (int k = 0; k < V; ++k) {
        if (views[k] < 0 || views[k] > 7) return PTB_EINVAL;
        codes |= views[k] << (3 * k);
        nT += views[k] & 1;
    }
    if (nT != 0 && nT != V && H != W) return PTB_EINVAL;      // transposing and row-preserving views of one call share one output shape
    if (elem_bytes != 1 && elem_bytes != 2 && elem_bytes != 4 && elem_bytes != 8 && elem_bytes != 16) return PTB_EINVAL;
    if (planes == 0 || H == 0 || W == 0) return PTB_OK;
    const uintptr_t both = reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out);
    if (both & (uintptr_t)(elem_bytes - 1)) return PTB_EUNSUPPORTED;      // (elements must be naturally aligned: torch storage is)
    PermArgs a{};
    a.in = in; a.out = out; a.V = V; a.codes = codes; a.in_is_batch = in_is_batch ? 1 : 0; a.planes = planes; a.H = H; a.W = W;
    hipStream_t s = (hipStream_t)stream;
    switch (elem_bytes) {
        case 1: return launch_permute<unsigned char>(a, run, s);
        case 2: return launch_permute<unsigned short>(a, run, s);
        case 4: return launch_permute<unsigned int>(a, run, s);
        case 8: return launch_permute<unsigned long long>(a, run, s);
        case 16: return launch_permute<E16>(a, run, s);
        default: return PTB_EINVAL;
    }
}
