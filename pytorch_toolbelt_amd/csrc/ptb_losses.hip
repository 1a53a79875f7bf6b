// ptb_losses.hip -- fused segmentation-loss reductions for gfx950 (MI355X).
//
// Reference: losses/functional.py:19-107 (focal_loss_with_logits), :110-173 (softmax_focal_loss_with_logits),
// :188-247 (soft_jaccard_score / soft_dice_score), losses/focal.py:77-105, losses/dice.py:59-131, losses/jaccard.py:48-103.
// The reference evaluates each loss as 12-15 full-tensor torch ops (and materialises an int64 one-hot, 1 GiB at
// [32,16,512,512]).  Here ONE pass over logits + labels produces every scalar the losses need:
//     sums[0] = sum of focal losses, sums[1] = sum of focal terms (normalised focal),
//     per class c: I_c = sum p*t, P_c = sum p, T_c = sum t   (p = softmax / sigmoid / given probabilities, masked)
// from which Dice, Jaccard and their log variants are [C]-sized scalar algebra.  Backward kernels recompute the
// activations from the logits (one more read, one gradient write) instead of storing any intermediate.
//
// Layout: logits [B, C, HW] fp32.  A wave owns 64*PIX consecutive pixels of one image (lane = PIX pixels, one 16 B load
// per class plane when PIX = 4).  For C <= CREG the lane first issues the loads of ALL class planes (CREG x 16 B in
// flight per lane -- this is what hides HBM latency), keeps them in registers, and does softmax with one exp per
// element.  Per-class sums are kept per lane across the wave's groups and reduced across lanes once, at the end;
// they leave the workgroup as one fp64 atomic per (statistic, class).  Streaming + transcendental work: no MFMA.
#include <initializer_list>

#include <type_traits>

#include "ptb_loss_device.h"

// the losses are tolerance-checked (1e-5), not bit-exact: let the compiler fuse multiply-adds in this file
#pragma clang fp contract(fast)

namespace ptb {

// focal contribution of one element; returns the (weighted) loss, adds the focal term to fsum
template <bool G2>
__device__ __forceinline__ float focal_one(float x, float t, bool ig, float cw, const FocalCfg& c, float& fsum) {
    float ce, f, df, p;
    focal_parts<G2, false>(x, ig ? 0.f : t, c, ce, f, df, p);
    float l = f * ce * (c.a1 * t + c.a0) * cw;
    l = ig ? 0.f : l;
    f = ig ? f * c.term_mask : f;
    fsum += f;
    return l;
}

// Focal contribution of one element with a HARD target from sigmoid(x) = ps and 1 - sigmoid(x) = qs (both already formed,
// see seg_loss_fwd_reg_kernel): BCE = -log(p_t), 1 - p_t without cancellation.  Same result as focal_one to rounding.
// PLAIN = gamma 2, no alpha, no reduced threshold, no ignore, no class weights (the default configuration): everything
// those options cost per element is compiled out.
template <bool G2, bool PLAIN = false>
__device__ __forceinline__ float focal_hard(float ps, float qs, bool t, bool ig, float cw, const FocalCfg& c, float& f) {
    if constexpr (PLAIN) {
        const float pt = t ? ps : qs, omp = t ? qs : ps;
        f = omp * omp;
        return f * (-lg2(pt) * kLn2);
    }
    const bool tt = t && !ig;
    const float pt = tt ? ps : qs, omp = tt ? qs : ps;
    const float ce = -lg2(pt) * kLn2;
    const float base = omp * c.sc;
    if (G2) f = base * base;
    else { f = pow_pos(base, c.gamma); f = c.g0 ? 1.0f : f; }
    f = pt < c.thr ? 1.0f : f;
    float l = f * ce * (t ? c.a1 + c.a0 : c.a0) * cw;
    l = ig ? 0.f : l;
    f = ig ? f * c.term_mask : f;
    return l;
}

// ------------------------------------------------------------------------------------------------ forward
// Opaque register barrier: stops the compiler from keeping exp()/sigmoid() results of one pass alive for the next
// (recomputing is cheaper than 64 extra VGPRs per lane, which would halve occupancy).
__device__ __forceinline__ void opaque(float& v) { asm volatile("" : "+v"(v)); }

// Register-resident variant: C <= CREG, the lane issues the loads of ALL class planes first (CREG x 16 B in flight).
// WHAT = SEG_FOCAL | SEG_STATS bits (compile time), DENSE = dense fp32 targets instead of int64 labels.
// SHARE (focal + softmax statistics on hard labels only): one exp per element, see the comment at `share` below.
template <int PIX, int CREG, int WHAT, bool DENSE, bool G2, bool SHARE = false, bool PLAIN = false>
__global__ __launch_bounds__(256, SHARE ? 3 : 1) void seg_loss_fwd_reg_kernel(const SegArgs a) {
    const FocalCfg cfg = focal_cfg(a);
    extern __shared__ float lds[];  // [4 waves][3][C]
    const int lane = threadIdx.x & 63, wave = wave_id();
    const int C = a.C;
    float* wl = lds + wave * 3 * C;
    constexpr bool stats = WHAT & SEG_STATS, focal = WHAT & SEG_FOCAL;
    const bool ignf = a.flags & SEG_HAS_IGNORE;
    double f_loss = 0.0, f_term = 0.0;
    float aI[CREG], aP[CREG], aT[CREG];
#pragma unroll
    for (int c = 0; c < CREG; ++c) { aI[c] = 0.f; aP[c] = 0.f; aT[c] = 0.f; }
    const long long per_img = (a.HW + 64 * PIX - 1) / (64 * PIX);
    const long long groups = per_img * a.B;
    for (long long g = (long long)blockIdx.x * 4 + wave; g < groups; g += (long long)gridDim.x * 4) {
        const Group<PIX> G = make_group<PIX>(g, per_img, lane, a.HW, DENSE ? nullptr : a.labels, ignf, a.ignore_label, C, a.error_flag);
        const long long base = (long long)G.b * C * a.HW + G.i0;
        int lab[PIX];
#pragma unroll
        for (int k = 0; k < PIX; ++k) lab[k] = (int)G.lab[k];
        float lsum = 0.f, fsum = 0.f;
        float xv[CREG][PIX];
#pragma unroll
        for (int c = 0; c < CREG; ++c) {
#pragma unroll
            for (int k = 0; k < PIX; ++k) xv[c][k] = 0.f;
            if (c < C) load_px<PIX>(a.logits + base + (long long)c * a.HW, xv[c], G.ok);
        }
        // Focal + softmax statistics on hard labels share ONE exp per element: with u = exp(x - m) (the softmax numerator)
        // and em = exp(-m), sigmoid(x) = u / (u + em) and 1 - sigmoid(x) = em / (u + em), so the class loop needs a
        // reciprocal and a log instead of two more exp and a log.  Pixels with |m| > 60 (em would leave the fp32 range) and label
        // elements whose sigmoid underflows are masked out of the class loop and redone exactly afterwards (never on sane logits).
        constexpr bool share = SHARE && focal && stats && !DENSE;   // the dispatcher sets SHARE only with prob == PROB_SOFTMAX
        float mx[PIX], inv[PIX], em[PIX];
        bool redo_all[PIX], redo[PIX];
#pragma unroll
        for (int k = 0; k < PIX; ++k) { mx[k] = 0.f; inv[k] = 0.f; em[k] = 1.f; redo_all[k] = false; redo[k] = false; }
        if (stats && a.prob == PROB_SOFTMAX) {
#pragma unroll
            for (int k = 0; k < PIX; ++k) {
                float m = -INFINITY;
#pragma unroll
                for (int c = 0; c < CREG; ++c) if (c < C) m = fmaxf(m, xv[c][k]);
                float d = 0.f;
#pragma unroll
                for (int c = 0; c < CREG; ++c) if (c < C) {
                    const float e = fexp(xv[c][k] - m);
                    d += e;
                    if (!focal || share) xv[c][k] = e;  // keep exp(x - m): one exp per element
                }
                mx[k] = m;
                inv[k] = rcp(d);
                em[k] = fexp(-m);
                redo_all[k] = share && !(fabsf(m) <= 60.f);
                redo[k] = redo_all[k];
                if (focal && !share) { opaque(mx[k]); } // dense targets: recompute exp in the class loop instead of caching 64 values
            }
        }
#pragma unroll
        for (int c = 0; c < CREG; ++c) {
            if (c < C) {
                const float cw = (!PLAIN && focal && a.class_weights) ? a.class_weights[c] : 1.0f;
                float tv[PIX], lv[PIX];
#pragma unroll
                for (int k = 0; k < PIX; ++k) { tv[k] = 0.f; lv[k] = 0.f; }
                if (DENSE) load_px<PIX>(a.dense + base + (long long)c * a.HW, tv, G.ok);
#pragma unroll
                for (int k = 0; k < PIX; ++k) {
                    if (!G.ok) continue;
                    const float x = xv[c][k];
                    float t;
                    bool ig = PLAIN ? false : G.ign[k];
                    if (!DENSE) t = lab[k] == c ? 1.f : 0.f;
                    else { t = tv[k]; if (ignf && t == a.ignore_value) ig = true; }
                    if (focal && share) {
                        const float r = rcp(x + em[k]);                 // x holds u = exp(logit - m) here
                        const float ps = x * r, qs = em[k] * r;
                        const bool hard = t != 0.f;
                        float f;
                        const float l = focal_hard<G2, PLAIN>(ps, qs, hard, ig, cw, cfg, f);
                        const bool skip = redo_all[k] || (hard && !ig && ps < 1e-36f);
                        redo[k] = redo[k] || skip;
                        lv[k] = skip ? 0.f : l;
                        fsum += skip ? 0.f : f;
                        lsum += lv[k];
                    } else if (focal) { lv[k] = focal_one<G2>(x, t, ig, cw, cfg, fsum); lsum += lv[k]; }
                    if (stats) {
                        float p;
                        if (a.prob == PROB_SOFTMAX) p = (focal && !share) ? fexp(x - mx[k]) * inv[k] : x * inv[k];
                        else if (a.prob == PROB_SIGMOID) p = sigmoid_parts(x).p;
                        else p = x;
                        if (ig) { p = 0.f; t = 0.f; }   // p*mask, t*mask (dice.py:85-111)
                        aI[c] += p * t; aP[c] += p; aT[c] += t;
                    }
                }
                if (focal && (a.flags & SEG_ELEMWISE)) store_px<PIX>(a.elem_out + base + (long long)c * a.HW, lv, G.ok);
            }
        }
        if (share && G.ok) {   // exact redo of what the class loop masked out: the logits are read again (rare)
#pragma unroll
            for (int k = 0; k < PIX; ++k) {
                if (!redo[k]) continue;
                for (int c = 0; c < C; ++c) {
                    if (!redo_all[k] && c != lab[k]) continue;
                    const float cw = a.class_weights ? a.class_weights[c] : 1.0f;
                    lsum += focal_one<G2>(a.logits[base + (long long)c * a.HW + k], lab[k] == c ? 1.f : 0.f, G.ign[k], cw, cfg, fsum);
                }
            }
        }
        if (focal) { f_loss += (double)lsum; f_term += (double)fsum; }
    }
    double* slot = a.sums + (size_t)(blockIdx.x % SUM_SLOTS) * (2 + 3 * C);
    if (focal) block_add2(f_loss, f_term, slot, lane, wave);
    if (stats) {
#pragma unroll
        for (int c = 0; c < CREG; ++c) {
            if (c < C) {
                const float sI = wave_sum(aI[c]), sP = wave_sum(aP[c]), sT = wave_sum(aT[c]);
                if (lane == 0) { wl[c] = sI; wl[C + c] = sP; wl[2 * C + c] = sT; }
            }
        }
        __syncthreads();
        for (int k = threadIdx.x; k < 3 * C; k += 256) {
            const double v = (double)lds[k] + (double)lds[3 * C + k] + (double)lds[6 * C + k] + (double)lds[9 * C + k];
            if (v != 0.0) slot_add(&slot[2 + k], v);
        }
    }
    if (a.tail_counter) region_tail(a, lds);
}

// Straight-line variant for the common case -- hard labels, C <= CREG, HW % 256 == 0 (every lane of every wave holds 4 valid
// pixels), softmax or given probabilities, and for FOCAL the default focal configuration (gamma 2, nothing else).  rocprof on
// the generic kernel above: 1 630 vector instructions per wave and pixel group (25 per element) and 71 % VALU busy -- the
// run-time switches (prob, c < C, ok, ignore) cut the unrolled class loop into hundreds of basic blocks.  Here everything
// that varies is a template parameter, classes beyond C are padded with -inf / 0 (they contribute exact zeros, so only
// their loads are guarded), and no control flow diverges: about 8 (statistics) / 20 (+ focal) vector instructions per
// element (2 pixels per lane, 98 VGPRs, measured the same as 4: 105 vs 106 us).  T_c is a label count: taken per WAVE from the compare mask the class loop needs anyway (s_bcnt1 on the scalar
// unit).  FOCAL shares the exp with the softmax as described at `share` above, with the same exact redo of extreme elements.
template <int CREG, int PROB, bool FOCAL, bool IGN, int PIX = 4, bool TERM = true, bool FULL = false, bool PF = false>   // FULL: C == CREG
__global__ __launch_bounds__(256, FOCAL ? 3 : 1) void seg_fwd_lean_kernel(const SegArgs a) {
    static_assert(!FOCAL || PROB == PROB_SOFTMAX, "the shared exp needs the softmax numerators");
    extern __shared__ float lds[];  // [4 waves][3][C]
    const int lane = threadIdx.x & 63, wave = wave_id();
    const int C = a.C;
    float* wl = lds + wave * 3 * C;
    float aI[CREG], aP[CREG];
    int nT[CREG];
#pragma unroll
    for (int c = 0; c < CREG; ++c) { aI[c] = 0.f; aP[c] = 0.f; nT[c] = 0; }
    double f_loss = 0.0, f_term = 0.0;
    const long long per_img = a.HW / (64 * PIX);
    const long long groups = per_img * a.B;
    const long long stride = (long long)gridDim.x * 4;
    // one pixel group = 64 * PIX pixels of one image: its CREG class planes and labels are fetched as one batch of loads
    auto fetch = [&](long long g, float (&xv)[CREG][PIX], long long (&l64)[PIX], long long& base) {
        const int b = (int)(g / per_img);
        const long long i0 = (g - (long long)b * per_img) * (64 * PIX) + (long long)lane * PIX;
        base = (long long)b * C * a.HW + i0;
        const long long* lp = a.labels + (long long)b * a.HW + i0;
#pragma unroll
        for (int k = 0; k < PIX; k += 2) {
            const longlong2 l2 = *reinterpret_cast<const longlong2*>(lp + k);
            l64[k] = l2.x; l64[k + 1] = l2.y;
        }
#pragma unroll
        for (int c = 0; c < CREG; ++c) {
            const float pad = PROB == PROB_SOFTMAX ? -INFINITY : 0.f;
#pragma unroll
            for (int k = 0; k < PIX; ++k) xv[c][k] = pad;
            if (FULL || c < C) load_px<PIX>(a.logits + base + (long long)c * a.HW, xv[c], true);
        }
    };
    auto process = [&](float (&xv)[CREG][PIX], const long long (&l64)[PIX], const long long base) {
        int lab[PIX];
        bool valid[PIX];
        {
            bool bad = false;
#pragma unroll
            for (int k = 0; k < PIX; ++k) {
                valid[k] = !IGN || l64[k] != a.ignore_label;
                bad = bad || (valid[k] && (l64[k] < 0 || l64[k] >= C));
                lab[k] = valid[k] ? (int)l64[k] : -1;
            }
            if (bad) raise_label_error(a.error_flag);
        }
        float inv[PIX], em[PIX];
        // FOCAL: the class loop below is exact only while every sigmoid and its complement stay normal fp32 numbers when formed
        // as u / (u + em), em / (u + em): all logits of the lane's pixels in [-80, 60] (then u = exp(x - m) >= e^-140 / e^-80 ... is
        // covered by x - m >= -80 as well).  A wave holding anything else redoes its focal sums from the re-read logits with the
        // generic formula (never on sane logits); the region statistics need no such care.
        bool tame = true;
#pragma unroll
        for (int k = 0; k < PIX; ++k) {
            inv[k] = 1.0f; em[k] = 1.0f;
            if (PROB == PROB_SOFTMAX) {
                float m = xv[0][k], lo = xv[0][k];
#pragma unroll
                for (int c = 1; c < CREG; ++c) {
                    m = fmaxf(m, xv[c][k]);
                    if (FOCAL) lo = fminf(lo, (FULL || c < C) ? xv[c][k] : lo);   // (the padding is -inf)
                }
                const float M = m * kLog2e;
                float d = 0.f;
#pragma unroll
                for (int c = 0; c < CREG; ++c) {
                    const float u = fexp_sub(xv[c][k], M);
                    xv[c][k] = u;
                    d += u;
                }
                inv[k] = rcp(d);
                if (FOCAL) {
                    em[k] = ex2(-M);
                    tame = tame && (m <= 60.f) && (lo >= -80.f) && (lo - m >= -80.f);
                }
            }
        }
        float lsum = 0.f, fsum = 0.f;    // lsum: sum of f * log2(p_t), scaled by -ln 2 once per group
#pragma unroll
        for (int c = 0; c < CREG; ++c) {
#pragma unroll
            for (int k = 0; k < PIX; ++k) {
                const float u = xv[c][k];
                const bool hit = lab[k] == c;
                if (FOCAL) {
                    const float r = rcp(u + em[k]);
                    const float ps = u * r, qs = em[k] * r;                  // sigmoid(x), 1 - sigmoid(x)
                    const float pt = hit ? ps : qs, omp = hit ? qs : ps;
                    const float f = omp * omp;
                    lsum = __builtin_fmaf(f, lg2(pt), lsum);
                    if (TERM) fsum += f;
                }
                const float pm = (IGN && !valid[k]) ? 0.f : u;
                aP[c] = __builtin_fmaf(pm, inv[k], aP[c]);
                aI[c] = __builtin_fmaf(hit ? u : 0.f, inv[k], aI[c]);
                nT[c] += __popcll(__ballot(hit));
            }
        }
        if (FOCAL) {
            lsum *= -kLn2;
            if (__any(!tame)) {   // the whole wave: exact sums from the re-read logits
                const FocalCfg cfg = focal_cfg(a);
                lsum = 0.f; fsum = 0.f;
#pragma unroll
                for (int k = 0; k < PIX; ++k)
                    for (int c = 0; c < C; ++c)
                        lsum += focal_one<true>(a.logits[base + (long long)c * a.HW + k], lab[k] == c ? 1.f : 0.f, false, 1.0f, cfg, fsum);
            }
            f_loss += (double)lsum;
            f_term += (double)fsum;
        }
    };
    const long long g0 = (long long)blockIdx.x * 4 + wave;
    if constexpr (PF) {
        // two register buffers: the loads of the NEXT group are in flight while this one is computed (the compute phase of a
        // group is ~1 us of VALU time during which the wave would otherwise have nothing outstanding)
        float xa[CREG][PIX], xb[CREG][PIX];
        long long la[PIX], lb[PIX], ba = 0, bb = 0;
        if (g0 < groups) fetch(g0, xa, la, ba);
        for (long long g = g0; g < groups; g += 2 * stride) {
            const bool hb = g + stride < groups;
            if (hb) fetch(g + stride, xb, lb, bb);
            process(xa, la, ba);
            if (g + 2 * stride < groups) fetch(g + 2 * stride, xa, la, ba);
            if (hb) process(xb, lb, bb);
        }
    } else {
        for (long long g = g0; g < groups; g += stride) {
            float xv[CREG][PIX];
            long long l64[PIX], base;
            fetch(g, xv, l64, base);
            process(xv, l64, base);
        }
    }
    double* slot = a.sums + (size_t)(blockIdx.x % SUM_SLOTS) * (2 + 3 * C);
    if (FOCAL) block_add2(f_loss, f_term, slot, lane, wave);
#pragma unroll
    for (int c = 0; c < CREG; ++c) {
        if (c < C) {
            const float sI = wave_sum(aI[c]), sP = wave_sum(aP[c]);
            if (lane == 0) { wl[c] = sI; wl[C + c] = sP; wl[2 * C + c] = (float)nT[c]; }
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < 3 * C; k += 256) {
        const double v = (double)lds[k] + (double)lds[3 * C + k] + (double)lds[6 * C + k] + (double)lds[9 * C + k];
        if (v != 0.0) slot_add(&slot[2 + k], v);
    }
    if (a.tail_counter) region_tail(a, lds);
}


// The cfg4 instance of the fused forward (hard labels, softmax statistics + default focal, 2 pixels per lane) with the vector
// instruction count cut by a quarter (rocprofv3 on seg_fwd_lean_kernel<.., FOCAL>: VALU 71 % busy, the kernel is co-bound by issue):
//  * everything per element is written on PAIRS (the lane's two pixels) so that the plain arithmetic issues as packed fp32
//    (v_pk_add / v_pk_mul / v_pk_fma_f32: two elements per full-rate slot); only v_exp / v_rcp / v_log stay per element;
//  * no per-element select on "is this the label's class": every element is summed with the t = 0 formula
//    sigma(x)^2 log2(1 - sigma(x)), where log2(1 - sigma) = log2(em / (u + em)) = -M - log2(u + em) needs no multiply -- so a pixel's
//    classes give -M F - FL with F = sum f, FL = sum f log2(u + em) -- and the label's class is corrected once per PIXEL
//    (its u picked from the registers by a 4-level select tree on the label's bits: 15 selects per pixel instead of 3 per element);
//  * I_c and T_c get one LDS add per pixel each into the lane's own column of a per-wave [class][lane] table (ds_add_f32, no
//    bank conflicts, no cross-lane traffic) instead of a select + fma and a compare + ballot per element.
// Results agree with seg_fwd_lean_kernel to rounding (same exp / rcp / log instructions, another association of the sums).
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f rcp2(v2f x) { return v2f{rcp(x.x), rcp(x.y)}; }
__device__ __forceinline__ v2f lg22(v2f x) { return v2f{lg2(x.x), lg2(x.y)}; }
__device__ __forceinline__ v2f ex22(v2f x) { return v2f{ex2(x.x), ex2(x.y)}; }
__device__ __forceinline__ v2f fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }

// FOCAL = false: the same streaming skeleton for the region statistics alone (DiceLoss / JaccardLoss on logits + hard labels): one
// exp per element, one rcp per pixel, P in packed registers, I and T through the per-lane LDS columns.
// STATS = false: the focal sums alone (BinaryFocalLoss in its default configuration on label maps): no softmax denominator, no P / I / T.
template <int CREG, bool TERM, bool FULL, bool PF = false, bool FOCAL = true, bool STATS = true>   // PF: the next pixel group's loads are in flight while this one is computed
__global__ __launch_bounds__(256, PF ? 2 : 4) void seg_focal_pk_kernel(const SegArgs a) {
    static_assert(FOCAL || STATS, "nothing to compute");
    extern __shared__ float lds[];  // [4 waves][2][CREG][64] per-lane columns of I and T; afterwards [4][3][C] wave sums + the tail's scratch
    const int lane = threadIdx.x & 63, wave = wave_id();
    const int C = a.C;
    float* colI = lds + wave * (2 * CREG * 64) + lane;
    float* colT = colI + CREG * 64;
#pragma unroll
    for (int c = 0; c < CREG; ++c) { colI[c * 64] = 0.f; colT[c * 64] = 0.f; }
    v2f aP[CREG];
#pragma unroll
    for (int c = 0; c < CREG; ++c) aP[c] = v2f{0.f, 0.f};
    double f_loss = 0.0, f_term = 0.0;
    const long long per_img = a.HW / 128;
    const long long groups = per_img * a.B;
    const long long stride = (long long)gridDim.x * 4;
    auto fetch = [&](long long g, v2f (&xv)[CREG], longlong2& l2, long long& base) {
        const int b = (int)(g / per_img);
        const long long i0 = (g - (long long)b * per_img) * 128 + (long long)lane * 2;
        base = (long long)b * C * a.HW + i0;
        l2 = *reinterpret_cast<const longlong2*>(a.labels + (long long)b * a.HW + i0);
#pragma unroll
        for (int c = 0; c < CREG; ++c) {
            xv[c] = v2f{-INFINITY, -INFINITY};
            if (FULL || c < C) xv[c] = __builtin_nontemporal_load(reinterpret_cast<const v2f*>(a.logits + base + (long long)c * a.HW));
        }
    };
    auto process = [&](v2f (&xv)[CREG], const longlong2 l2, const long long base) {
        if (l2.x < 0 || l2.x >= C || l2.y < 0 || l2.y >= C) raise_label_error(a.error_flag);
        const int lab[2] = {(int)l2.x & (CREG - 1), (int)l2.y & (CREG - 1)};     // (masked: a bad label must not leave the LDS table)
        v2f m = xv[0], lo = xv[0];
#pragma unroll
        for (int c = 1; c < CREG; ++c) {
            m = v2f{fmaxf(m.x, xv[c].x), fmaxf(m.y, xv[c].y)};
            if (FULL || c < C) lo = v2f{fminf(lo.x, xv[c].x), fminf(lo.y, xv[c].y)};
        }
        const v2f M = m * kLog2e;
        v2f d = v2f{0.f, 0.f};
#pragma unroll
        for (int c = 0; c < CREG; ++c) {
            xv[c] = ex22(fma2(xv[c], v2f{kLog2e, kLog2e}, -M));         // u = exp(x - m)
            if constexpr (STATS) d += xv[c];
        }
        const v2f inv = STATS ? rcp2(d) : v2f{0.f, 0.f};
        if constexpr (!FOCAL) {
#pragma unroll
            for (int c = 0; c < CREG; ++c) aP[c] = fma2(xv[c], inv, aP[c]);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                float sel[CREG];
#pragma unroll
                for (int c = 0; c < CREG; ++c) sel[c] = k ? xv[c].y : xv[c].x;
#pragma unroll
                for (int w = CREG / 2, bit = 1; w >= 1; w >>= 1, bit <<= 1) {
                    const bool up = lab[k] & bit;
#pragma unroll
                    for (int j = 0; j < w; ++j) sel[j] = up ? sel[2 * j + 1] : sel[2 * j];
                }
                atomicAdd(colI + lab[k] * 64, sel[0] * (k ? inv.y : inv.x));      // ds_add_f32 into this lane's own column
                atomicAdd(colT + lab[k] * 64, 1.0f);
            }
            return;
        }
        const v2f em = ex22(-M);
        const bool tame = m.x <= 60.f && m.y <= 60.f && lo.x >= -80.f && lo.y >= -80.f && lo.x - m.x >= -80.f && lo.y - m.y >= -80.f;
        v2f FL = v2f{0.f, 0.f}, F = v2f{0.f, 0.f};
#pragma unroll
        for (int c = 0; c < CREG; ++c) {
            const v2f u = xv[c];
            const v2f sm = u + em;
            const v2f r = rcp2(sm), L = lg22(sm);
            const v2f ps = u * r;                                       // sigmoid(x)
            const v2f f = ps * ps;
            FL = fma2(f, L, FL);
            F += f;
            if constexpr (STATS) aP[c] = fma2(u, inv, aP[c]);
        }
        // the label's class of each pixel: replace its t = 0 term by the t = 1 one, and feed I_c / T_c
        float lsum = 0.f, fsum = 0.f;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            float sel[CREG];
#pragma unroll
            for (int c = 0; c < CREG; ++c) sel[c] = k ? xv[c].y : xv[c].x;
#pragma unroll
            for (int w = CREG / 2, bit = 1; w >= 1; w >>= 1, bit <<= 1) {
                const bool up = lab[k] & bit;
#pragma unroll
                for (int j = 0; j < w; ++j) sel[j] = up ? sel[2 * j + 1] : sel[2 * j];
            }
            const float uh = sel[0], emk = k ? em.y : em.x, Mk = k ? M.y : M.x, invk = k ? inv.y : inv.x;
            const float sh = uh + emk, rh = rcp(sh), Lh = lg2(sh), lu = lg2(uh);
            const float psh = uh * rh, qsh = emk * rh;
            const float f0 = psh * psh, f1 = qsh * qsh;
            const float Fk = k ? F.y : F.x, FLk = k ? FL.y : FL.x;
            // sum_c f_c log2(q_c) = -M F - FL;  label's class: f0 (-M - Lh) out, f1 (lu - Lh) in
            lsum += __builtin_fmaf(f1, lu - Lh, __builtin_fmaf(f0, Mk + Lh, -__builtin_fmaf(Mk, Fk, FLk)));
            if (TERM) fsum += Fk + (f1 - f0);
            if constexpr (STATS) {
                atomicAdd(colI + lab[k] * 64, uh * invk);      // ds_add_f32 into this lane's own column
                atomicAdd(colT + lab[k] * 64, 1.0f);
            }
        }
        lsum *= -kLn2;
        if (__any(!tame)) {   // the whole wave: exact focal sums from the re-read logits (never on sane logits)
            const FocalCfg cfg = focal_cfg(a);
            lsum = 0.f; fsum = 0.f;
            const long long labs[2] = {l2.x, l2.y};
#pragma unroll
            for (int k = 0; k < 2; ++k)
                for (int c = 0; c < C; ++c)
                    lsum += focal_one<true>(a.logits[base + (long long)c * a.HW + k], labs[k] == c ? 1.f : 0.f, false, 1.0f, cfg, fsum);
        }
        f_loss += (double)lsum;
        f_term += (double)fsum;
    };
    const long long g0 = (long long)blockIdx.x * 4 + wave;
    if constexpr (PF) {
        v2f xa[CREG], xb[CREG];
        longlong2 la{}, lb{};
        long long ba = 0, bb = 0;
        if (g0 < groups) fetch(g0, xa, la, ba);
        for (long long g = g0; g < groups; g += 2 * stride) {
            const bool hb = g + stride < groups;
            if (hb) fetch(g + stride, xb, lb, bb);
            process(xa, la, ba);
            if (g + 2 * stride < groups) fetch(g + 2 * stride, xa, la, ba);
            if (hb) process(xb, lb, bb);
        }
    } else {
        for (long long g = g0; g < groups; g += stride) {
            v2f xv[CREG];
            longlong2 l2;
            long long base;
            fetch(g, xv, l2, base);
            process(xv, l2, base);
        }
    }
    double* slot = a.sums + (size_t)(blockIdx.x % SUM_SLOTS) * (2 + 3 * C);
    if constexpr (FOCAL) block_add2(f_loss, f_term, slot, lane, wave);
    if constexpr (!STATS) {
        if (a.tail_counter) region_tail(a, lds);
        return;
    }
    // Row sums of the wave's [2 CREG][64] table without 6-step shuffles per value: lane L adds up half (L / 32... for 2 CREG = 32
    // rows; in general 64 / ROWS parts) of row L % ROWS, walking the columns rotated by its row number so that the lanes of one
    // read hit distinct banks; the parts are then combined across lanes.  P (registers) goes through the same table afterwards.
    constexpr int ROWS = 2 * CREG, PARTS = 64 / ROWS, SPAN = 64 / PARTS;      // CREG 16: 32 rows x 2 halves of 32 columns
    const float* tab = lds + wave * (ROWS * 64);
    auto row_part_sum = [&](int rows) {                                        // sum of my part of row (lane % rows)
        const int r = lane % rows, part = lane / rows, span = 64 / (64 / rows);
        float acc = 0.f;
#pragma unroll 8
        for (int j = 0; j < span; ++j) acc += tab[r * 64 + part * span + ((j + r) & (span - 1))];
        for (int o = rows; o < 64; o <<= 1) acc += __shfl_xor(acc, o);        // combine the parts (lanes r, r + rows, ...)
        return acc;
    };
    static_assert(PARTS >= 1 && SPAN * PARTS == 64, "table rows");
    const float itsum = row_part_sum(ROWS);            // lanes 0 .. CREG-1: I_c, lanes CREG .. 2 CREG-1: T_c (of this wave)
#pragma unroll
    for (int c = 0; c < CREG; ++c) colI[c * 64] = aP[c].x + aP[c].y;          // (own column again: rows 0 .. CREG-1 now hold P)
    const float psum = row_part_sum(CREG);             // lanes 0 .. CREG-1: P_c
    __syncthreads();                 // every wave has read its table: it becomes the [4][3][C] wave sums
    float* wl = lds + wave * 3 * C;
    if (lane < CREG && lane < C) { wl[lane] = itsum; wl[C + lane] = psum; }
    if (lane >= CREG && lane < 2 * CREG && lane - CREG < C) wl[2 * C + lane - CREG] = itsum;
    __syncthreads();
    for (int k = threadIdx.x; k < 3 * C; k += 256) {
        const double v = (double)lds[k] + (double)lds[3 * C + k] + (double)lds[6 * C + k] + (double)lds[9 * C + k];
        if (v != 0.0) slot_add(&slot[2 + k], v);
    }
    if (a.tail_counter) region_tail(a, lds);
}

// Region statistics with DENSE targets and a per-element activation (multilabel / binary Dice and Jaccard: sigmoid or given
// probabilities): no cross-class dependency, so a wave simply streams 1024 consecutive elements of ONE (image, class) plane of
// logits and targets (2 x 4 x 16 B in flight per lane), keeps the three sums of that class in registers and leaves them with
// three slotted atomics.  The generic register-resident kernel above needs 198 VGPRs for this case and ran at 3.6 TB/s.
template <int PROB, bool IGN, bool FOCAL = false>
__global__ __launch_bounds__(256) void seg_stats_dense_lean_kernel(const SegArgs a) {
    static_assert(!FOCAL || PROB == PROB_SIGMOID, "the focal term shares the sigmoid of the statistics");
    extern __shared__ float lds[];  // only the in-launch tail uses it here
    const int lane = threadIdx.x & 63, wave = wave_id();
    const int C = a.C;
    const float term_mask = (a.flags & SEG_MASK_FOCAL_TERM) ? 0.0f : 1.0f;
    double f_loss = 0.0, f_term = 0.0;
    const long long segs_per_plane = a.HW / 1024;
    const long long segs = segs_per_plane * C * a.B;
    for (long long sg = (long long)blockIdx.x * 4 + wave; sg < segs; sg += (long long)gridDim.x * 4) {
        const long long plane = sg / segs_per_plane;           // b * C + c
        const int c = (int)(plane % C);
        const long long off = plane * a.HW + (sg - plane * segs_per_plane) * 1024 + (long long)lane * 4;
        float xv[4][4], tv[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            load_px<4>(a.logits + off + u * 256, xv[u], true);
            load_px<4>(a.dense + off + u * 256, tv[u], true);
        }
        float sI = 0.f, sP = 0.f, sT = 0.f, lsum = 0.f, fsum = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float x = xv[u][k];
                float t = tv[u][k];
                const bool ig = IGN && t == a.ignore_value;
                float p;
                if (PROB == PROB_SIGMOID) {
                    const Sig sgm = sigmoid_parts(x);
                    p = sgm.p;
                    if (FOCAL) {   // default focal configuration (gamma 2, nothing else), targets may be soft: functional.py:61-94
                        const float tt = ig ? 0.f : t;
                        const float ce = fmaxf(x, 0.f) - x * tt + sgm.log1pe;
                        const float pt = __builtin_fmaf(p, tt, (1.f - p) * (1.f - tt));
                        const float omp = fmaxf(1.f - pt, 0.f);
                        float f = omp * omp;
                        lsum += ig ? 0.f : f * ce;
                        fsum += ig ? f * term_mask : f;
                    }
                } else {
                    p = x;
                }
                if (ig) { p = 0.f; t = 0.f; }     // p * mask, t * mask (dice.py:85-111)
                sI = __builtin_fmaf(p, t, sI);
                sP += p;
                sT += t;
            }
        }
        sI = wave_sum(sI); sP = wave_sum(sP); sT = wave_sum(sT);
        if (lane == 0) {
            double* slot = a.sums + (size_t)((blockIdx.x * 4 + wave) % SUM_SLOTS) * (2 + 3 * C);
            slot_add(&slot[2 + c], (double)sI);
            slot_add(&slot[2 + C + c], (double)sP);
            slot_add(&slot[2 + 2 * C + c], (double)sT);
        }
        if (FOCAL) { f_loss += (double)lsum; f_term += (double)fsum; }
    }
    if (FOCAL) block_add2(f_loss, f_term, a.sums + (size_t)(blockIdx.x % SUM_SLOTS) * (2 + 3 * C), lane, wave);
    if (a.tail_counter) region_tail(a, lds);
}

// Generic variant: any C, class planes are streamed (softmax: one extra pass for the log-sum-exp), everything run time.
template <int PIX>
__global__ __launch_bounds__(256) void seg_loss_fwd_kernel(const SegArgs a) {
    const FocalCfg cfg = focal_cfg(a);
    extern __shared__ float lds[];  // [4 waves][3][C]
    const int lane = threadIdx.x & 63, wave = wave_id();
    const int C = a.C;
    float* wl = lds + wave * 3 * C;
    const bool stats = a.flags & SEG_STATS, focal = a.flags & SEG_FOCAL;
    const bool ignf = a.flags & SEG_HAS_IGNORE;
    if (stats) for (int k = lane; k < 3 * C; k += 64) wl[k] = 0.f;
    double f_loss = 0.0, f_term = 0.0;
    const long long per_img = (a.HW + 64 * PIX - 1) / (64 * PIX);
    const long long groups = per_img * a.B;
    for (long long g = (long long)blockIdx.x * 4 + wave; g < groups; g += (long long)gridDim.x * 4) {
        const Group<PIX> G = make_group<PIX>(g, per_img, lane, a.HW, a.labels, ignf, a.ignore_label, C, a.error_flag);
        const long long base = (long long)G.b * C * a.HW + G.i0;
        float lsum = 0.f, fsum = 0.f;
        float mx[PIX], inv[PIX];
#pragma unroll
        for (int k = 0; k < PIX; ++k) { mx[k] = -INFINITY; inv[k] = 0.f; }
        if (stats && a.prob == PROB_SOFTMAX) {
            for (int c = 0; c < C; ++c) {
                float xv[PIX];
                load_px<PIX>(a.logits + base + (long long)c * a.HW, xv, G.ok);
#pragma unroll
                for (int k = 0; k < PIX; ++k) if (G.ok) {
                    const float m2 = fmaxf(mx[k], xv[k]);
                    inv[k] = inv[k] * fexp(mx[k] - m2) + fexp(xv[k] - m2);
                    mx[k] = m2;
                }
            }
#pragma unroll
            for (int k = 0; k < PIX; ++k) inv[k] = rcp(inv[k]);
        }
        for (int c = 0; c < C; ++c) {
            const long long off = base + (long long)c * a.HW;
            const float cw = a.class_weights ? a.class_weights[c] : 1.0f;
            float xv[PIX], tv[PIX], lv[PIX];
            float sI = 0.f, sP = 0.f, sT = 0.f;
            load_px<PIX>(a.logits + off, xv, G.ok);
            if (!a.labels) load_px<PIX>(a.dense + off, tv, G.ok);
#pragma unroll
            for (int k = 0; k < PIX; ++k) {
                lv[k] = 0.f;
                if (!G.ok) continue;
                const float x = xv[k];
                float t;
                bool ig = G.ign[k];
                if (a.labels) t = G.lab[k] == c ? 1.f : 0.f;
                else { t = tv[k]; if (ignf && t == a.ignore_value) ig = true; }
                if (focal) { lv[k] = focal_one<false>(x, t, ig, cw, cfg, fsum); lsum += lv[k]; }
                if (stats) {
                    float p;
                    if (a.prob == PROB_SOFTMAX) p = fexp(x - mx[k]) * inv[k];
                    else if (a.prob == PROB_SIGMOID) p = sigmoid_parts(x).p;
                    else p = x;
                    if (ig) { p = 0.f; t = 0.f; }
                    sI += p * t; sP += p; sT += t;
                }
            }
            if (a.flags & SEG_ELEMWISE) store_px<PIX>(a.elem_out + off, lv, G.ok);
            if (stats) {
                sI = wave_sum(sI); sP = wave_sum(sP); sT = wave_sum(sT);
                if (lane == 0) { wl[c] += sI; wl[C + c] += sP; wl[2 * C + c] += sT; }
            }
        }
        f_loss += (double)lsum;
        f_term += (double)fsum;
    }
    double* slot = a.sums + (size_t)(blockIdx.x % SUM_SLOTS) * (2 + 3 * C);
    if (focal) block_add2(f_loss, f_term, slot, lane, wave);
    if (stats) {
        __syncthreads();
        for (int k = threadIdx.x; k < 3 * C; k += 256) {
            const double v = (double)lds[k] + (double)lds[3 * C + k] + (double)lds[6 * C + k] + (double)lds[9 * C + k];
            if (v != 0.0) slot_add(&slot[2 + k], v);
        }
    }
    if (a.tail_counter) region_tail(a, lds);
}

// ------------------------------------------------------------------------------------------------ focal-only forward
// No cross-class dependency: class planes are streamed 4 at a time (4 x 16 B in flight per lane, ~100 VGPRs).
template <int PIX, bool DENSE, bool G2>
__global__ __launch_bounds__(256) void focal_fwd_kernel(const SegArgs a) {
    const FocalCfg cfg = focal_cfg(a);
    const int lane = threadIdx.x & 63, wave = wave_id();
    const int C = a.C;
    const bool ignf = a.flags & SEG_HAS_IGNORE;
    const bool elem = a.flags & SEG_ELEMWISE;
    double f_loss = 0.0, f_term = 0.0;
    const long long per_img = (a.HW + 64 * PIX - 1) / (64 * PIX);
    const long long groups = per_img * a.B;
    for (long long g = (long long)blockIdx.x * 4 + wave; g < groups; g += (long long)gridDim.x * 4) {
        const Group<PIX> G = make_group<PIX>(g, per_img, lane, a.HW, DENSE ? nullptr : a.labels, ignf, a.ignore_label, C, a.error_flag);
        const long long base = (long long)G.b * C * a.HW + G.i0;
        float lsum = 0.f, fsum = 0.f;
        for (int c0 = 0; c0 < C; c0 += 4) {
            float xv[4][PIX], tv[DENSE ? 4 : 1][PIX];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int k = 0; k < PIX; ++k) xv[u][k] = 0.f;
                const bool on = G.ok && c0 + u < C;
                const long long off = base + (long long)(c0 + u) * a.HW;
                load_px<PIX>(a.logits + off, xv[u], on);
                if constexpr (DENSE) {
#pragma unroll
                    for (int k = 0; k < PIX; ++k) tv[u][k] = 0.f;
                    load_px<PIX>(a.dense + off, tv[u], on);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = c0 + u;
                if (c < C) {
                    const float cw = a.class_weights ? a.class_weights[c] : 1.f;
                    float lv[PIX];
#pragma unroll
                    for (int k = 0; k < PIX; ++k) {
                        lv[k] = 0.f;
                        if (!G.ok) continue;
                        float t;
                        bool ig = G.ign[k];
                        if constexpr (!DENSE) t = G.lab[k] == c ? 1.f : 0.f;
                        else { t = tv[u][k]; if (ignf && t == a.ignore_value) ig = true; }
                        lv[k] = focal_one<G2>(xv[u][k], t, ig, cw, cfg, fsum);
                        lsum += lv[k];
                    }
                    if (elem) store_px<PIX>(a.elem_out + base + (long long)c * a.HW, lv, G.ok);
                }
            }
        }
        f_loss += (double)lsum;
        f_term += (double)fsum;
    }
    block_add2(f_loss, f_term, a.sums + (size_t)(blockIdx.x % SUM_SLOTS) * (2 + 3 * C), lane, wave);
}

// Straight-line variant for hard labels with the default focal configuration (gamma 2; no alpha, threshold or class weights)
// and HW % 256 == 0.  With y = (label == c) ? -x : x the element is sigmoid(y)^2 * softplus(y): 1 - p_t = sigmoid(y) and
// BCE = softplus(y), so the target never enters the arithmetic again -- ~13 vector instructions + exp, rcp, log per element.
// CCH class planes are requested at once (the generic kernel above has 4 in flight); classes beyond C are padded with
// -inf, whose element is exactly 0.  IGN: ignored pixels add no loss, and their focal terms only count when not normalised.
template <int CCH, bool IGN>
__global__ __launch_bounds__(256) void focal_fwd_lean_kernel(const SegArgs a) {
    const int lane = threadIdx.x & 63, wave = wave_id();
    const int C = a.C;
    const float term_mask = (a.flags & SEG_MASK_FOCAL_TERM) ? 0.0f : 1.0f;
    double f_loss = 0.0, f_term = 0.0;
    const long long per_img = a.HW / 256;
    const long long groups = per_img * a.B;
    for (long long g = (long long)blockIdx.x * 4 + wave; g < groups; g += (long long)gridDim.x * 4) {
        const int b = (int)(g / per_img);
        const long long i0 = (g - (long long)b * per_img) * 256 + (long long)lane * 4;
        const long long base = (long long)b * C * a.HW + i0;
        int lab[4];
        bool valid[4];
        {
            const long long* lp = a.labels + (long long)b * a.HW + i0;
            const longlong2 l01 = *reinterpret_cast<const longlong2*>(lp), l23 = *reinterpret_cast<const longlong2*>(lp + 2);
            const long long l64[4] = {l01.x, l01.y, l23.x, l23.y};
            bool bad = false;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                valid[k] = !IGN || l64[k] != a.ignore_label;
                bad = bad || (valid[k] && (l64[k] < 0 || l64[k] >= C));
                lab[k] = valid[k] ? (int)l64[k] : -1;
            }
            if (bad) raise_label_error(a.error_flag);
        }
        float lrelu = 0.f, llog = 0.f, fsum = 0.f;   // loss = sum f * relu(y) + ln2 * sum f * log2(1 + e)
        for (int c0 = 0; c0 < C; c0 += CCH) {
            float xv[CCH][4];
#pragma unroll
            for (int u = 0; u < CCH; ++u) {
#pragma unroll
                for (int k = 0; k < 4; ++k) xv[u][k] = -INFINITY;
                if (c0 + u < C) load_px<4>(a.logits + base + (long long)(c0 + u) * a.HW, xv[u], true);
            }
#pragma unroll
            for (int u = 0; u < CCH; ++u) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float x = xv[u][k];
                    const float y = lab[k] == c0 + u ? -x : x;
                    const float e = ex2(-fabsf(y) * kLog2e);
                    const float s1 = 1.0f + e;
                    const float inv = rcp(s1);
                    const float sg = y >= 0.f ? inv : e * inv;          // sigmoid(y) = 1 - p_t
                    float f = sg * sg;
                    const float w = (IGN && !valid[k]) ? 0.f : f;        // weight of this element's loss
                    lrelu = __builtin_fmaf(w, fmaxf(y, 0.f), lrelu);
                    llog = __builtin_fmaf(w, lg2(s1), llog);
                    if (IGN) f = valid[k] ? f : f * term_mask;
                    fsum += f;
                }
            }
        }
        f_loss += (double)(lrelu + kLn2 * llog);
        f_term += (double)fsum;
    }
    block_add2(f_loss, f_term, a.sums + (size_t)(blockIdx.x % SUM_SLOTS) * (2 + 3 * C), lane, wave);
}

// ------------------------------------------------------------------------------------------------ focal backward
// grad[i] = coef[0] * (grad_elem ? grad_elem[i] : 1) * dL_i/dx_i + coef[1] * dF_i/dx_i  (coef on the device: no host sync).
// Same wave-group walk as the forward; class planes are processed 4 at a time so 4+ loads are in flight per lane.
template <int PIX, bool DENSE, bool GELEM, bool G2>
__global__ __launch_bounds__(256) void focal_bwd_kernel(const SegArgs a, const float* __restrict__ coef,
                                                        const float* __restrict__ grad_elem, float* __restrict__ grad) {
    const FocalCfg cfg = focal_cfg(a);
    const int lane = threadIdx.x & 63, wave = wave_id();
    const int C = a.C;
    const bool ignf = a.flags & SEG_HAS_IGNORE;
    const float k1 = coef[0], k2 = coef[1];
    const long long per_img = (a.HW + 64 * PIX - 1) / (64 * PIX);
    const long long groups = per_img * a.B;
    for (long long g = (long long)blockIdx.x * 4 + wave; g < groups; g += (long long)gridDim.x * 4) {
        const Group<PIX> G = make_group<PIX>(g, per_img, lane, a.HW, DENSE ? nullptr : a.labels, ignf, a.ignore_label, C, nullptr);
        const long long base = (long long)G.b * C * a.HW + G.i0;
        for (int c0 = 0; c0 < C; c0 += 4) {
            float xv[4][PIX], tv[DENSE ? 4 : 1][PIX], gv[GELEM ? 4 : 1][PIX];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int k = 0; k < PIX; ++k) xv[u][k] = 0.f;
                const bool on = G.ok && c0 + u < C;
                const long long off = base + (long long)(c0 + u) * a.HW;
                load_px<PIX>(a.logits + off, xv[u], on);
                if constexpr (DENSE) {
#pragma unroll
                    for (int k = 0; k < PIX; ++k) tv[u][k] = 0.f;
                    load_px<PIX>(a.dense + off, tv[u], on);
                }
                if constexpr (GELEM) {
#pragma unroll
                    for (int k = 0; k < PIX; ++k) gv[u][k] = 0.f;
                    load_px<PIX>(grad_elem + off, gv[u], on);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = c0 + u;
                if (c < C) {
                    const float w0 = a.class_weights ? a.class_weights[c] : 1.f;
                    float out[PIX];
#pragma unroll
                    for (int k = 0; k < PIX; ++k) {
                        out[k] = 0.f;
                        if (!G.ok) continue;
                        float t;
                        bool ig = G.ign[k];
                        if constexpr (!DENSE) t = G.lab[k] == c ? 1.f : 0.f;
                        else { t = tv[u][k]; if (ignf && t == a.ignore_value) ig = true; }
                        if (!ig) {
                            float ce, f, df, p;
                            focal_parts<G2, true>(xv[u][k], t, cfg, ce, f, df, p);
                            const float w = w0 * (cfg.a1 * t + cfg.a0);
                            const float dL = w * (df * ce + f * (p - t));
                            float g1 = k1;
                            if constexpr (GELEM) g1 = k1 * gv[u][k];
                            out[k] = g1 * dL + k2 * df;
                        }
                    }
                    store_px_nt<PIX>(grad + base + (long long)c * a.HW, out, G.ok);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ region-stat backward
// With G_c = (gI[c]*t_c + gP[c]) * mask:  softmax: p_k (G_k - sum_c G_c p_c);  sigmoid: G p (1-p);  identity: G.
// (the C <= 16 label instance is pinned at 5 waves per SIMD: the allocator took 101 VGPRs, 5 over the step; 95 without a spill)
template <int PIX, int CREG, bool DENSE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((PIX == 4 && CREG == 16 && !DENSE) ? 5 : 1))) void seg_stats_bwd_kernel(const SegArgs a, const float* __restrict__ gI,
                                                            const float* __restrict__ gP, float* __restrict__ grad) {
    const int lane = threadIdx.x & 63, wave = wave_id();
    const int C = a.C;
    const bool ignf = a.flags & SEG_HAS_IGNORE;
    const long long per_img = (a.HW + 64 * PIX - 1) / (64 * PIX);
    const long long groups = per_img * a.B;
    for (long long g = (long long)blockIdx.x * 4 + wave; g < groups; g += (long long)gridDim.x * 4) {
        const Group<PIX> G = make_group<PIX>(g, per_img, lane, a.HW, a.labels, ignf, a.ignore_label, C, nullptr);
        const long long base = (long long)G.b * C * a.HW + G.i0;
        if constexpr (CREG > 0) {
            float xv[CREG][PIX], tv[DENSE ? CREG : 1][PIX];
#pragma unroll
            for (int c = 0; c < CREG; ++c) {
#pragma unroll
                for (int k = 0; k < PIX; ++k) xv[c][k] = 0.f;
                if (c < C) load_px<PIX>(a.logits + base + (long long)c * a.HW, xv[c], G.ok);
            }
            if constexpr (DENSE) {
#pragma unroll
                for (int c = 0; c < CREG; ++c) {
#pragma unroll
                    for (int k = 0; k < PIX; ++k) tv[c][k] = 0.f;
                    if (c < C) load_px<PIX>(a.dense + base + (long long)c * a.HW, tv[c], G.ok);
                }
            }
            float dot[PIX];
#pragma unroll
            for (int k = 0; k < PIX; ++k) dot[k] = 0.f;
            float mx[PIX], inv[PIX];
#pragma unroll
            for (int k = 0; k < PIX; ++k) { mx[k] = 0.f; inv[k] = 0.f; }
            if (a.prob == PROB_SOFTMAX) {
#pragma unroll
                for (int k = 0; k < PIX; ++k) {
                    float m = -INFINITY;
#pragma unroll
                    for (int c = 0; c < CREG; ++c) if (c < C) m = fmaxf(m, xv[c][k]);
                    float d = 0.f;
#pragma unroll
                    for (int c = 0; c < CREG; ++c) if (c < C) d += fexp(xv[c][k] - m);
                    const float iv = rcp(d);
                    opaque(m);  // recompute exp(x - m) in each pass: cheaper than 64 live values (occupancy 2 -> 4)
                    float dd = 0.f;
#pragma unroll
                    for (int c = 0; c < CREG; ++c) if (c < C) {
                        const float t = !DENSE ? (G.lab[k] == c ? 1.f : 0.f) : tv[DENSE ? c : 0][k];
                        dd += (gI[c] * t + gP[c]) * fexp(xv[c][k] - m) * iv;
                    }
                    opaque(m);
                    mx[k] = m; inv[k] = iv; dot[k] = dd;
                }
            }
#pragma unroll
            for (int c = 0; c < CREG; ++c) {
                if (c < C) {
                    float out[PIX];
#pragma unroll
                    for (int k = 0; k < PIX; ++k) {
                        out[k] = 0.f;
                        if (!G.ok) continue;
                        const float t = !DENSE ? (G.lab[k] == c ? 1.f : 0.f) : tv[DENSE ? c : 0][k];
                        bool ig = G.ign[k];
                        if (DENSE && ignf && t == a.ignore_value) ig = true;
                        if (!ig) {
                            const float Gc = gI[c] * t + gP[c];
                            if (a.prob == PROB_SOFTMAX) out[k] = fexp(xv[c][k] - mx[k]) * inv[k] * (Gc - dot[k]);
                            else if (a.prob == PROB_SIGMOID) { const float p = sigmoid_parts(xv[c][k]).p; out[k] = Gc * p * (1.f - p); }
                            else out[k] = Gc;
                        }
                    }
#ifndef PTB_STATS_BWD_NT
#define PTB_STATS_BWD_NT 0
#endif
                    // (the pinned C <= 16 label instance keeps the plain store: the non-temporal one needs its 4 values in consecutive
                    // registers, which costs that instance a 44-byte spill at 5 waves per SIMD)
                    if constexpr (PIX == 4 && CREG == 16 && !DENSE && !PTB_STATS_BWD_NT) store_px<PIX>(grad + base + (long long)c * a.HW, out, G.ok);
                    else store_px_nt<PIX>(grad + base + (long long)c * a.HW, out, G.ok);
                }
            }
        } else {
            float mx[PIX], inv[PIX], dot[PIX];
#pragma unroll
            for (int k = 0; k < PIX; ++k) { mx[k] = -INFINITY; inv[k] = 0.f; dot[k] = 0.f; }
            if (a.prob == PROB_SOFTMAX) {
                for (int c = 0; c < C; ++c) {
                    float xv[PIX];
                    load_px<PIX>(a.logits + base + (long long)c * a.HW, xv, G.ok);
#pragma unroll
                    for (int k = 0; k < PIX; ++k) if (G.ok) {
                        const float m2 = fmaxf(mx[k], xv[k]);
                        inv[k] = inv[k] * fexp(mx[k] - m2) + fexp(xv[k] - m2);
                        mx[k] = m2;
                    }
                }
#pragma unroll
                for (int k = 0; k < PIX; ++k) inv[k] = rcp(inv[k]);
                for (int c = 0; c < C; ++c) {
                    float xv[PIX], tv[PIX];
                    load_px<PIX>(a.logits + base + (long long)c * a.HW, xv, G.ok);
                    if (!a.labels) load_px<PIX>(a.dense + base + (long long)c * a.HW, tv, G.ok);
#pragma unroll
                    for (int k = 0; k < PIX; ++k) if (G.ok && !G.ign[k]) {
                        const float p = fexp(xv[k] - mx[k]) * inv[k];
                        const float t = a.labels ? (G.lab[k] == c ? 1.f : 0.f) : tv[k];
                        dot[k] += (gI[c] * t + gP[c]) * p;
                    }
                }
            }
            for (int c = 0; c < C; ++c) {
                float xv[PIX], tv[PIX], out[PIX];
                load_px<PIX>(a.logits + base + (long long)c * a.HW, xv, G.ok);
                if (!a.labels) load_px<PIX>(a.dense + base + (long long)c * a.HW, tv, G.ok);
#pragma unroll
                for (int k = 0; k < PIX; ++k) {
                    out[k] = 0.f;
                    if (!G.ok) continue;
                    const float t = a.labels ? (G.lab[k] == c ? 1.f : 0.f) : tv[k];
                    bool ig = G.ign[k];
                    if (!a.labels && ignf && t == a.ignore_value) ig = true;
                    if (!ig) {
                        const float Gc = gI[c] * t + gP[c];
                        if (a.prob == PROB_SOFTMAX) out[k] = fexp(xv[k] - mx[k]) * inv[k] * (Gc - dot[k]);
                        else if (a.prob == PROB_SIGMOID) { const float p = sigmoid_parts(xv[k]).p; out[k] = Gc * p * (1.f - p); }
                        else out[k] = Gc;
                    }
                }
                store_px_nt<PIX>(grad + base + (long long)c * a.HW, out, G.ok);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ fused backward
// d/dlogits of (sigmoid focal sums, region statistics) in ONE pass (FocalDiceJaccardLoss): the class planes are read
// once and the gradient is written once, instead of two backward kernels plus an add (3.4x the traffic).
//   grad = coef[0] * dL/dx + coef[1] * dF/dx            (focal, see focal_bwd_kernel)
//        + p_k (G_k - sum_c G_c p_c) | G p (1 - p)       (region statistics with softmax | sigmoid probabilities)
template <int PIX, int CREG, bool DENSE, bool G2>
__global__ __launch_bounds__(256) void seg_fused_bwd_kernel(const SegArgs a, const float* __restrict__ coef, const float* __restrict__ gI,
                                                            const float* __restrict__ gP, float* __restrict__ grad) {
    const FocalCfg cfg = focal_cfg(a);
    const int lane = threadIdx.x & 63, wave = wave_id();
    const int C = a.C;
    const bool ignf = a.flags & SEG_HAS_IGNORE;
    const float k1 = coef[0], k2 = coef[1];
    const long long per_img = (a.HW + 64 * PIX - 1) / (64 * PIX);
    const long long groups = per_img * a.B;
    for (long long g = (long long)blockIdx.x * 4 + wave; g < groups; g += (long long)gridDim.x * 4) {
        const Group<PIX> G = make_group<PIX>(g, per_img, lane, a.HW, DENSE ? nullptr : a.labels, ignf, a.ignore_label, C, nullptr);
        const long long base = (long long)G.b * C * a.HW + G.i0;
        float xv[CREG][PIX];
#pragma unroll
        for (int c = 0; c < CREG; ++c) {
#pragma unroll
            for (int k = 0; k < PIX; ++k) xv[c][k] = 0.f;
            if (c < C) load_px<PIX>(a.logits + base + (long long)c * a.HW, xv[c], G.ok);
        }
        float mx[PIX], inv[PIX], dot[PIX];
#pragma unroll
        for (int k = 0; k < PIX; ++k) { mx[k] = 0.f; inv[k] = 0.f; dot[k] = 0.f; }
        if (a.prob == PROB_SOFTMAX && !DENSE) {
#pragma unroll
            for (int k = 0; k < PIX; ++k) {
                float m = -INFINITY;
#pragma unroll
                for (int c = 0; c < CREG; ++c) if (c < C) m = fmaxf(m, xv[c][k]);
                float d = 0.f;
#pragma unroll
                for (int c = 0; c < CREG; ++c) if (c < C) d += fexp(xv[c][k] - m);
                const float iv = rcp(d);
                opaque(m);  // recompute the exponentials below instead of keeping 64 of them alive
                float dd = 0.f;
#pragma unroll
                for (int c = 0; c < CREG; ++c) if (c < C) {
                    const float t = G.lab[k] == c ? 1.f : 0.f;
                    dd += (gI[c] * t + gP[c]) * fexp(xv[c][k] - m) * iv;
                }
                opaque(m);
                mx[k] = m; inv[k] = iv; dot[k] = G.ign[k] ? 0.f : dd;
            }
        }
#pragma unroll
        for (int c = 0; c < CREG; ++c) {
            if (c < C) {
                const float w0 = a.class_weights ? a.class_weights[c] : 1.f;
                float tv[PIX], out[PIX];
#pragma unroll
                for (int k = 0; k < PIX; ++k) { tv[k] = 0.f; out[k] = 0.f; }
                if (DENSE) load_px<PIX>(a.dense + base + (long long)c * a.HW, tv, G.ok);
#pragma unroll
                for (int k = 0; k < PIX; ++k) {
                    if (!G.ok) continue;
                    const float x = xv[c][k];
                    float t;
                    bool ig = G.ign[k];
                    if (!DENSE) t = G.lab[k] == c ? 1.f : 0.f;
                    else { t = tv[k]; if (ignf && t == a.ignore_value) ig = true; }
                    if (ig) continue;
                    float ce, f, df, p;
                    focal_parts<G2, true>(x, t, cfg, ce, f, df, p);   // p = sigmoid(x)
                    const float w = w0 * (cfg.a1 * t + cfg.a0);
                    float gx = k1 * w * (df * ce + f * (p - t)) + k2 * df;
                    const float Gc = gI[c] * t + gP[c];
                    if (!DENSE && a.prob == PROB_SOFTMAX) gx += fexp(x - mx[k]) * inv[k] * (Gc - dot[k]);
                    else if (a.prob == PROB_SIGMOID) gx += Gc * p * (1.f - p);
                    else gx += Gc;
                    out[k] = gx;
                }
                store_px_nt<PIX>(grad + base + (long long)c * a.HW, out, G.ok);
            }
        }
    }
}

// The same fused backward for hard labels + softmax statistics with ONE exp per element (see `share` in
// seg_loss_fwd_reg_kernel): u = exp(x - m) replaces the logits in registers, sigmoid(x) = u / (u + em), 1 - sigmoid(x) =
// em / (u + em) with em = exp(-m); BCE = -log(p_t).  Six transcendentals per element become three.  Pixels with |m| > 60
// and label elements whose sigmoid underflows are rewritten afterwards from the re-read logits with the exact formulas.
template <int PIX, int CREG, bool G2, bool PLAIN = false>
__global__ __launch_bounds__(256, 3) void seg_fused_bwd_shared_kernel(const SegArgs a, const float* __restrict__ coef,
                                                                      const float* __restrict__ gI, const float* __restrict__ gP,
                                                                      float* __restrict__ grad) {
    const FocalCfg cfg = focal_cfg(a);
    const int lane = threadIdx.x & 63, wave = wave_id();
    const int C = a.C;
    const bool ignf = a.flags & SEG_HAS_IGNORE;
    const float k1 = coef[0], k2 = coef[1];
    const long long per_img = (a.HW + 64 * PIX - 1) / (64 * PIX);
    const long long groups = per_img * a.B;
    for (long long g = (long long)blockIdx.x * 4 + wave; g < groups; g += (long long)gridDim.x * 4) {
        const Group<PIX> G = make_group<PIX>(g, per_img, lane, a.HW, a.labels, ignf, a.ignore_label, C, nullptr);
        const long long base = (long long)G.b * C * a.HW + G.i0;
        int lab[PIX];
#pragma unroll
        for (int k = 0; k < PIX; ++k) lab[k] = (int)G.lab[k];
        float xv[CREG][PIX];
#pragma unroll
        for (int c = 0; c < CREG; ++c) {
#pragma unroll
            for (int k = 0; k < PIX; ++k) xv[c][k] = 0.f;
            if (c < C) load_px<PIX>(a.logits + base + (long long)c * a.HW, xv[c], G.ok);
        }
        float mx[PIX], inv[PIX], dot[PIX], em[PIX];
        bool redo_all[PIX], redo[PIX];
#pragma unroll
        for (int k = 0; k < PIX; ++k) {
            float m = -INFINITY;
#pragma unroll
            for (int c = 0; c < CREG; ++c) if (c < C) m = fmaxf(m, xv[c][k]);
            float d = 0.f, dd = 0.f;
#pragma unroll
            for (int c = 0; c < CREG; ++c) if (c < C) {
                const float u = fexp(xv[c][k] - m);
                xv[c][k] = u;
                d += u;
                dd += (lab[k] == c ? gI[c] + gP[c] : gP[c]) * u;
            }
            const float iv = rcp(d);
            mx[k] = m; inv[k] = iv; dot[k] = G.ign[k] ? 0.f : dd * iv; em[k] = fexp(-m);
            redo_all[k] = !(fabsf(m) <= 60.f);
            redo[k] = redo_all[k];
        }
#pragma unroll
        for (int c = 0; c < CREG; ++c) {
            if (c < C) {
                const float w0 = (!PLAIN && a.class_weights) ? a.class_weights[c] : 1.f;
                float out[PIX];
#pragma unroll
                for (int k = 0; k < PIX; ++k) {
                    out[k] = 0.f;
                    if (!G.ok || (!PLAIN && G.ign[k])) continue;
                    const float u = xv[c][k];
                    const bool t = lab[k] == c;
                    const float r = rcp(u + em[k]);
                    const float ps = u * r, qs = em[k] * r;              // sigmoid(x), 1 - sigmoid(x)
                    const float pt = t ? ps : qs, omp = t ? qs : ps;
                    const float ce = -lg2(pt) * kLn2;
                    if constexpr (PLAIN) {
                        const float pq = ps * qs;
                        const float df = -2.0f * omp * (t ? pq : -pq);
                        float gx = k1 * (df * ce + omp * omp * (t ? -qs : ps)) + k2 * df;
                        gx += u * inv[k] * ((t ? gI[c] + gP[c] : gP[c]) - dot[k]);
                        out[k] = gx;
                        redo[k] = redo[k] || (t && ps < 1e-36f);
                        continue;
                    }
                    const float base_ = omp * cfg.sc;
                    const bool below = pt < cfg.thr;
                    float f, pw;
                    if (G2) { f = base_ * base_; pw = base_; }
                    else {
                        f = pow_pos(base_, cfg.gamma); f = cfg.g0 ? 1.0f : f;
                        pw = pow_pos(base_, cfg.gm1); pw = cfg.g1 ? 1.0f : pw;
                    }
                    f = below ? 1.0f : f;
                    const float dpt = t ? ps * qs : -(ps * qs);
                    float df = -cfg.gamma * pw * cfg.sc * dpt;
                    df = (below || (!G2 && cfg.g0)) ? 0.f : df;
                    const float w = w0 * (t ? cfg.a1 + cfg.a0 : cfg.a0);
                    float gx = k1 * w * (df * ce + f * (t ? -qs : ps)) + k2 * df;
                    const float Gc = t ? gI[c] + gP[c] : gP[c];
                    gx += u * inv[k] * (Gc - dot[k]);
                    out[k] = gx;
                    redo[k] = redo[k] || (t && ps < 1e-36f);
                }
                store_px_nt<PIX>(grad + base + (long long)c * a.HW, out, G.ok);
            }
        }
        if (G.ok) {   // exact rewrite of the elements the fast formulas cannot represent (never on sane logits)
#pragma unroll
            for (int k = 0; k < PIX; ++k) {
                if (!redo[k] || G.ign[k]) continue;
                for (int c = 0; c < C; ++c) {
                    if (!redo_all[k] && c != lab[k]) continue;
                    const float x = a.logits[base + (long long)c * a.HW + k];
                    const float t = lab[k] == c ? 1.f : 0.f;
                    float ce, f, df, p;
                    focal_parts<G2, true>(x, t, cfg, ce, f, df, p);
                    const float w = (a.class_weights ? a.class_weights[c] : 1.f) * (cfg.a1 * t + cfg.a0);
                    float gx = k1 * w * (df * ce + f * (p - t)) + k2 * df;
                    gx += fexp(x - mx[k]) * inv[k] * (gI[c] * t + gP[c] - dot[k]);
                    grad[base + (long long)c * a.HW + k] = gx;
                }
            }
        }
    }
}

// Straight-line instance of the fused backward for the common case (hard labels, C <= CREG padded with -inf, HW % 256 == 0,
// default focal configuration; see seg_fwd_lean_kernel): no divergent control flow, only the loads and the stores of the
// classes beyond C are guarded.
template <int CREG>
__global__ __launch_bounds__(256, 3) void seg_fused_bwd_lean_kernel(const SegArgs a, const float* __restrict__ coef, const float* __restrict__ gI,
                                                                    const float* __restrict__ gP, float* __restrict__ grad) {
    const int lane = threadIdx.x & 63, wave = wave_id();
    const int C = a.C;
    const float k1 = coef[0], k2 = coef[1];
    // dL/dP_c is wave-uniform (a scalar operand); dL/dI only matters for the label class of a pixel and is fetched per lane
    // (gI[label], 4 loads per group).  Keeping both as per-class values in vector registers costs 32 VGPRs and spilled.
    auto g_p = [&](int c) { return c < C ? gP[c] : 0.f; };
    const long long per_img = a.HW / 256;
    const long long groups = per_img * a.B;
    for (long long g = (long long)blockIdx.x * 4 + wave; g < groups; g += (long long)gridDim.x * 4) {
        const int b = (int)(g / per_img);
        const long long i0 = (g - (long long)b * per_img) * 256 + (long long)lane * 4;
        const long long base = (long long)b * C * a.HW + i0;
        int lab[4];
        {
            const long long* lp = a.labels + (long long)b * a.HW + i0;
            const longlong2 l01 = *reinterpret_cast<const longlong2*>(lp), l23 = *reinterpret_cast<const longlong2*>(lp + 2);
            lab[0] = (int)l01.x; lab[1] = (int)l01.y; lab[2] = (int)l23.x; lab[3] = (int)l23.y;
        }
        float gil[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) gil[k] = gI[min(max(lab[k], 0), C - 1)];
        float xv[CREG][4];
#pragma unroll
        for (int c = 0; c < CREG; ++c) {
#pragma unroll
            for (int k = 0; k < 4; ++k) xv[c][k] = -INFINITY;
            if (c < C) load_px<4>(a.logits + base + (long long)c * a.HW, xv[c], true);
        }
        float inv[4], em[4], dot[4], mx[4];
        bool redo_all[4], redo[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float m = xv[0][k];
#pragma unroll
            for (int c = 1; c < CREG; ++c) m = fmaxf(m, xv[c][k]);
            const float M = m * kLog2e;
            float d = 0.f, dd = 0.f, ulab = 0.f;
#pragma unroll
            for (int c = 0; c < CREG; ++c) {
                const float u = fexp_sub(xv[c][k], M);
                xv[c][k] = u;
                d += u;
                dd = __builtin_fmaf(g_p(c), u, dd);
                ulab = lab[k] == c ? u : ulab;
            }
            dd = __builtin_fmaf(gil[k], ulab, dd);
            inv[k] = rcp(d);
            dot[k] = dd * inv[k];
            em[k] = ex2(-M);
            mx[k] = m;
            redo_all[k] = !(fabsf(m) <= 60.f);
            redo[k] = redo_all[k];
        }
#pragma unroll
        for (int c = 0; c < CREG; ++c) {
            float out[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float u = xv[c][k];
                const bool t = lab[k] == c;
                const float r = rcp(u + em[k]);
                const float ps = u * r, qs = em[k] * r;
                const float pt = t ? ps : qs, omp = t ? qs : ps;
                const float ce = -lg2(pt) * kLn2;
                const float pq = ps * qs;
                const float df = -2.0f * omp * (t ? pq : -pq);
                float gx = k1 * (df * ce + omp * omp * (t ? -qs : ps)) + k2 * df;
                gx = __builtin_fmaf(u * inv[k], (t ? gil[k] : 0.f) + (g_p(c) - dot[k]), gx);
                out[k] = gx;
                redo[k] = redo[k] || (t && ps < 1e-36f);
            }
            if (c < C) { if (a.flags & SEG_NT_STORES) store_px_nt<4>(grad + base + (long long)c * a.HW, out, true); else store_px<4>(grad + base + (long long)c * a.HW, out, true); }
        }
        bool any_redo = false;
#pragma unroll
        for (int k = 0; k < 4; ++k) any_redo = any_redo || redo[k];
        if (__any(any_redo)) {   // exact rewrite of the elements the fast formulas cannot represent (never on sane logits)
            const FocalCfg cfg = focal_cfg(a);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (!redo[k]) continue;
                for (int c = 0; c < C; ++c) {
                    if (!redo_all[k] && c != lab[k]) continue;
                    const float x = a.logits[base + (long long)c * a.HW + k];
                    const float t = lab[k] == c ? 1.f : 0.f;
                    float ce, f, df, p;
                    focal_parts<true, true>(x, t, cfg, ce, f, df, p);
                    float gx = k1 * (df * ce + f * (p - t)) + k2 * df;
                    gx += fexp(x - mx[k]) * inv[k] * (gI[c] * t + gP[c] - dot[k]);
                    grad[base + (long long)c * a.HW + k] = gx;
                }
            }
        }
    }
}

// Backward of the dense-target statistics (and, FOCAL, of the default-configuration focal sums) as a plain stream: one wave =
// 1 024 consecutive elements of one (image, class) plane; grad = [k1 dL/dx + k2 dF/dx] + (gI_c t + gP_c) * dp/dx.
template <int PROB, bool IGN, bool FOCAL>
__global__ __launch_bounds__(256) void seg_dense_bwd_lean_kernel(const SegArgs a, const float* __restrict__ coef, const float* __restrict__ gI,
                                                                 const float* __restrict__ gP, float* __restrict__ grad) {
    static_assert(!FOCAL || PROB == PROB_SIGMOID, "the focal term shares the sigmoid of the statistics");
    const int lane = threadIdx.x & 63, wave = wave_id();
    const int C = a.C;
    const float k1 = FOCAL ? coef[0] : 0.f, k2 = FOCAL ? coef[1] : 0.f;
    const long long segs_per_plane = a.HW / 1024;
    const long long segs = segs_per_plane * C * a.B;
    for (long long sg = (long long)blockIdx.x * 4 + wave; sg < segs; sg += (long long)gridDim.x * 4) {
        const long long plane = sg / segs_per_plane;
        const int c = (int)(plane % C);
        const float gi = gI[c], gp = gP[c];
        const long long off = plane * a.HW + (sg - plane * segs_per_plane) * 1024 + (long long)lane * 4;
        float xv[4][4], tv[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            load_px<4>(a.logits + off + u * 256, xv[u], true);
            load_px<4>(a.dense + off + u * 256, tv[u], true);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float out[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float x = xv[u][k], t = tv[u][k];
                const bool ig = IGN && t == a.ignore_value;
                float gx;
                if (PROB == PROB_SIGMOID) {
                    const Sig sgm = sigmoid_parts(x);
                    const float p = sgm.p;
                    const float pq = p * (1.f - p);
                    gx = (gi * t + gp) * pq;
                    if (FOCAL) {   // gamma 2, no alpha / threshold / weights: f = (1 - pt)^2, df/dx = -2 (1 - pt) p (1 - p) (2t - 1)
                        const float ce = fmaxf(x, 0.f) - x * t + sgm.log1pe;
                        const float pt = __builtin_fmaf(p, t, (1.f - p) * (1.f - t));
                        const float omp = fmaxf(1.f - pt, 0.f);
                        const float df = -2.0f * omp * pq * (2.f * t - 1.f);
                        gx += k1 * (df * ce + omp * omp * (p - t)) + k2 * df;
                    }
                } else {
                    gx = gi * t + gp;
                }
                out[k] = ig ? 0.f : gx;
            }
            store_px_nt<4>(grad + off + u * 256, out, true);
        }
    }
}

// ------------------------------------------------------------------------------------------------ softmax focal
// softmax_focal_loss_with_logits (functional.py:110-173): per pixel sum_c pt_c^gamma * BCE(x_c, onehot_c) * w_c, masked by
// label != ignore_index.  sums[0] = sum of pixel losses, sums[1] = sum of ALL focal terms (the reference does not
// mask them, functional.py:161-164).  pixel_out (optional) receives the unreduced [B, HW] map.
struct SmfArgs {
    const float* logits; const long long* labels; const float* class_weights;
    double* sums; float* pixel_out; int* error_flag;
    int B, C; long long HW;
    int reduced; float gamma, threshold; long long ignore_label;
};

// f(pt) = pt^gamma, or the reduced variant (pt/thr)^gamma with f = 1 below thr; branch-free, G2 = gamma is exactly 2
template <bool G2>
__device__ __forceinline__ float smf_term(float pt, const SmfArgs& a, float sc, float thr) {
    const float base = pt * sc;
    float f = G2 ? base * base : (a.gamma == 0.f ? 1.0f : pow_pos(base, a.gamma));
    return pt < thr ? 1.0f : f;
}
template <bool G2>
__device__ __forceinline__ float smf_dterm(float pt, float t, const SmfArgs& a, float sc, float thr) {  // d f / d p
    const float base = pt * sc;
    float pw = G2 ? base : (a.gamma == 1.f ? 1.0f : pow_pos(base, a.gamma - 1.f));
    float df = a.gamma * pw * sc;
    df = (pt < thr || a.gamma == 0.f) ? 0.f : df;
    return df * (1.f - 2.f * t);
}

// MODE 0: forward.  MODE 1: backward,
//   grad_k = g1 * [p_k (h_k - sum_c h_c p_c) + w_k f_k (sigmoid(x_k) - t_k)] + k2 * p_k (d_k - sum_c d_c p_c),
//   h_c = w_c * bce_c * df_c/dp_c, d_c = df_c/dp_c; g1 = coef[0] (x per-pixel upstream gradient), k2 = coef[1].
template <int PIX, int CREG, int MODE, bool G2>
__global__ __launch_bounds__(256) void softmax_focal_kernel(const SmfArgs a, const float* __restrict__ coef,
                                                            const float* __restrict__ grad_pix, float* __restrict__ grad) {
    const float thr = a.reduced ? a.threshold : -INFINITY, sc = a.reduced ? 1.0f / a.threshold : 1.0f;
    const int lane = threadIdx.x & 63, wave = wave_id();
    const int C = a.C;
    double s_loss = 0.0, s_term = 0.0;
    const float k1 = MODE ? coef[0] : 0.f, k2 = MODE ? coef[1] : 0.f;
    const long long per_img = (a.HW + 64 * PIX - 1) / (64 * PIX);
    const long long groups = per_img * a.B;
    for (long long g = (long long)blockIdx.x * 4 + wave; g < groups; g += (long long)gridDim.x * 4) {
        const Group<PIX> G = make_group<PIX>(g, per_img, lane, a.HW, a.labels, true, a.ignore_label, C, MODE ? nullptr : a.error_flag);
        const long long base = (long long)G.b * C * a.HW + G.i0;
        float mx[PIX], inv[PIX], em[PIX], loss[PIX], dh[PIX], dd[PIX], gp[PIX];
        float tsum = 0.f;
        bool fast = false;
#pragma unroll
        for (int k = 0; k < PIX; ++k) { mx[k] = -INFINITY; inv[k] = 0.f; em[k] = 1.f; loss[k] = 0.f; dh[k] = 0.f; dd[k] = 0.f; gp[k] = 1.f; }
        if (MODE && grad_pix) load_px<PIX>(grad_pix + (long long)G.b * a.HW + G.i0, gp, G.ok);
        constexpr int NR = CREG > 0 ? CREG : 1;
        float xr[NR][PIX];
        if constexpr (CREG > 0) {
#pragma unroll
            for (int c = 0; c < CREG; ++c) {
#pragma unroll
                for (int k = 0; k < PIX; ++k) xr[c][k] = 0.f;
                if (c < C) load_px<PIX>(a.logits + base + (long long)c * a.HW, xr[c], G.ok);
            }
            // Fast path (wave-uniform): keep u = exp(x - m) in the registers instead of x -- the softmax probability is
            // u / d, and with em = exp(-m) the sigmoid the BCE term needs is u / (u + em), 1 - sigmoid = em / (u + em), BCE =
            // -log of one of the two: one exp per element instead of one per element and pass plus a sigmoid.  Needs u, em and
            // both quotients to be normal fp32 numbers: every logit in [-80, 60] and x - m >= -80, else the wave takes the exact path.
            bool tame = true;
#pragma unroll
            for (int k = 0; k < PIX; ++k) {
                float m = -INFINITY, lo = INFINITY;
#pragma unroll
                for (int c = 0; c < CREG; ++c) if (c < C) { m = fmaxf(m, xr[c][k]); lo = fminf(lo, xr[c][k]); }
                mx[k] = m;
                tame = tame && (m <= 60.f) && (lo >= -80.f) && (lo - m >= -80.f);   // every sigmoid and its complement a normal fp32 number
            }
            fast = !__any(!tame);
#pragma unroll
            for (int k = 0; k < PIX; ++k) {
                const float m = mx[k];
                float d = 0.f;
#pragma unroll
                for (int c = 0; c < CREG; ++c) if (c < C) {
                    const float u = fexp(xr[c][k] - m);
                    d += u;
                    if (fast) xr[c][k] = u;
                }
                inv[k] = rcp(d);
                em[k] = fexp(-m);
                if (!fast) opaque(mx[k]);  // exact path: recompute exp(x - m) per pass instead of keeping 64 values alive
            }
        } else {
            for (int c = 0; c < C; ++c) {
                float xv[PIX];
                load_px<PIX>(a.logits + base + (long long)c * a.HW, xv, G.ok);
#pragma unroll
                for (int k = 0; k < PIX; ++k) if (G.ok) {
                    const float m2 = fmaxf(mx[k], xv[k]);
                    inv[k] = inv[k] * fexp(mx[k] - m2) + fexp(xv[k] - m2);
                    mx[k] = m2;
                }
            }
#pragma unroll
            for (int k = 0; k < PIX; ++k) inv[k] = rcp(inv[k]);
        }
        // pass 1: losses (forward) or the two softmax-Jacobian dot products (backward)
        auto pass1 = [&](int c, const float (&xv)[PIX], auto fast_tag) {
            constexpr bool FAST = decltype(fast_tag)::value;
            const float w = a.class_weights ? a.class_weights[c] : 1.f;
#pragma unroll
            for (int k = 0; k < PIX; ++k) {
                if (!G.ok) continue;
                const long long tgt = G.ign[k] ? 0 : G.lab[k];   // masked_fill(target, ignore, 0), functional.py:139
                const float x = xv[k];                            // FAST: u = exp(logit - m)
                const float t = c == tgt ? 1.f : 0.f;
                float p, bce;
                if constexpr (FAST) {
                    p = x * inv[k];
                    const float r = rcp(x + em[k]);
                    bce = -lg2((c == tgt ? x : em[k]) * r) * kLn2;
                } else {
                    p = fexp(x - mx[k]) * inv[k];
                    bce = fmaxf(x, 0.f) - x * t + sigmoid_parts(x).log1pe;
                }
                const float pt = (1.f - t) * p + t * (1.f - p);
                if (MODE == 0) {
                    const float f = smf_term<G2>(pt, a, sc, thr);
                    loss[k] += f * bce * w;
                    tsum += f;
                } else {
                    const float df = smf_dterm<G2>(pt, t, a, sc, thr);
                    dh[k] += w * bce * df * p;
                    dd[k] += df * p;
                }
            }
        };
        if constexpr (CREG > 0) {
            if (fast) {
#pragma unroll
                for (int c = 0; c < CREG; ++c) if (c < C) pass1(c, xr[c], std::true_type{});
            } else {
#pragma unroll
                for (int c = 0; c < CREG; ++c) if (c < C) pass1(c, xr[c], std::false_type{});
            }
        } else {
            for (int c = 0; c < C; ++c) {
                float xv[PIX];
                load_px<PIX>(a.logits + base + (long long)c * a.HW, xv, G.ok);
                pass1(c, xv, std::false_type{});
            }
        }
        if (MODE == 0) {
            float ls = 0.f;
#pragma unroll
            for (int k = 0; k < PIX; ++k) {
                if (G.ign[k] || !G.ok) loss[k] = 0.f;
                ls += loss[k];
            }
            s_loss += (double)ls;
            s_term += (double)tsum;
            if (a.pixel_out) store_px<PIX>(a.pixel_out + (long long)G.b * a.HW + G.i0, loss, G.ok);
        } else {
            if (!fast) {
#pragma unroll
                for (int k = 0; k < PIX; ++k) { opaque(mx[k]); opaque(inv[k]); }  // pass 2 recomputes p, bce, df
            }
            auto pass2 = [&](int c, const float (&xv)[PIX], auto fast_tag) {
                constexpr bool FAST = decltype(fast_tag)::value;
                const float w = a.class_weights ? a.class_weights[c] : 1.f;
                float out[PIX];
#pragma unroll
                for (int k = 0; k < PIX; ++k) {
                    out[k] = 0.f;
                    if (!G.ok) continue;
                    const long long tgt = G.ign[k] ? 0 : G.lab[k];
                    const float x = xv[k];                        // FAST: u = exp(logit - m)
                    const float t = c == tgt ? 1.f : 0.f;
                    float p, bce, sig_minus_t;
                    if constexpr (FAST) {
                        p = x * inv[k];
                        const float r = rcp(x + em[k]);
                        const float ps = x * r, qs = em[k] * r;
                        bce = -lg2(c == tgt ? ps : qs) * kLn2;
                        sig_minus_t = c == tgt ? -qs : ps;
                    } else {
                        p = fexp(x - mx[k]) * inv[k];
                        const Sig sg = sigmoid_parts(x);
                        bce = fmaxf(x, 0.f) - x * t + sg.log1pe;
                        sig_minus_t = sg.p - t;
                    }
                    const float pt = (1.f - t) * p + t * (1.f - p);
                    const float df = smf_dterm<G2>(pt, t, a, sc, thr);
                    const float f = smf_term<G2>(pt, a, sc, thr);
                    const float g1 = G.ign[k] ? 0.f : (grad_pix ? k1 * gp[k] : k1);
                    const float gl = p * (w * bce * df - dh[k]) + w * f * sig_minus_t;
                    const float gf = p * (df - dd[k]);
                    out[k] = g1 * gl + k2 * gf;
                }
                store_px_nt<PIX>(grad + base + (long long)c * a.HW, out, G.ok);
            };
            if constexpr (CREG > 0) {
                if (fast) {
#pragma unroll
                    for (int c = 0; c < CREG; ++c) if (c < C) pass2(c, xr[c], std::true_type{});
                } else {
#pragma unroll
                    for (int c = 0; c < CREG; ++c) if (c < C) pass2(c, xr[c], std::false_type{});
                }
            } else {
                for (int c = 0; c < C; ++c) {
                    float xv[PIX];
                    load_px<PIX>(a.logits + base + (long long)c * a.HW, xv, G.ok);
                    pass2(c, xv, std::false_type{});
                }
            }
        }
    }
    if (MODE == 0) block_add2(s_loss, s_term, a.sums + (size_t)(blockIdx.x % SUM_SLOTS) * 2, lane, wave);
}

// Backward for C <= CREG with ONE evaluation of the transcendental terms: with A_c = p_c w_c bce_c df_c, D_c = p_c df_c and
// F_c = w_c f_c (sigmoid(x_c) - t_c) the gradient above is  grad_c = T_c - p_c S,  T_c = g1 (A_c + F_c) + k2 D_c,
// S = g1 sum_c A_c + k2 sum_c D_c.  Pass 1 leaves T_c in registers next to u_c = exp(x_c - m) (2 x CREG x PIX values), pass 2 is one
// fma and the store: exp + rcp + log per element instead of exp + 2 (rcp + log).  The exact (non-"tame") waves keep x_c and pay
// one more exp in pass 2.
template <int PIX, int CREG, bool G2, bool FULL>   // FULL: C == CREG (no padded classes)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PIX == 2 ? 4 : 2))) void softmax_focal_bwd_kernel(
    const SmfArgs a, const float* __restrict__ coef, const float* __restrict__ grad_pix, float* __restrict__ grad) {
    const float thr = a.reduced ? a.threshold : -INFINITY, sc = a.reduced ? 1.0f / a.threshold : 1.0f;
    const int lane = threadIdx.x & 63, wave = wave_id();
    const int C = FULL ? CREG : a.C;
    const float k1 = coef[0], k2 = coef[1];
    const long long per_img = (a.HW + 64 * PIX - 1) / (64 * PIX);
    const long long groups = per_img * a.B;
    for (long long g = (long long)blockIdx.x * 4 + wave; g < groups; g += (long long)gridDim.x * 4) {
        const Group<PIX> G = make_group<PIX>(g, per_img, lane, a.HW, a.labels, true, a.ignore_label, C, nullptr);
        if (!G.ok) continue;
        const long long base = (long long)G.b * C * a.HW + G.i0;
        float mx[PIX], inv[PIX], em[PIX], S[PIX], g1[PIX];
        int tg[PIX];
#pragma unroll
        for (int k = 0; k < PIX; ++k) { g1[k] = 1.f; tg[k] = G.ign[k] ? 0 : (int)G.lab[k]; }   // masked_fill(target, ignore, 0), functional.py:139
        if (grad_pix) load_px<PIX>(grad_pix + (long long)G.b * a.HW + G.i0, g1, true);
        float xr[CREG][PIX], T[CREG][PIX];
#pragma unroll
        for (int c = 0; c < CREG; ++c) {
#pragma unroll
            for (int k = 0; k < PIX; ++k) xr[c][k] = -INFINITY;     // padded classes: u = 0 exactly, every term below is 0
            if (FULL || c < C) load_px<PIX>(a.logits + base + (long long)c * a.HW, xr[c], true);
        }
        bool tame = true;
#pragma unroll
        for (int k = 0; k < PIX; ++k) {
            float m = xr[0][k], lo = xr[0][k];
#pragma unroll
            for (int c = 1; c < CREG; ++c) { m = fmaxf(m, xr[c][k]); lo = fminf(lo, (FULL || c < C) ? xr[c][k] : lo); }
            mx[k] = m;
            tame = tame && (m <= 60.f) && (lo >= -80.f) && (lo - m >= -80.f);   // every sigmoid and its complement a normal fp32 number
            g1[k] = G.ign[k] ? 0.f : k1 * g1[k];
            S[k] = 0.f;
        }
        if (__any(!tame)) {
            // exact formulas for this group (never on sane logits): rolled loops over the classes, logits re-read from L2 -- small
            // code and few registers, so the straight-line path below keeps its registers for the 2 x CREG x PIX stashed values
#pragma unroll
            for (int k = 0; k < PIX; ++k) {   // (unrolled: tg / g1 / mx are register arrays and need static indices)
                const float* xp = a.logits + base + k;
                float d = 0.f;
                for (int c = 0; c < C; ++c) d += fexp(xp[(long long)c * a.HW] - mx[k]);
                const float iv = rcp(d);
                float dh = 0.f, dd = 0.f;
                for (int c = 0; c < C; ++c) {
                    const float x = xp[(long long)c * a.HW], w = a.class_weights ? a.class_weights[c] : 1.f;
                    const float t = c == tg[k] ? 1.f : 0.f;
                    const float p = fexp(x - mx[k]) * iv;
                    const float pt = (1.f - t) * p + t * (1.f - p);
                    const float bce = fmaxf(x, 0.f) - x * t + sigmoid_parts(x).log1pe;
                    const float df = smf_dterm<G2>(pt, t, a, sc, thr);
                    dh += w * bce * df * p; dd += df * p;
                }
                for (int c = 0; c < C; ++c) {
                    const float x = xp[(long long)c * a.HW], w = a.class_weights ? a.class_weights[c] : 1.f;
                    const float t = c == tg[k] ? 1.f : 0.f;
                    const float p = fexp(x - mx[k]) * iv;
                    const float pt = (1.f - t) * p + t * (1.f - p);
                    const Sig sg = sigmoid_parts(x);
                    const float bce = fmaxf(x, 0.f) - x * t + sg.log1pe;
                    const float df = smf_dterm<G2>(pt, t, a, sc, thr), f = smf_term<G2>(pt, a, sc, thr);
                    grad[base + (long long)c * a.HW + k] = g1[k] * (p * (w * bce * df - dh) + w * f * (sg.p - t)) + k2 * (p * (df - dd));
                }
            }
            continue;
        }
#pragma unroll
        for (int k = 0; k < PIX; ++k) {
            const float M = mx[k] * kLog2e;
            float d = 0.f;
#pragma unroll
            for (int c = 0; c < CREG; ++c) {
                const float u = fexp_sub(xr[c][k], M);
                d += u;
                xr[c][k] = u;
            }
            inv[k] = rcp(d);
            em[k] = ex2(-M);
        }
#pragma unroll
        for (int c = 0; c < CREG; ++c) if (FULL || c < C) {
            const float w = a.class_weights ? a.class_weights[c] : 1.f;
#pragma unroll
            for (int k = 0; k < PIX; ++k) {
                const float u = xr[c][k];                         // exp(logit - m)
                const bool is_t = c == tg[k];
                const float t = is_t ? 1.f : 0.f;
                const float p = u * inv[k];
                const float r = rcp(u + em[k]);
                const float ps = u * r, qs = em[k] * r;           // sigmoid(x), 1 - sigmoid(x)
                const float bce = -lg2(is_t ? ps : qs) * kLn2;
                const float smt = is_t ? -qs : ps;
                const float pt = is_t ? 1.f - p : p;
                const float df = smf_dterm<G2>(pt, t, a, sc, thr);
                const float f = smf_term<G2>(pt, a, sc, thr);
                const float D = p * df;
                const float A = D * (w * bce);
                const float v = g1[k] * A + k2 * D;
                S[k] += v;
                T[c][k] = v + g1[k] * (w * f * smt);
            }
        }
#pragma unroll
        for (int k = 0; k < PIX; ++k) S[k] *= inv[k];
#pragma unroll
        for (int c = 0; c < CREG; ++c) if (FULL || c < C) {
            float out[PIX];
#pragma unroll
            for (int k = 0; k < PIX; ++k) out[k] = T[c][k] - xr[c][k] * S[k];
            store_px_nt<PIX>(grad + base + (long long)c * a.HW, out, true);
        }
    }
}

// Straight-line instances of the softmax focal loss for the common case (C <= CREG padded with -inf, HW % 256 == 0, gamma = 2,
// no class weights, no reduced threshold; see seg_fwd_lean_kernel).  u = exp(x - m) stays in the registers; the BCE term's
// sigmoid comes from u and em = exp(-m) like in the wave-uniform fast path of softmax_focal_kernel, and a wave that holds a
// pixel with |m| > 60 or a logit 80 below its maximum falls back to that kernel's exact formulas for the whole group.
template <int CREG, int MODE>
__global__ __launch_bounds__(256, 3) void softmax_focal_lean_kernel(const SmfArgs a, const float* __restrict__ coef,
                                                                    const float* __restrict__ grad_pix, float* __restrict__ grad) {
    const int lane = threadIdx.x & 63, wave = wave_id();
    const int C = a.C;
    double s_loss = 0.0, s_term = 0.0;
    const float k1 = MODE ? coef[0] : 0.f, k2 = MODE ? coef[1] : 0.f;
    const long long per_img = a.HW / 256;
    const long long groups = per_img * a.B;
    for (long long g = (long long)blockIdx.x * 4 + wave; g < groups; g += (long long)gridDim.x * 4) {
        const int b = (int)(g / per_img);
        const long long i0 = (g - (long long)b * per_img) * 256 + (long long)lane * 4;
        const long long base = (long long)b * C * a.HW + i0;
        int tgt[4];
        bool ign[4];
        {
            const long long* lp = a.labels + (long long)b * a.HW + i0;
            const longlong2 l01 = *reinterpret_cast<const longlong2*>(lp), l23 = *reinterpret_cast<const longlong2*>(lp + 2);
            const long long l64[4] = {l01.x, l01.y, l23.x, l23.y};
            bool bad = false;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                ign[k] = l64[k] == a.ignore_label;
                bad = bad || (!ign[k] && (l64[k] < 0 || l64[k] >= C));
                tgt[k] = ign[k] ? 0 : (int)l64[k];          // masked_fill(target, ignore, 0), functional.py:139
            }
            if (MODE == 0 && bad) raise_label_error(a.error_flag);
        }
        float gp[4] = {1.f, 1.f, 1.f, 1.f};
        if (MODE && grad_pix) load_px<4>(grad_pix + (long long)b * a.HW + i0, gp, true);
        float xv[CREG][4];
#pragma unroll
        for (int c = 0; c < CREG; ++c) {
#pragma unroll
            for (int k = 0; k < 4; ++k) xv[c][k] = -INFINITY;
            if (c < C) load_px<4>(a.logits + base + (long long)c * a.HW, xv[c], true);
        }
        float mx[4], inv[4], em[4];
        bool tame = true;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float m = xv[0][k], lo = xv[0][k];
#pragma unroll
            for (int c = 1; c < CREG; ++c) { m = fmaxf(m, xv[c][k]); lo = (c < C) ? fminf(lo, xv[c][k]) : lo; }
            mx[k] = m;
            tame = tame && (m <= 60.f) && (lo >= -80.f) && (lo - m >= -80.f);   // every sigmoid and its complement a normal fp32 number
        }
        if (__any(!tame)) {
            // exact formulas for this group (never on sane logits): rolled loops over the classes, logits re-read from L2 --
            // small code and few registers, so the fast path keeps its occupancy
            float lsum = 0.f, tsum = 0.f, pl[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {   // (unrolled: tgt / ign / mx / gp are register arrays and need static indices)
                const float* xp = a.logits + base + k;
                float d = 0.f;
                for (int c = 0; c < C; ++c) d += fexp(xp[(long long)c * a.HW] - mx[k]);
                const float iv = rcp(d);
                float loss = 0.f, dh = 0.f, dd = 0.f;
                for (int c = 0; c < C; ++c) {
                    const float x = xp[(long long)c * a.HW];
                    const float t = c == tgt[k] ? 1.f : 0.f;
                    const float p = fexp(x - mx[k]) * iv;
                    const float pt = (1.f - t) * p + t * (1.f - p);
                    const float bce = fmaxf(x, 0.f) - x * t + sigmoid_parts(x).log1pe;
                    const float df = 2.0f * pt * (1.f - 2.f * t);
                    loss += pt * pt * bce; tsum += pt * pt;
                    dh += bce * df * p; dd += df * p;
                }
                pl[k] = ign[k] ? 0.f : loss;
                lsum += pl[k];
                if (MODE) {
                    const float g1 = ign[k] ? 0.f : k1 * gp[k];
                    for (int c = 0; c < C; ++c) {
                        const float x = xp[(long long)c * a.HW];
                        const float t = c == tgt[k] ? 1.f : 0.f;
                        const float p = fexp(x - mx[k]) * iv;
                        const float pt = (1.f - t) * p + t * (1.f - p);
                        const Sig sg = sigmoid_parts(x);
                        const float bce = fmaxf(x, 0.f) - x * t + sg.log1pe;
                        const float df = 2.0f * pt * (1.f - 2.f * t);
                        grad[base + (long long)c * a.HW + k] = g1 * (p * (bce * df - dh) + pt * pt * (sg.p - t)) + k2 * (p * (df - dd));
                    }
                }
            }
            if (MODE == 0) {
                s_loss += (double)lsum; s_term += (double)tsum;
                if (a.pixel_out) store_px<4>(a.pixel_out + (long long)b * a.HW + i0, pl, true);
            }
            continue;
        }
        float loss[4], dh[4], dd[4];
        float tsum = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float M = mx[k] * kLog2e;
            float d = 0.f;
#pragma unroll
            for (int c = 0; c < CREG; ++c) {
                const float u = fexp_sub(xv[c][k], M);
                xv[c][k] = u;
                d += u;
            }
            inv[k] = rcp(d);
            em[k] = ex2(-M);
            loss[k] = 0.f; dh[k] = 0.f; dd[k] = 0.f;
        }
        // pass 1: the pixel loss (forward) or the two softmax-Jacobian dot products (backward)
#pragma unroll
        for (int c = 0; c < CREG; ++c) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float u = xv[c][k];
                const bool hit = c == tgt[k];
                const float p = u * inv[k];
                const float r = rcp(u + em[k]);
                const float bce = -lg2((hit ? u : em[k]) * r) * kLn2;
                const float pt = hit ? 1.f - p : p;
                if (MODE == 0) {
                    const float f = pt * pt;
                    loss[k] = __builtin_fmaf(f, bce, loss[k]);
                    tsum += f;
                } else {
                    const float dfp = (hit ? -2.0f : 2.0f) * pt * p;     // df * p
                    dh[k] = __builtin_fmaf(bce, dfp, dh[k]);
                    dd[k] += dfp;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE == 0) {
            float ls = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) { loss[k] = ign[k] ? 0.f : loss[k]; ls += loss[k]; }
            s_loss += (double)ls;
            s_term += (double)tsum;
            if (a.pixel_out) store_px<4>(a.pixel_out + (long long)b * a.HW + i0, loss, true);
        } else {
            float g1[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                g1[k] = ign[k] ? 0.f : k1 * gp[k];
                asm volatile("" : "+v"(tgt[k]));
                opaque(em[k]); opaque(inv[k]);   // pass 2 RECOMPUTES rcp(u + em) and u * inv: kept alive from pass 1 they cost 128 VGPRs (spills)
            }
#pragma unroll
            for (int c = 0; c < CREG; ++c) {
                float out[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float u = xv[c][k];
                    const bool hit = c == tgt[k];
                    const float p = u * inv[k];
                    const float r = rcp(u + em[k]);
                    const float ps = u * r, qs = em[k] * r;
                    const float bce = -lg2(hit ? ps : qs) * kLn2;
                    const float pt = hit ? 1.f - p : p;
                    const float df = (hit ? -2.0f : 2.0f) * pt;
                    const float gl = __builtin_fmaf(p, bce * df - dh[k], pt * pt * (hit ? -qs : ps));
                    out[k] = __builtin_fmaf(g1[k], gl, k2 * (p * (df - dd[k])));
                }
                if (c < C) store_px_nt<4>(grad + base + (long long)c * a.HW, out, true);
                __builtin_amdgcn_sched_barrier(0);   // one class at a time (interleaving all 64 element chains spills)
            }
        }
    }
    if (MODE == 0) block_add2(s_loss, s_term, a.sums + (size_t)(blockIdx.x % SUM_SLOTS) * 2, lane, wave);
}

// ------------------------------------------------------------------------------------------------ scalar epilogue
// The [C]-sized tail of DiceLoss / JaccardLoss / the fused focal+Dice+Jaccard loss (dice.py:112-131, jaccard.py:95-113) AND
// its derivative in one launch: as torch ops the tail is ~25 launches forward and ~40 in autograd's backward, more than the
// streaming kernels they follow.  One workgroup; fp32 arithmetic in the reference's order (the statistics arrive as fp64
// slot sums and are rounded to fp32 first, like `.float()`).
struct EpiArgs {
    const double* sums;   // [slots][2 + 3C]
    int slots, C;
    float focal_scale, dice_w, jacc_w, smooth, eps;
    int log_loss;
    const unsigned char* class_mask;  // [C] or null: classes that enter the mean
    int n_selected;
    float* loss;          // [1]
    float* coef;          // [2 + 2C]: d loss / d (focal loss sum, focal term sum, I[C], P[C])
    const int* error_flag;  // set by the forward kernel on a label outside [0, C): the loss becomes NaN (or null)
};

__device__ __forceinline__ void score_loss(float score, bool active, int log_loss, float eps, float& loss, float& dscore) {
    if (log_loss) {
        const float cl = fmaxf(score, eps);
        loss = -logf(cl);
        dscore = score >= eps ? -1.0f / cl : 0.f;
    } else {
        loss = 1.0f - score;
        dscore = -1.0f;
    }
    if (!active) { loss = 0.f; dscore = 0.f; }
}

__global__ __launch_bounds__(256) void region_epilogue_kernel(const EpiArgs a) {
    __shared__ double red[2][256];
    const int C = a.C, row = 2 + 3 * C;
    double dsum = 0.0, jsum = 0.0;
    const float inv_n = 1.0f / (float)a.n_selected;
    for (int c = threadIdx.x; c < C; c += 256) {
        double dI = 0.0, dP = 0.0, dT = 0.0;
        for (int sl = 0; sl < a.slots; ++sl) {
            const double* r = a.sums + (size_t)sl * row;
            dI += r[2 + c]; dP += r[2 + C + c]; dT += r[2 + 2 * C + c];
        }
        const float I = (float)dI, P = (float)dP, T = (float)dT;
        const bool sel = !a.class_mask || a.class_mask[c];
        const bool active = T > 0.f;
        float gI = 0.f, gP = 0.f;
        if (a.dice_w != 0.f) {
            const float num = 2.0f * I + a.smooth, card = P + T + a.smooth, den = fmaxf(card, a.eps);
            const float score = num / den;
            float l, ds;
            score_loss(score, active, a.log_loss, a.eps, l, ds);
            if (sel) {
                dsum += (double)l;
                gI += a.dice_w * ds * (2.0f / den);
                gP += a.dice_w * ds * (card >= a.eps ? -num / (den * den) : 0.f);
            }
        }
        if (a.jacc_w != 0.f) {
            const float num = I + a.smooth, uni = P + T - I + a.smooth, den = fmaxf(uni, a.eps);
            const float score = num / den;
            float l, ds;
            score_loss(score, active, a.log_loss, a.eps, l, ds);
            if (sel) {
                jsum += (double)l;
                const float dden = uni >= a.eps ? num / (den * den) : 0.f;
                gI += a.jacc_w * ds * (1.0f / den + dden);
                gP += a.jacc_w * ds * (-dden);
            }
        }
        a.coef[2 + c] = gI * inv_n;
        a.coef[2 + C + c] = gP * inv_n;
    }
    red[0][threadIdx.x] = dsum; red[1][threadIdx.x] = jsum;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { red[0][threadIdx.x] += red[0][threadIdx.x + o]; red[1][threadIdx.x] += red[1][threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double f = 0.0;
        for (int sl = 0; sl < a.slots; ++sl) f += a.sums[(size_t)sl * row];
        const float focal = a.focal_scale != 0.f ? a.focal_scale * (float)f : 0.f;
        const float dice = a.dice_w != 0.f ? a.dice_w * ((float)red[0][0] * inv_n) : 0.f;
        const float jacc = a.jacc_w != 0.f ? a.jacc_w * ((float)red[1][0] * inv_n) : 0.f;
        a.loss[0] = (a.error_flag && *a.error_flag) ? __builtin_nanf("") : focal + dice + jacc;
        a.coef[0] = a.focal_scale;
        a.coef[1] = 0.f;
    }
}

// ------------------------------------------------------------------------------------------------ host side
int g_loss_grid_cap = 0;  // 0 = per-kernel default; otherwise workgroups per launch (ptb_set_tunable key 4)
int g_loss_prefetch = 0;   // ptb_set_tunable key 8: register double buffering in the fused loss forward (measured: no gain, 0.146 vs 0.141-0.145 ms)
int g_smf_bwd_stash = 4;  // ptb_set_tunable key 7: 4 pixels per lane (251 VGPRs, 2 waves per SIMD) measured 0.372 ms fwd+bwd at cfg4, 2 pixels 0.54, the two-pass kernel 0.41-0.48
int g_nt_grad_stores = 1;    // ptb_set_tunable key 16: non-temporal stores of the gradient in the fused backward (A/B: 333 -> 315 us fwd+bwd at cfg4; the other backward kernels always use them)
int g_focal_pk_grid = 512;   // ptb_set_tunable key 13: workgroups of seg_focal_pk_kernel (2 per CU measured best: per-workgroup prologue / epilogue / slot atomics)
int g_stats_pk = 1;       // ptb_set_tunable key 20: Dice / Jaccard statistics and the default BinaryFocalLoss on label maps take the packed streaming kernel (0: the lean kernels)
int g_focal_pk = 1;       // ptb_set_tunable key 12 (2 = with register prefetch of the next pixel group): packed-fp32 / per-pixel-correction instance of the fused forward (seg_focal_pk_kernel); 0 = seg_fwd_lean_kernel
int g_fused_pix2 = 1;     // ptb_set_tunable key 5: fused focal + statistics forward with 2 pixels per lane (120 VGPRs, 4 waves per SIMD,
                          // instead of 4 pixels: 163 VGPRs, 3 waves): 0.164-0.171 vs 0.173-0.186 ms per FocalDiceJaccardLoss forward at cfg4
}  // namespace ptb

using namespace ptb;

static int seg_loss_fwd_launch(SegArgs& a, hipStream_t s);

// BinaryFocalLoss() on label maps in its default configuration (gamma 2, nothing else; mean / sum / normalized): served by the
// focal-only instance of the packed streaming kernel -- the only focal-only kernel that carries the in-launch tail.
static bool focal_only_pk(const SegArgs& a) {
    return (a.flags & (SEG_FOCAL | SEG_STATS)) == SEG_FOCAL && !g_force_scalar && g_stats_pk && a.labels && !a.dense && a.HW % 128 == 0 && a.C <= 16 &&
           a.gamma == 2.0f && !a.class_weights && !(a.flags & (SEG_HAS_ALPHA | SEG_REDUCED | SEG_ELEMWISE | SEG_HAS_IGNORE)) &&
           vec_ok(a.HW, {a.logits, a.dense, a.elem_out, a.labels});
}

extern "C" int ptb_seg_loss_fwd(const float* logits, const int64_t* labels, const float* dense, const float* class_weights,
                                double* sums, float* elem_out, int* error_flag, int B, int C, int64_t HW, int flags, int prob,
                                float gamma, float alpha, float threshold, int64_t ignore_label, float ignore_value,
                                ptb_stream_t stream) {
    SegArgs a{};
    if (int rc = fill_seg(a, logits, labels, dense, class_weights, B, C, HW, flags, prob, gamma, alpha, threshold, ignore_label, ignore_value)) return rc;
    if (!sums || !error_flag || ((flags & SEG_ELEMWISE) && !elem_out)) return PTB_EINVAL;
    if (C > 1024) return PTB_EUNSUPPORTED;
    a.sums = sums; a.elem_out = elem_out; a.error_flag = error_flag;
    hipStream_t s = (hipStream_t)stream;
    if (int rc = zero_sums(sums, 2 + 3 * C, error_flag, s)) return rc;   // the slot sums and the label flag start from zero
    if ((long long)B * HW == 0) return PTB_OK;
    return seg_loss_fwd_launch(a, s);
}

extern "C" int64_t ptb_region_workspace_bytes(int C) {
    if (C < 1 || C > 1024) return PTB_EINVAL;
    return (int64_t)SUM_SLOTS * (2 + 3 * C) * (int64_t)sizeof(double) + (SUM_SLOTS + 2) * (int64_t)sizeof(int);   // slot sums, tickets, label flag
}

// The whole Dice / Jaccard / focal + Dice + Jaccard forward in ONE launch: the streaming statistics kernel, whose last-arriving
// workgroup adds up the slots, evaluates the scalar epilogue and its derivative (what ptb_region_epilogue computes) and leaves the
// workspace zeroed.  `workspace`: ptb_region_workspace_bytes(C) bytes of device memory that were zero before the FIRST call and
// are used by one stream at a time; every call leaves them zero again.
extern "C" int ptb_region_loss_fwd(const float* logits, const int64_t* labels, const float* dense, const float* class_weights,
                                   void* workspace, int B, int C, int64_t HW, int flags, int prob, float gamma, float alpha,
                                   float threshold, int64_t ignore_label, float ignore_value, float focal_scale, float dice_weight,
                                   float jaccard_weight, float smooth, float eps, int log_loss, const unsigned char* class_mask,
                                   int n_selected, float* loss, float* coef, int* error_out, ptb_stream_t stream) {
    SegArgs a{};
    if (int rc = fill_seg(a, logits, labels, dense, class_weights, B, C, HW, flags, prob, gamma, alpha, threshold, ignore_label, ignore_value)) return rc;
    if (!workspace || !loss || !coef || n_selected < 1 || (reinterpret_cast<uintptr_t>(workspace) & 7u)) return PTB_EINVAL;
    if (!(flags & (SEG_STATS | SEG_FOCAL)) || (flags & SEG_ELEMWISE)) return PTB_EINVAL;
    if (!(flags & SEG_STATS) && !focal_only_pk(a)) return PTB_EUNSUPPORTED;      // (focal alone: only the packed instance ends with the tail)
    if (C > 1024) return PTB_EUNSUPPORTED;
    if ((long long)B * HW == 0) return PTB_EUNSUPPORTED;      // (nothing would launch: the caller composes the empty case)
    a.sums = static_cast<double*>(workspace);
    int* words = reinterpret_cast<int*>(a.sums + (size_t)SUM_SLOTS * (2 + 3 * C));
    a.tail_counter = reinterpret_cast<unsigned int*>(words);
    a.error_flag = words + SUM_SLOTS + 1;
    a.tail_error_out = error_out;
    a.tail_class_mask = class_mask;
    a.tail_loss = loss; a.tail_coef = coef;
    a.tail_focal_scale = (flags & SEG_FOCAL) ? focal_scale : 0.f;
    a.tail_dice_w = dice_weight; a.tail_jacc_w = jaccard_weight; a.tail_smooth = smooth; a.tail_eps = eps;
    a.tail_log_loss = log_loss; a.tail_n_selected = n_selected;
    return seg_loss_fwd_launch(a, (hipStream_t)stream);
}

static int seg_loss_fwd_launch(SegArgs& a, hipStream_t s) {
    const float *logits = a.logits, *dense = a.dense, *class_weights = a.class_weights;
    const long long* labels = a.labels;
    float* elem_out = a.elem_out;
    const int B = a.B, C = a.C, flags = a.flags, prob = a.prob;
    const int64_t HW = a.HW;
    const float gamma = a.gamma;
    // [4 waves][3][C] floats of per-wave sums; the in-launch tail reuses it for (2 + 3 C) doubles + 8 more (+ alignment slack)
    const size_t shmem = (size_t)4 * 3 * C * sizeof(float) + (a.tail_counter ? 128 : 0);
    const int what = flags & (SEG_FOCAL | SEG_STATS);
    if (!what) return PTB_EINVAL;
    const bool g2 = gamma == 2.0f;
    const bool vec = vec_ok(HW, {logits, dense, elem_out, labels});
    const int gcap = what == SEG_FOCAL ? kGridStream : kGridStats;
    const dim3 grid(vec ? grid_for_groups((HW + 255) / 256 * B, gcap) : grid_for_groups((HW + 63) / 64 * B, gcap)), block(256);
    if (!g_force_scalar && dense && !labels && what == (SEG_FOCAL | SEG_STATS) && vec && HW % 1024 == 0 && prob == PROB_SIGMOID && g2 &&
        !class_weights && !(flags & (SEG_HAS_ALPHA | SEG_REDUCED | SEG_ELEMWISE))) {
        const dim3 dgrid(grid_for_groups(HW / 1024 * C * B, kGridStream));
        if (flags & SEG_HAS_IGNORE) hipLaunchKernelGGL((seg_stats_dense_lean_kernel<PROB_SIGMOID, true, true>), dgrid, block, shmem, s, a);
        else hipLaunchKernelGGL((seg_stats_dense_lean_kernel<PROB_SIGMOID, false, true>), dgrid, block, shmem, s, a);
        return check_launch();
    }
    if (!g_force_scalar && dense && !labels && what == SEG_STATS && vec && HW % 1024 == 0 && (prob == PROB_SIGMOID || prob == PROB_IDENTITY)) {
        const bool ign = flags & SEG_HAS_IGNORE;
        const dim3 dgrid(grid_for_groups(HW / 1024 * C * B, kGridStream));
        if (prob == PROB_SIGMOID) { if (ign) hipLaunchKernelGGL((seg_stats_dense_lean_kernel<PROB_SIGMOID, true>), dgrid, block, shmem, s, a);
                                    else hipLaunchKernelGGL((seg_stats_dense_lean_kernel<PROB_SIGMOID, false>), dgrid, block, shmem, s, a); }
        else { if (ign) hipLaunchKernelGGL((seg_stats_dense_lean_kernel<PROB_IDENTITY, true>), dgrid, block, shmem, s, a);
               else hipLaunchKernelGGL((seg_stats_dense_lean_kernel<PROB_IDENTITY, false>), dgrid, block, shmem, s, a); }
        return check_launch();
    }
    // straight-line kernels for the common case (see seg_fwd_lean_kernel)
    if (!g_force_scalar && labels && !dense && vec && HW % 256 == 0 && C <= 16 && (what & SEG_STATS) && !(flags & SEG_ELEMWISE) &&
        (prob == PROB_SOFTMAX || prob == PROB_IDENTITY)) {
        const bool ign = flags & SEG_HAS_IGNORE;
        const bool plain_focal = what == (SEG_FOCAL | SEG_STATS) && prob == PROB_SOFTMAX && g2 && !class_weights &&
                                 !(flags & (SEG_HAS_IGNORE | SEG_HAS_ALPHA | SEG_REDUCED));
        const dim3 lgrid(grid_for_groups(HW / 256 * B, kGridStats));
        const bool no_term = flags & SEG_NO_TERM;
#define PTB_LEAN(CR) do { \
            if (plain_focal) { const dim3 g2(grid_for_groups(HW / 128 * B, kGridStats)); \
                               const size_t pk_lds = std::max(shmem, (size_t)4 * 2 * CR * 64 * sizeof(float)); \
                               const dim3 gpk(grid_for_groups(HW / 128 * B, g_focal_pk_grid)); \
                               if (g_fused_pix2 && g_focal_pk == 2 && no_term && C == CR) hipLaunchKernelGGL((seg_focal_pk_kernel<CR, false, true, true>), gpk, block, pk_lds, s, a); \
                               else if (g_fused_pix2 && g_focal_pk && no_term && C == CR) hipLaunchKernelGGL((seg_focal_pk_kernel<CR, false, true>), gpk, block, pk_lds, s, a); \
                               else if (g_fused_pix2 && g_focal_pk && no_term) hipLaunchKernelGGL((seg_focal_pk_kernel<CR, false, false>), gpk, block, pk_lds, s, a); \
                               else if (g_fused_pix2 && g_focal_pk) hipLaunchKernelGGL((seg_focal_pk_kernel<CR, true, false>), gpk, block, pk_lds, s, a); \
                               else if (!g_fused_pix2) hipLaunchKernelGGL((seg_fwd_lean_kernel<CR, PROB_SOFTMAX, true, false, 4, true, false>), lgrid, block, shmem, s, a); \
                               else if (no_term && C == CR && g_loss_prefetch) hipLaunchKernelGGL((seg_fwd_lean_kernel<CR, PROB_SOFTMAX, true, false, 2, false, true, true>), g2, block, shmem, s, a); \
                               else if (no_term && C == CR) hipLaunchKernelGGL((seg_fwd_lean_kernel<CR, PROB_SOFTMAX, true, false, 2, false, true>), g2, block, shmem, s, a); \
                               else if (no_term) hipLaunchKernelGGL((seg_fwd_lean_kernel<CR, PROB_SOFTMAX, true, false, 2, false, false>), g2, block, shmem, s, a); \
                               else hipLaunchKernelGGL((seg_fwd_lean_kernel<CR, PROB_SOFTMAX, true, false, 2, true, false>), g2, block, shmem, s, a); } \
            else if (prob == PROB_SOFTMAX && !ign && g_stats_pk && HW % 128 == 0) { \
                               const size_t pk_lds = std::max(shmem, (size_t)4 * 2 * CR * 64 * sizeof(float)); \
                               const dim3 gpk(grid_for_groups(HW / 128 * B, g_focal_pk_grid)); \
                               if (C == CR) hipLaunchKernelGGL((seg_focal_pk_kernel<CR, false, true, false, false>), gpk, block, pk_lds, s, a); \
                               else hipLaunchKernelGGL((seg_focal_pk_kernel<CR, false, false, false, false>), gpk, block, pk_lds, s, a); } \
            else if (prob == PROB_SOFTMAX) { if (ign) hipLaunchKernelGGL((seg_fwd_lean_kernel<CR, PROB_SOFTMAX, false, true>), lgrid, block, shmem, s, a); \
                                             else hipLaunchKernelGGL((seg_fwd_lean_kernel<CR, PROB_SOFTMAX, false, false>), lgrid, block, shmem, s, a); } \
            else { if (ign) hipLaunchKernelGGL((seg_fwd_lean_kernel<CR, PROB_IDENTITY, false, true>), lgrid, block, shmem, s, a); \
                   else hipLaunchKernelGGL((seg_fwd_lean_kernel<CR, PROB_IDENTITY, false, false>), lgrid, block, shmem, s, a); } } while (0)
        if (what == SEG_STATS || plain_focal) {
            if (C <= 4) PTB_LEAN(4); else if (C <= 8) PTB_LEAN(8); else PTB_LEAN(16);
            return check_launch();
        }
#undef PTB_LEAN
    }
    if (focal_only_pk(a)) {
        // BinaryFocalLoss() on label maps in its default configuration: the packed streaming kernel of the fused loss without its
        // statistics half (the shared exponent shift max_c x keeps one exp per element for sigmoid and its complement)
        const bool term = !(flags & SEG_NO_TERM);
        const dim3 gpk(grid_for_groups(HW / 128 * B, g_focal_pk_grid));
#define PTB_FPK(CR) do { const size_t pk_lds = std::max(shmem, (size_t)4 * 2 * CR * 64 * sizeof(float)); \
                         if (term) { if (C == CR) hipLaunchKernelGGL((seg_focal_pk_kernel<CR, true, true, false, true, false>), gpk, block, pk_lds, s, a); \
                                     else hipLaunchKernelGGL((seg_focal_pk_kernel<CR, true, false, false, true, false>), gpk, block, pk_lds, s, a); } \
                         else { if (C == CR) hipLaunchKernelGGL((seg_focal_pk_kernel<CR, false, true, false, true, false>), gpk, block, pk_lds, s, a); \
                                else hipLaunchKernelGGL((seg_focal_pk_kernel<CR, false, false, false, true, false>), gpk, block, pk_lds, s, a); } } while (0)
        if (C <= 4) PTB_FPK(4); else if (C <= 8) PTB_FPK(8); else PTB_FPK(16);
#undef PTB_FPK
        return check_launch();
    }
    if (what == SEG_FOCAL && !g_force_scalar && labels && !dense && vec && HW % 256 == 0 && g2 && !class_weights &&
        !(flags & (SEG_HAS_ALPHA | SEG_REDUCED | SEG_ELEMWISE))) {
        const bool ign = flags & SEG_HAS_IGNORE;
        const dim3 lgrid(grid_for_groups(HW / 256 * B, kGridStream));
#define PTB_FL(CH) do { if (ign) hipLaunchKernelGGL((focal_fwd_lean_kernel<CH, true>), lgrid, block, 0, s, a); \
                        else hipLaunchKernelGGL((focal_fwd_lean_kernel<CH, false>), lgrid, block, 0, s, a); } while (0)
        if (C <= 4) PTB_FL(4); else PTB_FL(8);   // 16 at once (139 VGPRs, 3 waves / SIMD) measured 119 us vs 108 us for 2 x 8 at C = 16
#undef PTB_FL
        return check_launch();
    }
    if (what == SEG_FOCAL) {
#define PTB_FF(P, D) do { if (g2) hipLaunchKernelGGL((focal_fwd_kernel<P, D, true>), grid, block, 0, s, a); \
                          else hipLaunchKernelGGL((focal_fwd_kernel<P, D, false>), grid, block, 0, s, a); } while (0)
        if (vec) { if (labels) PTB_FF(4, false); else PTB_FF(4, true); }
        else { if (labels) PTB_FF(1, false); else PTB_FF(1, true); }
#undef PTB_FF
    } else if (vec && C <= 16) {
#define PTB_FWD(W, D) do { if (g2) hipLaunchKernelGGL((seg_loss_fwd_reg_kernel<4, 16, W, D, true>), grid, block, shmem, s, a); \
                           else hipLaunchKernelGGL((seg_loss_fwd_reg_kernel<4, 16, W, D, false>), grid, block, shmem, s, a); } while (0)
        const bool plain = g2 && !class_weights && !(flags & (SEG_HAS_IGNORE | SEG_HAS_ALPHA | SEG_REDUCED));
        if (labels && what != SEG_STATS && prob == PROB_SOFTMAX && !(flags & SEG_ELEMWISE)) {
            if (plain) hipLaunchKernelGGL((seg_loss_fwd_reg_kernel<4, 16, 3, false, true, true, true>), grid, block, shmem, s, a);
            else if (g2) hipLaunchKernelGGL((seg_loss_fwd_reg_kernel<4, 16, 3, false, true, true>), grid, block, shmem, s, a);
            else hipLaunchKernelGGL((seg_loss_fwd_reg_kernel<4, 16, 3, false, false, true>), grid, block, shmem, s, a);
        } else if (labels) { if (what == SEG_STATS) hipLaunchKernelGGL((seg_loss_fwd_reg_kernel<4, 16, 2, false, true>), grid, block, shmem, s, a); else PTB_FWD(3, false); }
        else { if (what == SEG_STATS) hipLaunchKernelGGL((seg_loss_fwd_reg_kernel<4, 16, 2, true, true>), grid, block, shmem, s, a); else PTB_FWD(3, true); }
#undef PTB_FWD
    } else if (vec) {
        hipLaunchKernelGGL((seg_loss_fwd_kernel<4>), grid, block, shmem, s, a);
    } else {
        hipLaunchKernelGGL((seg_loss_fwd_kernel<1>), grid, block, shmem, s, a);
    }
    return check_launch();
}

extern "C" int ptb_focal_bwd(const float* logits, const int64_t* labels, const float* dense, const float* class_weights,
                             const float* coef, const float* grad_elem, float* grad, int B, int C, int64_t HW, int flags,
                             float gamma, float alpha, float threshold, int64_t ignore_label, float ignore_value,
                             ptb_stream_t stream) {
    SegArgs a{};
    if (int rc = fill_seg(a, logits, labels, dense, class_weights, B, C, HW, flags, PROB_SIGMOID, gamma, alpha, threshold, ignore_label, ignore_value)) return rc;
    if (!coef || !grad) return PTB_EINVAL;
    if ((long long)B * HW == 0) return PTB_OK;
    hipStream_t s = (hipStream_t)stream;
    const bool vec = vec_ok(HW, {logits, dense, grad_elem, grad, labels});
    const dim3 grid(vec ? grid_for_groups((HW + 255) / 256 * B, kGridStream) : grid_for_groups((HW + 63) / 64 * B, kGridStream)), block(256);
    const bool g2 = gamma == 2.0f;
#define PTB_FBWD(P, D, G) do { if (g2) hipLaunchKernelGGL((focal_bwd_kernel<P, D, G, true>), grid, block, 0, s, a, coef, grad_elem, grad); \
                               else hipLaunchKernelGGL((focal_bwd_kernel<P, D, G, false>), grid, block, 0, s, a, coef, grad_elem, grad); } while (0)
    if (vec) {
        if (labels) { if (grad_elem) PTB_FBWD(4, false, true); else PTB_FBWD(4, false, false); }
        else { if (grad_elem) PTB_FBWD(4, true, true); else PTB_FBWD(4, true, false); }
    } else {
        if (labels) { if (grad_elem) PTB_FBWD(1, false, true); else PTB_FBWD(1, false, false); }
        else { if (grad_elem) PTB_FBWD(1, true, true); else PTB_FBWD(1, true, false); }
    }
#undef PTB_FBWD
    return check_launch();
}

extern "C" int ptb_seg_stats_bwd(const float* logits, const int64_t* labels, const float* dense, const float* gI, const float* gP,
                                 float* grad, int B, int C, int64_t HW, int flags, int prob, int64_t ignore_label,
                                 float ignore_value, ptb_stream_t stream) {
    SegArgs a{};
    if (int rc = fill_seg(a, logits, labels, dense, nullptr, B, C, HW, flags, prob, 0.f, 0.f, 0.f, ignore_label, ignore_value)) return rc;
    if (!gI || !gP || !grad) return PTB_EINVAL;
    if ((long long)B * HW == 0) return PTB_OK;
    hipStream_t s = (hipStream_t)stream;
    if (!g_force_scalar && dense && !labels && HW % 1024 == 0 && (prob == PROB_SIGMOID || prob == PROB_IDENTITY) && vec_ok(HW, {logits, dense, grad})) {
        const dim3 dgrid(grid_for_groups(HW / 1024 * C * B, kGridStream)), block(256);
        const bool ign = flags & SEG_HAS_IGNORE;
        if (prob == PROB_SIGMOID) { if (ign) hipLaunchKernelGGL((seg_dense_bwd_lean_kernel<PROB_SIGMOID, true, false>), dgrid, block, 0, s, a, gI, gI, gP, grad);
                                    else hipLaunchKernelGGL((seg_dense_bwd_lean_kernel<PROB_SIGMOID, false, false>), dgrid, block, 0, s, a, gI, gI, gP, grad); }
        else { if (ign) hipLaunchKernelGGL((seg_dense_bwd_lean_kernel<PROB_IDENTITY, true, false>), dgrid, block, 0, s, a, gI, gI, gP, grad);
               else hipLaunchKernelGGL((seg_dense_bwd_lean_kernel<PROB_IDENTITY, false, false>), dgrid, block, 0, s, a, gI, gI, gP, grad); }
        return check_launch();
    }
    if (vec_ok(HW, {logits, dense, grad, labels})) {
        const int grid = grid_for_groups((HW + 255) / 256 * B, kGridStats);
        if (C <= 16 && labels) hipLaunchKernelGGL((seg_stats_bwd_kernel<4, 16, false>), dim3(grid), dim3(256), 0, s, a, gI, gP, grad);
        else hipLaunchKernelGGL((seg_stats_bwd_kernel<4, 0, false>), dim3(grid), dim3(256), 0, s, a, gI, gP, grad);
    } else {
        hipLaunchKernelGGL((seg_stats_bwd_kernel<1, 0, false>), dim3(grid_for_groups((HW + 63) / 64 * B, kGridStats)), dim3(256), 0, s, a, gI, gP, grad);
    }
    return check_launch();
}

template <int MODE>
static int launch_smf(const SmfArgs& a, const float* coef, const float* grad_pix, float* grad, hipStream_t s) {
    // forward only: the straight-line backward needs 168 VGPRs + spills for its two passes over 64 elements and measured 332 us
    // against 275 us for the kernel below
    if (MODE == 0 && !g_force_scalar && a.HW % 256 == 0 && a.C <= 16 && a.gamma == 2.0f && !a.class_weights && !a.reduced &&
        vec_ok(a.HW, {a.logits, a.pixel_out, grad_pix, grad, a.labels})) {
        const dim3 lgrid(grid_for_groups(a.HW / 256 * a.B, kGridStream)), block(256);
        if (a.C <= 4) hipLaunchKernelGGL((softmax_focal_lean_kernel<4, 0>), lgrid, block, 0, s, a, coef, grad_pix, grad);
        else if (a.C <= 8) hipLaunchKernelGGL((softmax_focal_lean_kernel<8, 0>), lgrid, block, 0, s, a, coef, grad_pix, grad);
        else hipLaunchKernelGGL((softmax_focal_lean_kernel<16, 0>), lgrid, block, 0, s, a, coef, grad_pix, grad);
        return check_launch();
    }
    if (vec_ok(a.HW, {a.logits, a.pixel_out, grad_pix, grad, a.labels})) {
        const int grid = grid_for_groups((a.HW + 255) / 256 * a.B, kGridStream);
        // (2 pixels per lane for the backward: 151 -> fewer VGPRs but the same 442 us forward + backward at cfg4: not used)
        if (MODE == 1 && a.C <= 16 && !g_force_scalar && g_smf_bwd_stash) {   // one transcendental pass, T_c kept in registers
            const int pix = g_smf_bwd_stash;
            const int g2 = grid_for_groups((a.HW + 64 * pix - 1) / (64 * pix) * a.B, kGridStream);
#define PTB_SMF_BWD(P, G, F) hipLaunchKernelGGL((softmax_focal_bwd_kernel<P, 16, G, F>), dim3(g2), dim3(256), 0, s, a, coef, grad_pix, grad)
            const bool gm2 = a.gamma == 2.0f;
            // (FULL = true, all 16 loads unconditional, lets the scheduler hoist everything: 256 VGPRs + scratch against 184 -- not used)
            if (pix == 2) { if (gm2) PTB_SMF_BWD(2, true, false); else PTB_SMF_BWD(2, false, false); }
            else { if (gm2) PTB_SMF_BWD(4, true, false); else PTB_SMF_BWD(4, false, false); }
#undef PTB_SMF_BWD
            return check_launch();
        }
        if (a.C <= 16 && a.gamma == 2.0f) hipLaunchKernelGGL((softmax_focal_kernel<4, 16, MODE, true>), dim3(grid), dim3(256), 0, s, a, coef, grad_pix, grad);
        else if (a.C <= 16) hipLaunchKernelGGL((softmax_focal_kernel<4, 16, MODE, false>), dim3(grid), dim3(256), 0, s, a, coef, grad_pix, grad);
        else hipLaunchKernelGGL((softmax_focal_kernel<4, 0, MODE, false>), dim3(grid), dim3(256), 0, s, a, coef, grad_pix, grad);
    } else {
        hipLaunchKernelGGL((softmax_focal_kernel<1, 0, MODE, false>), dim3(grid_for_groups((a.HW + 63) / 64 * a.B, kGridStream)), dim3(256), 0, s, a, coef, grad_pix, grad);
    }
    return check_launch();
}

extern "C" int ptb_softmax_focal_fwd(const float* logits, const int64_t* labels, const float* class_weights, double* sums,
                                     float* pixel_out, int* error_flag, int B, int C, int64_t HW, int reduced, float gamma,
                                     float threshold, int64_t ignore_label, ptb_stream_t stream) {
    if (!logits || !labels || !sums || !error_flag || B < 0 || C < 1 || HW < 0) return PTB_EINVAL;
    if (int rc = zero_sums(sums, 2, error_flag, (hipStream_t)stream)) return rc;
    if ((long long)B * HW == 0) return PTB_OK;
    SmfArgs a{logits, (const long long*)labels, class_weights, sums, pixel_out, error_flag, B, C, HW, reduced, gamma, threshold, ignore_label};
    return launch_smf<0>(a, nullptr, nullptr, nullptr, (hipStream_t)stream);
}

extern "C" int ptb_softmax_focal_bwd(const float* logits, const int64_t* labels, const float* class_weights, const float* coef,
                                     const float* grad_pix, float* grad, int B, int C, int64_t HW, int reduced, float gamma,
                                     float threshold, int64_t ignore_label, ptb_stream_t stream) {
    if (!logits || !labels || !coef || !grad || B < 0 || C < 1 || HW < 0) return PTB_EINVAL;
    if ((long long)B * HW == 0) return PTB_OK;
    SmfArgs a{logits, (const long long*)labels, class_weights, nullptr, nullptr, nullptr, B, C, HW, reduced, gamma, threshold, ignore_label};
    return launch_smf<1>(a, coef, grad_pix, grad, (hipStream_t)stream);
}

// Fused backward of ptb_seg_loss_fwd(PTB_SEG_FOCAL | PTB_SEG_STATS): returns PTB_EUNSUPPORTED when the fused kernel does
// not apply (C > 16, unaligned, dense targets with softmax), in which case the caller uses the two separate kernels.
extern "C" int ptb_seg_fused_bwd(const float* logits, const int64_t* labels, const float* dense, const float* class_weights,
                                 const float* coef, const float* gI, const float* gP, float* grad, int B, int C, int64_t HW, int flags,
                                 int prob, float gamma, float alpha, float threshold, int64_t ignore_label, float ignore_value,
                                 ptb_stream_t stream) {
    SegArgs a{};
    if (int rc = fill_seg(a, logits, labels, dense, class_weights, B, C, HW, flags, prob, gamma, alpha, threshold, ignore_label, ignore_value)) return rc;
    if (!coef || !gI || !gP || !grad) return PTB_EINVAL;
    if ((long long)B * HW == 0) return PTB_OK;
    if (!g_force_scalar && dense && !labels && HW % 1024 == 0 && prob == PROB_SIGMOID && gamma == 2.0f && !class_weights &&
        !(flags & (SEG_HAS_ALPHA | SEG_REDUCED)) && vec_ok(HW, {logits, dense, grad})) {
        const dim3 dgrid(grid_for_groups(HW / 1024 * C * B, kGridStream)), dblock(256);
        if (flags & SEG_HAS_IGNORE) hipLaunchKernelGGL((seg_dense_bwd_lean_kernel<PROB_SIGMOID, true, true>), dgrid, dblock, 0, (hipStream_t)stream, a, coef, gI, gP, grad);
        else hipLaunchKernelGGL((seg_dense_bwd_lean_kernel<PROB_SIGMOID, false, true>), dgrid, dblock, 0, (hipStream_t)stream, a, coef, gI, gP, grad);
        return check_launch();
    }
    if (C > 16 || !vec_ok(HW, {logits, dense, grad, labels}) || (dense && prob == PROB_SOFTMAX)) return PTB_EUNSUPPORTED;
    const dim3 grid(grid_for_groups((HW + 255) / 256 * B, kGridStats)), block(256);
    hipStream_t s = (hipStream_t)stream;
    const bool g2 = gamma == 2.0f;
#define PTB_FUSED(D) do { if (g2) hipLaunchKernelGGL((seg_fused_bwd_kernel<4, 16, D, true>), grid, block, 0, s, a, coef, gI, gP, grad); \
                          else hipLaunchKernelGGL((seg_fused_bwd_kernel<4, 16, D, false>), grid, block, 0, s, a, coef, gI, gP, grad); } while (0)
    if (labels && prob == PROB_SOFTMAX) {
        const bool plain = g2 && !class_weights && !(flags & (SEG_HAS_IGNORE | SEG_HAS_ALPHA | SEG_REDUCED));
        if (plain && !g_force_scalar && HW % 256 == 0) {
            if (g_nt_grad_stores) a.flags |= SEG_NT_STORES;
            const dim3 lgrid(grid_for_groups(HW / 256 * B, kGridStats));
            if (C <= 4) hipLaunchKernelGGL((seg_fused_bwd_lean_kernel<4>), lgrid, block, 0, s, a, coef, gI, gP, grad);
            else if (C <= 8) hipLaunchKernelGGL((seg_fused_bwd_lean_kernel<8>), lgrid, block, 0, s, a, coef, gI, gP, grad);
            else hipLaunchKernelGGL((seg_fused_bwd_lean_kernel<16>), lgrid, block, 0, s, a, coef, gI, gP, grad);
        } else if (plain) hipLaunchKernelGGL((seg_fused_bwd_shared_kernel<4, 16, true, true>), grid, block, 0, s, a, coef, gI, gP, grad);
        else if (g2) hipLaunchKernelGGL((seg_fused_bwd_shared_kernel<4, 16, true>), grid, block, 0, s, a, coef, gI, gP, grad);
        else hipLaunchKernelGGL((seg_fused_bwd_shared_kernel<4, 16, false>), grid, block, 0, s, a, coef, gI, gP, grad);
    } else if (labels) PTB_FUSED(false); else PTB_FUSED(true);
#undef PTB_FUSED
    return check_launch();
}

extern "C" int ptb_region_epilogue(const double* sums, int slots, int C, float focal_scale, float dice_weight, float jaccard_weight,
                                   float smooth, float eps, int log_loss, const uint8_t* class_mask, int n_selected, float* loss,
                                   float* coef, const int* error_flag, ptb_stream_t stream) {
    if (!sums || !loss || !coef || slots < 1 || C < 1 || n_selected < 1 || n_selected > C) return PTB_EINVAL;
    EpiArgs a{sums, slots, C, focal_scale, dice_weight, jaccard_weight, smooth, eps, log_loss, class_mask, n_selected, loss, coef, error_flag};
    hipLaunchKernelGGL(region_epilogue_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch();
}
