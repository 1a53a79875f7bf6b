// ptb_focal_softmax.hip -- focal_loss_with_logits(activation="softmax", softmax_dim=...) for gfx950 (MI355X).
//
// Reference: losses/functional.py:61-107 with `p = torch.softmax(output, dim=softmax_dim)` (:63-64): the BCE term still uses the
// logits (:66), only the focal term's probability is the softmax over one dimension, so an element's loss depends on the
// whole softmax fibre through p.  Layout: the tensor is viewed as [B, C, HW] with C the softmax dimension (any dimension of a
// contiguous tensor is such a view: B = product of the dimensions before it, HW = of those after it).  A wave owns 64 * PIX
// consecutive positions of one b; each lane walks the C planes three times at most -- (A) running maximum and sum of
// exponentials, (B) losses (forward) or the softmax-Jacobian dot product (backward), (C) gradients -- with 16-byte loads when
// HW % 4 == 0.  The planes of one b are 4 * HW bytes apart, so passes B and C mostly hit L2 / Infinity Cache for the usual
// segmentation shapes; it is a non-default option and a streaming kernel, not a tuned one (sigmoid focal: ptb_losses.hip).
//
// class_weights follow dim 1 of the ORIGINAL tensor (functional.py:83-88), which need not be the softmax dimension: cw_mode
// 1 = indexed by the softmax channel, 2 = by (b / cw_div) % cw_n, 3 = by (position / cw_div) % cw_n.
#include "ptb_loss_device.h"

namespace ptb {

struct FsmArgs {
    SegArgs s;
    int cw_mode, cw_n;
    long long cw_div;
};

__device__ __forceinline__ float fsm_weight(const FsmArgs& a, int b, int c, long long pos) {
    if (a.cw_mode == 0) return 1.f;
    if (a.cw_mode == 1) return a.s.class_weights[c];
    if (a.cw_mode == 2) return a.s.class_weights[(b / a.cw_div) % a.cw_n];
    return a.s.class_weights[(pos / a.cw_div) % a.cw_n];
}

// element pieces: BCE with logits, focal term f, d f / d p (p = softmax probability), sigmoid(x)
template <bool G2>
__device__ __forceinline__ void fsm_parts(float x, float p, float t, const FocalCfg& c, float& ce, float& f, float& dfdp, float& sg) {
    const Sig s = sigmoid_parts(x);
    sg = s.p;
    ce = fmaxf(x, 0.f) - x * t + s.log1pe;
    const float pt = p * t + (1.f - p) * (1.f - t);
    const float base = fmaxf(1.f - pt, 0.f) * c.sc;
    const bool below = pt < c.thr;
    float pw;                                    // base^(gamma - 1)
    if (G2) { f = base * base; pw = base; }
    else {
        f = c.g0 ? 1.0f : pow_pos(base, c.gamma);
        pw = c.g1 ? 1.0f : pow_pos(base, c.gm1);
    }
    f = below ? 1.0f : f;
    dfdp = -c.gamma * pw * c.sc * (2.f * t - 1.f);
    dfdp = (below || (!G2 && c.g0) || 1.f - pt <= 0.f) ? 0.f : dfdp;
}

// MODE 0: forward (sums[0] += loss, sums[1] += focal terms, optional unreduced map).  MODE 1: backward,
//   grad_j = p_j (H_j - sum_c H_c p_c) + G_j w_j f_j (sigmoid(x_j) - t_j),   H_c = (G_c w_c ce_c + K m_c) df_c/dp_c,
//   G_c = coef[0] * (grad_elem ? grad_elem_c : 1) on kept elements (0 on ignored ones), K = coef[1], m_c = 1 or the ignore mask
//   of the normalised focal term, w_c = class weight * alpha weight.
template <int PIX, int MODE, bool G2>
__global__ __launch_bounds__(256) void focal_softmax_kernel(const FsmArgs fa, const float* __restrict__ coef, const float* __restrict__ grad_elem,
                                                            float* __restrict__ grad) {
    const SegArgs& a = fa.s;
    const FocalCfg cfg = focal_cfg(a);
    const int lane = threadIdx.x & 63, wave = wave_id();
    const int C = a.C;
    const bool ignf = a.flags & SEG_HAS_IGNORE;
    const bool elem = a.flags & SEG_ELEMWISE;
    const bool dense = a.dense != nullptr;
    const float k1 = MODE ? coef[0] : 0.f, k2 = MODE ? coef[1] : 0.f;
    double f_loss = 0.0, f_term = 0.0;
    const long long per_img = (a.HW + 64 * PIX - 1) / (64 * PIX);
    const long long groups = per_img * a.B;
    for (long long g = (long long)blockIdx.x * 4 + wave; g < groups; g += (long long)gridDim.x * 4) {
        const Group<PIX> G = make_group<PIX>(g, per_img, lane, a.HW, dense ? nullptr : a.labels, ignf, a.ignore_label, C, MODE ? nullptr : a.error_flag);
        const long long base = (long long)G.b * C * a.HW + G.i0;
        float mx[PIX], den[PIX], S[PIX];
#pragma unroll
        for (int k = 0; k < PIX; ++k) { mx[k] = -INFINITY; den[k] = 0.f; S[k] = 0.f; }
        // pass A: running maximum and sum of exponentials over the softmax fibre
        for (int c = 0; c < C; ++c) {
            float xv[PIX];
#pragma unroll
            for (int k = 0; k < PIX; ++k) xv[k] = 0.f;
            load_px<PIX>(a.logits + base + (long long)c * a.HW, xv, G.ok);
#pragma unroll
            for (int k = 0; k < PIX; ++k) {
                const float m2 = fmaxf(mx[k], xv[k]);
                den[k] = den[k] * fexp(mx[k] - m2) + fexp(xv[k] - m2);
                mx[k] = m2;
            }
        }
#pragma unroll
        for (int k = 0; k < PIX; ++k) den[k] = rcp(den[k]);
        float lsum = 0.f, fsum = 0.f;
        // passes B (and C for the backward): per element pieces; `final` = write the gradient with the finished dot product S
        auto walk = [&](bool final) {
            for (int c = 0; c < C; ++c) {
                float xv[PIX], tv[PIX], gv[PIX], out[PIX];
#pragma unroll
                for (int k = 0; k < PIX; ++k) { xv[k] = 0.f; tv[k] = 0.f; gv[k] = 1.f; out[k] = 0.f; }
                const long long off = base + (long long)c * a.HW;
                load_px<PIX>(a.logits + off, xv, G.ok);
                if (dense) load_px<PIX>(a.dense + off, tv, G.ok);
                if (MODE && grad_elem) load_px<PIX>(grad_elem + off, gv, G.ok);
#pragma unroll
                for (int k = 0; k < PIX; ++k) {
                    if (!G.ok) continue;
                    float t;
                    bool ig = G.ign[k];
                    if (dense) { t = tv[k]; if (ignf && t == a.ignore_value) ig = true; }
                    else t = G.lab[k] == c ? 1.f : 0.f;
                    if (ig) t = 0.f;
                    const float p = fexp(xv[k] - mx[k]) * den[k];
                    float ce, f, dfdp, sg;
                    fsm_parts<G2>(xv[k], p, t, cfg, ce, f, dfdp, sg);
                    const float w = fsm_weight(fa, G.b, c, G.i0 + k) * (cfg.a1 * t + cfg.a0);
                    if (MODE == 0) {
                        const float l = ig ? 0.f : f * ce * w;
                        out[k] = l;
                        lsum += l;
                        fsum += ig ? f * cfg.term_mask : f;
                    } else {
                        const float Gc = ig ? 0.f : k1 * gv[k];
                        const float H = (Gc * w * ce + k2 * (ig ? cfg.term_mask : 1.f)) * dfdp;
                        if (!final) S[k] += H * p;
                        else out[k] = p * (H - S[k]) + Gc * w * f * (sg - t);
                    }
                }
                if (MODE == 0) { if (elem) store_px<PIX>(a.elem_out + off, out, G.ok); }
                else if (final) store_px_nt<PIX>(grad + off, out, G.ok);
            }
        };
        walk(false);
        if (MODE) walk(true);
        f_loss += (double)lsum;
        f_term += (double)fsum;
    }
    if (MODE == 0) block_add2(f_loss, f_term, a.sums + (size_t)(blockIdx.x % SUM_SLOTS) * 2, lane, wave);
}

}  // namespace ptb

using namespace ptb;

static int fsm_fill(FsmArgs& fa, const float* logits, const int64_t* labels, const float* dense, const float* class_weights, int cw_mode,
                    int cw_n, int64_t cw_div, int B, int C, int64_t HW, int flags, float gamma, float alpha, float threshold,
                    int64_t ignore_label, float ignore_value) {
    if (int rc = fill_seg(fa.s, logits, labels, dense, class_weights, B, C, HW, flags, PROB_SOFTMAX, gamma, alpha, threshold, ignore_label, ignore_value)) return rc;
    if (cw_mode < 0 || cw_mode > 3 || (cw_mode && (!class_weights || cw_n < 1 || cw_div < 1))) return PTB_EINVAL;
    if (cw_mode == 1 && cw_n != C) return PTB_EINVAL;
    if (!cw_mode) fa.s.class_weights = nullptr;
    fa.cw_mode = cw_mode; fa.cw_n = cw_n; fa.cw_div = cw_div;
    return PTB_OK;
}

extern "C" int ptb_focal_softmax_fwd(const float* logits, const int64_t* labels, const float* dense, const float* class_weights,
                                     int cw_mode, int cw_n, int64_t cw_div, double* sums, float* elem_out, int* error_flag, int B, int C,
                                     int64_t HW, int flags, float gamma, float alpha, float threshold, int64_t ignore_label,
                                     float ignore_value, ptb_stream_t stream) {
    FsmArgs fa{};
    if (int rc = fsm_fill(fa, logits, labels, dense, class_weights, cw_mode, cw_n, cw_div, B, C, HW, flags, gamma, alpha, threshold, ignore_label, ignore_value)) return rc;
    if (!sums || !error_flag || ((flags & SEG_ELEMWISE) && !elem_out)) return PTB_EINVAL;
    fa.s.sums = sums; fa.s.elem_out = elem_out; fa.s.error_flag = error_flag;
    hipStream_t s = (hipStream_t)stream;
    if (int rc = zero_sums(sums, 2, error_flag, s)) return rc;
    if ((long long)B * HW == 0) return PTB_OK;
    const bool vec = vec_ok(HW, {logits, dense, elem_out, labels});
    const dim3 grid(vec ? grid_for_groups((HW + 255) / 256 * B, kGridStream) : grid_for_groups((HW + 63) / 64 * B, kGridStream)), block(256);
    const bool g2 = gamma == 2.0f;
    if (vec) { if (g2) hipLaunchKernelGGL((focal_softmax_kernel<4, 0, true>), grid, block, 0, s, fa, nullptr, nullptr, nullptr);
               else hipLaunchKernelGGL((focal_softmax_kernel<4, 0, false>), grid, block, 0, s, fa, nullptr, nullptr, nullptr); }
    else { if (g2) hipLaunchKernelGGL((focal_softmax_kernel<1, 0, true>), grid, block, 0, s, fa, nullptr, nullptr, nullptr);
           else hipLaunchKernelGGL((focal_softmax_kernel<1, 0, false>), grid, block, 0, s, fa, nullptr, nullptr, nullptr); }
    return check_launch();
}

extern "C" int ptb_focal_softmax_bwd(const float* logits, const int64_t* labels, const float* dense, const float* class_weights,
                                     int cw_mode, int cw_n, int64_t cw_div, const float* coef, const float* grad_elem, float* grad, int B,
                                     int C, int64_t HW, int flags, float gamma, float alpha, float threshold, int64_t ignore_label,
                                     float ignore_value, ptb_stream_t stream) {
    FsmArgs fa{};
    if (int rc = fsm_fill(fa, logits, labels, dense, class_weights, cw_mode, cw_n, cw_div, B, C, HW, flags, gamma, alpha, threshold, ignore_label, ignore_value)) return rc;
    if (!coef || !grad) return PTB_EINVAL;
    if ((long long)B * HW == 0) return PTB_OK;
    hipStream_t s = (hipStream_t)stream;
    const bool vec = vec_ok(HW, {logits, dense, grad_elem, grad, labels});
    const dim3 grid(vec ? grid_for_groups((HW + 255) / 256 * B, kGridStream) : grid_for_groups((HW + 63) / 64 * B, kGridStream)), block(256);
    const bool g2 = gamma == 2.0f;
    if (vec) { if (g2) hipLaunchKernelGGL((focal_softmax_kernel<4, 1, true>), grid, block, 0, s, fa, coef, grad_elem, grad);
               else hipLaunchKernelGGL((focal_softmax_kernel<4, 1, false>), grid, block, 0, s, fa, coef, grad_elem, grad); }
    else { if (g2) hipLaunchKernelGGL((focal_softmax_kernel<1, 1, true>), grid, block, 0, s, fa, coef, grad_elem, grad);
           else hipLaunchKernelGGL((focal_softmax_kernel<1, 1, false>), grid, block, 0, s, fa, coef, grad_elem, grad); }
    return check_launch();
}
