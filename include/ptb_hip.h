/*
 * ptb_hip.h -- C ABI of libptb_hip.so: hand-written gfx950 (MI355X / CDNA4) HIP kernels for the
 * pytorch-toolbelt large-image inference hot path (tile merge, TTA de-augment/augment, loss reductions).
 *
 * The reference (BloodAxe/pytorch-toolbelt) is pure Python: it has no FFI / operator registry, so the
 * drop-in boundary is its Python module surface.  Each entry point below replaces the torch
 * op chain of the cited reference function; `pytorch_toolbelt_amd` binds them with ctypes (see INTEGRATION.md
 * for the stub a reference maintainer would add).
 *
 * Conventions
 *   - plain pointers + sizes only; every tensor is contiguous row-major ("NCHW"); device pointers unless marked HOST.
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*); nothing synchronises, nothing allocates.
 *   - returns 0 on success, a negative PTB_E* code otherwise; never throws, never aborts.
 *   - float = IEEE fp32.  Arithmetic is not contracted (no FMA fusion) where the reference's result is
 *     reproduced bit-for-bit (tile accumulation, merge division).
 */
#ifndef PTB_HIP_H
#define PTB_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PTB_OK 0
#define PTB_EINVAL (-1)      /* bad argument (null pointer, non-positive size, unknown enum) */
#define PTB_EUNSUPPORTED (-2) /* legal for the reference but not implemented natively (caller must raise) */
#define PTB_ELAUNCH (-3)     /* HIP launch failed; see ptb_last_hip_error() */
#define PTB_EBOUNDS (-4)     /* tile rectangle leaves the accumulator */

typedef void* ptb_stream_t; /* hipStream_t */

/* View transform code: bit0 = transpose, bit1 = flip source rows, bit2 = flip source cols.
 *   out[i][j] = src[r][c], (r,c) = (j,i) if transpose else (i,j); r -> rows-1-r if bit1; c -> cols-1-c if bit2.
 * The 8 codes are the dihedral group D4 (inference/functional.py:47-132 of the reference). */
#define PTB_VIEW_IDENT 0
#define PTB_VIEW_TRANSPOSE 1
#define PTB_VIEW_FLIPUD 2
#define PTB_VIEW_ROT90_CW 3      /* x[N-1-j][i]   */
#define PTB_VIEW_FLIPLR 4
#define PTB_VIEW_ROT90_CCW 5     /* x[j][N-1-i]   */
#define PTB_VIEW_ROT180 6
#define PTB_VIEW_ANTITRANSPOSE 7 /* x[N-1-j][N-1-i] */

/* Reductions of _deaugment_averaging (inference/tta.py:63-96, inference/functional.py:250-333). */
#define PTB_RED_SUM 0
#define PTB_RED_MEAN 1
#define PTB_RED_GMEAN 2
#define PTB_RED_HMEAN 3
#define PTB_RED_HARMONIC1P 4
#define PTB_RED_LOGODD 5
#define PTB_RED_LOG1P 6

int ptb_version(void);
/* hipGetErrorString of the last failing HIP call made by this library on this thread ("" if none). */
const char* ptb_last_hip_error(void);
/* Tuning knob for benchmarks/tests: key 0 = chunk rows of the view kernels (16|32|64), 1 = force scalar kernels (0|1),
 * 2 = non-temporal streaming loads in the view kernels (0|1, default 1). */
int ptb_set_tunable(int key, int value);

/* ---- TileMerger.integrate_batch / accumulate_single (inference/tiles.py:310-339) -------------------------------
 * for b in 0..B-1 (in order):  image[:, y:y+th, x:x+tw] += tiles[b] * weight ;  norm[0, y:y+th, x:x+tw] += weight
 * image [C,H,W], norm [H,W], weight [th,tw], tiles [B,C,th,tw]; xs/ys HOST int64[B] (top-left corner of each tile).
 * Overlapping tiles of one batch are handled race-free and in batch order: bit-identical to the sequential loop. */
int ptb_tile_accumulate(float* image, float* norm, const float* weight, const float* tiles, const int64_t* xs,
                        const int64_t* ys, int B, int C, int th, int tw, int H, int W, ptb_stream_t stream);

/* ---- TileMerger.merge / merge_ (inference/tiles.py:345-350): out[c] = image[c] / norm (no eps clamp) -----------
 * out may alias image (merge_). */
int ptb_merge_div(const float* image, const float* norm, float* out, int C, int64_t HW, ptb_stream_t stream);
/* Band variant used by the multi-GPU merger (no reference counterpart; must equal the single-device merge):
 * out[c][p] = (image[c][p] + (p < extra_n ? extra[c][p] : 0)) / norm[p] for p in [0,HW), with explicit channel strides
 * (elements) so a row band of a larger accumulator can be merged; extra = the halo strip received from the
 * neighbouring rank (may be NULL with extra_n = 0). */
int ptb_merge_div_ex(const float* image, const float* norm, float* out, int C, int64_t HW, int64_t image_cs, int64_t out_cs,
                     const float* extra, int64_t extra_cs, int64_t extra_n, ptb_stream_t stream);

/* ---- {fliplr,flipud,flips,d2,d4}_image_deaugment (inference/tta.py:287-316,344-365,442-467,503-524) -------------
 * in [V*B, C, H, W] (chunk-major: rows [k*B,(k+1)*B) are view k), views HOST int[V] = inverse transform of each chunk.
 * out [B, C, H, W] = reduce_k view_k(in[k*B + b]).  V <= 8.  Transposing views require H == W. */
int ptb_deaug_reduce(const float* in, float* out, int V, const int* views, int reduction, int B, int C, int H, int W,
                     ptb_stream_t stream);

/* ---- per-view transform without reduction ----------------------------------------------------------------------
 * out[k*B + b] = scale * view_k(in[src]) with src = b (in_is_batch = 1: *_image_augment, inference/tta.py:257-284,
 * 319-341,385-422,470-484; in [B,C,H,W]) or src = k*B + b (in_is_batch = 0: de-augment with reduction=None).
 * out [V*B, C, Ho, Wo] where (Ho,Wo) = (W,H) for transposing views (so H == W is required for them). */
int ptb_view_transform(const float* in, float* out, int V, const int* views, int in_is_batch, float scale, int B, int C,
                       int H, int W, ptb_stream_t stream);

/* ---- fused: de-augment + reduce + TileMerger.integrate_batch (tta.py:442-467 feeding tiles.py:321-339) ----------
 * image[:, y:y+th, x:x+tw] += reduce_k view_k(in[k*B+b]) * weight ; norm += weight.  The reduced tile never goes
 * to HBM.  Same tensors as ptb_deaug_reduce (H=th, W=tw) and ptb_tile_accumulate. */
int ptb_deaug_accumulate(float* image, float* norm, const float* weight, const float* in, int V, const int* views,
                         int reduction, const int64_t* xs, const int64_t* ys, int B, int C, int th, int tw, int H,
                         int W, ptb_stream_t stream);

/* ---- bilinear resize of ms_image_augment / ms_image_deaugment (inference/tta.py:599-621, 645-689) ----------------
 * = torch.nn.functional.interpolate(in, size=(hout,wout), mode="bilinear", align_corners=...) on `planes` = B*C maps. */
int ptb_resize_bilinear(const float* in, float* out, int64_t planes, int hin, int win, int hout, int wout,
                        int align_corners, ptb_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PTB_HIP_H */
