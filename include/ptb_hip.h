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
#define PTB_EHELD (-6)       /* ptb_band_plan_submit_next: the batch lives in memory of a batch a later launch still reads */
#define PTB_EFRESH (-5)      /* first-touch bitmap cannot be honoured for this batch: zero-fill the fresh blocks, then call
                                again with fresh = NULL */

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

/* element type of the model outputs a `_t` entry point reads (accumulators and results are always fp32) */
#define PTB_F32 0
#define PTB_F16 1
#define PTB_BF16 2
/* or-ed into the `in_dtype` of ptb_deaug_accumulate_t / ptb_accumulate_planned(2) / ptb_merge_band / ptb_band_plan_submit(_rank): the
 * reduced value is rounded to the (fp16 / bf16) source type -- round to nearest even, NaN -> quiet NaN, like torch's `.to(dtype)` --
 * before it is multiplied by the window.  That is what the reference's two calls compute on half-precision model outputs
 * (torch.autocast): `tta.*_image_deaugment(y)` returns a HALF tensor (inference/tta.py:442-467), `integrate_batch` widens it
 * (inference/tiles.py:334-335).  No effect on PTB_F32 sources. */
#define PTB_ROUND_SRC 0x100

int ptb_version(void);
/* hipGetErrorString of the last failing HIP call made by this library on this thread ("" if none). */
const char* ptb_last_hip_error(void);
/* Tuning knob for benchmarks/tests: key 0 = chunk rows of the view kernels (16|32|64), 1 = force scalar kernels (0|1),
 * 2 = non-temporal streaming loads in the view kernels (0|1, default 1), 3 = LDS-staged multiscale kernel (0|1, default 1),
 * 4 = workgroups per loss-kernel launch (0 = per-kernel default), 5 = fused focal + statistics forward with 2 pixels per lane
 * (0|1, default 1), 6 = output tile rows of the fused multiscale kernel (16|32|64, default 32),
 * 7 = softmax focal backward that keeps the per-class terms in registers: pixels per lane (0 = off | 2 | 4, default 4),
 * 8 = the fused loss forward prefetches the next pixel group into a second register buffer (0|1, default 0: measured no gain),
 * 9 = XCD-aware tile order of the fused multiscale kernel: strip width in tile columns (0 = row-major order over all XCDs, default 64),
 * 10 = workgroup order of the band plan kernel (A/B): 0 = channels of a work item adjacent (default, fastest), 1 = every XCD a contiguous
 * eighth of the list, 2 = channel-major,
 * 11 = rows per work item of band plans created afterwards (32 | 64, default 64),
 * 12 = fused focal + Dice + Jaccard forward: 0 = lean kernel, 1 = packed-fp32 kernel (default), 2 = packed-fp32 with prefetch,
 * 13 = workgroups of the packed-fp32 forward (default 512), 15 = output tile width of the fused multiscale kernel (64 | 128, default 128),
 * 16 = non-temporal gradient stores in the fused loss backward (0|1, default 1), 17 = XCD-contiguous tile order of the Lovasz radix
 * scatter (0|1, default 1), 18 = one-launch finish of a rank's image in ptb_band_plan_finish_rank (0|1, default 1),
 * 19 = the gradient-binning scatter of the Lovasz training path also evaluates the loss (0|1, default 1; 0: separate lovasz_dot_kernel),
 * 20 = Dice / Jaccard statistics (logits + labels, no ignore) and the default BinaryFocalLoss on label maps run as the statistics-only /
 * focal-only instances of the packed streaming kernel of the fused loss (0|1, default 1; 0: the lean kernels),
 * 21 = the band plan kernel requests the next covering tile before it finishes the current one (0: never, 1: half / bf16 model outputs,
 *      2: fp32 as well; default 2),
 * 22 = (A/B, round 6) odd work items of the band plan kernel issue the loads of their views starting at view V / 2 instead of view 0
 *      (registers, reduction order and bits unchanged; default 0),
 * 23 = the last 8-bit level of the key-only Lovasz forward (ptb_lovasz_fwd_keys) evaluates the loss from every key's final rank and the
 *      foreground count in front of it -- both from the scanned histograms -- instead of scattering the keys a fourth time (0|1, default 1),
 * 25 = the prefetching instances of the band plan kernel (one 1024-thread workgroup per CU) alternate between two sets of LDS tiles for
 *      the transposing views: one barrier per covering tile instead of two (0|1, default 1),
 * 27 = band launches of the identity view (the plain loop without TTA) run one workgroup per work item over ALL channels: the window and
 *      the normaliser are loaded once per pixel instead of once per channel, every covering tile of a channel is requested at once (0|1,
 *      default 1).
 * Every setting computes the same values (key 19: bit for bit for segments of up to 2^24 elements -- above that the separate dot kernel's
 * (float)(i + 1) positions round and the two settings may differ in the last bits, the default being the reference's telescoping
 * difference); the keys exist for same-box A/B runs and for tests that compare two code paths bit for bit. */
int ptb_set_tunable(int key, int value);

/* ---- TileMerger.integrate_batch / accumulate_single (inference/tiles.py:310-339) -------------------------------
 * for b in 0..B-1 (in order):  image[:, y:y+th, x:x+tw] += tiles[b] * weight ;  norm[0, y:y+th, x:x+tw] += weight
 * image [C,H,W], norm [H,W], weight [th,tw], tiles [B,C,th,tw]; xs/ys HOST int64[B] (top-left corner of each tile).
 * Overlapping tiles of one batch are handled race-free and in batch order: bit-identical to the sequential loop.
 * First-touch stores (optional): `fresh` is a HOST bitmap owned by the caller, one byte per accumulator block of
 * 64 columns x fresh_rows rows (row-major, ceil(H/fresh_rows) x ceil(W/64)); 1 = the block was never written since the
 * accumulators were (logically) zeroed.  Cells made only of fresh blocks are written with plain stores -- the
 * accumulators then never need a memset and are not read on first touch -- and the library clears the bits it wrote.
 * NULL = plain read-modify-write everywhere.  PTB_EFRESH: see above.
 * norm may be NULL: the normaliser depends only on the crop list and the window, never on the predictions, so a caller
 * can keep the crop list and materialise it later (ptb_norm_accumulate) -- or reuse the one of the previous image. */
int ptb_tile_accumulate(float* image, float* norm, const float* weight, const float* tiles, const int64_t* xs,
                        const int64_t* ys, int B, int C, int th, int tw, int H, int W, uint8_t* fresh, int fresh_rows,
                        ptb_stream_t stream);

/* Planned accumulation (no reference counterpart; the result must equal integrate_batch(...) followed by merge()): the
 * caller knows the complete crop list of the image up front.  `remaining` / `done` are HOST byte maps on the block grid of
 * the first-touch bitmap (64 columns x fresh_rows rows): remaining[b] = planned tiles that have not yet touched block b
 * (initialised by the caller from the crop list), done[b] = the merged value of b has been written.  The call works like
 * ptb_deaug_accumulate (V = 1, views = {PTB_VIEW_IDENT}, PTB_RED_SUM for plain tiles) except that cells whose blocks
 * receive their last planned tile in this launch are written as (sum / norm_full) into `merged` [C,H,W] instead of into
 * `image` -- the separate merge pass disappears for them -- and no norm is accumulated (norm_full [H,W] = the complete,
 * data-independent normaliser in integration order).  The library updates both maps.  One launch group per call:
 * PTB_EUNSUPPORTED (nothing launched) if the batch needs several groups, is not block aligned, or does not fit the plan;
 * PTB_EFRESH as above.  Blocks with done == 0 at the end are merged by ptb_merge_div_masked. */
int ptb_accumulate_planned(float* image, const float* norm_full, float* merged, const float* weight, const void* in, int in_dtype,
                           int V, const int* views, int reduction, const int64_t* xs, const int64_t* ys, int B, int C, int th, int tw,
                           int H, int W, uint8_t* fresh, int fresh_rows, uint8_t* remaining, uint8_t* done, ptb_stream_t stream);
/* ptb_accumulate_planned2 = ptb_accumulate_planned + flags.  PTB_PLANNED_KEEP_SUMS (bit 0): a finalised block also stores its weighted
 * sum in `image` -- the accumulator stays complete and exact whenever it is read (TileMerger.image of a merger that planned itself). */
#define PTB_PLANNED_KEEP_SUMS 1
int ptb_accumulate_planned2(float* image, const float* norm_full, float* merged, const float* weight, const void* in, int in_dtype,
                           int V, const int* views, int reduction, const int64_t* xs, const int64_t* ys, int B, int C, int th, int tw,
                           int H, int W, uint8_t* fresh, int fresh_rows, uint8_t* remaining, uint8_t* done, int flags, ptb_stream_t stream);
/* out[c] = image[c] / norm on the blocks (64 columns x rows rows, row-major grid) whose byte in the DEVICE map `mask` is
 * non-zero; other blocks of `out` are left untouched. */
int ptb_merge_div_masked(const float* image, const float* norm, float* out, int C, int H, int W, const uint8_t* mask, int rows,
                         ptb_stream_t stream);

/* norm[0, y:y+th, x:x+tw] += weight for b in 0..B-1, in order (the norm_mask half of integrate_batch: same race-free cell
 * ownership, same summation order, same optional first-touch bitmap -- a SEPARATE bitmap from the image's). */
int ptb_norm_accumulate(float* norm, const float* weight, const int64_t* xs, const int64_t* ys, int B, int th, int tw, int H,
                        int W, uint8_t* fresh, int fresh_rows, ptb_stream_t stream);

/* Test hook, host only (no device work): the launch plan of the two calls above for one batch.  out: up to `cap`
 * records of 12 ints {launch group, ox, oy, w, h, fresh, chunk_end, ntiles, tile[4] (batch indices, -1 padded)}.
 * Returns the number of cells or a PTB_E* code. */
int ptb_debug_plan(const int64_t* xs, const int64_t* ys, int B, int th, int tw, int H, int W, int chunk_rows,
                   uint8_t* fresh, int fresh_rows, int* out, int cap);

/* ---- TileMerger.merge / merge_ (inference/tiles.py:345-350): out[c] = image[c] / norm (no eps clamp) -----------
 * out may alias image (merge_). */
int ptb_merge_div(const float* image, const float* norm, float* out, int C, int64_t HW, ptb_stream_t stream);
/* Band variant used by the multi-GPU merger (no reference counterpart; must equal the single-device merge):
 * out[c][p] = (image[c][p] + (p < extra_n ? extra[c][p] : 0)) / norm[p] for p in [0,HW), with explicit channel strides
 * (elements) so a row band of a larger accumulator can be merged; extra = the halo strip received from the
 * neighbouring rank (may be NULL with extra_n = 0). */
int ptb_merge_div_ex(const float* image, const float* norm, float* out, int C, int64_t HW, int64_t image_cs, int64_t out_cs,
                     const float* extra, int64_t extra_cs, int64_t extra_n, ptb_stream_t stream);

/* Deferred merge of one horizontal band (rows y0:y1 of the accumulator-sized map, lying between two consecutive tile edges) from
 * ALL the tiles that cover it: == the rows y0:y1 of `merge()` after `integrate_batch(<group>_image_deaugment(batch), crops)` of
 * those n tiles in the given order (inference/tiles.py:321-346 after inference/tta.py:442-467), without an accumulator in
 * HBM.  tile_src HOST array of n DEVICE pointers (view 0 of tile t, [C, th, tw] of in_dtype; view v lies tile_view_stride[t]
 * elements further -- the tiles may belong to different batch tensors), xs / ys HOST tile origins; every tile must cover
 * all rows of the band.  norm_full [H, W] = the complete normaliser, merged [C, H, W].  Returns PTB_EUNSUPPORTED when the
 * band does not fit the fast kernel (alignment, more than 48 tiles, more than 4 tiles over one pixel): the caller then
 * integrates those tiles through the incremental entry points. */
int ptb_merge_band(float* merged, const float* norm_full, const float* weight, const void* const* tile_src,
                   const int64_t* tile_view_stride, int in_dtype, int V, const int* views, int reduction, const int64_t* xs,
                   const int64_t* ys, int n, int C, int th, int tw, int H, int W, int y0, int y1, ptb_stream_t stream);

/* Diagnostic (no reference counterpart): streams `bytes` of device memory through a read-only kernel (16 B per lane, eight
 * non-temporal loads in flight, 8192 workgroups) so that a benchmark can time what this GPU's memory system gives a pure
 * read stream and quote its kernels against that as well as against the 8 TB/s spec.  `sink` = 4 writable device bytes. */
int ptb_read_probe(const void* buf, int64_t bytes, float* sink, ptb_stream_t stream);

/* The same read-only stream over `n` (<= 64) buffers in ONE launch (pointer table in the kernel arguments, a persistent grid of
 * `workgroups` (0: 8192) workgroups walking 32 KiB chunks of the concatenation), so that a pass over the per-batch model outputs
 * of an image pays ramp-up and tail once, like the band kernel it is compared with.  Returns the bytes the launch reads (whole
 * 32 KiB chunks of every buffer) or a negative error code. */
int64_t ptb_read_probe_multi(const void* const* bufs, const int64_t* bytes, int n, float* sink, int workgroups, ptb_stream_t stream);

/* out[i] = sum_s slot_sums[s][i] for the [nslots][n] slotted fp64 sums the loss entry points produce (they zero the slots and
 * the label flag themselves, on the launch stream: callers hand in uninitialised workspaces).  When error_flag (may be NULL) was
 * raised by the forward kernel -- a label outside [0, C) that is not ignore_index, where the reference's F.one_hot raises -- every
 * sum becomes NaN, so the condition can never produce a finite loss, even before the asynchronous host check reports it. */
int ptb_sums_finalize(const double* slot_sums, int nslots, int n, double* out, const int* error_flag, ptb_stream_t stream);

/* ---- Deferred merge of a whole image, planned once (TileMerger(crops=tiler.crops, defer=True)).
 * Replaces the reference's per-batch loop `merger.integrate_batch(<group>_image_deaugment(pred), crops)` + `merger.merge()`
 * (inference/tta.py:442-467, inference/tiles.py:321-346) when the crop list of the image is known up front.
 *
 * ptb_band_plan_create: HOST-side planning from the n tile origins (xs, ys; every tile th x tw, in integration order) of an
 *   H x W merged map with C channels.  The rows between consecutive tile edges are bands; consecutive bands are grouped into
 *   launches of about rows_per_launch rows (at most 224 covering tiles each).  Bands lying completely inside rows
 *   [final_lo, final_hi) are normalised (sum / norm); all others write the UN-normalised weighted sum (multi-GPU merger: rows
 *   whose sums are completed by, or belong to, a neighbouring rank; a single-GPU merger passes 0, H).  cuts [ncuts] (may be
 *   NULL / 0): additional row positions (multiples of 4) that are band edges and launch-group boundaries, so that a caller can
 *   have rows it must hand over early finished by their own launch.  Returns the size in bytes of the device table the plan needs
 *   (>= 0; *out receives the plan), PTB_EUNSUPPORTED when the geometry is off the 4-pixel grid / more than 4 tiles cover a pixel.
 * ptb_band_plan_upload: copies the work-item table into caller-provided device memory (64-byte aligned, the size create
 *   returned; the library never allocates device memory).  Once per plan.
 * ptb_band_plan_info: number of launch groups / bands / work items; last_group_of_tile [n] = the last group that reads tile t
 *   (the caller may release a batch once that group was launched); group_rows [3 * groups] = y0, y1 and the last tile (the
 *   one whose arrival completes it; -1: no tile covers those rows) of every group.
 * ptb_band_plan_submit: takes the B tiles pos .. pos+B-1 of the plan -- `batch` = view 0 of the first tile, tile b at
 *   b * tile_stride elements, view v of a tile view_stride elements further (chunk-major [V*B, C, th, tw] model output:
 *   tile_stride = C*th*tw, view_stride = B*C*th*tw) -- and launches every group whose last tile is now in: the group's rows of
 *   merged [C, H, W] = sum_t w * reduce_v(view_v^-1(tile_t)) / norm_full (integration order per pixel, fp32, no contraction:
 *   bit-identical to the incremental entry points).  The batches must stay alive and unmodified until their last group was
 *   launched.  Returns the number of launches (>= 0); PTB_EUNSUPPORTED when `pos` is not the next planned tile or the
 *   configuration (dtype, views, reduction, merged / norm_full / weight pointers) differs from the image's first batch -- nothing
 *   is launched then and the caller integrates incrementally.
 * ptb_band_plan_reset: next image.  ptb_band_plan_state: tiles taken / launches issued for the current image. */
typedef struct ptb_band_plan ptb_band_plan;
int64_t ptb_band_plan_create(const int64_t* xs, const int64_t* ys, int n, int C, int th, int tw, int H, int W, int rows_per_launch,
                             int final_lo, int final_hi, const int64_t* cuts, int ncuts, ptb_band_plan** out);
int ptb_band_plan_upload(ptb_band_plan* plan, void* dev_table, ptb_stream_t stream);
int ptb_band_plan_info(const ptb_band_plan* plan, int* n_groups, int* n_bands, int64_t* n_items, int64_t* last_group_of_tile,
                       int64_t* group_rows);
int ptb_band_plan_reset(ptb_band_plan* plan);
int ptb_band_plan_state(const ptb_band_plan* plan, int* pos, int* launched);
int ptb_band_plan_submit(ptb_band_plan* plan, int pos, int B, const void* batch, int64_t tile_stride, int64_t view_stride, int in_dtype,
                         int V, const int* views, int reduction, float* merged, const float* norm_full, const float* weight,
                         ptb_stream_t stream);
/* ptb_band_plan_submit_next: the next B planned tiles of an image whose configuration the image's first ptb_band_plan_submit has set
 * (same dtype / views / reduction / merged / norm / weight), `batch` = a contiguous [V * B, C, th, tw] model output -- one integrate_batch
 * of the reference's loop (inference/tiles.py:321-339) as a four-argument host call.  The plan keeps the byte ranges of the batches a later
 * launch group still reads: a batch overlapping one of them is refused with PTB_EHELD before anything is recorded.  Returns the launches
 * issued (>= 0); PTB_EUNSUPPORTED without a configuration or past the plan's last tile; else ptb_band_plan_submit's codes. */
int ptb_band_plan_submit_next(ptb_band_plan* plan, const void* batch, int B, ptb_stream_t stream);
void ptb_band_plan_destroy(ptb_band_plan* plan);

/* Multi-GPU (no reference counterpart; the single-device result of inference/tiles.py:321-346 is the specification).
 * ptb_band_plan_create2 = ptb_band_plan_create + `early`: n_early row ranges [lo, hi) of the plan's rows (ends among the cuts) that a
 * neighbouring rank waits for.  With them launch groups are formed by class: the bands of every early range in one launch of their own
 * (issued as soon as the tiles feeding that range are in), the others in groups of ~rows_per_launch rows that do not break at the cuts.
 * ptb_band_plan_rows_launched: 1 when every group writing rows r0 .. r1-1 has been issued for the current image, else 0.
 * ptb_halo_pack: strided rectangle -> contiguous send buffer, dst[c][r][x] = src[c * chan_stride + r * row_stride + x].
 * ptb_band_plan_submit_rank: ptb_band_plan_submit, then packs every outgoing rectangle (rects: n_sends x {r0, r1, c0, c1}, the plan's
 * rows) whose rows are complete into send_bufs[k] (`packed` [n_sends] in/out, zero at the start of an image) and, when the last one
 * has just been packed, records ready_event (a hipEvent_t; NULL only when n_sends == 0: a communication stream that did not wait
 * for it would send rectangles the pack kernels have not written yet -> PTB_EINVAL) on the stream.  Returns the band launches issued (>= 0).
 * ptb_band_plan_finish_rank: the end of a rank's image -- adds the n_recvs received rectangles of partial sums ({r0, r1, c0, c1} in
 * the plan's rows, packed [C][rows][cols] buffers) to `merged` and divides the n_ranges row ranges {r0, r1} that held partial sums by
 * `norm` [H][W] in place (launches only). */
/* The exchange itself over RCCL, bound at run time (dlopen of the process's librccl: no link-time dependency).
 * ptb_rccl_available: 1 when an RCCL library could be bound.  ptb_rccl_unique_id: 128 bytes produced on ONE rank and handed to the
 * others by the job's own channel.  ptb_rccl_comm_init: collective over the nranks processes (HIP current device = the rank's GPU).
 * ptb_halo_exchange: all outgoing (ptb_halo_pack'ed) and incoming rectangles of this rank -- contiguous fp32 buffers, element counts,
 * peer ranks -- as ONE ncclGroup of ncclSend / ncclRecv on `stream` (every pair concurrently, each on its own xGMI link; stream-ordered,
 * returns at once).  PTB_EUNSUPPORTED: no RCCL library; PTB_ELAUNCH: RCCL reported an error (text in ptb_last_hip_error()). */
int ptb_rccl_available(void);
int ptb_rccl_unique_id(void* id128);
int ptb_rccl_comm_init(const void* id128, int nranks, int rank, void** comm);
int ptb_rccl_comm_destroy(void* comm);
int ptb_halo_exchange(void* comm, int n_sends, const float* const* send_bufs, const int64_t* send_counts, const int* send_peers, int n_recvs,
                      float* const* recv_bufs, const int64_t* recv_counts, const int* recv_peers, ptb_stream_t stream);
int64_t ptb_band_plan_create2(const int64_t* xs, const int64_t* ys, int n, int C, int th, int tw, int H, int W, int rows_per_launch,
                              int final_lo, int final_hi, const int64_t* cuts, int ncuts, const int64_t* early, int n_early,
                              ptb_band_plan** out);
/* ptb_band_plan_create3 = ptb_band_plan_create2 + flags.  PTB_PLAN_CLIP_ROWS (bit 0): the plan's rows [0, H) are a window of the image;
 * tiles may hang over its top / bottom (ys < 0, ys + th > H) and only what lies inside is read and merged -- a rank that owns a range
 * of PIXEL rows and holds every tile touching them merges exactly those rows with no exchange (parallel: partition="pixel_rows"). */
#define PTB_PLAN_CLIP_ROWS 1
int64_t ptb_band_plan_create3(const int64_t* xs, const int64_t* ys, int n, int C, int th, int tw, int H, int W, int rows_per_launch,
                              int final_lo, int final_hi, const int64_t* cuts, int ncuts, const int64_t* early, int n_early, int flags,
                              ptb_band_plan** out);
int ptb_band_plan_rows_launched(const ptb_band_plan* plan, int r0, int r1);
int ptb_halo_pack(const float* src, int64_t chan_stride, int64_t row_stride, int C, int rows, int cols, float* dst, ptb_stream_t stream);
int ptb_band_plan_finish_rank(const ptb_band_plan* plan, float* merged, const float* norm, int n_recvs, const int64_t* rects,
                              const float* const* recv_bufs, int n_ranges, const int64_t* ranges, ptb_stream_t stream);
int ptb_band_plan_submit_rank(ptb_band_plan* plan, int pos, int B, const void* batch, int64_t tile_stride, int64_t view_stride, int in_dtype,
                              int V, const int* views, int reduction, float* merged, const float* norm_full, const float* weight,
                              int n_sends, const int64_t* rects, float* const* send_bufs, int* packed, void* ready_event,
                              int* all_packed, ptb_stream_t stream);

/* dst[c][r][x] += src[c][r][x] for a packed src [C, rows, cols] and a rectangle of a larger fp32 accumulator (element
 * strides dst_cs per channel, dst_rs per row): folds a halo rectangle received from another rank into the band
 * accumulator (multi-GPU merger; no reference counterpart). */
int ptb_rect_add(float* dst, const float* src, int C, int rows, int cols, int64_t dst_cs, int64_t dst_rs, ptb_stream_t stream);

/* ---- Loop edges on the device (compositions of reference functions, no single counterpart) ---------
 * ptb_split_tiles_u8 == ImageSlicer.split (inference/tiles.py:177-204, BORDER_CONSTANT only) -> image_to_tensor
 * (utils/torch_utils.py:204-231, HWC->CHW) -> .float() [-> * scale[c] + bias[c]] [-> *_image_augment, tta.py:385-422]:
 * image DEVICE uint8 [IH, IW, IC] contiguous; xs/ys HOST int64[B] = tile origins in IMAGE coordinates (the slicer's
 * bbox_crops: negative / overhanging parts read pad_value); views HOST int[V] (augment view codes; {PTB_VIEW_IDENT} for
 * none); scale/bias HOST float[IC] or both NULL; out DEVICE fp32 [V*B, IC, th, tw] chunk-major.  IC <= 16, V <= 8. */
int ptb_split_tiles_u8(const uint8_t* image, int IH, int IW, int IC, const int64_t* xs, const int64_t* ys, int B, int th, int tw,
                       int V, const int* views, const float* scale, const float* bias, int pad_value, float* out,
                       ptb_stream_t stream);
/* ptb_merge_crop == TileMerger.merge (tiles.py:345-346) -> np.moveaxis(.., 0, -1) -> .astype(uint8) | argmax
 * -> ImageSlicer.crop_to_orignal_size (tiles.py:271-280; README.md:225-226): window [top, top+OH) x [left, left+OW) of
 * image[C,H,W] / norm[H,W] (norm == NULL: image is already normalised).  layout 0: out [C, OH, OW], 1: out [OH, OW, C].  kind 0: float32; 1: uint8 by truncating
 * cast (numpy .astype on x86-64: low byte of the int32 truncation, 0 for NaN / out of int32 range); 2 / 3: argmax over
 * channels as uint8 / int64 [OH, OW] (first maximum, NaN counts as maximum; layout ignored). */
int ptb_merge_crop(const float* image, const float* norm, int C, int H, int W, int top, int left, int OH, int OW, int layout,
                   int kind, void* out, ptb_stream_t stream);

/* ---- Ensembler (+ ApplySigmoidTo / ApplySoftmaxTo) (inference/ensembling.py:12-123) ---------------
 * out = reduce_t act(inputs[t]) over T model outputs read in place (the reference stacks them first, :111,:117):
 * inputs HOST array of T DEVICE pointers, each [B, C, HW] contiguous fp32; reduction = PTB_RED_*; activation 0: none,
 * 1: sigmoid(x * temperature) (:65), 2: softmax over C of x * temperature (:41).  Summation in list order.  T <= 16. */
int ptb_ensemble_reduce(const float* const* inputs, int T, int reduction, int activation, float temperature, int B, int C,
                        int64_t HW, float* out, ptb_stream_t stream);

/* ---- VolumeMerger.integrate_batch (inference/tiles_3d.py:195-208) -----------------------------------
 * volume [C, D, H, W], norm [D, H, W], weight [d, h, w], tiles [B, C, d, h, w] DEVICE fp32; zs/ys/xs HOST int64[B] = tile
 * origin (roi start) per axis.  For b = 0..B-1 in order: volume[:, roi] += tiles[b] * weight (product rounded, then
 * added); norm[roi] += weight.  Out-of-range rois -> PTB_EBOUNDS.  merge = ptb_merge_div with HW = D*H*W. */
int ptb_volume_accumulate(float* volume, float* norm, const float* weight, const float* tiles, const int64_t* zs, const int64_t* ys,
                          const int64_t* xs, int B, int C, int d, int h, int w, int D, int H, int W, ptb_stream_t stream);

/* ---- {fliplr,flipud,flips,d2,d4}_image_deaugment (inference/tta.py:287-316,344-365,442-467,503-524) -------------
 * in [V*B, C, H, W] (chunk-major: rows [k*B,(k+1)*B) are view k), views HOST int[V] = inverse transform of each chunk.
 * out [B, C, H, W] = reduce_k view_k(in[k*B + b]).  V <= 8.  Transposing views require H == W. */
int ptb_deaug_reduce(const float* in, float* out, int V, const int* views, int reduction, int B, int C, int H, int W,
                     ptb_stream_t stream);

/* Backward of ptb_deaug_reduce for the NON-linear reductions (the TTA functions "respect gradients flow",
 * inference/tta.py:3-4); `in` / `out` are the forward input / output, `views` the forward views.  The linear reductions
 * (sum, mean) are differentiated with ptb_view_transform. */
int ptb_deaug_reduce_bwd(const float* in, const float* out, const float* grad_out, float* grad_in, int V, const int* views,
                         int reduction, int B, int C, int H, int W, ptb_stream_t stream);

/* Half-precision sources: the same as ptb_deaug_reduce / ptb_deaug_accumulate (and ptb_tile_accumulate with V = 1,
 * views = {PTB_VIEW_IDENT}, PTB_RED_SUM) with `in` holding in_dtype elements -- what the reference does with
 * `batch.type_as(self.image)` (inference/tiles.py:334-335) / its dtype-agnostic TTA ops, minus the fp32 copy: fp16 / bf16
 * values are widened in registers (exact), so the result equals the fp32 entry point on in.float() bit for bit.
 * PTB_EUNSUPPORTED (nothing launched) when the shape needs the scalar kernels or a non-default chunk size. */
int ptb_deaug_reduce_t(const void* in, int in_dtype, float* out, int V, const int* views, int reduction, int B, int C, int H,
                       int W, ptb_stream_t stream);
int ptb_deaug_accumulate_t(float* image, float* norm, const float* weight, const void* in, int in_dtype, int V, const int* views,
                           int reduction, const int64_t* xs, const int64_t* ys, int B, int C, int th, int tw, int H, int W,
                           uint8_t* fresh, int fresh_rows, ptb_stream_t stream);

/* ---- reductions over a stack of ANY length, explicit eps (inference/functional.py:247-331; tta.py:63-95) ----------
 * out[i] = post(mean_t pre(src[t, i])) for src [T, n] contiguous fp32: geometric_mean, harmonic_mean(eps), harmonic1p_mean,
 * logodd_mean(eps), log1p_mean, mean, sum along dim 0.  The fused view kernels above take at most 8 planes and the default
 * eps = 1e-6; this entry is `_deaugment_averaging` for tencrop TTA (T = 10), ensembles of more than 8 predictions, and the
 * reductions called with their eps argument.  eps is the Python double: the kernel clamps at (float)eps / (float)(1 - eps).
 * The backward takes the forward output: grad[t, i] = grad_out[i] * post'(out[i]) * pre'(src[t, i]) / T. */
int ptb_stack_reduce(const float* src, int T, int64_t n, int reduction, double eps, float* out, ptb_stream_t stream);
int ptb_stack_reduce_bwd(const float* src, const float* out, const float* grad_out, int T, int64_t n, int reduction, double eps,
                         float* grad, ptb_stream_t stream);

/* ---- per-view transform without reduction ----------------------------------------------------------------------
 * out[k*B + b] = scale * view_k(in[src]) with src = b (in_is_batch = 1: *_image_augment, inference/tta.py:257-284,
 * 319-341,385-422,470-484; in [B,C,H,W]) or src = k*B + b (in_is_batch = 0: de-augment with reduction=None).
 * out [V*B, C, Ho, Wo] where (Ho,Wo) = (W,H) for transposing views (so H == W is required for them). */
int ptb_view_transform(const float* in, float* out, int V, const int* views, int in_is_batch, float scale, int B, int C,
                       int H, int W, ptb_stream_t stream);

/* The same views as pure data movement for elements of ANY type: torch_fliplr ... torch_rot180_transpose, *_image_augment and
 * *_image_deaugment(reduction=None) on integer / boolean / float64 / half tensors and on tensors of more than four dims
 * (inference/functional.py:47-132: x.flip(3), x.flip(2), x.rot90(k, dims=(2, 3)), x.transpose(2, 3) -- index permutations of dims 2
 * and 3 whatever the dtype, dims beyond the fourth ride along with their pixel).  in [planes or V*planes, H, W, run] elements of
 * elem_bytes (1 | 2 | 4 | 8 | 16) bytes, planes = B * C, run = product of the dims beyond the fourth (1 for 4-D tensors);
 * out [V*planes, Ho, Wo, run] with (Ho, Wo) = (W, H) for transposing views -- non-square planes are fine when the views of one call
 * agree on the output shape (all transposing or none; PTB_EINVAL otherwise).  Bit-exact by construction. */
int ptb_view_permute(const void* in, void* out, int V, const int* views, int in_is_batch, int64_t planes, int H, int W,
                     int elem_bytes, int64_t run, ptb_stream_t stream);

/* ---- fused: de-augment + reduce + TileMerger.integrate_batch (tta.py:442-467 feeding tiles.py:321-339) ----------
 * image[:, y:y+th, x:x+tw] += reduce_k view_k(in[k*B+b]) * weight ; norm += weight.  The reduced tile never goes
 * to HBM.  Same tensors as ptb_deaug_reduce (H=th, W=tw) and ptb_tile_accumulate (incl. the first-touch bitmap). */
int ptb_deaug_accumulate(float* image, float* norm, const float* weight, const float* in, int V, const int* views,
                         int reduction, const int64_t* xs, const int64_t* ys, int B, int C, int th, int tw, int H,
                         int W, uint8_t* fresh, int fresh_rows, ptb_stream_t stream);

/* ---- bilinear resize of ms_image_augment / ms_image_deaugment (inference/tta.py:599-621, 645-689) ----------------
 * = torch.nn.functional.interpolate(in, size=(hout,wout), mode="bilinear", align_corners=...) on `planes` = B*C maps. */
int ptb_resize_bilinear(const float* in, float* out, int64_t planes, int hin, int win, int hout, int wout,
                        int align_corners, ptb_stream_t stream);

/* ---- fused ms_image_deaugment (inference/tta.py:645-689): out = reduce_s resize(inputs[s] -> (hout,wout)) -----------
 * inputs: HOST array of n (<= 8) device pointers, map s is [planes, hs[s], ws[s]]; maps that already have the output
 * size are taken as they are (the reference skips F.interpolate for offset 0).  The resized maps never go to HBM. */
int ptb_ms_deaug_reduce(const float* const* inputs, const int* hs, const int* ws, int n, float* out, int64_t planes, int hout,
                        int wout, int align_corners, int reduction, ptb_stream_t stream);
/* Row-strip variant (cfg5 over several GPUs, no collective: every rank reduces its own strip of output rows): computes
 * output rows [out_row0, out_row0 + out_rows) of the hout_full x wout result into out [planes, out_rows, wout].  Map s has
 * hs_full[s] rows in total, of which inputs[s] holds src_rows[s] rows starting at global row src_row0[s] ([planes,
 * src_rows[s], ws[s]]); the strip must contain every source row the 4-tap footprints of the output rows touch
 * (pytorch_toolbelt_amd.parallel.ms_strip_plan computes the ranges).  Same arithmetic as the full call. */
int ptb_ms_deaug_reduce_strip(const float* const* inputs, const int* hs_full, const int* ws, const int* src_row0,
                              const int* src_rows, int n, float* out, int64_t planes, int hout_full, int wout, int out_row0,
                              int out_rows, int align_corners, int reduction, ptb_stream_t stream);

/* Adjoints of the resize entry points ("TTA respects gradient flow", inference/tta.py:3-4; the reference gets them from autograd
 * through F.interpolate).  grad_in / grad_inputs[s] must be ZEROED by the caller: the 4 taps of several output pixels land on
 * the same source pixel and are added with fp32 atomics (summation order, hence the last bit, not deterministic -- as in
 * ATen's upsample_bilinear2d_backward).  ptb_ms_deaug_reduce_bwd: gradient of ptb_ms_deaug_reduce w.r.t. every scale's map;
 * fwd_out = the forward result (needed by the non-linear reductions, may be NULL for sum / mean). */
int ptb_resize_bilinear_bwd(const float* grad_out, float* grad_in, int64_t planes, int hin, int win, int hout, int wout,
                            int align_corners, ptb_stream_t stream);
int ptb_ms_deaug_reduce_bwd(const float* const* inputs, const int* hs, const int* ws, int n, const float* fwd_out, const float* grad_out,
                            float* const* grad_inputs, int64_t planes, int hout, int wout, int align_corners, int reduction,
                            ptb_stream_t stream);

/* F.interpolate(x, size, mode="nearest") for [planes, hin, win] -> [planes, hout, wout] (multiscale TTA with mode="nearest",
 * inference/tta.py:599-621, 645-689): out[p, y, x] = in[p, min(floor(y * hin / hout), hin - 1), min(floor(x * win / wout), win - 1)].
 * backward != 0: `in` is grad_out [planes, hout, wout] and `out` the ZEROED grad_in [planes, hin, win] (atomic adds). */
int ptb_resize_nearest(const float* in, float* out, int64_t planes, int hin, int win, int hout, int wout, int backward,
                       ptb_stream_t stream);

/* The two remaining 4-D modes F.interpolate offers (the reference forwards any `mode`, inference/tta.py:599-621, 645-689), same layout
 * and backward convention as ptb_resize_nearest:
 * "nearest-exact": out[p, y, x] = in[p, min(floor((y + 0.5) * hin / hout), hin - 1), min(floor((x + 0.5) * win / wout), win - 1)];
 * "area" (= adaptive_avg_pool2d): out[p, y, x] = mean of in[p, floor(y hin / hout) : ceil((y + 1) hin / hout),
 *                                                           floor(x win / wout) : ceil((x + 1) win / wout)]. */
int ptb_resize_nearest_exact(const float* in, float* out, int64_t planes, int hin, int win, int hout, int wout, int backward,
                             ptb_stream_t stream);
int ptb_resize_area(const float* in, float* out, int64_t planes, int hin, int win, int hout, int wout, int backward, ptb_stream_t stream);

/* F.interpolate(x, size, mode="bicubic", align_corners) for [planes, hin, win] -> [planes, hout, wout] (multiscale TTA with
 * mode="bicubic"): 4 x 4 taps around floor(src) (src without the clamp at 0 of the linear modes), indices clamped into the image,
 * cubic convolution weights with A = -0.75, rows first (aten UpSampleBicubic2d).  backward != 0: `in` is grad_out
 * [planes, hout, wout], `out` the ZEROED grad_in [planes, hin, win] (atomic adds, like aten's backward). */
int ptb_resize_bicubic(const float* in, float* out, int64_t planes, int hin, int win, int hout, int wout, int align_corners, int backward,
                       ptb_stream_t stream);

/* Multiscale + flip TTA in one pass (BASELINE configs[4]; extension): inputs[s] = the model output for the flip-augmented batch
 * of scale s, [V * B, C, hs[s], ws[s]] chunk-major (view v of plane p at (v * planes + p) * hs * ws; planes = B * C),
 *   out = reduction_s( bilinear_s( inner_reduction_v( view_v^-1( inputs[s][v] ) ) ) )
 * == ms_image_deaugment([<group>_image_deaugment(y_s, inner_reduction) for s], ...) (inference/tta.py:287-316, 344-365, 503-524 then
 * :645-689) without the flip-reduced maps ever reaching HBM.  views[V] = the group's INVERSE view codes (row-preserving only:
 * PTB_VIEW_IDENT / FLIPLR / FLIPUD / ROT180).  Returns PTB_EUNSUPPORTED for transposing views, widths not divisible by 4, or a
 * scale ratio above ~1.3 (source window of a 64 x 64 tile larger than 88 x 96): the caller then composes the two entry points. */
int ptb_ms_flip_deaug_reduce(const float* const* inputs, const int* hs, const int* ws, int n, int V, const int* views,
                             int inner_reduction, float* out, int64_t planes, int hout, int wout, int align_corners, int reduction,
                             ptb_stream_t stream);

/* One rank's rows of the same pass (BASELINE configs[4] over several GPUs: output row strips, no collective): inputs[s] holds rows
 * src_row0[s] .. src_row0[s] + src_rows[s] - 1 of every view plane of scale s ([V * planes, src_rows[s], ws[s]]; hs_full[s] is the full
 * height: the taps are those of the full-size call, so the strips of the result concatenate to it bit for bit); out receives rows
 * out_row0 .. out_row0 + out_rows - 1 of the [planes, hout_full, wout] result.  PTB_EBOUNDS when a strip lacks a source row the taps of
 * those output rows read; views that flip rows do not come in strips (PTB_EUNSUPPORTED). */
int ptb_ms_flip_deaug_reduce_strip(const float* const* inputs, const int* hs_full, const int* ws, const int* src_row0, const int* src_rows,
                                   int n, int V, const int* views, int inner_reduction, float* out, int64_t planes, int hout_full, int wout,
                                   int out_row0, int out_rows, int align_corners, int reduction, ptb_stream_t stream);

/* ================================= segmentation losses (pytorch_toolbelt.losses) =================================
 * logits [B, C, HW] fp32; targets are either labels int64 [B, HW] (one-hot is formed on the fly, never materialised)
 * or dense fp32 [B, C, HW]; exactly one of `labels` / `dense` is non-NULL.  Scalars are accumulated in fp64.
 * flags: */
#define PTB_SEG_FOCAL 1            /* accumulate sigmoid focal sums: sums[0] = sum loss, sums[1] = sum focal_term */
#define PTB_SEG_STATS 2            /* accumulate region statistics: sums[2 + k*C + c], k = 0: I=sum p*t, 1: P=sum p, 2: T=sum t */
#define PTB_SEG_HAS_IGNORE 4       /* ignore_label (labels) / ignore_value (dense) is active */
#define PTB_SEG_HAS_ALPHA 8
#define PTB_SEG_REDUCED 16         /* reduced focal loss with `threshold` */
#define PTB_SEG_MASK_FOCAL_TERM 32 /* normalized focal: ignored elements add 0 to sums[1] (functional.py:90-98) */
#define PTB_SEG_ELEMWISE 64        /* also write the unreduced focal loss to elem_out [B, C, HW] */
#define PTB_SEG_NO_TERM 128       /* the caller will not read sums[1] (no normalized=True): kernels may leave it unset */
/* prob (activation used for the region statistics): */
#define PTB_PROB_SOFTMAX 0         /* log_softmax(dim=1).exp()  (dice.py:68-72 multiclass) */
#define PTB_PROB_SIGMOID 1         /* logsigmoid.exp()          (dice.py:73-75 binary / multilabel) */
#define PTB_PROB_IDENTITY 2        /* from_logits=False: `logits` already holds probabilities */

/* Scalars are accumulated into PTB_SUM_SLOTS interleaved copies (same-address atomics serialise on the device); the
 * caller zeroes the whole buffer and adds the slots: total[k] = sum_s sums[s * n + k]. */
#define PTB_SUM_SLOTS 64

/* One pass over logits+targets for focal_loss_with_logits (losses/functional.py:19-107, sigmoid activation),
 * BinaryFocalLoss (losses/focal.py:77-105) and the sums of soft_dice_score / soft_jaccard_score with dims=(0,2)
 * (losses/functional.py:188-247; DiceLoss losses/dice.py:59-131, JaccardLoss losses/jaccard.py:48-103).
 * sums: double[PTB_SUM_SLOTS][2 + 3*C] (zeroed by this call, like error_flag; reduce with ptb_sums_finalize); error_flag: int, set to 1 when a label is outside [0, C)
 * and not ignore_label (the reference's F.one_hot raises).  class_weights [C] fp32 or NULL. */
int ptb_seg_loss_fwd(const float* logits, const int64_t* labels, const float* dense, const float* class_weights,
                     double* sums, float* elem_out, int* error_flag, int B, int C, int64_t HW, int flags, int prob,
                     float gamma, float alpha, float threshold, int64_t ignore_label, float ignore_value,
                     ptb_stream_t stream);

/* Gradient of the sigmoid focal loss w.r.t. logits: grad[i] = coef[0] * (grad_elem ? grad_elem[i] : 1) * dL_i/dx_i
 * + coef[1] * dF_i/dx_i, coef = DEVICE float[2] (reduction / normalisation factors times the upstream gradient). */
int ptb_focal_bwd(const float* logits, const int64_t* labels, const float* dense, const float* class_weights,
                  const float* coef, const float* grad_elem, float* grad, int B, int C, int64_t HW, int flags,
                  float gamma, float alpha, float threshold, int64_t ignore_label, float ignore_value,
                  ptb_stream_t stream);

/* focal_loss_with_logits(..., activation="softmax", softmax_dim=d) (losses/functional.py:61-107 with :63-64
 * `p = torch.softmax(output, dim=softmax_dim)`): the tensor is passed as the [B, C, HW] view whose C is the softmax dimension.
 * Same flags / sums[0..1] / elem_out / coef / grad_elem contract as ptb_seg_loss_fwd (PTB_SEG_FOCAL) and ptb_focal_bwd, except
 * that `sums` is [PTB_SUM_SLOTS, 2].  class_weights follow dim 1 of the ORIGINAL tensor (functional.py:83-88): cw_mode 0 =
 * none, 1 = indexed by the softmax channel c (cw_n == C), 2 = by (b / cw_div) % cw_n, 3 = by (position / cw_div) % cw_n. */
int ptb_focal_softmax_fwd(const float* logits, const int64_t* labels, const float* dense, const float* class_weights, int cw_mode,
                          int cw_n, int64_t cw_div, double* sums, float* elem_out, int* error_flag, int B, int C, int64_t HW, int flags,
                          float gamma, float alpha, float threshold, int64_t ignore_label, float ignore_value, ptb_stream_t stream);
int ptb_focal_softmax_bwd(const float* logits, const int64_t* labels, const float* dense, const float* class_weights, int cw_mode,
                          int cw_n, int64_t cw_div, const float* coef, const float* grad_elem, float* grad, int B, int C, int64_t HW,
                          int flags, float gamma, float alpha, float threshold, int64_t ignore_label, float ignore_value,
                          ptb_stream_t stream);

/* Gradient of any function of the region statistics w.r.t. logits, given DEVICE arrays gI[C] = dLoss/dI_c and
 * gP[C] = dLoss/dP_c (T does not depend on the logits). */
int ptb_seg_stats_bwd(const float* logits, const int64_t* labels, const float* dense, const float* gI, const float* gP,
                      float* grad, int B, int C, int64_t HW, int flags, int prob, int64_t ignore_label,
                      float ignore_value, ptb_stream_t stream);

/* Both of the above in one pass (backward of a ptb_seg_loss_fwd call with PTB_SEG_FOCAL | PTB_SEG_STATS);
 * PTB_EUNSUPPORTED when the fused kernel does not apply (C > 16, HW % 4 != 0, ...): use the two kernels above. */
int ptb_seg_fused_bwd(const float* logits, const int64_t* labels, const float* dense, const float* class_weights,
                      const float* coef, const float* gI, const float* gP, float* grad, int B, int C, int64_t HW, int flags,
                      int prob, float gamma, float alpha, float threshold, int64_t ignore_label, float ignore_value,
                      ptb_stream_t stream);

/* Scalar tail of DiceLoss (losses/dice.py:112-131), JaccardLoss (losses/jaccard.py:95-113) and their weighted sum with a
 * mean focal loss, together with its derivative, from the slot sums a ptb_seg_loss_fwd call produced:
 *   loss[0] = focal_scale * sum_focal + dice_weight * mean_c dice_loss_c + jaccard_weight * mean_c jaccard_loss_c
 *   score_c = (2 I + smooth) / max(P + T + smooth, eps)  |  (I + smooth) / max(P + T - I + smooth, eps);
 *   loss_c = [T_c > 0] * (log_loss ? -log(max(score_c, eps)) : 1 - score_c); the mean runs over the n_selected classes with
 *   class_mask[c] != 0 (class_mask NULL: all C).  fp32 arithmetic on the fp64 sums rounded to fp32, like the reference.
 * coef DEVICE float[2 + 2C] receives d loss / d (focal loss sum, focal term sum, I[C], P[C]) -- the arrays
 * ptb_focal_bwd / ptb_seg_stats_bwd / ptb_seg_fused_bwd take (after scaling by the upstream gradient).  error_flag (may be NULL): the
 * forward kernel's label flag; when raised the loss is NaN. */
int ptb_region_epilogue(const double* sums, int slots, int C, float focal_scale, float dice_weight, float jaccard_weight,
                        float smooth, float eps, int log_loss, const uint8_t* class_mask, int n_selected, float* loss,
                        float* coef, const int* error_flag, ptb_stream_t stream);

/* The two above in ONE launch (no reference counterpart: losses/dice.py:66-131 and losses/jaccard.py:62-113 are ~40 torch ops): the
 * streaming statistics kernel of ptb_seg_loss_fwd, whose last-arriving workgroup adds up the slot sums, evaluates the scalar tail
 * of ptb_region_epilogue and its derivative, and leaves the workspace zeroed for the next call -- no memset, finalize or epilogue
 * launches.  `workspace`: ptb_region_workspace_bytes(C) bytes of 8-byte aligned DEVICE memory that were ZERO before the first call
 * and are used by one stream at a time; every call leaves them zero.  flags must contain SEG_STATS (2), optionally SEG_FOCAL (1)
 * and the option bits of ptb_seg_loss_fwd (not ELEMWISE).  loss DEVICE float[1], coef DEVICE float[2 + 2C] as for
 * ptb_region_epilogue; error_out (may be NULL): int[1], device OR pinned-host memory, receives the call's label flag (a label
 * outside [0, C) that is not ignore_index also turns the loss into NaN).  Returns PTB_EUNSUPPORTED for an empty input. */
int64_t ptb_region_workspace_bytes(int C);
int ptb_region_loss_fwd(const float* logits, const int64_t* labels, const float* dense, const float* class_weights, void* workspace,
                        int B, int C, int64_t HW, int flags, int prob, float gamma, float alpha, float threshold, int64_t ignore_label,
                        float ignore_value, float focal_scale, float dice_weight, float jaccard_weight, float smooth, float eps,
                        int log_loss, const unsigned char* class_mask, int n_selected, float* loss, float* coef, int* error_out,
                        ptb_stream_t stream);

/* softmax_focal_loss_with_logits / CrossEntropyFocalLoss (losses/functional.py:110-173, losses/focal.py:108-161).
 * sums double[PTB_SUM_SLOTS][2] (zeroed by this call): sum of per-pixel losses, sum of all focal terms; pixel_out [B, HW] optional. */
int ptb_softmax_focal_fwd(const float* logits, const int64_t* labels, const float* class_weights, double* sums,
                          float* pixel_out, int* error_flag, int B, int C, int64_t HW, int reduced, float gamma,
                          float threshold, int64_t ignore_label, ptb_stream_t stream);
int ptb_softmax_focal_bwd(const float* logits, const int64_t* labels, const float* class_weights, const float* coef,
                          const float* grad_pix, float* grad, int B, int C, int64_t HW, int reduced, float gamma,
                          float threshold, int64_t ignore_label, ptb_stream_t stream);

/* ---- Remaining elementwise + reduce losses -------------------------------------------------------
 * kind: 0 SoftBCEWithLogitsLoss (losses/soft_bce.py:29-46; p0 = smooth factor when flags & 2, chan_w / chan_pw = DEVICE
 * per-channel weight / pos_weight [C] or NULL with element channel = (i / HW) % C), 1 balanced BCE (losses/
 * balanced_bce.py:27-40), 2 QualityFocalLoss (losses/quality_focal_loss.py:33-35; p0 = beta), 3 wing_loss (losses/
 * functional.py:260-269; p0 = width, p1 = curvature, p2 = width - width*log(1 + width/curvature)), 4 log_cosh_loss
 * (losses/functional.py:338-341), 5 soft F1 counts of one class (losses/soft_f1.py:22-24, 63-78: p = clamp(sigmoid(x), p0,
 * 1 - p0), or p = x when flags & 2).  flags: 1 = elements whose target == ignore_value contribute 0; 2 = label smoothing.
 * x, t: DEVICE fp32 [n].  sums: DEVICE double [PTB_SUM_SLOTS][4], zeroed by this call, slot-wise partial sums of
 *   kind 0,3,4: {loss};  kind 1: {sum t*logsigmoid(x), sum (1-t)*logsigmoid(-x), #(t == 1), #(t == 0)};  kind 2: {loss, focal};
 *   kind 5: {sum p t, sum p, sum t, #kept} (TP = s0, FP = s1 - s0, FN = s2 - s0); its apply gives d(coef[0] s0 + coef[1] s1)/dx.
 * elem_out (optional, not for kind 1): per-element loss. */
int ptb_pointwise_loss_fwd(int kind, const float* x, const float* t, const float* chan_w, const float* chan_pw, double* sums,
                           float* elem_out, int64_t n, int C, int64_t HW, int flags, float p0, float p1, float p2,
                           float ignore_value, ptb_stream_t stream);
/* out[i] = coef[0] * (grad_elem ? grad_elem[i] : 1) * dloss_i/dx_i (+ coef[1] * dfocal_i/dx_i for kind 2); kind 1: coef =
 * {pos class weight, neg class weight} (each already multiplied by the upstream gradient); emit_loss = 1 (kind 1 only)
 * writes the per-element loss -(coef[0]*t*logsigmoid(x) + coef[1]*(1-t)*logsigmoid(-x)) instead.  coef: DEVICE float[2]. */
int ptb_pointwise_loss_apply(int kind, int emit_loss, const float* x, const float* t, const float* chan_w, const float* chan_pw,
                             const float* coef, const float* grad_elem, float* out, int64_t n, int C, int64_t HW, int flags,
                             float p0, float p1, float p2, float ignore_value, ptb_stream_t stream);
/* SoftCrossEntropyLoss = label_smoothed_nll_loss(log_softmax(x, 1)) (losses/soft_ce.py:24-33, losses/functional.py:280-323):
 * logits [B, C, HW], labels int64 [B, HW]; sums double [PTB_SUM_SLOTS][4]: {sum nll, sum smooth} over non-ignored pixels;
 * pixel_out (optional) [B, HW] = (1-eps)*nll + eps/C*smooth; error_flag set to 1 on a label outside [0, C) that is not ignored.
 * bwd: grad[b,c,i] = coef[0] * (grad_pix ? grad_pix[b,i] : 1) * d pixel_loss / d x[b,c,i]; coef DEVICE float[1]. */
int ptb_soft_ce_fwd(const float* logits, const int64_t* labels, double* sums, float* pixel_out, int* error_flag, int B, int C,
                    int64_t HW, float eps, int has_ignore, int64_t ignore_label, ptb_stream_t stream);
int ptb_soft_ce_bwd(const float* logits, const int64_t* labels, const float* coef, const float* grad_pix, float* grad, int B, int C,
                    int64_t HW, float eps, int has_ignore, int64_t ignore_label, ptb_stream_t stream);

/* BinaryBiTemperedLogisticLoss (losses/bitempered_loss.py:223-284; bi_tempered_logistic_loss :135-180 with the two
 * activations (-x, x) and targets (1-t, t)): x, t DEVICE fp32 [n]; sums double [PTB_SUM_SLOTS][4] {sum of per-element
 * losses}; elem_out optional; iters = normalisation iterations (the reference uses 5); elements whose target equals
 * ignore_value contribute 0.  bwd: grad[i] = coef[0] * (grad_elem ? grad_elem[i] : 1) * dloss_i/dx_i, coef DEVICE float[1]. */
int ptb_bitempered_binary_fwd(const float* x, const float* t, double* sums, float* elem_out, int64_t n, float t1, float t2,
                              float smoothing, int iters, int has_ignore, float ignore_value, ptb_stream_t stream);
int ptb_bitempered_binary_bwd(const float* x, const float* t, const float* coef, const float* grad_elem, float* grad, int64_t n,
                              float t1, float t2, float smoothing, int iters, int has_ignore, float ignore_value,
                              ptb_stream_t stream);

/* bi_tempered_logistic_loss (losses/bitempered_loss.py:135-180) for activations [R, K] (classes last) with DENSE targets [R, K]
 * (one-hot or soft; label smoothing applied inside): out[r] = the unreduced loss of row r.  One wave per row; the tempered softmax's
 * normalisation runs `iters` iterations (the reference's num_iters = 5): fixed point for t2 > 1, bisection for t2 < 1, log-sum-exp
 * for t2 = 1.  backward != 0: grad_loss [R] in, out = d loss / d activations [R, K] (closed form via the escort distribution,
 * :94-104).  t1 = 2 is rejected (the loss formula divides by 2 - t1). */
int ptb_bitempered_rows(const float* activations, const float* onehot, const float* grad_loss, float* out, int64_t R, int K, float t1,
                        float t2, float smoothing, int iters, int backward, ptb_stream_t stream);

/* ---- Lovasz hinge / Lovasz-softmax (losses/lovasz.py:23-184) ---------------------------------------------------
 * mode 0 (softmax): pred = probabilities [B, C, HW], labels int64 [B, HW]; mode 1 (hinge): pred = logits [B, HW],
 * flabels = float 0/1 labels [B, HW], C = 1.  A segment is one (group, class): group = image when per_image else the
 * whole batch; S = groups*C segments of P = (per_image ? HW : B*HW) elements, n = S*P < 2^31.
 * seg_loss[s] (double, zeroed by this call) = dot(relu(errors_sorted), lovasz_grad(fg_sorted)); fg_total[s] = number
 * of foreground pixels (class presence); grad_at_pixel[s*P + i] = Lovasz gradient at the rank of pixel i (for backward;
 * may be NULL when no backward will follow: the scatter of the gradients to pixel order is then skipped).
 * Workspaces are caller-provided device buffers: keys_a/keys_b u32[n] (key = ~kappa, kappa = bits(max(error, +0)) << 1 | fg: the
 * errors in descending order, ties by fg, then by index -- the reference's torch.sort leaves the order of ties open),
 * vals_a/vals_b u32[n] (index << 1 | fg), chunk u32[S*ceil(P/2048)], temp of ptb_lovasz_temp_bytes(P, S) bytes (digit histograms of the
 * hand-written segmented radix sort: four stable 8-bit passes per segment). */
int64_t ptb_lovasz_temp_bytes(int64_t per_segment, int segments);
int ptb_lovasz_fwd(const float* pred, const int64_t* labels, const float* flabels, int B, int C, int64_t HW, int mode,
                   int per_image, int has_ignore, int64_t ignore_label, float ignore_value, uint32_t* keys_a, uint32_t* keys_b,
                   unsigned* vals_a, unsigned* vals_b, unsigned* chunk, unsigned* fg_total,
                   double* seg_loss, float* grad_at_pixel, void* temp, int64_t temp_bytes, ptb_stream_t stream);
/* ptb_lovasz_fwd with the gradient left BINNED instead of scattered to pixel order (16.7 M random 4-byte writes at [4,16,512,512]
 * become one more pass of the sort's scatter): on return keys_b[] holds (index << 1 | fg) and vals_b[] the fp32 gradient bits of the
 * same element, grouped by blocks of 2^r pixels, r = the RETURN VALUE (12..14): the pairs of block b of segment s are at
 * s*P + (b << r) ..., in arbitrary order inside the block.  scratch is unused (may be NULL).  PTB_EUNSUPPORTED when a segment
 * has more than 256 * 2^14 elements (use ptb_lovasz_fwd).  ptb_lovasz_bwd_binned consumes (keys_b, vals_b, r). */
/* ptb_lovasz_fwd_keys: the forward WITHOUT a gradient (evaluation / torch.no_grad()) as a key-only sort -- no (index, fg) value travels
 * with the key: errors <= 0 and ignored pixels share the key of +0 (they contribute relu(e) * grad = 0 wherever they sort among
 * themselves), a positive error's 31 significant bits move up by one and the foreground flag takes the freed bit.  Same seg_loss /
 * fg_total as ptb_lovasz_fwd (the order of equal errors differs, which the loss does not depend on); 4 bytes per element and pass. */
int ptb_lovasz_fwd_keys(const float* pred, const int64_t* labels, const float* flabels, int B, int C, int64_t HW, int mode, int per_image,
                        int has_ignore, int64_t ignore_label, float ignore_value, uint32_t* keys_a, uint32_t* keys_b, unsigned* chunk,
                        unsigned* fg_total, double* seg_loss, void* temp, int64_t temp_bytes, ptb_stream_t stream);
int ptb_lovasz_fwd_binned(const float* pred, const int64_t* labels, const float* flabels, int B, int C, int64_t HW, int mode,
                          int per_image, int has_ignore, int64_t ignore_label, float ignore_value, uint32_t* keys_a, uint32_t* keys_b,
                          unsigned* vals_a, unsigned* vals_b, unsigned* chunk, unsigned* fg_total,
                          double* seg_loss, float* scratch, void* temp, int64_t temp_bytes, ptb_stream_t stream);
int ptb_lovasz_bwd_binned(const float* pred, const int64_t* labels, const float* flabels, const float* coef, const uint32_t* binned_vals,
                          const float* binned_grad, float* grad, int B, int C, int64_t HW, int mode, int per_image, int has_ignore,
                          int64_t ignore_label, float ignore_value, int block_log2, ptb_stream_t stream);
/* ptb_lovasz_bwd_binned with the incoming gradient of the scalar loss as a DEVICE scalar gscale and ptb_lovasz_reduce's coef_out as
 * coef_unit: coefficient of segment s = gscale[0] * coef_unit[s], formed inside the kernel (the rounding of the [S]-sized product the
 * caller would otherwise launch in front of it). */
int ptb_lovasz_bwd_binned2(const float* pred, const int64_t* labels, const float* flabels, const float* coef_unit, const float* gscale,
                           const uint32_t* binned_vals, const float* binned_grad, float* grad, int B, int C, int64_t HW, int mode,
                           int per_image, int has_ignore, int64_t ignore_label, float ignore_value, int block_log2, ptb_stream_t stream);
/* The scalar the modules return from seg_loss / fg_total (losses/lovasz.py:92-108, :110-140): per group the mean of seg_loss over
 * the classes with fg_total > 0 (present_only = 1, classes="present") or over all classes (0; the hinge loss is C = 1), 0 when none
 * is selected; then the mean over the groups.  loss_out = DEVICE float; coef_out = DEVICE float[groups*C] = d(loss)/d(seg_loss). */
int ptb_lovasz_reduce(const double* seg_loss, const unsigned* fg_total, int groups, int C, int present_only, float* loss_out,
                      float* coef_out, ptb_stream_t stream);
/* grad[pred layout] = coef[s] * grad_at_pixel * d(error)/d(pred); coef = DEVICE float[S].  Every element of grad is written. */
int ptb_lovasz_bwd(const float* pred, const int64_t* labels, const float* flabels, const float* coef,
                   const float* grad_at_pixel, float* grad, int B, int C, int64_t HW, int mode, int per_image,
                   int has_ignore, int64_t ignore_label, float ignore_value, ptb_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PTB_HIP_H */
