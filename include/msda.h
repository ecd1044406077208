/*
 * libmsda — multi-scale deformable attention (MSDeformAttn) for AMD Instinct MI355X (gfx950).
 *
 * C ABI of the hot path.  Plain pointers and sizes only; every buffer is allocated and owned by
 * the caller and lives in device (HBM) memory of the GPU that is current when the call is made.
 * The library never allocates, frees or retains pointers, never synchronises the device, and
 * launches asynchronously on the HIP stream it is given (NULL = the default stream).  There is
 * no global mutable state: the entry points may be called concurrently from several host
 * threads (autograd runs the backward on its own engine thread).  The one exception is
 * msda_backward_noalloc, whose scratch is a static array inside the library (see there).
 *
 * Tensor conventions (contiguous, row-major):
 *   value   (N, S, M, D)         S = sum_l H_l * W_l
 *   shapes  (L, 2) int64         (H_l, W_l), device memory
 *   lsi     (L,)   int64         first index of level l inside S, device memory
 *   loc     (N, Lq, M, L, P, 2)  (x, y) normalised to [0, 1]
 *   attn    (N, Lq, M, L, P)
 *   out     (N, Lq, M * D)
 *
 * Return value of every int-returning entry point: 0 on success, a negative MSDA_E* code for an
 * argument error (nothing was launched), a positive hipError_t if a launch failed.
 */
#ifndef MSDA_H_
#define MSDA_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* element types of value/out/grad tensors (dtype_val) and of loc/attn tensors (dtype_loc) */
enum {
  MSDA_F32 = 0,
  MSDA_F64 = 1,
  MSDA_BF16 = 2,
  MSDA_F16 = 3,
};

/* argument errors */
enum {
  MSDA_OK = 0,
  MSDA_ENULL = -1,      /* a required pointer is NULL */
  MSDA_ESHAPE = -2,     /* a dimension is < 1 or a product overflows 2^31-1 where it must not */
  MSDA_EDTYPE = -3,     /* unsupported dtype or dtype combination */
  MSDA_EALIGN = -4,     /* a pointer is not aligned to its element type (16 B for fast paths) */
  MSDA_EWORKSPACE = -5, /* workspace missing or smaller than msda_backward_workspace_bytes() */
  MSDA_ESTEP = -6,      /* im2col_step < 1 or N % min(N, im2col_step) != 0 */
  MSDA_ELEVELS = -7,    /* more levels or points than the library supports (L <= 16, L*P <= 128) */
  MSDA_ELAYOUT = -8,    /* MSDA_FLAG_VALUE_HEAD_MAJOR on a shape/path that only takes (N,S,M,D) */
  MSDA_ENOALLOC = -9,   /* msda_backward_noalloc: scratch need beyond the static array's 128 MiB */
};

/* flags for msda_backward */
enum {
  /* Bitwise reproducible grad_value: two calls on the same inputs give identical bits (grad_loc
   * and grad_attn always do).  Shapes the binned path covers (msda_head_major_supported) keep
   * that path and run a tile kernel that adds every position's contributions in an order fixed
   * by the data (runs in query-block order, rounds made of whole runs, entries sorted per
   * position): about 2.5 times the backward's time at any size (measured on MI355X at the
   * Cityscapes shapes: 2.04 -> 5.15 ms); pass the flag to msda_backward_workspace_bytes too, the
   * size differs.  Other shapes (F64,
   * head dims without a tuned kernel) use a per-destination gather in a fixed order, O(S * Lq * P)
   * per head and channel — for small problems (gradcheck, debugging) only. */
  MSDA_FLAG_DETERMINISTIC = 1,
  /* Force the generic (any D, one lane per channel) kernels even where a tuned path exists. */
  MSDA_FLAG_GENERIC = 2,
  /* Tuned path only: scatter grad_value with global fp32 atomics instead of the binned,
   * LDS-accumulated path (for A/B measurements; much slower on MI355X). */
  MSDA_FLAG_ATOMIC_SCATTER = 4,
  /* `value` is laid out (N, M, S, D) — one contiguous slab per head — instead of (N, S, M, D).
   * Accepted by msda_forward and msda_backward where msda_head_major_supported() says so;
   * grad_value is (N, S, M, D) either way.  A head's slab then spreads over all L2 channels and
   * x-neighbouring bilinear corners are contiguous: the gathers run ~25 % faster on MI355X. */
  MSDA_FLAG_VALUE_HEAD_MAJOR = 8,
  /* Locations and weights come as ONE packed tensor (N, Lq, M, 3*L*P): per (batch, query, head)
   * row the L*P (x, y) pairs, then the L*P weights — the input of the reference's
   * `MultiScaleFlashAttnFunction`
   * (tests/unipercept/nn/layers/test_nn_layers_deform_attn.py:105-111,186-192 of the reference:
   * `cat([loc.flatten(-3), attn.flatten(-2)], -1)`).  Pass it as `loc`; `attn` is ignored.  In
   * msda_backward `grad_loc` receives the packed gradient (same layout) and `grad_attn` is
   * ignored.  Weights are used as given (no softmax), as that test compares against
   * `reference.deform_attn` with the same weights.  Where msda_packed_points_supported() says so. */
  MSDA_FLAG_PACKED_POINTS = 16,
  /* msda_backward / msda_backward_workspace_bytes: ignore `shapes_host`, i.e. cut grad_value into
   * runs of the flattened sequence even when the level shapes are known on the host (for A/B
   * measurements and tests of that geometry). */
  MSDA_FLAG_TILE_1D = 32,
  /* The points come RAW, as the layer's projection left them: `loc` is one tensor
   * (N, Lq, M, 3*L*P) holding per (batch, query, head) row the L*P pixel offsets (dx, dy) and then
   * the L*P attention logits (the packed row form of MSDA_FLAG_PACKED_POINTS), and `ref_points`
   * (N, Lq, L, 2), fp32, holds the query's reference point on every level.  The kernels compute
   *     (x, y) = ref + (dx / W_l, dy / H_l),   weights = softmax over the row's L*P logits
   * as they read the points, and msda_backward writes the gradient of the raw tensor (offsets:
   * d(x, y) / (W_l, H_l); logits: w_i * (g_i - sum_k w_k g_k)) into `grad_loc`: neither the
   * locations nor the softmax weights exist as a tensor in either direction.  This is the
   * arithmetic of the canonical layer around the op (see msda_points_forward); the entry point
   * of the reference that takes packed points, `MultiScaleFlashAttnFunction`, applies no softmax
   * (tests/unipercept/nn/layers/test_nn_layers_deform_attn.py:105-122) and is served without
   * this flag.  Where msda_raw_points_supported() says so; `attn` / `grad_attn` are ignored. */
  MSDA_FLAG_RAW_POINTS = 64,
  /* msda_backward in two calls, for callers that want the binning pass of the binned path out of
   * the backward (e.g. issued right behind the forward of a training step) and for measuring
   * what it costs (tools/exp_fwdbin.py).  MSDA_FLAG_BIN_ONLY: only the tile directory and the
   * scatter records are written into `workspace`; no gradient is written (the gradient pointers
   * must still be valid).  MSDA_FLAG_BINNED: `workspace` already holds them — from a
   * MSDA_FLAG_BIN_ONLY call with the same points, shapes, `shapes_host` and layout flags — and
   * only the gradients are computed.  Both need the binned path in its register-resident form
   * (tuned shapes, L*P <= 32; MSDA_ELAYOUT otherwise).  Measured on MI355X at the Cityscapes
   * shapes: the backward gets 0.13 ms shorter (2.32 -> 2.19 ms) and the separate pass costs
   * 0.30 ms, so one call is the faster way; the kernels never read outside `workspace` or the
   * tensors whatever the workspace holds. */
  MSDA_FLAG_BINNED = 128,
  MSDA_FLAG_BIN_ONLY = 256,
};

/* Library ABI version (bumped when a signature changes or entry points are added). */
int msda_version(void);

/* Human-readable text for a return code of this library (static storage). */
const char* msda_error_string(int code);

/*
 * Batch-chunking knob of the reference signature
 * (`MultiScaleDeformAttnFunction.apply(..., im2col_step)`,
 * tests/unipercept/nn/layers/test_nn_layers_deform_attn.py:51-58 and :34,142 of the reference).
 * The kernels here take the whole batch in one launch, so the value only has to be valid:
 * step = min(N, im2col_step) must be >= 1 and divide N.  Returns 0 or MSDA_ESTEP.
 */
int msda_check_im2col_step(int N, int im2col_step);

/*
 * Forward: out[b,q,m*D:(m+1)*D] = sum_{l,p} attn[b,q,m,l,p] * bilinear(value_l[b,:,m,:], loc[b,q,m,l,p]).
 * Replaces `MultiScaleDeformAttnFunction.forward`
 * (tests/unipercept/nn/layers/test_nn_layers_deform_attn.py:51-58 of the reference; numerically the
 * same as `reference.deform_attn`, same file :40-49).
 *
 * dtype_val in {F32, F64, BF16, F16}; dtype_loc == dtype_val, or F32 with a 16-bit dtype_val.
 * Accumulation is fp32 (fp64 for F64).  `ref_points` is only read with MSDA_FLAG_RAW_POINTS
 * (NULL otherwise).
 */
int msda_forward(const void* value, const int64_t* shapes_dev, const int64_t* lsi_dev,
                 const void* loc, const void* attn, const void* ref_points, void* out, int N,
                 int S, int M, int D, int Lq, int L, int P, int dtype_val, int dtype_loc,
                 int flags, void* hip_stream);

/*
 * 1 when both msda_forward and msda_backward take MSDA_FLAG_VALUE_HEAD_MAJOR for this problem
 * (tuned kernels + binned scatter), else 0.  No counterpart in the reference: its value tensor is
 * always (N, S, M, D) (tests/unipercept/nn/layers/test_nn_layers_deform_attn.py:30).
 */
int msda_head_major_supported(int N, int S, int M, int D, int Lq, int L, int P, int dtype_val,
                              int dtype_loc);

/*
 * 1 when msda_forward and msda_backward take MSDA_FLAG_PACKED_POINTS for this problem (tuned
 * kernels, even L*P), else 0; callers then unpack into separate location / weight tensors.
 */
int msda_packed_points_supported(int N, int S, int M, int D, int Lq, int L, int P, int dtype_val,
                                 int dtype_loc);

/*
 * 1 when msda_forward and msda_backward take MSDA_FLAG_RAW_POINTS for this problem (tuned
 * kernels, even L*P <= 32, binned backward), else 0; callers then run msda_points_forward /
 * msda_points_backward around the packed-points form.  `dtype_loc` is the raw tensor's type.
 */
int msda_raw_points_supported(int N, int S, int M, int D, int Lq, int L, int P, int dtype_val,
                              int dtype_loc);

/*
 * Copies value between the two layouts: to_head_major != 0: src (N,S,M,D) -> dst (N,M,S,D);
 * to_head_major == 0: the reverse.  D * element size must be a multiple of 16 bytes and both
 * pointers 16-byte aligned.  `padding_mask` is optional (NULL allowed): N * S bytes, device
 * memory; the rows of a position whose byte is non-zero are written as zeros — the
 * `masked_fill(padding_mask, 0)` of the layer that calls the op, done in the same pass, so that
 * the value projection's output reaches the per-head layout, masked, in one epilogue pass.  No
 * counterpart in the reference (see above).
 */
int msda_value_relayout(const void* src, void* dst, const unsigned char* padding_mask, int N,
                        int S, int M, int D, int dtype_val, int to_head_major, void* hip_stream);

/*
 * Scratch the backward needs, in bytes (0 when none).  The binned grad_value path keeps its
 * scatter records (64 bytes per sampled point, worst case) and a tile directory there, and —
 * without `shapes_host`, with MSDA_FLAG_TILE_1D or MSDA_FLAG_DETERMINISTIC, and for head dims
 * above 32 channels per four lanes — a per-head copy of grad_out (as large as grad_out); the
 * atomic paths need an fp32 accumulation image of grad_value for 16-bit value dtypes.
 * `flags`, `shapes_host`: as for msda_backward (the size depends on them); pass the same values
 * to both.
 */
size_t msda_backward_workspace_bytes(int N, int S, int M, int D, int Lq, int L, int P,
                                     int dtype_val, int dtype_loc, int flags,
                                     const int64_t* shapes_host);

/*
 * Backward: grad_value (N,S,M,D), grad_loc (N,Lq,M,L,P,2), grad_attn (N,Lq,M,L,P) for a given
 * grad_out (N,Lq,M*D).  Replaces `MultiScaleDeformAttnFunction.backward` (exercised by gradcheck
 * at tests/unipercept/nn/layers/test_nn_layers_deform_attn.py:149-159 of the reference).
 *
 * All three outputs are fully overwritten; grad_value is zeroed by the callee (no need to
 * pre-zero).  `workspace` must hold msda_backward_workspace_bytes(...) bytes, 256-B aligned
 * (may be NULL when that is 0); its contents on entry are ignored and on return undefined.
 * grad_loc / grad_attn have dtype_loc, grad_value has dtype_val.
 *
 * `shapes_host` is optional (NULL allowed): a HOST copy of the L (H, W) pairs in `shapes_dev`.
 * With it, and when `lsi_dev` is the exclusive prefix sum of H*W (the usual pyramid), the binned
 * scatter cuts grad_value into 8 x 8 pixel patches of a level instead of runs of the
 * flattened sequence, which saves about a third of its record traffic (1.27 records per sampled
 * point instead of about 2 on a 1024 x 2048 pyramid).  It must match
 * `shapes_dev`; a table that disagrees gives undefined grad_value contents (never an
 * out-of-bounds access).  The pointer is only read during the call.
 */
int msda_backward(const void* grad_out, const void* value, const int64_t* shapes_dev,
                  const int64_t* lsi_dev, const void* loc, const void* attn,
                  const void* ref_points, void* grad_value, void* grad_loc, void* grad_attn,
                  void* workspace, size_t workspace_bytes, int N, int S, int M, int D, int Lq,
                  int L, int P, int dtype_val, int dtype_loc, int flags,
                  const int64_t* shapes_host, void* hip_stream);

/*
 * msda_backward for callers that cannot allocate device memory: code under HIP graph capture,
 * tight serving / fine-tuning loops and harnesses that forbid hipMalloc.  It has no counterpart
 * in the reference.  Same arguments and results as msda_backward without the workspace pair: the
 * scratch (directory, scatter records, per-head grad_out copy, or the fp32 accumulation image
 * of the atomic paths) lives in a static 128 MiB device array inside the library, one copy per
 * GPU, reserved on every GPU of a process that loads the library (whether or not this entry
 * point is ever called) and held until the process ends.  The array used is that of the GPU
 * `hip_stream` belongs to (the current GPU for the default stream).  The call itself allocates
 * and frees nothing (no hipMalloc / hipFree, no allocating library routine).
 *
 * Limit: msda_backward_workspace_bytes(...) for the problem must be <= 128 MiB (134 217 728
 * bytes), else MSDA_ENOALLOC is returned and nothing is launched; the backward of the
 * CPU-reference configuration (512 x 1024 input: S = Lq = 10752, 8 heads x 32, 3 levels x 4
 * points, batch 1, fp32) needs 78 MiB.  The first call on a device looks the array's address up
 * (hipGetSymbolAddress): make that call outside graph capture.
 *
 * Every call on one GPU uses that GPU's one array: calls that overlap in time on the same GPU
 * (two streams, two host threads) race on it and give undefined gradients; calls on one stream,
 * or on different GPUs, are safe.  The array keeps the last call's contents between calls; the
 * kernels never read a scratch word they have not written in the same call (the atomic paths
 * clear their accumulation image first).
 */
int msda_backward_noalloc(const void* grad_out, const void* value, const int64_t* shapes_dev,
                          const int64_t* lsi_dev, const void* loc, const void* attn,
                          const void* ref_points, void* grad_value, void* grad_loc,
                          void* grad_attn, int N, int S, int M, int D, int Lq, int L, int P,
                          int dtype_val, int dtype_loc, int flags, const int64_t* shapes_host,
                          void* hip_stream);

/*
 * Sampling-point epilogue of the layer that calls the op.  `raw` (N, Lq, M, 3*L*P) is the output
 * of ONE projection that computes sampling offsets and attention logits together, laid out per
 * (batch, query, head) row as L*P (dx, dy) offsets followed by L*P logits; `ref_points`
 * (N, Lq, L, 2) holds the query's reference point on every level.  Writes the packed
 * locations+weights tensor that MSDA_FLAG_PACKED_POINTS takes:
 *     (x, y) = ref + (dx / W_l, dy / H_l),   weights = softmax over the L*P logits of the row.
 * One pass over the tensor instead of the softmax / divide / add / concatenate sequence of the
 * canonical layer.  The reference's own layer body is absent from the snapshot (its package
 * `unipercept.nn.layers.deform_attn` is imported at
 * tests/unipercept/nn/layers/test_nn_layers_deform_attn.py:6-10 only); this follows the
 * Deformable-DETR / Mask2Former pixel-decoder convention its models use.  All tensors have
 * `raw` has `dtype_raw` (F32, BF16 or F16); `ref_points` and `packed` have `dtype_packed`, which
 * is `dtype_raw` or F32 (a 16-bit projection output with full-resolution locations and weights:
 * a bf16 location near 1.0 moves in steps of 1/256, a whole pixel on a 256-wide level).  `raw`
 * and `packed` must be 16-byte aligned and may not overlap.
 */
int msda_points_forward(const void* raw, const void* ref_points, const int64_t* shapes_dev,
                        void* packed, int N, int Lq, int M, int L, int P, int dtype_raw,
                        int dtype_packed, void* hip_stream);

/*
 * Gradient of msda_points_forward with respect to `raw`, from the gradient of `packed` and
 * `packed` itself (its weights part):  d(dx, dy) = d(x, y) / (W_l, H_l),
 * d logit_i = w_i * (g_i - sum_k w_k g_k).  The gradient of `ref_points`, where a caller needs
 * it, is the sum of d(x, y) over heads and points.  `grad_packed` and `packed` have
 * `dtype_packed`, `grad_raw` has `dtype_raw` (combinations as for msda_points_forward).
 */
int msda_points_backward(const void* grad_packed, const void* packed, const int64_t* shapes_dev,
                         void* grad_raw, int N, int Lq, int M, int L, int P, int dtype_raw,
                         int dtype_packed, void* hip_stream);

/* ---------------------------------------------------------------------------------------------
 * Point sampling (PointRend): the same bilinear gather on one NCHW map with border padding.
 * --------------------------------------------------------------------------------------------- */

/* interpolation modes of msda_point_sample_* */
enum {
  MSDA_SAMPLE_BILINEAR = 0,
  MSDA_SAMPLE_NEAREST = 1,
};

/*
 * out[n, c, p] = input[n, c] sampled at point_coords[n, p] = (x, y) in [0, 1]^2.  Replaces
 * `unipercept.vision.pointrend.sample(input, point_coords, align_corners, mode=...)`
 * (sources/unipercept/vision/pointrend.py:25-76 of the reference), i.e.
 * grid_sample(input, 2 * point_coords - 1, mode, padding_mode="border", align_corners):
 * pixel coordinate ((2c - 1 + 1) * size - 1) / 2 (align_corners: (2c - 1 + 1) / 2 * (size - 1)),
 * clamped to [0, size - 1]; bilinear, or nearest with ties to even.
 *
 * input (N, C, H, W), coords (N, P, 2), out (N, C, P), all of `dtype` (F32, F64, BF16, F16;
 * 16-bit types compute in fp32), contiguous.  A (N, Hg, Wg, 2) coordinate grid is the same call
 * with P = Hg * Wg.
 */
int msda_point_sample_forward(const void* input, const void* coords, void* out, int N, int C,
                              int H, int W, int P, int dtype, int mode, int align_corners,
                              void* hip_stream);

/*
 * Gradients of msda_point_sample_forward for a given grad_out (N, C, P): grad_input (N, C, H, W)
 * and grad_coords (N, P, 2), both fully overwritten (zeroed by the callee where it accumulates).
 * The coordinate gradient is 0 where the border clamp acts (pixel coordinate <= 0 or
 * >= size - 1) and for MSDA_SAMPLE_NEAREST, as in aten::grid_sampler_2d_backward, and includes
 * the factor 2 of `2 * point_coords - 1`.  dtype F32 or F64 (a 16-bit caller converts).
 * grad_input is accumulated with hardware float atomics: sums of more than two contributions to
 * one pixel may differ in the last bits from run to run.
 */
int msda_point_sample_backward(const void* grad_out, const void* input, const void* coords,
                               void* grad_input, void* grad_coords, void* workspace,
                               size_t workspace_bytes, int N, int C, int H, int W, int P,
                               int dtype, int mode, int align_corners, void* hip_stream);

/*
 * Optional scratch of msda_point_sample_backward, in bytes (0 when it has no use for any): the
 * points of every sample listed per band of map rows, so that the workgroup that owns a band
 * walks only the points that touch it.  `workspace` may be NULL (or too small): the call is
 * then still correct, every band walks all points of its sample (about 1.5x the time on a
 * 256 x 512 map with 12 544 points).  4-byte aligned; contents on entry ignored, on return
 * undefined.  No counterpart in the reference.
 */
size_t msda_point_sample_workspace_bytes(int N, int C, int H, int W, int P, int dtype);

#ifdef __cplusplus
}
#endif
#endif /* MSDA_H_ */
