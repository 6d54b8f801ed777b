/*
 * irn_hip.h — C ABI of libirn_hip.so: the MI355X (gfx950) implementation of IRN's pseudo-label
 * hot path (inter-pixel affinity random walk + label epilogues + instance front-end).
 *
 * The reference (jiwoon-ahn/irn) is pure Python and has no FFI layer of its own; the seam this
 * library sits behind is the reference's *operator tier* (SURVEY.md §8b):
 *   misc/indexing.py            PathIndex, edge_to_affinity, affinity_sparse2dense,
 *                               to_transition_matrix, propagate_to_edge
 *   step/make_sem_seg_labels.py lines 43-49   (x4 upsample, /max, bg plane, argmax, keys LUT)
 *   step/make_ins_seg_labels.py find_centroids_with_refinement, cluster_centroids,
 *                               separte_score_by_mask, lines 137-147
 * Each entry point below names the reference lines it replaces.  A reference-side binding is a
 * ctypes stub; irn_amd/_lib.py is the one this repo ships.
 *
 * Conventions
 *   - every function returns an int status (IRN_OK = 0); irn_last_error() gives the message of
 *     the calling thread's last failure.  Nothing throws across the boundary.
 *   - all `dev` pointers are caller-owned device memory (fp32 unless noted), contiguous,
 *     row-major.  The library never allocates or frees device memory the caller can see;
 *     scratch comes from the caller through a workspace pointer whose size the library reports.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); all work is
 *     enqueued on it and nothing synchronises unless stated.
 *   - one host thread per context; contexts are independent (one process per GPU, no
 *     communication on the path — reference misc/torchutils.py:66-68 sharding).
 */
#ifndef IRN_HIP_H
#define IRN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IRN_OK 0
#define IRN_ERR_ARG 1          /* bad argument (null pointer, non-positive size, unsupported radius/beta) */
#define IRN_ERR_HIP 2          /* a HIP runtime call or kernel launch failed */
#define IRN_ERR_STATE 3        /* call order violated (e.g. run before configure / workspace too small) */

#define IRN_MAX_RADIUS 16

int irn_version(void);
const char *irn_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * PathIndex  (replaces misc/indexing.py:6-56, `PathIndex.get_search_paths_dst`)
 * order = 0: the reference's channel order (paths grouped by length, discovery order inside a
 *            group) — the order of edge_to_affinity's channel axis and of `search_dst`;
 * order = 1: raster (dy, dx) order — the plane order of the walk's internal weight table.
 * irn_path_count reports sizes; irn_path_table fills host arrays:
 *   dst_dydx   [n_dirs][2]   destination (dy,dx) of every direction
 *   path_start [n_dirs+1]    CSR offsets into cells_dydx
 *   cells_dydx [n_cells][2]  path cells, far-to-near (element 0 of a path is its destination)
 * ------------------------------------------------------------------------------------------- */
int irn_path_count(int radius, int *n_dirs, int *n_cells);
int irn_path_table(int radius, int order, int32_t *dst_dydx, int32_t *path_start, int32_t *cells_dydx);

/* ---------------------------------------------------------------------------------------------
 * edge_to_affinity  (replaces misc/indexing.py:91-109 and the identical gather in
 * net/resnet50_irn.py:162-175).  edge: dev [B, Hp, Wp]; aff: dev [B, n_dirs, (Hp-rf)*(Wp-2rf)],
 * rf = radius-1, channel order 0 (reference).  aff[b,d,s] = 1 - max over path(d) of edge.
 * No index tensors are built or uploaded (reference :58-88, :96-99).
 * ------------------------------------------------------------------------------------------- */
int irn_edge_to_affinity(const float *edge_dev, int batch, int hp, int wp, int radius,
                         float *aff_dev, void *stream);

/* Its vector-Jacobian product, for the training seam (net/resnet50_irn.py:162-175 under autograd: the
 * gradient of aff[b,d,s] goes, negated, to the first path cell that attains the maximum).
 * grad_aff: dev [B, n_dirs, (Hp-rf)*(Wp-2rf)] -> grad_edge: dev [B, Hp, Wp] (overwritten).  The arg-max
 * is recomputed from `edge`; nothing from the forward call is needed. */
int irn_edge_to_affinity_backward(const float *edge_dev, const float *grad_aff_dev, int batch, int hp, int wp,
                                  int radius, float *grad_edge_dev, void *stream);

/* Pair displacement of the training seam (AffinityDisplacementLoss.to_pair_displacement,
 * net/resnet50_irn.py:177-193).  disp: dev fp32 [batch, channels, hp, wp]; out: dev fp32
 * [batch, channels, |S|, (hp-rf)*(wp-2rf)], out[b,c,d,y,x] = disp[b,c,y,rf+x] - disp[b,c,y+dy_d,rf+x+dx_d]
 * with d in the reference's channel order (search_dst).  The backward writes the full gradient
 * [batch, channels, hp, wp] (no accumulation, no atomics). */
int irn_pair_displacement(const float *disp_dev, int batch, int channels, int hp, int wp, int radius, float *out_dev,
                          void *stream);
int irn_pair_displacement_backward(const float *grad_out_dev, int batch, int channels, int hp, int wp, int radius,
                                   float *grad_disp_dev, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Random-walk context  (replaces misc/indexing.py:141-165 `propagate_to_edge` and everything it
 * calls: PathIndex :148, edge_to_affinity :151, affinity_sparse2dense :154, to_transition_matrix
 * :160, the final matmul :164 — as sparse stencil sweeps, never densified).
 *
 * A context is bound to one radius and serves batches of independent images ("units").
 *   irn_walk_create      host-side tables for `radius` are built and uploaded once
 *   irn_walk_configure   describe a batch: n images with grid (h[i], w[i]) and c[i] walk channels
 *                        (c = #classes for semantic labels, #classes x #instances for instance
 *                        labels); returns the scratch bytes the batch needs
 *   irn_walk_run         for every image i:  out[i][c,h,w] = (cam[i][c] * (1-edge[i])) . T_i^n_sweeps
 *                        with T_i the column-normalised beta-powered affinity of edge[i]
 *                        (n_sweeps = 2^exp_times).  Pointer arrays are HOST arrays of device
 *                        pointers.  `inst_map` (optional, may be NULL or hold NULLs): per image an
 *                        int32 [h,w] cluster map with k_inst[i] instances — then cam[i] is
 *                        [c/k_inst, h, w] and channel (cls*k_inst + k) starts from
 *                        cam[cls] * (inst_map == k)   (step/make_ins_seg_labels.py:77-80,:133).
 * beta must be > 0 (beta == 0 would make the reference's dense matrix all-ones).
 * ------------------------------------------------------------------------------------------- */
typedef struct irn_walk_ctx irn_walk_ctx;

int irn_walk_create(int radius, irn_walk_ctx **ctx_out);
int irn_walk_destroy(irn_walk_ctx *ctx);
int irn_walk_configure(irn_walk_ctx *ctx, int n_images, const int32_t *h, const int32_t *w,
                       const int32_t *c, size_t *workspace_bytes);
int irn_walk_run(irn_walk_ctx *ctx, const float *const *edge_dev, const float *const *cam_dev,
                 const int32_t *const *inst_map_dev, const int32_t *k_inst,
                 float *const *out_dev, float beta, int n_sweeps,
                 void *workspace_dev, size_t workspace_bytes, void *stream);

/* How x . T^n_sweeps is evaluated (round 3).  T is similar to a symmetric matrix with spectrum in [-1, 1], and
 * lambda^n = sum_k c_k T_k(lambda) (Chebyshev polynomials) with c_k = 2^(1-n) C(n, (n-k)/2) ~ exp(-k^2 / 2n): the terms
 * beyond K ~ sqrt(2 n ln(1/tol)) are dropped (their sum is the bound `tol` on the change of the result, 1e-7 by default:
 * below the rounding of the fp32 state) and the walk runs the three-term recurrence y_{t+1} = 2 T y_t - y_{t-1} for K
 * operator applications instead of n (84 instead of 256 at exp_times = 8), accumulating s = sum_k c_k y_k.  Measured
 * against the fp64 oracle the result is as close as that of the plain iteration (tests/test_gpu_schedule.py).
 *   option "accel" = 1 (default) / 0: truncated Chebyshev series / plain powers (n applications, bit-identical to the
 *   recurrence-free kernels of rounds 1-2); "accel_tol_exp" = e: tol = 10^-e (default 7: 84 applications for n = 256; 6: 78).
 *   irn_walk_steps reports the operator applications a run with `n_sweeps` would execute (for flop accounting).
 * The caller's edge / cam / inst_map device buffers must stay valid and unmodified until irn_walk_sync (or the next
 * irn_walk_run) returns: a run that irn_walk_sync has to repeat reads them again. */
int irn_walk_steps(irn_walk_ctx *ctx, int n_sweeps, int *n_steps);
/* The series itself (host only, no device needed): *n_steps = K operator applications, *recurrence = 1 for the
 * three-term recurrence / 0 for plain powers (n < 8 or nothing to gain), coef_out[0..K] = c_k renormalised to sum 1
 * (may be NULL; coef_cap = its capacity). */
int irn_power_series(int n, int tol_exp, double *coef_out, int coef_cap, int *n_steps, int *recurrence);

/* Tuning knobs (performance only, never results): name/value pairs, e.g. "variant" = 0 generic table-driven sweep
 * (any radius; the default for radii other than 5 and 10), 1 = register-blocked streaming sweep (radius 5/10),
 * 2 = weights-stationary persistent walk (radius 5/10; the DEFAULT there, with a per-batch fall-back to variant 1
 * when an image does not fit one round of workgroups or the grid cannot be co-resident); "cooperative" = 1 (default)
 * launches the persistent kernel with hipLaunchCooperativeKernel.  Unknown names fail. */
int irn_walk_set_option(irn_walk_ctx *ctx, const char *name, int value);

/* The weights-stationary persistent kernel ("variant" = 2) runs one workgroup per compute unit and its tiles wait for
 * each other inside the launch, with a bounded wait: when the grid does not become co-resident in time (another
 * process, stream or partition holds compute units) a launch gives up instead of hanging.
 *   irn_walk_sync   waits for the last irn_walk_run; if that launch gave up, runs the batch again on the streaming
 *                   sweeps (same operator, kernel boundaries instead of in-launch hand-offs) and waits for it, so the
 *                   outputs are valid whenever it returns IRN_OK.  *fell_back (may be NULL) = 1 when that happened.
 *                   This is the call to make before consuming the outputs of a run.
 *   irn_walk_check  no waiting, no repair: IRN_ERR_STATE if a launch completed so far gave up and irn_walk_sync has
 *                   not handled it (the outputs of that run are invalid) — what a benchmark wants, where a silent
 *                   fall-back would misreport.  Always IRN_OK for the streaming variants.
 *   irn_walk_fallback_runs   how many batches irn_walk_sync has re-run so far (monitoring). */
int irn_walk_sync(irn_walk_ctx *ctx, int *fell_back);
int irn_walk_check(irn_walk_ctx *ctx);
int irn_walk_fallback_runs(irn_walk_ctx *ctx);

/* Start-up self-checks of the weights-stationary walk's two measured assumptions (speed only, never results; the
 * reference has no counterpart — its walk is dense sgemm, misc/indexing.py:132-139):
 *   placement   0 not checked yet, 1 "blocks with equal b % 8 share an XCD, the eight residues eight XCDs" holds on this device (the tiles of an image are
 *               then packed onto one XCD), 2 it does not (tiles keep launch order; one line on stderr says so);
 *   poll_delay  the delay (units of 64 clocks) between a single-channel tile's stores and its first poll in use now;
 *   probe_ms4   launch times in ms the start-up probe measured for delays 8, 10, 12, 14 on this context's first representative
 *               batch (zeros: this context did not probe — too small a batch, another context of the process probed
 *               before, or option "poll_delay" pinned the value; "poll_delay_auto" = 0 switches the probe off).
 * Any pointer may be null. */
int irn_walk_tuning(irn_walk_ctx *ctx, int *poll_delay, int *placement, float *probe_ms4);

/* How the weights-stationary walk packs a batch onto the device — host arithmetic only, no device and no context needed
 * (the CPU tests check it; the reference has no counterpart: it walks one image at a time, step/make_sem_seg_labels.py:41).
 * Images i = 0..n_images-1 of h[i] x w[i] grid pixels and channels[i] walk channels are cut into tiles (one workgroup each)
 * and packed into rounds of at most n_workgroups tiles; the workgroups run their rounds back to back, so inside a round the
 * heaviest image goes to the slot range that is free first.  placement as irn_walk_tuning reports it (1: consecutive slots
 * share an XCD, 2: launch order).  jobs_out: int32 [cap_rounds][n_workgroups][4] = {image or -1, tile y0, tile x0, tiles of
 * the image}; *n_rounds = rounds needed, 0 when the batch does not fit the persistent launch (an image with more tiles than
 * workgroups, or narrower than the radius: such batches run on the streaming sweeps).  cap_rounds = 0 only asks for *n_rounds. */
int irn_walk_plan_rounds(int radius, int n_images, const int32_t *h, const int32_t *w, const int32_t *channels,
                         int n_workgroups, int placement, int32_t *jobs_out, int cap_rounds, int *n_rounds);

/* Diagnostic (option "profile" = 1, resident walk only): per-sweep time stamps of two workgroups of
 * the first round — host_out is int64 [2][256][4] = {sweep start, state staged, first partial sums
 * in LDS, sweep stored} in ticks of the 100 MHz wall clock.  Synchronises the device. */
int irn_walk_read_profile(irn_walk_ctx *ctx, long long *host_out);

/* Kernel timing hook for bench.py: when enabled, every irn_walk_run brackets its sweep kernels
 * with a pair of HIP events on `stream` (no synchronisation).  irn_walk_last_sweep_ms waits for the
 * pairs recorded since its previous call, returns their summed elapsed time and the number of
 * sweeps (one sweep = one pass of the transition operator over the whole batch) they cover, and
 * forgets them. */
int irn_walk_enable_timing(irn_walk_ctx *ctx, int enable);
int irn_walk_last_sweep_ms(irn_walk_ctx *ctx, float *ms, int *n_launches);

/* Intermediate products of the last irn_walk_run, for parity tests of the build stage:
 * weights of image i as [n_dirs, h, w] in raster direction order (order = 1) and 1/degree
 * as fp64 [h, w], copied device-to-device out of the workspace. */
int irn_walk_export_weights(irn_walk_ctx *ctx, int image, float *w_dev, double *inv_deg_dev,
                            void *workspace_dev, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Label epilogue  (replaces step/make_sem_seg_labels.py:43-49 and
 * step/make_ins_seg_labels.py:137-145).  For every image: bilinear x4 upsample
 * (align_corners=False) of rw [c,h,w] cropped to (out_h,out_w); divide by the global max;
 * argmax over {bg_thres, channels} with the first maximum winning.
 *   labels_dev[i]  uint8 [out_h,out_w] = 0 for background else keys[i][argmax-1]+1
 *                  (keys: host array of device int64 pointers, the CAM dict's `keys`); may be NULL
 *   argmax_dev[i]  int32 [out_h,out_w] raw argmax index (0 = background); may be NULL
 *   rw_up_dev[i]   fp32 [c,out_h,out_w] normalised upsampled scores (instance scoring); may be NULL
 * scratch: n_images * 4 bytes of device memory (per-image max), provided by the caller.
 * ------------------------------------------------------------------------------------------- */
int irn_label_epilogue(int n_images, const float *const *rw_dev, const int32_t *c, const int32_t *h,
                       const int32_t *w, const int32_t *out_h, const int32_t *out_w, float bg_thres,
                       const int64_t *const *keys_dev, uint8_t *const *labels_dev,
                       int32_t *const *argmax_dev, float *const *rw_up_dev, void *scratch_dev,
                       void *stream);

/* ---------------------------------------------------------------------------------------------
 * Multi-scale CAM merge  (replaces step/make_cam.py:38-52).  src[s]: dev fp32 [n_classes, hs[s], ws[s]]
 * activation maps of scale s (original + flipped-back, net/resnet50_cam.py:66-70); keys: dev int64
 * [n_keys] present classes (torch.nonzero(label)).  Writes
 *   cam      dev fp32 [n_keys, ceil(out_h/4), ceil(out_w/4)]   sum over scales of bilinear resizes
 *   high_res dev fp32 [n_keys, out_h, out_w]                   same at 16*ceil(./16), cropped
 * each channel divided by (its maximum + 1e-5).  scratch: 8 * n_keys bytes.  At most 8 scales.
 * ------------------------------------------------------------------------------------------- */
int irn_cam_merge(int n_scales, const float *const *src_dev, const int32_t *hs, const int32_t *ws, int n_classes,
                  const int64_t *keys_dev, int n_keys, int out_h, int out_w, float *cam_dev, float *high_res_dev,
                  void *scratch_dev, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Multi-scale input pipeline  (replaces the per-scale loop of VOC12ClassificationDatasetMSF.__getitem__,
 * voc12/dataloader.py:191-201: pil_rescale misc/imutils.py:8-22 -> TorchvisionNormalize
 * voc12/dataloader.py:65-78 -> HWC_to_CHW -> stack([img, flip(img, -1)])).
 *
 * The resampling is Pillow's 8-bit bicubic (Image.resize(size, Image.BICUBIC): separable, 22-bit fixed-point
 * weights, int32 accumulation, 8-bit intermediate) and is bit-exact.
 *
 *   irn_bicubic_plan        host only: the fixed-point tap table of one axis (lo[out], count[out],
 *                           weights[out * ksize]); call with null arrays to query ksize.
 *   irn_bicubic_resize_u8   dev u8 [h, w, channels] -> dev u8 [hs, ws, channels]  (= misc/imutils.py pil_resize
 *                           order 3); channels 1, 3 or 4; scratch: irn_bicubic_scratch_bytes.
 *   irn_msf_pack            dev u8 [h, w, 3] -> for each scale s a dev fp32 [2, 3, hs[s], ws[s]]: resized,
 *                           normalised through lut (dev fp32 [3 * 256], lut[c * 256 + v] = the fp32 value of
 *                           (v / 255. - mean[c]) / std[c] computed in double), channel-major, image followed by
 *                           its horizontal flip.  hs/ws equal to h/w skip the resize like the reference's
 *                           `s == 1` branch.  scratch: max over scales of irn_bicubic_scratch_bytes(.., 3).
 * ------------------------------------------------------------------------------------------- */
int irn_bicubic_plan(int in_size, int out_size, int32_t *ksize, int32_t *lo, int32_t *count, int32_t *weights,
                     size_t weights_capacity);
size_t irn_bicubic_scratch_bytes(int h, int w, int hs, int ws, int channels);
int irn_bicubic_resize_u8(const uint8_t *img_dev, int h, int w, int channels, int hs, int ws, uint8_t *out_dev,
                          void *scratch_dev, void *stream);
int irn_msf_pack(const uint8_t *img_dev, int h, int w, int n_scales, const int32_t *hs, const int32_t *ws,
                 const float *lut_dev, float *const *out_dev, void *scratch_dev, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Trunk epilogue  (replaces the elementwise tail of reference net/resnet50.py:34-54 Bottleneck.forward —
 * FixedBatchNorm :11-14, `out += residual`, ReLU — and of the stem :94-97, on the inference path)
 *
 *   x dev fp32 [n_images, n_channels, plane_elems] (a convolution's output, contiguous NCHW), IN PLACE:
 *       x[n, c, i] = act(x[n, c, i] * scale[c] + shift[c] (+ r[n, c, i]))      act = ReLU if relu else identity
 *       r = res, or res * res_scale[c] + res_shift[c] when res_scale / res_shift are given (the projection shortcut's own
 *       batch norm, net/resnet50.py:48-49, folded into the same pass)
 *   scale / shift (/ res_scale / res_shift) dev fp32 [n_channels]: weight / sqrt(running_var + eps) and
 *   bias - running_mean * scale, folded by the caller; res (may be NULL) like x.  One fused multiply-add per element and
 *   operand, then one addition; NaNs propagate.  x and res 16-byte aligned; at most 2^31 - 1 elements per call.
 * ------------------------------------------------------------------------------------------- */
int irn_bn_act(float *x_dev, const float *res_dev, const float *scale_dev, const float *shift_dev, const float *res_scale_dev,
               const float *res_shift_dev, int64_t n_images, int n_channels, int64_t plane_elems, int relu, void *stream);

/* The same pass over a channels-last tensor: x dev fp32 [n_pixels, n_channels] (= an [N, C, H, W] tensor in
 * torch.channels_last memory format, n_pixels = N*H*W), n_channels a multiple of 4, constants 16-byte aligned.  Same
 * arithmetic (one fmaf per element), same reference lines (net/resnet50.py:11-14, :34-54); for a trunk whose convolutions
 * run on MIOpen's NHWC solvers (IRN_CHANNELS_LAST=1). */
int irn_bn_act_nhwc(float *x_dev, const float *res_dev, const float *scale_dev, const float *shift_dev, const float *res_scale_dev,
                    const float *res_shift_dev, int64_t n_pixels, int n_channels, int relu, void *stream);

/* 1x1 convolution of a channels-last activation with its whole elementwise tail in the GEMM's epilogue (hipBLASLt, fp32
 * compute): conv1 -> bn1 -> ReLU and conv3 -> bn3 -> (+ residual) -> ReLU of Bottleneck.forward, reference
 * net/resnet50.py:34-54 (FixedBatchNorm :11-14 is a constant affine map: its scale is folded into `w` by the caller in
 * double precision, its shift is `bias`), and the projection shortcut :48-49.
 *   x dev fp32 [m, cin] (= [N, cin, H, W] in torch.channels_last, m = N*H*W), w dev fp32 [cout, cin] (the convolution's
 *   weight, scale folded in), bias dev fp32 [cout] or NULL, residual dev fp32 [m, cout] or NULL (may be `out`),
 *   out dev fp32 [m, cout]:      out = act(x . w^T + bias (+ residual)),   act = ReLU if relu else identity.
 *   algo_rank: 0 = hipBLASLt's first heuristic choice for the problem, k = its k-th (irn_conv1x1_algo_count gives how many
 *   there are): a function of the problem only, never of a timing made in this process, so that every process computes
 *   the same bits.  workspace: caller device memory of irn_conv1x1_workspace_bytes() (may be 0 / NULL: fewer kernels
 *   qualify).  Enqueued on `stream`; nothing synchronises. */
size_t irn_conv1x1_workspace_bytes(void);
int irn_conv1x1_algo_count(int64_t m, int cin, int cout, int has_bias, int has_residual, int relu, size_t workspace_bytes,
                           int *count_out);
int irn_conv1x1_nhwc(const float *x_dev, const float *w_dev, const float *bias_dev, const float *residual_dev, float *out_dev,
                     int64_t m, int cin, int cout, int relu, int algo_rank, void *workspace_dev, size_t workspace_bytes,
                     void *stream);

/* Split-precision form of the same layers (round 6): the fp32 GEMM runs at <= 0.9 of the 157 TFLOP/s fp32 matrix peak; an fp16
 * MFMA GEMM with fp32 accumulation runs ~2.5x faster over a 3x longer K, and two fp16 numbers carry 22 mantissa bits:
 *       x = x_hi + 2^-11 x_lo',  x_hi = fp16(x),  x_lo' = fp16((x - x_hi) 2^11)          (irn_split16, per element)
 *       x . w ~ x_hi w_hi + x_hi w_lo + x_lo' (w_hi 2^-11)                                (the dropped term: 2^-22 |x w|)
 * Measured on every 1x1 layer of the CAM network: 5.4e-6 / 9.2e-6 from fp64 on the normalised CAM where the fp32 GEMMs give
 * 5.9e-6 / 7.6e-6 (tools/bf16x3_cam_error.py; bar 1e-4); per layer 3.4e-7 .. 1.9e-6 against 2.6e-7 .. 1.3e-6 (profiles/r06_s2_*).
 * Same reference lines as irn_conv1x1_nhwc (net/resnet50.py:34-54).
 *
 * irn_split16: x dev fp32 [n_pixels, n_channels] (channels-last activation, n_channels a multiple of 8, 16-byte aligned)
 *   -> out dev fp16 [n_pixels, 3 n_channels] = [hi | hi | lo'] per pixel.  With scale / shift (dev fp32 [n_channels], both or
 *   neither) the element is first y = x * scale[c] + shift[c] (one fmaf, the FixedBatchNorm of the 3x3 convolution in front,
 *   net/resnet50.py:40-42) and, with relu, max(y, 0) — the fp32 tensor is read once and never written back.  overflow_dev
 *   (dev uint32, may be NULL) has bit 0 set when an element was beyond fp16's range (|y| > 65504) or NaN: the caller's
 *   signal that this activation needs the fp32 path.
 * irn_gemm16_nhwc: a16 dev fp16 [m, k] (irn_split16's output, k = 3 cin), b16 dev fp16 [cout, k] = [w_hi | w_lo | w_hi 2^-11] of
 *   the batch-norm-folded weight scaled by 2^p (prepared by the caller in double precision), alpha = 2^-p:
 *       out = act(alpha a16 . b16^T + bias (+ residual)),  fp32 [m, cout]; algo_rank / workspace as irn_conv1x1_nhwc. */
int irn_split16(const float *x_dev, const float *scale_dev, const float *shift_dev, int relu, void *out_dev, int64_t n_pixels,
                int n_channels, unsigned *overflow_dev, void *stream);
/* The same pass between a dense map [n_images, h, w, n_channels] and its zero-bordered form [n_images, h+2, w+2, .] (pixel (n, y, x)
 * at row n (h+2)(w+2) + (y+1)(w+2) + x+1): in_padded = x is the bordered fp32 form (interior read), out_padded = out is the
 * bordered fp16 form (interior split from the pixels, border rows written as zeros: the buffer needs no preparation).  On the
 * bordered form a 3x3 / pad 1 / stride 1 convolution (conv2 of Bottleneck.forward, net/resnet50.py:40) is NINE fp16 GEMMs that
 * accumulate over row-shifted views of one operand — tap (ky, kx) reads row r + (ky-1)(w+2) + (kx-1):
 * irn_conv3x3_split_gemm: a16 dev fp16 = the bordered operand [n (h+2)(w+2), 3 cin] with (w+3) rows of valid memory in front of
 *   and behind it (any content), w16 dev fp16 [9, cout, 3 cin] (one irn_gemm16_nhwc operand per tap, raster order, one common
 *   scale), out dev fp32 [n (h+2)(w+2), cout] = alpha sum_taps a16[shifted] . w16[tap]^T, accumulated tap after tap in fp32 in a
 *   fixed order; border rows of `out` hold garbage.  row_fused = 1: w16 = [3, cout, 9 cin] (the three taps of a kernel row side by
 *   side) and THREE GEMMs over K = 9 cin — the taps (ky, 0..2) of a pixel are three consecutive memory rows, so a kernel row's
 *   operand is the same buffer read with leading dimension 3 cin and 9 cin columns.  algo_rank / workspace as irn_conv1x1_nhwc. */
int irn_split16_pad(const float *x_dev, const float *scale_dev, const float *shift_dev, int relu, void *out_dev, int64_t n_images,
                    int h, int w, int n_channels, int in_padded, int out_padded, unsigned *overflow_dev, void *stream);
int irn_conv3x3_split_gemm(const void *a16_dev, const void *w16_dev, float *out_dev, int64_t n_images, int h, int w, int cin, int cout,
                           float alpha, int row_fused, int algo_rank, void *workspace_dev, size_t workspace_bytes, void *stream);
int irn_gemm16_algo_count(int64_t m, int k, int cout, int has_bias, int has_residual, int relu, size_t workspace_bytes,
                          int *count_out);
int irn_gemm16_nhwc(const void *a16_dev, const void *b16_dev, const float *bias_dev, const float *residual_dev, float *out_dev,
                    int64_t m, int k, int cout, int relu, float alpha, int algo_rank, void *workspace_dev, size_t workspace_bytes,
                    void *stream);

/* Stem: batch norm + ReLU + max pool 3x3 / stride 2 / pad 1 in one pass (reference net/resnet50.py:94-97; the nets'
 * stage1, net/resnet50_cam.py:14, net/resnet50_irn.py:15).
 *   x dev fp32 [n_images, n_channels, h, w] (conv1's output) -> out dev fp32 [n_images, n_channels, (h-1)/2+1, (w-1)/2+1]
 *   = max over the window's in-image taps of relu(x * scale[c] + shift[c]); NaNs propagate. */
int irn_stem_pool(const float *x_dev, const float *scale_dev, const float *shift_dev, int64_t n_images, int n_channels, int h,
                  int w, float *out_dev, void *stream);

/* IRNet heads: bilinear upsampling by an integer factor, align_corners = False, (+ ReLU) — nn.Upsample(scale_factor = f,
 * mode = 'bilinear') followed by nn.ReLU, reference net/resnet50_irn.py:36,42,48,72,78,84.  ATen's source index and
 * weights (src = (dst + 0.5) / f - 0.5 clamped at 0) and its operation order, in fp32 without contraction.
 *   x dev fp32 [n_planes, h, w] -> out dev fp32 [n_planes, h * f, w * f] (16-byte aligned) */
int irn_upsample_bilinear(const float *x_dev, int64_t n_planes, int h, int w, int factor, int relu, float *out_dev,
                          void *stream);

/* ---------------------------------------------------------------------------------------------
 * Instance front-end
 *   irn_find_centroids   replaces step/make_ins_seg_labels.py:18-56
 *       dp dev [2,h,w] -> centroids dev int32 [2,h,w]; float32 state, float64 increment in the
 *       reference's operation order, no FMA contraction, round-half-even at the end.
 *   irn_cluster_centroids replaces step/make_ins_seg_labels.py:58-75 (+ misc/imutils.py:182-190):
 *       weak = |dp| < thres; 4-connected components numbered in raster order of first pixel;
 *       label at each pixel's centroid; renumbered 0..K-1 ascending.
 *       -> cluster_map dev int32 [h,w], *k_out (host) = K.  Synchronises `stream` (K is returned).
 *       scratch: irn_cluster_scratch_bytes(h, w) bytes.
 * ------------------------------------------------------------------------------------------- */
int irn_find_centroids(const float *dp_dev, int h, int w, int iterations, int32_t *centroids_dev,
                       void *stream);
size_t irn_cluster_scratch_bytes(int h, int w);
int irn_cluster_centroids(const int32_t *centroids_dev, const float *dp_dev, int h, int w, float thres,
                          int32_t *cluster_map_dev, int *k_out, void *scratch_dev, void *stream);

/* Batched forms (the loop over images of step/make_ins_seg_labels.py:119-133): HOST arrays of device pointers and
 * sizes, one launch sequence for the whole batch, and NOTHING synchronises — the instance counts stay on the device
 * (k_dev: dev int32 [n_images]) and the caller fetches all of them with one transfer before it configures the walk
 * (c[i] = n_classes[i] * K[i]).  scratch: irn_cluster_batch_scratch_bytes. */
int irn_find_centroids_batch(int n_images, const float *const *dp_dev, const int32_t *h, const int32_t *w,
                             int iterations, int32_t *const *centroids_dev, void *stream);
size_t irn_cluster_batch_scratch_bytes(int n_images, const int32_t *h, const int32_t *w);
int irn_cluster_centroids_batch(int n_images, const int32_t *const *centroids_dev, const float *const *dp_dev,
                                const int32_t *h, const int32_t *w, float thres, int32_t *const *cluster_map_dev,
                                int32_t *k_dev, void *scratch_dev, void *stream);

/* 4-connected component labelling of a byte mask [n,h,w] (non-zero = foreground), ids 1.. per
 * image in raster order of first pixel, 0 = background — the skimage.measure.label(connectivity=1,
 * background=0) call of step/make_ins_seg_labels.py:66,92.  n_labels_dev: int32 [n] component
 * counts.  scratch: irn_ccl_scratch_bytes(n,h,w). */
size_t irn_ccl_scratch_bytes(int n, int h, int w);
int irn_label4(const uint8_t *mask_dev, int n, int h, int w, int32_t *labels_dev, int32_t *n_labels_dev,
               void *scratch_dev, void *stream);

/* ---------------------------------------------------------------------------------------------
 * detect_instance  (replaces step/make_ins_seg_labels.py:82-105 and the one-hot of :145-147).
 * rw_up dev fp32 [n_channels,h,w] (normalised scores), argmax dev int32 [h,w] (0 = background,
 * c+1 = channel c) -> one detection per 4-connected component of every channel's mask, ordered
 * channel ascending then raster order of the component's first pixel (skimage label order).
 *   _count : labels the components (state kept in `scratch`); *n_det_out (host) = number of
 *            detections.  Synchronises `stream`.
 *   _emit  : after _count with the same inputs and scratch: score[d] = area < min_area ? 0 :
 *            max(score * mask) (:96-99), channel[d] (index into the caller's class_id array), mask
 *            dev uint8 [n_det,h,w].  Only call with n_det >= 1.
 * scratch: irn_detect_scratch_bytes(n_channels, h, w).
 * ------------------------------------------------------------------------------------------- */
size_t irn_detect_scratch_bytes(int n_channels, int h, int w);
int irn_detect_instance_count(const float *rw_up_dev, const int32_t *argmax_dev, int n_channels, int h, int w,
                              int *n_det_out, void *scratch_dev, void *stream);
int irn_detect_instance_emit(const float *rw_up_dev, const int32_t *argmax_dev, int n_channels, int h, int w, int n_det,
                             double min_area, float *score_dev, int32_t *channel_dev, uint8_t *mask_dev,
                             void *scratch_dev, void *stream);


/* Batched forms: one labelling pass sequence for all images of a batch.  _batch_count writes the detection counts
 * to n_det_dev (dev int32 [n_images]) and does NOT synchronise; the caller reads them with one transfer, sizes the
 * outputs and calls _batch_emit with the same inputs and scratch plus the counts (host) — images with n_det[i] == 0
 * are skipped (their output pointers may be NULL).  min_area: host [n_images].
 * scratch: irn_detect_batch_scratch_bytes. */
size_t irn_detect_batch_scratch_bytes(int n_images, const int32_t *n_channels, const int32_t *h, const int32_t *w);
int irn_detect_instance_batch_count(int n_images, const float *const *rw_up_dev, const int32_t *const *argmax_dev,
                                    const int32_t *n_channels, const int32_t *h, const int32_t *w, int32_t *n_det_dev,
                                    void *scratch_dev, void *stream);
int irn_detect_instance_batch_emit(int n_images, const float *const *rw_up_dev, const int32_t *const *argmax_dev,
                                   const int32_t *n_channels, const int32_t *h, const int32_t *w, const int32_t *n_det,
                                   const double *min_area, float *const *score_dev, int32_t *const *channel_dev,
                                   uint8_t *const *mask_dev, void *scratch_dev, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* IRN_HIP_H */
