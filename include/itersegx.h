/*
 * itersegx.h -- C ABI of libitersegx.so: the MI355X (gfx950) implementation of iterseg's
 * affinity-U-Net -> watershed hot path.
 *
 * Every entry point names the reference interface it replaces (paths relative to the
 * reference checkout, AbigailMcGovern/iterseg v0.3.0).  Plain pointers and sizes only.
 *
 * Conventions
 *   - All array arguments are C-contiguous.  Unless a function name ends in `_host`, data
 *     pointers are DEVICE pointers (HBM) valid on the context's device, and work is
 *     enqueued on the context's stream (isx_ctx_set_stream); scalar results that the
 *     signature returns through a host pointer imply a stream synchronisation.
 *   - Volumes are (Z, Y, X); "padded" means one zero voxel on every side: (Z+2, Y+2, X+2),
 *     the layout segment_single_volume hands to the processing function
 *     (src/iterseg/segmentation.py:885-900).
 *   - Return value: ISX_OK (0) or a negative ISX_ERR_* code; isx_last_error() returns a
 *     thread-local message.  There is no CPU fallback: without a usable HIP device
 *     isx_ctx_create fails (ISX_ERR_NO_DEVICE) and nothing else can be called.
 */
#ifndef ITERSEGX_H
#define ITERSEGX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ISX_VERSION 100

#define ISX_OK 0
#define ISX_ERR_NO_DEVICE (-1)   /* no HIP device / context creation failed            */
#define ISX_ERR_HIP (-2)         /* a HIP runtime call or kernel launch failed          */
#define ISX_ERR_ARG (-3)         /* invalid argument (shape, null pointer, geometry)    */
#define ISX_ERR_NO_SEEDS (-4)    /* zero seeds survive: the reference raises here
                                    (np.apply_along_axis in watershed.py:50)            */
#define ISX_ERR_STATE (-5)       /* weights not loaded / workspace not prepared         */
#define ISX_ERR_NOMEM (-6)       /* device allocation failed                            */
#define ISX_ERR_INTERNAL (-7)    /* an internal invariant failed (a bug: never expected) */

typedef struct isx_ctx isx_ctx;

/* ---------------------------------------------------------------------------------- *
 * Context (no reference counterpart: the reference keeps its state in the `config`     *
 * dict built by affinity_watershed_prep_config, segmentation.py:80-135; the context is *
 * what that dict owns in this implementation).                                         *
 * ---------------------------------------------------------------------------------- */
int isx_version(void);
const char *isx_last_error(void);
int isx_device_count(void);
isx_ctx *isx_ctx_create(int device);               /* NULL on failure                   */
void isx_ctx_destroy(isx_ctx *ctx);
/* Work is enqueued on a context-owned non-blocking stream until a caller stream is bound;
 * NULL binds the HIP null stream (what torch.cuda.current_stream().cuda_stream is by default). */
int isx_ctx_set_stream(isx_ctx *ctx, void *hip_stream);
int isx_ctx_synchronize(isx_ctx *ctx);
int isx_ctx_device(const isx_ctx *ctx);
/* Bytes currently held by the context's device workspace cache. */
int64_t isx_ctx_workspace_bytes(const isx_ctx *ctx);
void isx_ctx_release_workspace(isx_ctx *ctx);
/* Development / test hook: the ISX_* environment knobs (kernel variants, the flood's stress modes) are read ONCE, at
 * isx_ctx_create; this reads them again for an existing context.  No reference counterpart (the reference has no knobs). */
int isx_ctx_reload_knobs(isx_ctx *ctx);

/* Kernel timing by class, measured with HIP events on the context's stream
 * (bench.py's roofline numbers).  Classes: see ISX_PROF_*.  enable=0 stops and clears. */
#define ISX_PROF_CONV_MFMA 0
#define ISX_PROF_CONV_DIRECT 1
#define ISX_PROF_ELEMENTWISE 2
#define ISX_PROF_WS_FILTERS 3
#define ISX_PROF_WS_CCL 4
#define ISX_PROF_WS_FLOOD 5
#define ISX_PROF_NCLASS 6
int isx_prof_enable(isx_ctx *ctx, int enable);
int isx_prof_get(isx_ctx *ctx, int cls, double *total_ms, int64_t *launches);

/* ---------------------------------------------------------------------------------- *
 * Chunk geometry -- replaces make_chunks (src/iterseg/predict.py:38-61).               *
 * table: int32 [cap][9] rows (sz, sy, sx, lo_z, hi_z, lo_y, hi_y, lo_x, hi_x), host     *
 * memory, z-major order like itertools.product.  Returns the number of chunks (may be  *
 * > cap: call again with a larger table) or a negative error.                          *
 * ---------------------------------------------------------------------------------- */
int64_t isx_make_chunks(const int64_t shape[3], const int64_t chunk[3], const int64_t margin[3],
                        int32_t *table, int64_t cap);

/* ---------------------------------------------------------------------------------- *
 * U-Net -- replaces load_unet (predict.py:25-35), UNet.forward (unet.py:284-364) and    *
 * the chunk loop process_chunks / predict_chunk_feature_map (predict.py:64-126).        *
 * ---------------------------------------------------------------------------------- */
/* One state_dict entry (host pointer, float32, reference layout e.g. OIDHW for conv
 * weights; keys as in unet.py: "c0.conv0.weight", "up1.bias", ...).  Buffers
 * ("...running_mean", "...running_var") are accepted; "num_batches_tracked" is int64. */
int isx_load_weight_host(isx_ctx *ctx, const char *key, const void *data, const int64_t *shape,
                         int ndim);
/* Call after the last isx_load_weight_host: checks the schema strictly (like
 * load_state_dict(strict=True)) and repacks into kernel layouts on the device. */
int isx_weights_commit(isx_ctx *ctx, int out_channels);
/* The forked network of the reference -- UNet(out_channels=(a, b)) (unet.py:152-157,192-210) as train_unet builds it for
 * fork_channels (train.py:139-141): one encoder, shared transposed convolutions, one decoder c5_i..c8_i per fork, outputs
 * concatenated (unet.py:313-325).  nforks = 1 or 2 (the reference's decoder() serves forks 0 and 1), at most 8 channels
 * in total; state-dict keys "c5_1.conv0.weight" ... for the second decoder.  nforks = 1 is isx_weights_commit. */
int isx_weights_commit_forked(isx_ctx *ctx, const int32_t *fork_channels, int nforks);
/* Read back a state_dict entry (running statistics are updated by forward passes, U2). */
int isx_get_weight_host(isx_ctx *ctx, const char *key, void *data, int64_t nbytes);

/* Forward `n` chunks cut from a device volume (Z,Y,X) f32 at the given starts
 * (int32 [n][3], host memory), each of size chunk[3]; train-mode BatchNorm per chunk
 * (SURVEY.md F2).  out: device float32 [n][out_channels][cz][cy][cx] (NCDHW, like the
 * tensor predict_chunk_feature_map returns, predict.py:123). */
int isx_unet_forward_chunks(isx_ctx *ctx, const float *volume, const int64_t shape[3],
                            const int32_t *starts, int64_t n, const int64_t chunk[3], float *out);
/* P2-P4 in one call: every chunk of `table` (isx_make_chunks rows, host memory) is pushed
 * through the U-Net and its centre crop written into feature_map (5, Z, Y, X) on the
 * device (process_chunks' `output_volume`, segmentation.py:138-140).  Chunks
 * [first, first+count) only: the multi-GPU shard of one volume. */
int isx_predict_volume(isx_ctx *ctx, const float *volume, const int64_t shape[3],
                       const int32_t *table, int64_t first, int64_t count,
                       const int64_t chunk[3], float *feature_map);
/* Tap a named intermediate of the most recent single-chunk forward (tests): copies the
 * activation as NCDHW float32 to host.  Names: "c0.conv0", "c0", "d0", "c4", "up0", ... */
int isx_unet_tap_host(isx_ctx *ctx, const char *name, float *out, int64_t cap_elems,
                      int64_t shape_out[4]);
/* Max chunks processed concurrently by one launch set (workspace sizing). */
int isx_unet_set_batch(isx_ctx *ctx, int batch);
/* Arithmetic of the wide convolutions at inference.  ISX_PRECISION_SPLIT_F16 (default): every fp32 operand split into
 * two halves (hi + lo/2^11), three v_mfma_f32_32x32x16_f16 per K block with separate fp32 accumulators -- fp32-class
 * results (about 2^-21 per product, below the fp32 accumulation noise; measured error against a float64 oracle at or
 * below the fp32 kernel's) at the f16 matrix rate; layers whose operands could leave the f16 range run the fp32 kernel.
 * ISX_PRECISION_FP32: v_mfma_f32_32x32x2_f32, exact fp32 products.
 * Reduced-precision side modes (never the default; stated tolerances in tests/test_gpu_unet.py, test_gpu_train.py):
 * ISX_PRECISION_F16: plain f16 operands (round to nearest), one MFMA per product, fp32 accumulate -- the "fp16 U-Net" of
 * BASELINE config 5; same f16 range fallback as the split mode.  ISX_PRECISION_BF16: plain bf16 operands
 * (v_mfma_f32_32x32x16_bf16), fp32 accumulate, fp32 BatchNorm statistics.
 * Training runs fp32 except in ISX_PRECISION_BF16, where the wide convolutions of the forward pass and their data
 * gradients run on the bf16 matrix cores from bf16 copies of the fp32 master weights (weight gradients, BatchNorm,
 * the loss and Adam stay fp32): the mixed-precision mode of BASELINE config 4.  Same reference operator in every
 * mode: nn.Conv3d in ConvModule.forward (unet.py:63-79,91-106). */
#define ISX_PRECISION_FP32 0
#define ISX_PRECISION_SPLIT_F16 1
#define ISX_PRECISION_F16 2
#define ISX_PRECISION_BF16 3
int isx_unet_set_precision(isx_ctx *ctx, int mode);
/* Final activation of each decoder's last ConvModule -- UNet(chan_final_activations=[...]) (src/iterseg/unet.py:205-209,
 * ConvModule.forward :91-106; train_unet hands the list through, train.py:134-141).  One entry per decoder (1 or 2), default
 * sigmoid.  ISX_FINAL_SOFTMAX is nn.Softmax() with dim=None on the 5-D activation: over the module's channels (dim 1).
 * ISX_FINAL_NONE is what the reference does for any other string: BatchNorm output unchanged.  Applies to inference
 * (isx_unet_forward*, isx_predict_volume) and to training (loss and backward). */
#define ISX_FINAL_SIGMOID 0
#define ISX_FINAL_RELU 1
#define ISX_FINAL_TANH 2
#define ISX_FINAL_NONE 3
#define ISX_FINAL_SOFTMAX 4
int isx_unet_set_final_activations(isx_ctx *ctx, const int32_t *acts, int n_forks);
/* How each convolution of the LAST forward actually ran (layer order: c0.conv0, c0.conv1, c1.conv0 ... c4.conv1,
 * c5_0.conv0 ... c8_0.conv1, then c5_1 ... c8_1 of a second fork): the reduced modes send a layer whose operands could
 * leave the f16 range (bounds from the BatchNorm affine parameters of its producer) to the fp32 kernel, at about half the
 * throughput -- this makes that decision visible.  Returns the number of layers (modes may be NULL);
 * *n_fallback = layers that ran ISX_ARITH_FP32_RANGE_FALLBACK.  Same operator as above (unet.py:63-79,91-106). */
#define ISX_ARITH_FP32 0                  /* fp32 kernel: the mode asked for it, the layer is a thin (direct) one, or training */
#define ISX_ARITH_REDUCED 1               /* the split-f16 / f16 / bf16 kernel                                                 */
#define ISX_ARITH_FP32_RANGE_FALLBACK 2   /* a reduced mode was asked for but the range check sent the layer to fp32          */
int isx_unet_precision_report(isx_ctx *ctx, int32_t *modes, int cap, int32_t *n_fallback);

/* ---------------------------------------------------------------------------------- *
 * Training -- replaces _train_step + the duplicated optimiser.step() (src/iterseg/train.py:344-353,
 * :310), nn.BCELoss / nn.MSELoss / WeightedBCELoss (train.py:197-217, custom_loss.py:56-84) and
 * torch.optim.Adam (train.py:145) for one (1,1,D,H,W) patch per step (the reference's batch size).
 * Parameters, gradients and Adam moments are four flat float32 DEVICE buffers owned by the caller,
 * holding the trainable state-dict entries in state-dict order (per module: conv0.weight, conv0.bias,
 * conv1.weight, conv1.bias, batch0.weight, batch0.bias, batch1.weight, batch1.bias; then up0..3
 * weight, bias) in their reference layouts -- so a data-parallel gradient all-reduce is one RCCL call
 * on the gradient buffer.
 * ---------------------------------------------------------------------------------- */
int64_t isx_train_param_count(isx_ctx *ctx);
/* Binds the buffers (n = isx_train_param_count), copies the committed weights into `params`, allocates
 * the backward workspace for patches of `chunk` voxels. */
int isx_train_bind(isx_ctx *ctx, const int64_t chunk[3], float *params, float *grads, float *adam_m,
                   float *adam_v, int64_t n);
#define ISX_LOSS_BCE 0
#define ISX_LOSS_MSE 1
#define ISX_LOSS_WEIGHTED_BCE 2
#define ISX_LOSS_DICE 3         /* custom_loss.py:24-53 (smooth = 1, mean over channels) */
/* One step on a device patch x (D,H,W) with targets y (C,D,H,W): forward (train-mode BN), loss,
 * and if do_backward: gradients into the bound buffer followed by n_adam Adam updates with the same
 * gradients (the reference applies two, train.py:353 + :310).  loss_out (host, 1 + C floats): the loss
 * and the per-channel means logged by channel_losses_to_dict (custom_loss.py:8-17). */
int isx_train_step(isx_ctx *ctx, const float *x, const float *y, int loss_kind, const float *chan_weights,
                   float lr, int n_adam, int do_backward, float *loss_out);
/* ChannelwiseLoss (custom_loss.py:131-159) and any other per-channel mix:
 *   loss = sum_c chan_weight[c] * mean_m elem_loss(chan_kind[c])(y_hat[c, m], y[c, m]),
 * chan_kind[c] in {ISX_LOSS_BCE, ISX_LOSS_MSE}.  For the reference's `losses` / `chan_losses` lists with L
 * entries, a channel of entry i covering n_i channels gets weight 1 / (L * n_i).  Otherwise as isx_train_step. */
int isx_train_step_channelwise(isx_ctx *ctx, const float *x, const float *y, const int32_t *chan_kind,
                               const float *chan_weight, float lr, int n_adam, int do_backward, float *loss_out);
/* n_steps Adam updates from the bound gradient buffer (after an external all-reduce), then repack. */
int isx_adam_step(isx_ctx *ctx, float lr, int n_steps);
/* Re-derive the kernel-layout weights after the caller changed the bound parameter buffer. */
int isx_train_refresh_weights(isx_ctx *ctx);
/* Gradient tensors of the last backward pass of isx_train_step, channels-last [M][C] float32, to HOST memory (the
 * counterpart of isx_unet_tap_host for loss.backward(), train.py:351; tests pin the bf16 data-gradient convolutions layer
 * by layer with it).  name: "draw:<layer>" d loss / d raw convolution output, "gact:<layer>" d loss / d activation,
 * "dpool:<k>", "dcat:<k>" gradients of the pooled / concatenated inputs; layer numbering as isx_unet_precision_report.
 * shape_out = {M, C}. */
int isx_train_tap_host(isx_ctx *ctx, const char *name, float *out, int64_t cap_elems, int64_t shape_out[2]);

/* ---------------------------------------------------------------------------------- *
 * Training targets -- src/iterseg/labels.py (the producer side of train_unet). *
 * ---------------------------------------------------------------------------------- */
/* get_training_labels (labels.py:23-68) for int32 instance labels (Z,Y,X) on the device -> float32 (C,Z,Y,X):
 *   ISX_TARGET_AFFINITY   "<axis>-<n>"  nth_affinity (labels.py:87-108): 1 where voxel i and voxel i - n along the axis
 *                                       differ (reflected at the start), else 0;
 *   ISX_TARGET_CENTRENESS "centreness"  get_centreness (labels.py:142-205, log/power off): per object
 *                                       1 - d / d_max, d = |(centroid - voxel) * scale| in float64; 0 for background and
 *                                       for single-voxel objects;
 *   ISX_TARGET_MASK       "mask"        get_semantic_labels (labels.py:283-285): labels > 1;
 *   ISX_TARGET_CENTRENESS_LOG "centreness-log"  get_centreness(log=True) (labels.py:178-205): d -> ln d for d > 0, shifted
 *                                       by |min| over the object, then 1 - d / d_max;
 *   ISX_TARGET_OFFSET     "offsets-<axis>"  get_centre_offsets (labels.py:212-276), axis in axis[c]: per object
 *                                       o = (centroid - voxel) * scale along the axis, o / max(o) for o > 0, -(o / min(o))
 *                                       for o < 0, mapped to [0, 1] by (v + 1) / 2 and rounded to float32; background 0.5;
 *   ISX_TARGET_CENTROID_GAUSS "centroid-gauss"  get_gauss_centroids (labels.py:293-309): 1 at every object's rounded
 *                                       centroid, Gaussian (sigma 1) plane by plane, shifted and scaled to [0, 1].
 * The reference returns float64 holding the same values (affinities / mask are 0 or 1, the others are float32). */
#define ISX_TARGET_AFFINITY 0
#define ISX_TARGET_CENTRENESS 1
#define ISX_TARGET_MASK 2
#define ISX_TARGET_CENTRENESS_LOG 3
#define ISX_TARGET_OFFSET 4
#define ISX_TARGET_CENTROID_GAUSS 5
int isx_training_labels(isx_ctx *ctx, const int32_t *labels, const int64_t shape[3], int n_channels,
                        const int32_t *kind, const int32_t *axis, const int32_t *n, const double scale[3], float *out);

/* Patch sampler of the training data (src/iterseg/train_io.py:224-275, get_image_chunks).  `vol` is a device (Z, Y, X)
 * float32 image, `starts` a HOST list of n (z, y, x) crop offsets, `shape` the crop shape.
 * isx_patch_stats: for every candidate crop the sum (float64), the maximum and (optionally) the minimum of its voxels into the
 * HOST arrays sum_out / max_out / min_out (the minimum decides skimage's clip range in augment_intensity, augment.py:66) -- the reference's brightness test is  x.mean() / x.max() > min_brightness_prop  (train_io.py:249-250); the random
 * draws stay with the caller so that the same draws pick the same crops.
 * isx_patch_gather: the crops themselves, `out` = device [n][shape] of 4-byte elements (float32 image or int32 labels --
 * the reference slices the image, the ground truth and every training-label channel with the same slices, :255,:297-308). */
int isx_patch_stats(isx_ctx *ctx, const float *vol, const int64_t dims[3], const int32_t *starts, int n,
                    const int64_t shape[3], double *sum_out, float *max_out, float *min_out /* may be NULL */);
int isx_patch_gather(isx_ctx *ctx, const void *vol, const int64_t dims[3], const int32_t *starts, int n,
                     const int64_t shape[3], void *out);

/* One array of the reference's augment_images (src/iterseg/augment.py:8-118) in one pass, device 4-D array (C, Z, Y, X) of
 * 4-byte elements in -> out (3-D arrays: C = 1):
 * intensity != 0 (the image, float32; augment_intensity :36-73): v * gain + shift with gain = scale / image.max(), then the noise
 * the caller drew (noise_mode 0 none, 1 additive v + g, 2 "speckle" v + v * g; `noise` = device float32 array of the INPUT's
 * shape), clipped to [clip_low, 1]: clip_low = 0, or -1 where skimage's random_noise (called at augment.py:66) would see a
 * negative value and keep the signed range; without noise upstream clips to [0, 1] (:70-71).
 * Then augment_order (:76-118): mirror along input axis mirror_axis (-1: none), then np.transpose with out axis k = in axis
 * perm[k] (NULL: identity).  intensity == 0 moves raw 32-bit words (labels and ground truth keep every bit).  The image, every
 * label set and the ground truth of a chunk take the same mirror / permutation. */
int isx_augment_array(isx_ctx *ctx, const void *in, const int64_t shape[4], int intensity, float gain, float shift,
                      int noise_mode, const float *noise, float clip_low, int mirror_axis, const int32_t *perm, void *out);

/* ---------------------------------------------------------------------------------- *
 * Segmentation assessment -- src/iterseg/metrics.py (the pinnable part). *
 * ---------------------------------------------------------------------------------- */
/* Contingency table of two int32 label volumes of n voxels (device): the distinct (gt, mr) pairs, as
 * (uint32 gt << 32 | uint32 mr) in ascending order, and their voxel counts, into device buffers of `cap` entries;
 * *n_pairs (host) receives the number of distinct pairs (ISX_ERR_ARG if it exceeds cap).  This is the counting
 * behind skimage.metrics.variation_of_information(gt, mr) and the np.unique object counts of get_accuracy_metrics
 * (metrics.py:105-119); the entropies are computed from it in float64 on the host. */
int isx_label_contingency(isx_ctx *ctx, const int32_t *gt, const int32_t *mr, int64_t n, uint64_t *pairs,
                          int64_t *counts, int64_t cap, int64_t *n_pairs);

/* ---------------------------------------------------------------------------------- *
 * Watershed post-processor -- src/iterseg/watershed.py.                                 *
 * ---------------------------------------------------------------------------------- */
/* Multi-GPU flood: the flood decomposes exactly over 6-connected mask components (the heap order inside one
 * component does not depend on the others: watershed.py:95-159), so with world > 1 this context
 * floods only the components of its rank -- contiguous ranges of the root-ordered component list balanced by
 * voxel count, the same on every rank -- and leaves the voxels of the others 0 (seed voxels are labelled on every
 * rank).  The caller combines the label volumes with an element-wise max (RCCL all-reduce).  Applies to
 * isx_affinity_watershed / isx_segment_output_image / isx_segment_volume*; (0, 1) restores the default. */
int isx_ws_set_shard(isx_ctx *ctx, int rank, int world);

/* ---------------------------------------------------------------------------------- *
 * Collectives on RCCL (xGMI), one process per GPU -- for hosts that do not already hold a communicator (the Python host
 * of this repository uses torch.distributed, which is the same RCCL).  They are the three exchanges of the sharded path:
 *   isx_allreduce_grads       data-parallel training: the reference's step (train.py:344-363) on every rank, gradients
 *                             averaged between isx_train_step(..., n_adam = 0) and isx_adam_step
 *   isx_gather_tiles          all-gather of equal-sized per-rank tiles: feature-map z-slabs of ONE volume predicted
 *                             chunk-sharded (predict.py:64-126), or the label tiles of a stack segmented frame-sharded
 *                             (segmentation.py:873-882)
 *   isx_allreduce_max_labels  merges label volumes flooded component-sharded (isx_ws_set_shard): disjoint supports
 * RCCL is loaded (dlopen) by the first isx_comm_* call; all calls are enqueued on the context's stream.
 * ---------------------------------------------------------------------------------- */
#define ISX_COMM_ID_BYTES 128
/* Rank 0 creates the id and the host distributes it (MPI, a socket, a file); then every rank calls isx_comm_init. */
int isx_comm_unique_id(uint8_t *id /* [ISX_COMM_ID_BYTES] */);
int isx_comm_init(isx_ctx *ctx, const uint8_t *id, int rank, int world);
int isx_comm_destroy(isx_ctx *ctx);
/* grads (device, n floats): sum over the ranks, then divided by the world size, in place. */
int isx_allreduce_grads(isx_ctx *ctx, float *grads, int64_t n);
/* recv (device) = the ranks' send buffers (device, bytes_per_rank each) in rank order. */
int isx_gather_tiles(isx_ctx *ctx, const void *send, void *recv, int64_t bytes_per_rank);
int isx_allreduce_max_labels(isx_ctx *ctx, int32_t *labels, int64_t n);
/* skimage.filters.gaussian as used at watershed.py:227 / :234: float32 in/out, per-axis
 * sigma in {0,1,2}, border 'nearest', truncate 4, double accumulation per pass.  img and out must be different buffers
 * (a pass reads neighbours that the same pass writes): ISX_ERR_ARG otherwise. */
int isx_ws_gaussian(isx_ctx *ctx, const float *img, float *out, const int64_t shape[3],
                    const int32_t sigma[3]);
/* skimage.filters.threshold_otsu (256 bins) of a float32 array; *thresh is HOST memory. */
int isx_ws_threshold_otsu(isx_ctx *ctx, const float *img, int64_t n, float *thresh);
/* W1 _get_mask (watershed.py:226-229): mask = img > otsu(gaussian(img, 2)), uint8 (Z,Y,X).
 * thresh (host, optional) receives the threshold. */
int isx_ws_mask(isx_ctx *ctx, const float *img, const int64_t shape[3], uint8_t *mask,
                float *thresh);
/* W2 _get_centroids (watershed.py:232-236): gaussian(0,1,1) (if smooth != 0) +
 * peak_local_max(threshold_abs).  coords: device int64 [cap][3] (z,y,x), ordered by
 * descending intensity then ascending raveled index; *count is HOST memory (true count,
 * may exceed cap). */
int isx_ws_seeds(isx_ctx *ctx, const float *cent, const int64_t shape[3], int smooth,
                 float threshold_abs, int64_t *coords, int64_t cap, int64_t *count);
/* W3 _remove_unwanted_objects (watershed.py:239-251) on a padded mask: 6-connected
 * components with min_area <= size < max_area survive; seeds (device int64 [n][3], padded
 * coordinates) outside them are dropped, order preserved.  *n_out is HOST memory. */
int isx_ws_filter(isx_ctx *ctx, const uint8_t *mask, const int64_t pshape[3], int64_t min_area,
                  int64_t max_area, uint8_t *mask_out, const int64_t *seeds, int64_t n_seeds,
                  int64_t *seeds_out, int64_t *n_out);
/* ndi.label(mask) of watershed.py:240 (6-connectivity): roots uint32 padded, the smallest
 * raveled index of each voxel's component (0xFFFFFFFF outside the mask); sizes (optional)
 * uint32 padded, the voxel count stored at each root. */
int isx_ws_components(isx_ctx *ctx, const uint8_t *mask, const int64_t pshape[3], uint32_t *roots,
                      uint32_t *sizes);
/* W4 affinity_watershed / raveled_affinity_watershed (watershed.py:17-35, :95-159):
 * aff (3, PZ, PY, PX) float32 zero-padded, seeds int64 [n][3] padded coordinates
 * (label = row + 1), mask uint8 padded with a zero border, out uint32 padded labels
 * (zero on entry except where the caller wants pre-labelled voxels kept).
 * scale (host, optional, 3 floats): |scale[a]| multiplies affinity channel a (:23-24). */
int isx_affinity_watershed(isx_ctx *ctx, const float *aff, const int64_t *seeds, int64_t n_seeds,
                           const uint8_t *mask, uint32_t *out, const int64_t pshape[3],
                           const float *scale);
/* W0 segment_output_image (watershed.py:165-223), plugin-path arguments
 * (segmentation.py:190-194): feature_map (5,Z,Y,X) with channels (z,y,x affinity, mask,
 * centres); labels (Z+2,Y+2,X+2) uint32, written in place (border stays 0).
 * Optional device outputs (may be NULL): seeds_out int64 [cap][3] unpadded coords,
 * mask_out uint8 padded.  Optional host outputs: n_seeds, thresh.
 * abs_thresh: NULL for Otsu, else the `absolute_thresh` value. */
int isx_segment_output_image(isx_ctx *ctx, const float *feature_map, const int64_t shape[3],
                             uint32_t *labels, int64_t min_area, int64_t max_area,
                             const float *abs_thresh, int64_t *seeds_out, int64_t seeds_cap,
                             int64_t *n_seeds, uint8_t *mask_out, float *thresh);
/* The same with segment_output_image's `scale=` argument (watershed.py:170,217-219 -> affinity_watershed :23-24):
 * scale (host, 3 floats, may be NULL): |scale[a]| multiplies the normalised, padded affinity channel a before the flood. */
int isx_segment_output_image_scaled(isx_ctx *ctx, const float *feature_map, const int64_t shape[3],
                                    uint32_t *labels, int64_t min_area, int64_t max_area,
                                    const float *abs_thresh, int64_t *seeds_out, int64_t seeds_cap,
                                    int64_t *n_seeds, uint8_t *mask_out, float *thresh, const float *scale);

/* ---------------------------------------------------------------------------------- *
 * End to end -- replaces segment_single_volume + affinity_watershed_for_chunks          *
 * (segmentation.py:885-900, :147-195).                                                  *
 * ---------------------------------------------------------------------------------- */
/* volume /= max(volume) on the device (segmentation.py:889). */
int isx_normalise_volume(isx_ctx *ctx, float *volume, int64_t n);
/* Device-resident frame: normalised volume (Z,Y,X) -> labels (Z+2,Y+2,X+2) uint32.
 * feature_map: caller-provided (5,Z,Y,X) scratch (the config dict's `output_volume`);
 * it is zeroed on return like the reference does (segmentation.py:195). */
int isx_segment_volume(isx_ctx *ctx, const float *volume, const int64_t shape[3],
                       const int64_t chunk[3], const int64_t margin[3], float *feature_map,
                       uint32_t *labels);
/* Host-buffer form a maintainer binds from affinity_watershed_for_chunks: input_volume
 * float32 (Z,Y,X) already normalised, current_output uint32 (Z+2,Y+2,X+2) zeros. */
int isx_segment_volume_host(isx_ctx *ctx, const float *input_volume, const int64_t shape[3],
                            const int64_t chunk[3], const int64_t margin[3],
                            uint32_t *current_output);

/* ---------------------------------------------------------------------------------- *
 * Frame pipeline for host-resident (t,z,y,x) stacks -- replaces the per-frame host<->device traffic of
 * segmentation_loop (src/iterseg/segmentation.py:833-882: for every t, segment_single_volume(frame) ->
 * output_labels[t] = labels).  Frames of a stack are independent there, so while frame t computes, frame t+1 is
 * staged into pinned memory and uploaded and the labels of frame t-1 are downloaded, on a copy stream and library
 * threads.  Between isx_pipe_create and isx_pipe_destroy the pipe's compute thread is the only user of the context.
 *   isx_pipe_submit  queues one float32 (Z,Y,X) HOST frame (it must stay valid until its isx_pipe_wait); at most
 *                    `depth` frames are in flight.  ISX_PIPE_NORMALISE: frame /= max(frame) on the device
 *                    (segmentation.py:889) instead of on the host.
 *   isx_pipe_wait    blocks until the OLDEST frame in flight is done and writes its labels to HOST memory:
 *                    uint32 (Z+2,Y+2,X+2) like current_output (segmentation.py:890-899), or with ISX_PIPE_CROPPED the
 *                    (Z,Y,X) interior that segment_single_volume returns (:900) -- 32-bit either way, so an int32
 *                    label store row (segmentation.py:881) can be the destination.  Returns the frame's status
 *                    (ISX_ERR_NO_SEEDS: labels are all zero, the reference raises for such a frame).
 * ---------------------------------------------------------------------------------- */
typedef struct isx_pipe isx_pipe;
#define ISX_PIPE_NORMALISE 1
#define ISX_PIPE_CROPPED 2
int isx_pipe_create(isx_ctx *ctx, const int64_t shape[3], const int64_t chunk[3], const int64_t margin[3],
                    int depth, isx_pipe **pipe);
int isx_pipe_submit(isx_pipe *pipe, const float *frame, int flags);
int isx_pipe_wait(isx_pipe *pipe, void *labels, int flags);
void isx_pipe_destroy(isx_pipe *pipe);

#ifdef __cplusplus
}
#endif
#endif /* ITERSEGX_H */
