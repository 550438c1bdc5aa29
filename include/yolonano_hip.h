/*
 * yolonano_hip.h — C ABI of libyolonano_hip.so, the MI355X (gfx950) YOLO-Nano hot path.
 *
 * The reference (yjh0410/YOLO-Nano) has no FFI/plugin interface: its boundary is the Python
 * surface of `YOLONano` (models/yolo_nano.py:12-376).  This header is the C boundary that sits
 * directly under that surface; every entry point names the reference code it replaces.  The host
 * shim `yolo_nano_amd.YOLONano` binds these with ctypes.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch types.
 *   - every function returns 0 on success, non-zero on failure; yn_last_error(h) gives the text
 *     (yn_last_error(NULL) for a failed yn_create).
 *   - one handle = one device + one HIP stream; a handle is not thread-safe.
 *   - "dev" pointers are device (HBM) pointers owned by the caller; the handle owns only its
 *     weights and its activation workspace.  Nothing here synchronises the stream unless noted.
 *   - activations cross this boundary as float32.  Raw head tensors are NHWC
 *     [B, H, W, A*(1+C+4)] — the layout models/yolo_nano.py:312 permutes to before splitting.
 *   - candidate index  n = off_s + (y*W_s + x)*A + a ,  scales in order stride 8, 16, 32
 *     (models/yolo_nano.py:308-330);  N = A * sum_s (S/stride_s)^2.
 */
#ifndef YOLONANO_HIP_H
#define YOLONANO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct yn_handle yn_handle;

enum { YN_BACKBONE_0_5X = 0, YN_BACKBONE_1_0X = 1, YN_BACKBONE_1_5X = 2, YN_BACKBONE_2_0X = 3 };
enum { YN_ACT_NONE = 0, YN_ACT_RELU = 1, YN_ACT_LEAKY = 2 };
enum { YN_F32 = 0, YN_F16 = 1 };
/* return values: 0 = success, 1 = failure (yn_last_error has the text), YN_STATUS_RANGE = yn_infer / yn_pack_detections refused because an EARLIER
 * yn_infer on this handle left the split-f16 range and nobody has acknowledged it through yn_range_status yet (see there) */
enum { YN_STATUS_OK = 0, YN_STATUS_ERROR = 1, YN_STATUS_RANGE = 2 };

/* Mirrors YOLONano.__init__(device, input_size, num_classes, trainable, conf_thresh, nms_thresh,
 * anchor_size, backbone, diou_nms)  — models/yolo_nano.py:13-27. */
typedef struct yn_config {
    int   input_size;      /* S, multiple of 32 */
    int   num_classes;     /* C */
    int   num_anchors;     /* A per scale (3) */
    float anchors[18];     /* [3 scales][A][w,h] in input pixels (data/config.py:11-17) */
    int   backbone;        /* YN_BACKBONE_* (backbone/shufflenetv2.py:95-102) */
    float conf_thresh;     /* models/yolo_nano.py:19 */
    float nms_thresh;      /* models/yolo_nano.py:20 */
    int   diou_nms;        /* models/yolo_nano.py:21 */
    int   max_batch;       /* workspace is sized for this many images at input_size */
    int   device;          /* HIP device ordinal */
    void* stream;          /* hipStream_t; NULL = default stream */
} yn_config;

/* ---- lifetime / configuration -------------------------------------------------------------- */
int  yn_abi_version(void);
int  yn_create(const yn_config* cfg, yn_handle** out);          /* YOLONano.__init__            :13  */
void yn_destroy(yn_handle* h);
const char* yn_last_error(yn_handle* h);
int  yn_set_grid(yn_handle* h, int input_size);                 /* YOLONano.set_grid            :115 */
int  yn_set_stream(yn_handle* h, void* stream);                /* drains the previous stream first  */
int  yn_set_thresholds(yn_handle* h, float conf_thresh, float nms_thresh, int diou_nms);
int  yn_num_predictions(yn_handle* h);                          /* N for the current grid            */
int  yn_use_graph(yn_handle* h, int enable);                    /* hipGraph-capture yn_infer/forward */
int  yn_synchronize(yn_handle* h);
/* Inside one forward the executor forks independent kernel chains (the two branches of a stride-2 unit, the laterals, the heads)
 * onto two side streams of the handle (default on: +4 % for a single handle).  A caller that already runs several handles
 * concurrently on its own streams should turn it off — nine streams contending cost 10 % at three handles (24.1 k vs 21.7 k images/s). */
int  yn_multi_stream(yn_handle* h, int enable);
/* The MFMA-bound convolutions (the dense 3x3 neck layers) run on the f16 matrix pipe with SPLIT fp32 operands by default —
 * x = hi + lo*2^-11, three f16 MFMAs per product into fp32 accumulators: fp32-class results (per-product error <= ~3*2^-22; the
 * f32 MFMA of gfx950 runs at 1/16 of the f16 rate and there is no TF32).  enable != 0 pins every conv to the f32 MFMA. */
int  yn_exact_f32(yn_handle* h, int enable);
/* RANGE of the split-f16 family, and its guard.  hi = (f16)x is finite only for |x| < 65520, so the default path needs every folded
 * GEMM weight and every activation that enters a GEMM-shaped conv below 65504 (normalised inputs, BatchNorm-folded weights and the
 * activations they produce are O(1)..O(100); the reference's fp32 has no such limit).  Checked, not assumed:
 *   - weights: yn_fold_bn tests every folded pointwise / dense-3x3 weight on the device; if one is >= 65504 (or not finite) the handle
 *     runs the f32-MFMA family from then on, exactly as under yn_exact_f32(1)  (*weights_exceed_f16 = 1);
 *   - activations: every kernel that splits activations tracks the largest |x| it split and raises a flag in HBM when it reached
 *     65504; *activation_overflow returns that flag for everything issued since the last call and clears it.  The call synchronises
 *     the handle's stream.  A set flag means the results of those calls are NOT valid: re-run them after yn_exact_f32(h, 1)
 *     (the host shim yolo_nano_amd.YOLONano does this by itself, once, and stays on the f32-MFMA family).
 *     yn_infer also delivers the flag WITH its results, at no extra synchronisation: while it is set every count_dev[b] comes back
 *     NEGATIVE (-1 - K_b), and yn_pack_detections carries the mark on as offsets_dev[B] = -1 - total.  The flag is sticky until
 *     yn_range_status clears it.  The same fact OUT OF BAND, for callers that loop `i < count[b]` without looking at the sign: the kernel
 *     that writes the negative counts also sets one word of pinned host memory, and from then on every yn_infer / yn_pack_detections
 *     on the handle returns YN_STATUS_RANGE (checked on the host, no synchronisation) until yn_range_status has been called - or
 *     yn_exact_f32(h, 1) is in force (the f32-MFMA family cannot leave the range: running again under it is the recovery).  The out-of-band
 *     check is BEST EFFORT for pipelined calls: a yn_infer enqueued before the marking kernel has run is not refused; the negative counts
 *     always travel with the results themselves.
 *     yn_range_status also reports (return 1 + yn_last_error) an expired bounded wait of stage_pipe_kernel (yn_stage_fuse).
 * Tiny values need no guard: below the f16 normal range lo = (x - hi) * 2^11 still carries x. */
int  yn_range_status(yn_handle* h, int* weights_exceed_f16, int* activation_overflow);
/* yn_infer only: the last pointwise conv of each detection head (models/yolo_nano.py:299-301) and the decode of that scale's
 * candidates (:308-330, 362-367) run as ONE kernel, so the raw head tensors are neither written nor re-read (default on; needs
 * the split-f16 family and A(5+C) <= 256, otherwise yn_infer runs head GEMM + decode kernel as before).  Outputs are bit-identical
 * either way: a speed switch for A/B runs.  enable = 1: when the stride-8 head has >= 8192 pixels (below that three GEMMs + one decode
 * launch are faster), 2: always.  Env: YN_FUSE_DECODE=0/1/2. */
int  yn_fuse_decode(yn_handle* h, int enable);
/* Layer k of the three detection heads (models/yolo_nano.py:299-301: same operator, three pyramid levels) and the three FPN
 * laterals (:286-288) run as ONE grouped launch each instead of three (default on; split-f16 family only).  Bit-identical outputs:
 * a speed switch for A/B runs.  Env: YN_GROUP=0/1. */
int  yn_group_launch(yn_handle* h, int enable);
/* A stride-2 ShuffleV2 unit (backbone/shufflenetv2.py:30-51, 73-74: branch 2 = pointwise -> depthwise stride 2 -> pointwise, branch 1 =
 * depthwise stride 2 -> pointwise, then concat + channel shuffle) as ONE kernel where its tile fits (input channels <= 32, branch
 * width <= 64: stage 2, whose intermediate is the largest tensor of the network); the wider units (stages 3 / 4, branch width <= 256) as
 * their first pointwise conv + ONE kernel for everything behind it.  Default on, split-f16 family only, bit-identical to the five
 * launches.  Env: YN_DOWN_FUSE=0/1 (YN_DOWN2=0: only the wide units back to five launches). */
int  yn_down_fuse(yn_handle* h, int enable);
/* Layers .2 + .3 + .4 of the three detection heads and the candidate decode as ONE grouped kernel — depthwise 3x3 + pointwise conv +
 * last conv + decode on an 8 x 4 pixel tile; layer .3's activation never reaches memory (models/yolo_nano.py:60-82, 299-330, 362-367).
 * Default on; needs yn_fuse_decode and yn_group_launch in effect and a head of 129..256 columns (COCO); otherwise, or with 0, two
 * grouped kernels (depthwise + pointwise, last conv + decode).  Bit-identical either way.  Env: YN_TAIL_FUSE=0/1. */
int  yn_tail_fuse(yn_handle* h, int enable);
/* Per-class NMS (models/yolo_nano.py:159-188, 263-272): resolve the 64 best-scored boxes of every class first and drop every later box
 * one of their KEPT boxes suppresses before the dense pairwise phase (exact: a removed box suppresses nothing).  mode 0 = off, 1 (default) =
 * for batches of >= 4 images, 2 = always.  Same kept sets either way.  Env: YN_NMS_PREFILTER=0/1/2. */
int  yn_nms_prefilter(yn_handle* h, int mode);
/* Large class segments (> 1 024 boxes behind the prefilter) whose boxes are spread out get their suppression words from a sweep over bins of
 * the boxes' left edges - only pairs whose x-extents intersect are evaluated, with the same exact predicate (models/yolo_nano.py:159-188: `ovr <=
 * thresh` keeps) - instead of the dense 64 x 64 tiles; the kernel decides per segment from a pair-count estimate.  1 (default) / 0: every segment
 * dense.  Kept sets are identical either way.  Env: YN_NMS_SWEEP=0/1. */
int  yn_nms_sweep(yn_handle* h, int enable);
/* Testing aid: how many (image, class) segments of the LAST yn_infer / yn_postprocess call (B images, C classes) the sweep handled
 * (synchronises the handle's stream); -1 on a failed copy. */
int  yn_nms_sweep_segments(yn_handle* h, int B, int C);
/* Per-layer tile autotuning of the pointwise-conv GEMM (default on): the first eager execution of a layer
 * shape times every instantiated tile configuration of the layer's family (split-f16 by default, f32-MFMA under yn_exact_f32) on
 * the handle's stream and caches the fastest.  All configurations of a family produce bit-identical results; disabling falls back
 * to a static heuristic. */
int  yn_autotune(yn_handle* h, int enable);
/* Testing aid: pin every pointwise GEMM of this handle to tile configuration `index` (0 <= index < yn_pw_config_count();
 * a configuration that does not cover a layer's strides falls back to the heuristic one); index < 0 restores the autotuner. */
int  yn_set_pw_config(yn_handle* h, int index);
/* The process-wide autotune table (layer shape -> tile configuration) of `device` to / from a small text file, the device ordinal
 * left out: the ranks of a multi-GPU job adopt ONE rank's choices instead of each timing the same shapes at once (bench.py).
 * yn_tune_load returns the number of entries adopted (existing ones are kept), -1 if the file cannot be read. */
int  yn_tune_save(const char* path, int device);
int  yn_tune_load(const char* path, int device);
int  yn_pw_config_count(void);
/* Configurations [0, yn_pw_f32_config_count()) are the f32-MFMA family (LDS-tiled, then register-direct), the rest the split-f16
 * family (gemm_split_kernel); results are bit-identical INSIDE a family. */
int  yn_pw_f32_config_count(void);
/* The stride-1 ShuffleV2 units run as one kernel each (depthwise -> pw2 -> concat+shuffle -> next unit's pw1).  mode 1
 * (default): on the stages whose map is large enough for that to pay; 0: three kernels per unit everywhere; 2: one kernel per
 * unit everywhere.  All three give bit-identical results (A/B measurements, tests). */
int  yn_unit_chain(yn_handle* h, int mode);
/* The form of that one-kernel-per-unit launch (backbone/shufflenetv2.py:53-63, 70-72): unit_pipe_kernel, the persistent software-pipelined
 * tile walk - mode 1 (default): by its size rule; 0: never (unit_chain2_kernel everywhere); 2: also for few tiles.  Bit-identical. */
int  yn_chain_pipe(yn_handle* h, int mode);
/* All but the last stride-1 unit of a backbone stage (backbone/shufflenetv2.py:118-125: the `for i in range(numrepeat)` loop) as ONE
 * persistent launch (stage_pipe_kernel: (unit, tile) work items by ticket, tile-level ready flags between the units).  mode 1 (default):
 * from 256 tiles; 0: one launch per unit; 2: at every size.  publish_early 0 (default): a tile raises its ready flag under the next tile's
 * depthwise phase (at once when the workgroup has to wait for its next item's inputs); 1: right behind its stores.  Bit-identical to the
 * per-unit launches. */
int  yn_stage_fuse(yn_handle* h, int mode, int publish_early);
/* pw_pipe_kernel (the persistent form of a pointwise conv, utils/modules.py:8-18 folded) among the autotuner's candidates: 1 (default) / 0. */
int  yn_pw_pipe(yn_handle* h, int enable);

/* ---- weights ------------------------------------------------------------------------------- */
/* nn.Module.load_state_dict (eval.py:127, benchmark.py:132): one call per state-dict entry, using
 * the reference's 469 key names; `host_ptr` float32 (int64 for num_batches_tracked, ignored).   */
int  yn_load_param(yn_handle* h, const char* state_dict_key, const void* host_ptr,
                   const int64_t* shape, int ndim);
/* Same, source already in HBM (e.g. a torch Parameter's data_ptr()). */
int  yn_load_param_dev(yn_handle* h, const char* state_dict_key, const void* dev_ptr,
                       const int64_t* shape, int ndim);
/* utils/fuse_conv_bn.py:6-53 — fold every BN into its conv and repack for the kernels.  Must be
 * called after loading parameters and before inference.  A conv whose BN keys were never loaded
 * is taken as already folded (the state dict of a model that went through fuse_conv_bn()).      */
int  yn_fold_bn(yn_handle* h);
/* Read back the folded weight/bias of one conv in the reference layout [Cout,Cin/g,k,k] / [Cout]
 * (parity check of utils/fuse_conv_bn.py:17-21). `conv_key` e.g. "smooth_1.convs.0".           */
int  yn_get_folded(yn_handle* h, const char* conv_key, float* host_weight, float* host_bias);

/* ---- the network --------------------------------------------------------------------------- */
/* YOLONano.forward lines 284-301: backbone, FPN+PAN neck, three heads.  x_dev: NCHW
 * [B,3,S,S] float32.  Outputs: NHWC raw head tensors [B,S/8,S/8,A(5+C)], [B,S/16,..], [B,S/32,..]. */
int  yn_forward_raw(yn_handle* h, const float* x_dev, int B,
                    float* head_s8_dev, float* head_s16_dev, float* head_s32_dev);

/* ShuffleNetV2.forward (backbone/shufflenetv2.py:157-167): the same network pass, additionally copying the three backbone
 * taps the neck consumes — c3 [B,S/8,S/8,C3], c4 [B,S/16,S/16,C4], c5 [B,S/32,S/32,C5], NHWC float32 (C = 116/232/464 for
 * 1.0x, 48/96/192 for 0.5x) — so that a backbone failure localises (parity tests; never graph-captured). */
int  yn_forward_taps(yn_handle* h, const float* x_dev, int B, float* c3_dev, float* c4_dev, float* c5_dev);

/* Lines 308-330 + 362-367 for EVERY image of the batch (the reference only finishes image 0):
 * all_bbox [B,N,4] = clamp(decode_boxes/S, 0, 1); all_class [B,N,C] = softmax(cls)*sigmoid(obj). */
int  yn_score_full(yn_handle* h, const float* head_s8_dev, const float* head_s16_dev,
                   const float* head_s32_dev, int B, float* all_bbox_dev, float* all_class_dev);

/* YOLONano.decode_boxes :139-156 — txtytwth [B, sum HW, A, 4] -> xyxy pixels [B, N, 4]. */
int  yn_decode_boxes(yn_handle* h, const float* txtytwth_dev, int B, float* xyxy_dev);

/* YOLONano.create_grid :86-112 — host arrays grid [HWtot,2], stride [HWtot,A,2], anchors [HWtot,A,2]. */
int  yn_create_grid(yn_handle* h, int input_size, float* grid_host, float* stride_host, float* anchor_host);

/* ---- post-processing ----------------------------------------------------------------------- */
/* YOLONano.nms :159-188 / diou_nms :191-242 — one class.  dets [n,4] xyxy, scores [n]; writes the
 * kept indices in pick order (descending score; equal scores: higher index first) and the count. */
int  yn_nms(yn_handle* h, const float* dets_dev, const float* scores_dev, int n, float nms_thresh,
            int diou, int32_t* keep_dev, int32_t* count_dev);

/* Per-class NMS over an arbitrary detection list — the merge step of TestTimeAugmentation (utils/misc.py:132-146, nms of
 * utils/misc.py:8-37 = the arithmetic of YOLONano.nms): boxes [n,4], scores [n], cls [n] (0 <= cls < num_classes) ->
 * kept detections in ascending input order, count[0] = K.  Output buffers have capacity n. */
int  yn_nms_merge(yn_handle* h, const float* boxes_dev, const float* scores_dev, const int32_t* cls_dev, int n, int num_classes,
                  float nms_thresh, int diou, float* out_boxes, float* out_scores, int32_t* out_cls, int32_t* out_index, int32_t* count_dev);

/* ValTransforms (data/transforms.py:445-458 = Resize :73-119 + Normalize :59-70 + ToTensor :394-398; call sites
 * benchmark.py:58, evaluator/vocapi_evaluator.py:64): img_dev = uint8 [h0][w0][3] BGR on the device -> x_dev float32
 * [3][side][side] RGB, normalised, letterboxed.  The caller supplies Resize's integer geometry (rw x rh resized extent placed
 * at (left, top) inside the side x side square, padded with mean*255) — it is host arithmetic the reference does in Python
 * (`int(r * size)`, `//`); the resize itself is cv2's 8-bit INTER_LINEAR.  mean / std: 3 host floats each, BGR order. */
int  yn_preprocess(yn_handle* h, const uint8_t* img_dev, int h0, int w0, int rw, int rh, int left, int top, int side,
                   const float* mean_host, const float* std_host, float* x_dev);

/* The same for n images in one launch per 32: imgs_host[i] = device pointer of image i, geom_host[i] = {h0, w0, rw, rh, left,
 * top}; x_dev = float32 [n][3][side][side] (the network's input batch). */
int  yn_preprocess_batch(yn_handle* h, int n, const uint8_t* const* imgs_host, const int32_t* geom_host, int side,
                         const float* mean_host, const float* std_host, float* x_dev);

/* YOLONano.postprocess :245-279, batched: all_local [B,N,4], all_conf [B,N,C] ->
 * per image b: count[b] = K_b and, in ascending candidate order, out_boxes[b,0:K_b,4],
 * out_scores[b,0:K_b], out_cls[b,0:K_b], out_index[b,0:K_b] (candidate index; may be NULL).
 * Output buffers have capacity N per image. */
int  yn_postprocess(yn_handle* h, const float* all_local_dev, const float* all_conf_dev, int B, int N, int C,
                    float* out_boxes_dev, float* out_scores_dev, int32_t* out_cls_dev,
                    int32_t* out_index_dev, int32_t* count_dev);

/* The whole eval-mode YOLONano.forward :282-376 for a batch: network + score head + per-class NMS,
 * all on device, no host round trip.  Outputs as yn_postprocess. */
int  yn_infer(yn_handle* h, const float* x_dev, int B,
              float* out_boxes_dev, float* out_scores_dev, int32_t* out_cls_dev,
              int32_t* out_index_dev, int32_t* count_dev);

/* The hand-over that ends YOLONano.forward (`.to('cpu').numpy()`, models/yolo_nano.py:370-376) for a whole batch: gathers the
 * kept rows of the yn_infer / yn_postprocess outputs of all B images into ONE contiguous record list
 * rec_dev [total][6] float32 = x1, y1, x2, y2, score, class (image order; ascending candidate order inside an image; the class
 * as a float, exact below 2^24) and offsets_dev[B+1] = exclusive prefix of the counts ([B] = total).  The host then needs two
 * copies per batch (the offsets, then total*24 bytes) instead of three per image.  rec_dev has capacity B*N records. */
int  yn_pack_detections(yn_handle* h, const float* out_boxes_dev, const float* out_scores_dev, const int32_t* out_cls_dev,
                        const int32_t* count_dev, int B, int N, float* rec_dev, int32_t* offsets_dev);

/* ---- training loss (train.py:219-229, forward value + gradient w.r.t. the raw predictions) ---------- */
/* models/yolo_nano.py:332-358 + tools.iou_score (tools.py:219-233) + tools.loss (tools.py:236-276).
 * Predictions in the reference's split layout: conf [B,N] (= [B,N,1]), cls [B,N,C], txtytwth [B,N,4];
 * target [B,N,11] = [obj, cls, tx,ty,tw,th, weight, x1,y1,x2,y2] as tools.multi_gt_creator builds it (tools.py:108).
 * losses_dev[4] = conf, cls, bbox (txty+twth), iou — each already divided by B.  The three gradient buffers
 * (same shapes as the predictions; all or none) receive d(conf+cls+bbox+iou)/d(prediction), i.e. what
 * `total_loss.backward()` (train.py:222-229) leaves in the prediction tensors; gt_conf = iou.detach(). */
int  yn_loss(yn_handle* h, const float* conf_dev, const float* cls_dev, const float* txtytwth_dev,
             const float* target_dev, int B, float* losses_dev,
             float* g_conf_dev, float* g_cls_dev, float* g_txtytwth_dev);
/* Same, reading the predictions from / writing the gradients to the three raw NHWC head tensors
 * (layout of yn_forward_raw), i.e. without the re-layout copies of models/yolo_nano.py:308-330. */
int  yn_loss_heads(yn_handle* h, const float* head_s8_dev, const float* head_s16_dev, const float* head_s32_dev,
                   const float* target_dev, int B, float* losses_dev,
                   float* g_s8_dev, float* g_s16_dev, float* g_s32_dev);

/* torch.optim.SGD(lr, momentum=0.9, weight_decay=5e-4).step() (train.py:167-171, 230) on ONE flat float32 bucket
 * holding every parameter (1.27-1.33 M elements), fused with the 1/world_size averaging of the all-reduced
 * gradient sum:  g = grads*grad_scale + wd*p ; buf = first_step ? g : momentum*buf + g ; p -= lr*buf.
 * A bucket that holds a NaN or Inf leaves parameters and momentum untouched — the reference skips an iteration whose loss is
 * NaN (train.py:225-226); after the data-parallel all-reduce every rank sees the same non-finite bucket, so all ranks skip
 * together without a host round trip.  yn_train_skipped_steps reads the number of skipped updates (synchronises). */
int  yn_sgd_step(yn_handle* h, float* params_dev, const float* grads_dev, float* momentum_buf_dev, int64_t n,
                 float lr, float momentum, float weight_decay, float grad_scale, int first_step);

/* ---- ModelEMA.update (utils/misc.py:76-86): ema[i] = ema[i] * d + (1 - d) * model[i] over one float32 tensor (or one flat
 * buffer), d = decay * (1 - exp(-updates / 2000)) computed by the caller in double like the reference; the kernel keeps
 * torch's rounding sequence, so the result is bit-identical to the reference's in-place update. */
int  yn_ema_update(yn_handle* h, float* ema_dev, const float* model_dev, int64_t n, double decay);

/* ---- training labels: tools.multi_gt_creator (tools.py:97-216, call site train.py:212) --------------------------------
 * labels_dev float64 [total][5] = xmin, ymin, xmax, ymax (fractions of the image), class — the objects of image b are rows
 * offsets_dev[b] .. offsets_dev[b+1]-1 IN LIST ORDER (a later object overwrites the slot of an earlier one, as in the
 * reference).  anchors_host: the 9 [w,h] pairs in input pixels as float64 (the reference's Python floats; the float32
 * copy inside the handle would move IoU ties).  target_dev float32 [B][N][11] = obj, cls, tx, ty, tw, th, weight,
 * xmin, ymin, xmax, ymax is fully overwritten.  N and the grid come from the handle's current input size. */
int  yn_make_targets(yn_handle* h, const double* labels_dev, const int32_t* offsets_dev, int B, const double* anchors_host, float* target_dev);

/* ---- training step (train.py:212-231) ---------------------------------------------------------------- */
/* Parameters, gradients and SGD momentum live in three caller-owned FLAT float32 device buffers of
 * yn_train_param_count() elements, in nn.Module.named_parameters() order of the reference model (per layer:
 * conv.weight, [conv.bias], [bn.weight, bn.bias]); yn_train_param_offset maps a state-dict key to its slice.
 * yn_train_bind copies the loaded state dict into `params`, zeroes `grads` / `momentum`.  BatchNorm running statistics
 * stay inside the handle (updated in place, momentum 0.1; read back with yn_read_param). */
int64_t yn_train_param_count(yn_handle* h);
int  yn_train_param_offset(yn_handle* h, const char* state_dict_key, int64_t* offset, int64_t* numel);
int  yn_train_bind(yn_handle* h, float* params_dev, float* grads_dev, float* momentum_dev, int64_t n);
/* One step: train-mode forward (BatchNorm batch statistics) of x [B,3,S,S], the four losses of tools.loss against
 * target [B,N,11] into losses_dev[4], backward into `grads` (overwritten).  do_update != 0 also applies
 * SGD(lr, momentum, weight_decay) with grads*grad_scale; data-parallel callers pass do_update = 0, all-reduce `grads`
 * (RCCL, one flat bucket) and then call yn_sgd_step(grad_scale = 1/world).  Call yn_fold_bn before the next inference. */
int  yn_train_step(yn_handle* h, const float* x_dev, const float* target_dev, int B, float lr, float momentum,
                   float weight_decay, float grad_scale, int do_update, float* losses_dev);
int  yn_read_param(yn_handle* h, const char* state_dict_key, float* host, int64_t numel);
/* The fp16 step's dynamic loss scale (initial 1024; halved on an overflowing step, doubled after 2000 clean ones; all on the device).
 * The decision is taken by yn_sgd_step from the finite-scan of the bucket it applies - after the data-parallel all-reduce that bucket is
 * the same on every rank, so the replicas' scales move together.  get / set (both synchronise) let a checkpoint carry the scale and its
 * clean-step counter across a resume; set before the first fp16 step replaces the default start value.  set also discards the overflow
 * flag / pending mark of a step that has not been settled yet (the restored state starts clean).  Accepted range [1, 2^30]; the device
 * only ever DOUBLES up to 65536, so a larger restored value can only shrink. */
int  yn_train_get_loss_scale(yn_handle* h, float* scale, float* clean_steps);
int  yn_train_set_loss_scale(yn_handle* h, float scale, float clean_steps);
/* The gradient exchange of the data-parallel step (train.py:13-14 imports DistributedDataParallel; BASELINE configs[2]: "DDP grad
 * all-reduce over xGMI") for callers WITHOUT torch: all-reduce(sum), in place, of the bound flat gradient buffer over an RCCL
 * communicator the caller owns (`nccl_comm` is an ncclComm_t), enqueued on the handle's stream — after yn_train_step(do_update = 0),
 * before yn_sgd_step(grad_scale = 1/world).  One 5.3 MB bucket per step: no bucketing, no overlap needed at this model size.
 * The library does not link RCCL: ncclAllReduce is resolved at run time from the librccl already loaded in the process (the one
 * that created the communicator), else from librccl.so.1.  torch users keep torch.distributed (parallel.dp_train_step). */
int  yn_allreduce_grads(yn_handle* h, void* nccl_comm);
/* The forward half of yn_train_step on its own — `model.train(); model.backbone/neck/heads(x)` (models/yolo_nano.py:284-301 with
 * BatchNorm batch statistics; the running statistics ARE updated) in the precision selected by yn_train_precision: the three
 * raw NHWC head tensors as dense float32 [B,S/8,S/8,A(5+C)], [B,S/16,..], [B,S/32,..].  Parity hook for the train-mode network. */
int  yn_train_forward(yn_handle* h, const float* x_dev, int B, float* head_s8_dev, float* head_s16_dev, float* head_s32_dev);
int  yn_train_skipped_steps(yn_handle* h, int64_t* count_host);
/* The fp16 step forks the head towers of levels 3 / 4 onto streams of their own when that measured faster on this device (a handle's
 * steps 3-6 time the step both ways, so the choice - and with it the order of some atomic sums - depends on the machine).  Query the
 * decision (force = 0; *decision: -1 undecided, 0 one stream, 1 forked) or pin it for reproducible runs (force = 1: one stream, 2: forked). */
int  yn_train_head_fork(yn_handle* h, int force, int* decision_host);
/* Arithmetic of yn_train_step (BASELINE configs[2] names fp16; train.py itself runs fp32).  YN_F32 (default): fp32 end to end.
 * YN_F16: activations and activation gradients are STORED as fp16 (channel-padded NHWC, half the HBM bytes), every GEMM-shaped
 * conv (forward, input gradient, weight gradient) runs on the f16 MFMA with fp32 accumulation, BatchNorm statistics / parameter
 * gradients / the optimiser stay fp32 on the fp32 master weights, and the loss gradient is multiplied by a dynamic loss scale
 * kept on the device (halved when a step's gradients overflow — that step is skipped — doubled after 2000 clean steps). */
int  yn_train_precision(yn_handle* h, int dtype);
/* Opt-in: the fp16 step replays everything between its host-side preparation and the optimiser (~540 launches on two streams) from a
 * hipGraph once the same (x, target, batch, grid) has been seen twice on a handle with a stream of its own; callers whose tensors'
 * addresses never repeat stay on direct launches.  Off by default: measured slower than direct launches on this runtime.
 * enable: 1 / 0 switch it, -1 leaves it; replays (optional) receives the number of steps served from a graph so far. */
int  yn_train_graph(yn_handle* h, int enable, int64_t* replays);

/* ---- single operators (op-level parity tests; NHWC float32 device tensors) ------------------ */
/* weights in the reference (torch) layout on the DEVICE: dw [C,1,3,3], pw [Cout,Cin,1,1],
 * dense [Cout,Cin,3,3]; bias [Cout] or NULL. */
int  yn_op_dwconv3x3(yn_handle* h, const float* x, int B, int H, int W, int C, int stride,
                     const float* w, const float* bias, int act, float* y);
int  yn_op_pwconv(yn_handle* h, const float* x, int B, int H, int W, int Cin, int Cout,
                  const float* w, const float* bias, int act, float* y);
/* The tail of a ShuffleV2Block (backbone/shufflenetv2.py:72,74 + channel_shuffle :14-28) exactly as the network runs it:
 * y = channel_shuffle(cat(pass, act(pw(x))), 2), i.e. y[..., 2j] = pass[..., j], y[..., 2j+1] = act(pw(x))[..., j], written by
 * the GEMM epilogue (the shuffle is never materialised on its own).  pass [B,H,W,Cout], y [B,H,W,2*Cout]. */
int  yn_op_pwconv_shuffle(yn_handle* h, const float* x, const float* pass, int B, int H, int W, int Cin, int Cout,
                          const float* w, const float* bias, int act, float* y);
/* x2/resample: 0 none, 1 add nearest-up2 of x2 [B,H/2,W/2,Cin], 2 add nearest-down of x2 [B,2H,2W,Cin]
 * (models/yolo_nano.py:291-296 fused into the conv's prologue).  Cin must be a multiple of 32. */
int  yn_op_conv3x3(yn_handle* h, const float* x, const float* x2, int resample, int B, int H, int W,
                   int Cin, int Cout, const float* w, const float* bias, int act, float* y);
/* stem: x NCHW [B,3,H,W] -> y NHWC [B,Ho,Wo,Cout], 3x3 stride 2 pad 1 (backbone/shufflenetv2.py:109) */
int  yn_op_stem(yn_handle* h, const float* x_nchw, int B, int H, int W, int Cout,
                const float* w, const float* bias, int act, float* y);
int  yn_op_maxpool3x3s2(yn_handle* h, const float* x, int B, int H, int W, int C, float* y);
/* ShuffleV2Block (backbone/shufflenetv2.py:31-78), weights taken from the handle's loaded params:
 * block = "backbone.stage2.1" etc.  x [B,H,W,Cin] -> y [B,H/stride,W/stride,Cout]. */
int  yn_op_shuffle_block(yn_handle* h, const char* block, const float* x, int B, int H, int W, float* y);
/* NCHW <-> NHWC helpers for the tests and the host shim */
int  yn_op_nchw_to_nhwc(yn_handle* h, const float* x, int B, int C, int H, int W, float* y);
int  yn_op_nhwc_to_nchw(yn_handle* h, const float* x, int B, int C, int H, int W, float* y);

/* Single kernels of the fp16 training step (yn_train_precision YN_F16) behind fp32 NHWC device tensors, for op-level parity: the
 * inputs are rounded to fp16 into the step's channel-padded layout (gapped != 0: the two-plane layout of a ShuffleV2 unit output,
 * Cin = 2*bf), ONE kernel per requested result runs, the fp16 results return as fp32.  kind 0 pointwise, 1 depthwise 3x3
 * (stride 1|2), 2 dense 3x3; w / dw in the reference layouts; y = conv(x) + bias, dx / dw = the gradients for dy (null = skip). */
int  yn_op_h16_conv(yn_handle* h, int kind, const float* x, int B, int H, int W, int Cin, int gapped, const float* w, const float* bias,
                    int Cout, int stride, const float* dy, float* y, float* dx, float* dw);
/* The column sums the fp16 step takes in its GEMM epilogues instead of separate reduction launches, on their own (kind 0 pointwise,
 * 2 dense 3x3; layouts as yn_op_h16_conv): y = conv(x) with sums_fwd[0][c] = sum y, sums_fwd[1][c] = sum y^2 over the STORED fp16
 * values (the train-mode BatchNorm statistics, utils/modules.py:12-21);  and, when dy is given, dx = the input gradient with
 * sums_bwd[0][c] = sum d, sums_bwd[1][c] = sum d * xhat, d = dx * act'(BN(y_below)), xhat = (y_below - mean) * invstd — the two sums
 * the BatchNorm backward of the layer BELOW needs (y_below [B,H,W,Cin] its pre-BN output; mean / invstd / gamma / beta [Cin] device
 * pointers).  sums are host double [2][channels]. */
int  yn_op_h16_gemm_stats(yn_handle* h, int kind, const float* x, int B, int H, int W, int Cin, int gapped, const float* w, int Cout,
                          float* y, double* sums_fwd, const float* dy, const float* y_below, const float* mean, const float* invstd,
                          const float* gamma, const float* beta, int act, float* dx, double* sums_bwd);
/* Train-mode BatchNorm (+ activation) forward over y [M][C] and, when dz is given, its backward: z, dy [M][C], dgamma, dbeta [C]. */
int  yn_op_h16_bn(yn_handle* h, const float* y, const float* dz, int64_t M, int C, const float* gamma, const float* beta, int act,
                  float* z, float* dy, float* dgamma, float* dbeta);
/* The same BatchNorm as the last layer of a ShuffleV2 unit (backbone/shufflenetv2.py:69-78 with :14-28): forward writes the unit
 * output unit[m][2c] = pass[m][c], unit[m][2c+1] = act(BN(y))[m][c]  ([M][2C]: concat + channel_shuffle(2));  backward takes the unit
 * output's gradient dunit [M][2C] and returns dy [M][C] (through activation and BatchNorm), deven [M][C] = dunit[:, 0::2] (the
 * pass-through half, written by the backward kernel on its way: it loads those values anyway), dgamma, dbeta [C].  C <= 128. */
int  yn_op_h16_bn_unit(yn_handle* h, const float* y, const float* pass, const float* dunit, int64_t M, int C, const float* gamma, const float* beta,
                       int act, float* unit, float* dy, float* deven, float* dgamma, float* dbeta);

/* ---- measurement -------------------------------------------------------------------------- */
/* When enabled, every kernel launch of yn_forward_raw / yn_infer is bracketed by a pair of HIP
 * events recorded on the handle's stream (graph replay is bypassed while enabled).  After the
 * call, yn_profile_get(i) returns the launch's layer name, the kernel symbol, its measured duration and its
 * ALGORITHMIC flops / bytes (each conv reads its input once and writes its output once, weights
 * once — DESIGN.md §Measurement).  bench.py builds its `roofline` block from these. */
int  yn_profile_enable(yn_handle* h, int enable);
int  yn_profile_count(yn_handle* h);
int  yn_profile_get(yn_handle* h, int i, char* name, int name_cap, char* kernel, int kernel_cap,
                    float* ms, double* alg_flops, double* alg_bytes);

#ifdef __cplusplus
}
#endif
#endif /* YOLONANO_HIP_H */
