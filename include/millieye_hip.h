/*
 * millieye_hip.h - C ABI of libmillieye_hip.so (gfx950 / MI355X).
 *
 * The reference (sxontheway/milliEye) is pure Python and owns no native layer: every
 * FLOP of its hot path is a call into PyTorch / torchvision.
 * This header is therefore the FFI a maintainer binds with ctypes (see INTEGRATION.md
 * and millieye_amd/hip.py): each entry point replaces the library call(s) made at
 * the reference call site quoted above it.  All paths are relative to /root/reference.
 *
 * Conventions
 *   - plain C: raw device pointers, sizes, a hipStream_t passed as void*; no torch types.
 *   - every function returns 0 on success, a positive hipError_t value if a HIP call
 *     failed, or a negative ME_E_* code for argument errors; me_last_error() gives text.
 *   - activations are fp32 channels-last ("NHWC"): element (n,y,x,c) of a tensor lives at
 *     ptr[((n*H + y)*W + x)*pitch + c], pitch >= C (pitch > C addresses a channel slice
 *     of a wider concat buffer - that is how darknet [route] concatenation costs nothing).
 *   - launches are asynchronous on `stream`; nothing here allocates or synchronises.
 */
#ifndef MILLIEYE_HIP_H
#define MILLIEYE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ME_ABI_VERSION 13

#define ME_E_BADARG (-1)   /* inconsistent / unsupported descriptor            */
#define ME_E_NULLPTR (-2)  /* required pointer is NULL                          */
#define ME_E_ALIGN (-3)    /* pointer or pitch not aligned as the kernel needs  */
#define ME_E_TOOBIG (-4)   /* size exceeds a kernel capacity (stated per call)  */

/* activation codes (conv epilogue) */
#define ME_ACT_LINEAR 0
#define ME_ACT_LEAKY 1   /* LeakyReLU(0.1): v > 0 ? v : 0.1f * v */
#define ME_ACT_SIGMOID 2 /* 1 / (1 + expf(-v))                    */

int me_abi_version(void);
const char* me_last_error(void);
/* number of CUs / max clock (kHz) / LDS bytes per workgroup of the current device */
int me_device_query(int32_t* cu_count, int32_t* clock_khz, int32_t* lds_bytes);

/* ------------------------------------------------------------------------------------
 * me_conv2d_f32 - fused  y = act(conv(x, w) * scale + shift) [+ residual]  [nearest x2]
 *
 * replaces: nn.Conv2d + nn.BatchNorm2d(eval) + nn.LeakyReLU(0.1) blocks built at
 *   module3_our_dataset/yolov3/models.py:22-41 and run at :252-253; the [shortcut] add
 *   at :258-260 (residual); the Upsample at :82-92 when it directly follows the conv;
 *   cnn_layers_1 (module3_our_dataset/my_models.py:62-77), cnn_layers_3 (:132-157).
 * BatchNorm (eval) and/or the conv bias are folded by the caller into per-output-channel
 *   (scale, shift):  scale = gamma / sqrt(var + eps), shift = beta - mean * scale
 *   (+ bias * scale); plain bias conv: scale = 1, shift = bias.
 * weights: fp32, packed [cout][ky][kx][cin] (i.e. OIHW -> OHWI), row length ksize*ksize*cin.
 * x: NHWC (pitch x_pitch), or NCHW when x_nchw != 0 (the network input, models.py:247).
 * residual (optional, NHWC [n,ho,wo,cout]) is added AFTER the activation (models.py:260).
 * upsample == 2 writes every output pixel to the 2x2 block (2y..2y+1, 2x..2x+1) of a
 *   [n,2*ho,2*wo] tensor (F.interpolate(scale_factor=2, mode="nearest")).
 * constraints: cin % 4 == 0 unless cin <= 4 (direct small-cin kernel, ksize 3, stride 1);
 *   x / wgt / y 16-byte aligned, pitches % 4 == 0 (NHWC).
 */
typedef struct me_conv_desc {
  const float* x;
  const float* wgt;
  const float* scale;
  const float* shift;
  const float* res; /* may be NULL */
  float* y;
  int64_t x_pitch;   /* elements between consecutive pixels of x (NHWC); ignored for NCHW */
  int64_t res_pitch;
  int64_t y_pitch;
  int32_t n, h, w, cin;
  int32_t cout, ksize, stride, pad;
  int32_t ho, wo;
  int32_t act;
  int32_t upsample; /* 1 or 2 */
  int32_t x_nchw;   /* 0 / 1 */
  int32_t tile;     /* 0 = auto; else force a tile config id (testing / tuning): 1-5 per-tap tiles 128x128 / 128x64 / 64x64 /
                       128x32 / 256x128; 41-45 the same with the last partial round of tiles cut split_k ways along K ("tail
                       split": compact slabs, me_conv2d_workspace_bytes says how much); >= 100 patch-resident tiles */
  int32_t split_k;  /* 0 = auto; 1 = never split; k > 1 forces k K-splits (needs the workspace) */
  void* workspace;  /* optional scratch for deterministic split-K slabs (256-byte aligned), or NULL */
  int64_t workspace_bytes;
  const float* wgt_tiled; /* second packing of the same weights, [ksize*ksize][cin/16][cout][16]: every (tap, 16-channel
                             chunk) slab of cout rows x 64 bytes is contiguous.  Tile ids >= 100 (patch-resident 3x3 / stride 1
                             kernels, csrc/conv_p8_f32.hip) need it; the per-tap MFMA kernel streams its weight tiles from it
                             when given (cin % 16 == 0): the OHWI rows of a tile lie ksize^2*cin*4 bytes apart and thrash the
                             L2 sets on the deep layers.  May be NULL. */
  int32_t* tile_counters; /* optional arrival counters for the in-launch split-K reduction: tile_counters_len ints, ALL ZERO on
                             entry, left all zero on return (the last workgroup of a tile to arrive sums the slabs in the fixed
                             order 0..k-1 and applies the epilogue - same bits as the two-pass form, one launch less).  NULL, or
                             fewer counters than tiles: the slabs are reduced by a second launch.  One array per stream. */
  int64_t tile_counters_len;
  /* ABI 10: per-column-class tap masks.  tap_mask_cols > 0: the output channels form cout / tap_mask_cols (<= 4) classes of
     tap_mask_cols consecutive channels; bit t of tap_mask[class] clear = tap t (ky * ksize + kx) of that class's filters is all
     zero and is SKIPPED (its K stages are neither fetched nor multiplied) - the caller guarantees the zeros.  Used by the data
     gradient of the 3x3 / stride-2 layers (one 2x2 convolution computes the four output-parity classes; 7 of its 16
     (class, tap) pairs are structurally zero, detector_train.py).  Needs ksize^2 <= 16, cin % 16 == 0, upsample 1, no K split
     (split_k <= 1), tile 0 - 5 with a tile width that divides tap_mask_cols; 0 = off (every tap of every channel). */
  uint32_t tap_mask[4];
  int32_t tap_mask_cols;
  int32_t reserved0;
} me_conv_desc;
int me_conv2d_f32(const me_conv_desc* d, void* stream);
/* scratch the automatic plan would like for this descriptor (0 = none). Small-M layers (13x13, 26x26
 * maps at small batch) are split along K so that every CU gets work; partial sums go to
 * workspace[split][n*ho*wo][cout] and a second launch reduces them in a FIXED order (bit-reproducible)
 * before applying the same fused epilogue.  Without a workspace the call never splits. */
int64_t me_conv2d_workspace_bytes(const me_conv_desc* d);
/* algorithmic FLOPs (2*MAC) of the descriptor - used by bench.py for the roofline */
int64_t me_conv2d_flops(const me_conv_desc* d);

/* ------------------------------------------------------------------------------------
 * me_maxpool_f32 - NHWC max pooling.
 * replaces: nn.MaxPool2d(size, stride, padding=(size-1)//2) and, for size 2 / stride 1, the
 *   preceding nn.ZeroPad2d((0,1,0,1)) (models.py:43-49): zero_ext != 0 extends the input by
 *   one row/column of ZEROS (value 0.0 takes part in the max - quirk q16), while `pad`
 *   behaves like torch's implicit -inf padding.
 */
typedef struct me_pool_desc {
  const float* x;
  float* y;
  int64_t x_pitch, y_pitch;
  int32_t n, h, w, c;
  int32_t size, stride, pad, zero_ext;
  int32_t ho, wo;
} me_pool_desc;
int me_maxpool_f32(const me_pool_desc* d, void* stream);

/* nearest-neighbour x`factor` upsample, NHWC (models.py:82-92) - stand-alone fallback */
int me_upsample_f32(const float* x, int64_t x_pitch, float* y, int64_t y_pitch, int32_t n, int32_t h,
                    int32_t w, int32_t c, int32_t factor, void* stream);
/* y = a + b over [pixels, c] with independent pitches ([shortcut] fallback, models.py:258-260) */
int me_add_f32(const float* a, int64_t a_pitch, const float* b, int64_t b_pitch, float* y, int64_t y_pitch,
               int64_t pixels, int32_t c, void* stream);
/* copy a channel slice ([route] fallback, torch.cat at models.py:257) */
int me_copy_f32(const float* x, int64_t x_pitch, float* y, int64_t y_pitch, int64_t pixels, int32_t c,
                void* stream);
/* NHWC (pitch) -> dense NCHW and back: API-boundary layout conversion only */
int me_nhwc_to_nchw_f32(const float* x, int64_t x_pitch, float* y, int32_t n, int32_t h, int32_t w, int32_t c,
                        void* stream);

/* ------------------------------------------------------------------------------------
 * me_yolo_decode_f32 - YOLO head decode for one scale.
 * replaces: YOLOLayer.forward inference branch, models.py:132-179 (+ compute_grid_offsets
 *   :119-129): x = raw detection conv output NHWC [n,g,g,A*(5+C)], channel a*(5+C)+k.
 *   out row r = row_offset + a*g*g + gy*g + gx of out[n, rows_total, 5+C]:
 *     k=0: (sigmoid(v) + gx) * stride      k=1: (sigmoid(v) + gy) * stride
 *     k=2: exp(v) * (anchor_w/stride) * stride   k=3: same with anchor_h
 *     k>=4: sigmoid(v)
 *   anchors: A pairs (w/stride, h/stride) - the reference's `scaled_anchors`, divided on the
 *   host in double precision and rounded to fp32 exactly like models.py:126 does (A <= 8).
 */
typedef struct me_yolo_desc {
  const float* x;
  float* out;
  int64_t x_pitch;
  int32_t n, g, num_anchors, num_classes;
  int32_t rows_total, row_offset;
  float stride;
  float anchors[16];
} me_yolo_desc;
int me_yolo_decode_f32(const me_yolo_desc* d, void* stream);

/* ------------------------------------------------------------------------------------
 * me_nms_batched_f32 - confidence filter + class-aware greedy NMS, per image.
 * replaces: non_max_suppression_cpp (module3_our_dataset/utils/utils.py:337-378) including
 *   xywh2xyxy (:68-74) and torchvision.ops.boxes.batched_nms / nms (utils.py:372; semantics
 *   restated in SURVEY.md Appendix C: offset trick boxes + label*(max+1), IoU without +1,
 *   suppress when IoU > iou_thresh, rank by objectness, ties -> lower row first).
 * pred: [n, rows, 5+C] (cx,cy,w,h,obj,cls...).  If writeback_xyxy != 0 the first four
 *   columns of every row are overwritten with (x1,y1,x2,y2) like the reference does in place.
 * det:  [n, max_det, 7+C] rows (x1,y1,x2,y2,obj,cls_conf,cls_pred,C scores); count[n] valid rows
 *   each (rows beyond count are left untouched).
 * workspace: me_nms_workspace_bytes(n, rows) bytes, 256-byte aligned, contents irrelevant.
 * capacity: rows <= 32768.
 */
typedef struct me_nms_desc {
  float* pred;
  float* det;
  int32_t* count;
  void* workspace;
  int32_t n, rows, num_classes, max_det;
  float conf_thresh, iou_thresh;
  int32_t writeback_xyxy;
} me_nms_desc;
int64_t me_nms_workspace_bytes(int32_t n, int32_t rows);
int me_nms_batched_f32(const me_nms_desc* d, void* stream);
/* The same split over the decode: me_yolo_decode_cand_f32 = me_yolo_decode_f32 of one [yolo] scale that also appends the rows
 * with objectness >= conf_thresh to the NMS candidate lists in `nms_workspace` (me_nms_workspace_bytes(n, rows_total) bytes,
 * 256-byte aligned; `first` != 0 on the first scale of a forward resets the lists) while the row is in registers - the
 * confidence filter of non_max_suppression_cpp (utils/utils.py:351-366) without re-reading [n, rows, 5 + C].  5 + C <= 128.
 * me_nms_batched_prepped_f32 then runs selection + emit on those lists (same desc as me_nms_batched_f32, same workspace,
 * writeback_xyxy must be 0).  Results are identical to me_yolo_decode_f32 + me_nms_batched_f32. */
int me_yolo_decode_cand_f32(const me_yolo_desc* y, float conf_thresh, void* nms_workspace, int32_t first, void* stream);
/* the same for 1 - 3 [yolo] scales of one forward in ONE launch (Darknet.forward's three decodes, yolov3/models.py:259-262, once
 * the last head convolution has run); the scales share n, rows_total and out. */
int me_yolo_decode_cand_multi_f32(const me_yolo_desc* const* ys, int32_t count, float conf_thresh, void* nms_workspace,
                                  int32_t first, void* stream);
int me_nms_batched_prepped_f32(const me_nms_desc* d, void* stream);

/* plain torchvision-style nms / batched_nms on explicit boxes (box_ops.* re-export used by
 *   run_sp.py:214 / run_mp.py:320).  boxes [m,4] xyxy, scores [m], labels [m] (float class ids,
 *   may be NULL -> plain nms).  keep[m] receives kept indices in descending-score order,
 *   *keep_count (device int32) their number.  workspace: me_nms_workspace_bytes(1, m). m <= 32768. */
int me_nms_boxes_f32(const float* boxes, const float* scores, const float* labels, int32_t m, float iou_thresh,
                     int64_t* keep, int32_t* keep_count, void* workspace, void* stream);

/* ------------------------------------------------------------------------------------
 * me_gather_class_boxes_f32 - proposal assembly.
 * replaces: the per-image python loop of Network.forward, my_models.py:459-473: keep detections
 *   with cls_pred == class_idx, emit rows (image_i, x1,y1,x2,y2, obj, cls_conf, cls_pred,
 *   score[0:class_num]) image-major / NMS order.  boxes: [n*max_det, 8+class_num]; *total (device).
 */
int me_gather_class_boxes_f32(const float* det, const int32_t* count, int32_t n, int32_t max_det,
                              int32_t num_classes, int32_t class_idx, int32_t class_num, float* boxes,
                              int32_t* total, void* stream);

/* ------------------------------------------------------------------------------------
 * me_roi_heads_f32 - RoI pooling + refinement head + ensemble head + output rows (inference).
 * replaces: torchvision.ops.ps_roi_align / roi_align calls at my_models.py:495-496 (semantics:
 *   SURVEY.md Appendix C), refinement_head.forward (:260-284), ensemble_head.forward (:202-210),
 *   mask/threshold/box_regress/output assembly (:502-535, box_regress :378-391).
 *
 * RoIs = the first *n_img rows of img_boxes (from me_gather_class_boxes_f32) followed by n_radar
 *   rows of radar_boxes [n_radar,5] (image_i, x1,y1,x2,y2 in pixels).
 * img_map:   NHWC [n, fh, fw, 490] (pitch img_pitch)  - cnn_layers_1 output
 * radar_map: NHWC [n, fh, fw, 10]  (pitch radar_pitch) - cnn_layers_3 output
 * per-RoI results (arrays sized cap = n_img_cap + n_radar, row k = RoI k):
 *   regress [cap,4], refine [cap,2], mask1 [cap] (p foreground), out_rows [cap,8]
 *   (image_i, x1,y1,x2,y2, p, cls_score, cls_pred), keep [cap] uint8 (p > threshold),
 *   sort_key [cap] (p for image proposals, p/5 for radar proposals - quirk q4).
 * weights (fp32, device): see me_heads_weights.
 */
typedef struct me_heads_weights {
  const float* w0t;   /* refinement_head.net0.0.weight transposed: [490][256] */
  const float* b0;    /* [256] */
  const float* w1;    /* net1.0.weight [4][256]  */
  const float* b1;    /* [4] */
  const float* w2;    /* net2.0.weight [13][256] (only rows 0,1 are used - quirk q2) */
  const float* b2;    /* [13] */
  const float* rw;    /* radar_net.0.weight [10][490] (10x10x7x7 flattened) */
  const float* rscale; /* [10] folded BN(eval) + conv bias: v*rscale + rshift */
  const float* rshift; /* [10] */
  const float* rw2;   /* radar_net.3.weight [10] */
  const float* rb2;   /* [1] */
  const float* e1w;   /* ensemble_head.fc1.0.weight [32][2] */
  const float* e1b;   /* [32] */
  const float* e2w;   /* ensemble_head.fc2.0.weight [2][64] */
  const float* e2b;   /* [2] */
  const float* rb;    /* radar_net.0.bias [10] - training mode only (inference folds it into rshift) */
} me_heads_weights;

typedef struct me_heads_desc {
  const float* img_map;
  const float* radar_map;
  int64_t img_pitch, radar_pitch;
  int32_t n, fh, fw;           /* batch, size of img_map */
  int32_t rh, rw;              /* size of radar_map (equal to fh, fw in training / evaluation; 32 x 32 in the
                                  reference's demos, run_sp.py:199-204 - same spatial_scale, quirk q15) */
  float spatial_scale;         /* 1/16 */
  const float* img_boxes;      /* [n_img_cap, box_cols] */
  const int32_t* n_img;        /* device scalar */
  int32_t n_img_cap, box_cols; /* box_cols = 8 + class_num (9) */
  const float* radar_boxes;    /* [n_radar,5] or NULL */
  int32_t n_radar;
  float thr_img, thr_radar;
  int32_t regress;             /* 1: box_regress the kept boxes (modes 0,3); 0: mode 2 */
  me_heads_weights wts;
  float* regress_out;
  float* refine_out;
  float* mask1_out;
  float* out_rows;
  uint8_t* keep;               /* [cap]: 1 = row kept; the launch also writes 0 into the slots behind the last RoI */
  float* sort_key;
  /* training mode (all four non-NULL): the launch stops after the pooled features / hidden layer /
   * small dot products and stores them for the backward pass instead of running the scalar tail:
   * save_feat_img [cap,490], save_feat_rad [cap,490], save_hidden [cap,256] (post LeakyReLU),
   * save_small [cap,16] = (reg0..3, z2_0, z2_1, rconv0..9) with biases added, pre-activation.
   * The tail then runs as me_heads_tail_f32 once the RoI-wise BatchNorm statistics are known. */
  float* save_feat_img;
  float* save_feat_rad;
  float* save_hidden;
  float* save_small;
  /* optional scratch [n_img_cap + n_radar, 980] floats: when given, the RoI pooling runs as its own launch (one workgroup per
   * RoI - 8 RoIs' 7840 bilinear samples per workgroup were two thirds of the fused launch, which has less than one workgroup
   * per CU to hide them behind) and the heads kernel reads the pooled features back; same samples, same arithmetic.  NULL =
   * the single fused launch. */
  float* pool_scratch;
  /* 1: img_map holds the 490 score-map channels bin-major - channel (ph * 7 + pw) * 10 + c_out instead of the reference's
   * (c_out * 7 + ph) * 7 + pw (my_models.py:47-52 feeds ps_roi_align) - so that the ten values a PS-RoIAlign sample point needs
   * are 40 contiguous bytes; the caller permutes the output channels of the 1x1 convolution that writes the map (its weight
   * rows), nothing else changes.  0: reference order. */
  int32_t img_bin_major;
} me_heads_desc;
int me_roi_heads_f32(const me_heads_desc* d, void* stream);
/* me_compact_sort_rows_f32 - the tail of Network.forward (my_models.py:517-539): keep the rows whose `keep` byte is set
 * and order them by descending `key`, equal keys in ascending row order ( = torch.sort(key[nonzero(keep)], descending,
 * stable) ).  rows [cap, cols], keep [cap] uint8, key [cap] (no NaN) -> out [cap, cols] with the first *count rows
 * valid; count [1] int32.  One launch: every kept row computes its own rank against all keys (broadcast LDS reads). */
int me_compact_sort_rows_f32(const float* rows, const uint8_t* keep, const float* key, int32_t cap, int32_t cols, float* out,
                             int32_t* count, void* stream);

/* Stage 2 (module2_mixed/my_models.py:299-364) heads for every box of every class: PS-RoIAlign (7x7, 490 -> 10 maps) on
 * img_map, refinement_head((490,256,class_num+1)) (net0 LeakyReLU, net1 -> regress [cap,4], net2 sigmoid -> refine
 * [cap,class_num+1]), ensemble_head((2,32,32*(class_num+1),2)) on (refine, (obj_conf, class scores)) with the LeakyReLU
 * after fc2 and the softmax, mask [cap] = masks[:,1]; out_rows [cap,8] = (image_i, box_regress(x1,y1,x2,y2), mask,
 * class_conf, class_pred), keep = mask > refine_threshold, sort_key = mask.  boxes [cap, box_cols >= 8+class_num] rows
 * (image_i, x1,y1,x2,y2, obj, cls_conf, cls_pred, class scores...), *n_boxes of them valid (device scalar).
 * Weights: w0t [490,256] (transposed), b0, w1 [4,256], b1, w2 [class_num+1,256], b2, e1w [32,2], e1b, e2w
 * [2,32*(class_num+1)], e2b of me_heads_weights; the radar fields are ignored. */
/* pool_scratch (round 3): optional [boxes_cap, 490] floats, 16-byte aligned - the PS-RoIAlign then runs as its own launch (one
 * workgroup per box), like me_heads_desc.pool_scratch; NULL = the single fused launch. */
int me_m2_heads_f32(const float* img_map, int64_t img_pitch, int32_t n, int32_t fh, int32_t fw, float spatial_scale,
                    const float* boxes, const int32_t* n_boxes, int32_t boxes_cap, int32_t box_cols, int32_t class_num,
                    const me_heads_weights* w, float refine_threshold, float* regress_out, float* refine_out,
                    float* mask_out, float* out_rows, uint8_t* keep, float* sort_key, float* pool_scratch, void* stream);

/* ---- stage-2 training building blocks (module2_mixed/my_models.py:96-164 heads, :366-459 objective) ------------------
 * me_linear_f32: y [rows,out] = act(x [rows,in] . w[out,in]^T + bias) (nn.Linear + Linear / LeakyReLU / Sigmoid).
 * me_mask_scale_f32: y = mask ? x * scale : 0 (nn.Dropout(0.5) in train mode, forward and backward; the mask is drawn
 *   by the caller - with torch's CPU generator like the reference's CPU run, or by me_dropout_mask_u8).
 * me_dropout_mask_u8 (ABI 13): the keep mask of nn.Dropout(1 - keep_prob) on the device - Philox4x32-10 (the generator family
 *   of torch's CUDA dropout) keyed by `seed`, counter = element index / 4, one 32-bit word per element: keep iff
 *   (word >> 8) / 2^24 < keep_prob.  Same (seed, count) -> same mask; the caller draws a fresh seed per step.
 * me_m2_loss_f32: per-RoI terms [k,5] = (focal on softmax(o), confidence BCE, category BCE, SmoothL1 xy, SmoothL1 wh) and
 *   the gradients d_o [k,2], d_refine [k,c1], d_regress [k,4] of
 *   focal + (conf + category)/lambda0 + (xy + wh)/lambda1, times grad_scale; pos / sample are the IoU-positive and the
 *   sampled rows, class_label [k,c1-1] and target_location [k,4] come from the host-side labelling. */
int me_linear_f32(const float* x, int64_t ldx, int64_t rows, int32_t in_features, const float* w, const float* bias,
                  int32_t out_features, int32_t act, float* y, int64_t ldy, void* stream);
int me_mask_scale_f32(const float* x, const uint8_t* mask, float scale, int64_t count, float* y, void* stream);
int me_dropout_mask_u8(uint64_t seed, float keep_prob, int64_t count, uint8_t* mask, void* stream);
/* me_m2_pairs_f32 (ABI 12): the ensemble head's input x2 [k * c1, 2] = (refinement_vector, yolo_vector) pairs
 *   (module2_mixed/my_models.py:333-339; yolo_vector = columns 5, 8 .. of the proposal rows).
 * me_m2_rows_f32 (ABI 12): the tail of the stage-2 forward (:341-364) per proposal - masks = softmax(o) [k,2], keep = masks[:,1] >
 *   threshold, box_regress, rows [k,8] = (image_i, x1,y1,x2,y2, masks[:,1], cls_score, cls_pred), key = masks[:,1]; order the kept
 *   rows with me_compact_sort_rows_f32 (= torch.sort(descending, stable)). */
int me_m2_pairs_f32(const float* refine, const float* boxes, int32_t box_cols, int32_t k, int32_t c1, float* x2, void* stream);
int me_m2_rows_f32(const float* o, const float* regress, const float* boxes, int32_t box_cols, int32_t k, float threshold,
                   float* masks, float* rows, uint8_t* keep, float* key, void* stream);
int me_m2_loss_f32(const float* o, const float* refine, int32_t c1, const float* regress, const float* boxes,
                   int32_t box_cols, const float* target_location, const float* class_label, const uint8_t* pos,
                   const uint8_t* sample, int32_t k, float alpha, float lambda0, float lambda1, float grad_scale,
                   float* terms, float* d_o, float* d_refine, float* d_regress, void* stream);

/* scalar tail of the heads for `k` RoIs from a saved `small` [k,16] block (training forward; identical
 * arithmetic to the fused inference tail).  d->wts.rscale/rshift must hold the radar_net BatchNorm as an
 * affine on rconv (batch statistics in train mode); reads d->img_boxes / n_img / radar_boxes, writes
 * regress_out, refine_out, mask1_out, out_rows, keep, sort_key. */
int me_heads_tail_f32(const me_heads_desc* d, const float* small, int32_t k, void* stream);

/* per-RoI loss terms and gradient seeds of the stage-3 objective (my_models.py:606-635):
 *   focal (alpha, gamma=2, sum) on [1-p, p] for rows with in_focal, BCE(sum)/lambda on conf for rows with
 *   in_conf; label_pos marks IoU-positive rows.  terms [k,2] = (focal_i, bce_i) (sum them with
 *   me_colsum_f32); seed_p [k] = dL/dp, seed_conf [k] = dL/dconf (already divided by lambda). */
/* IoU labels of the stage-3 proposals (my_models.py:317-375 with the call site's truthy multi_boxes, quirk q5; +1-pixel
 * IoU of utils.py:269-274): proposals = n_img rows of img_boxes [*, cols] (image_i, x1,y1,x2,y2, conf, cls_score, cls_pred, ..)
 * followed by n_radar rows of radar_boxes [*, 5] (image_i, x1,y1,x2,y2; class 0); targets [q,6] = (image_i, class, x1,y1,x2,y2)
 * in pixels.  out [k,4] = (first-maximum IoU over the same image / class targets or 0, keep flag, conf_1, conf_2) - everything
 * the host-side metric and negative sampling (python `random`, q7) read, in one buffer.  Bit-identical with the host code. */
int me_iou_labels_f32(const float* img_boxes, int32_t n_img, int32_t cols, const float* radar_boxes, int32_t n_radar,
                      const float* targets, int32_t q, const float* refine, const float* mask1, const uint8_t* keep, float* out,
                      void* stream);
int me_heads_loss_f32(const float* mask1, const float* refine, const uint8_t* label_pos, const uint8_t* in_focal,
                      const uint8_t* in_conf, int32_t k, float alpha, float conf_lambda, float* terms,
                      float* seed_p, float* seed_conf, void* stream);

/* backward of the scalar tail, one thread per RoI.  Inputs: saved small [k,16], refine [k,2], mask1 [k],
 * seeds, d (weights incl. rscale/rshift, img_boxes, n_img).  Outputs (row k = RoI k):
 *   g_o [k,2] dL/d(fc2 logits), g_hpre [k,64] dL/d(fc1 pre-activation), h_act [k,64] fc1 activations,
 *   xin [k,4] = (refine0, yolo0, refine1, yolo1), g_z2 [k,2] dL/d(net2 logits 0,1),
 *   g_rl [k,10] dL/d(radar_net LeakyReLU output), rl [k,10] that output, g_rlogit [k] dL/d(1x1 logit). */
int me_heads_tail_bwd_f32(const me_heads_desc* d, const float* small, const float* refine, const float* mask1,
                          const float* seed_p, const float* seed_conf, int32_t k, float* g_o, float* g_hpre,
                          float* h_act, float* xin, float* g_z2, float* g_rl, float* rl, float* g_rlogit,
                          void* stream);

/* ---- generic training building blocks (millieye_amd/csrc/train.hip) ----------------------- */
/* C[m,n] = alpha * op(A) * op(B) + beta * C, row-major with leading dimensions; op = transpose if
 * trans_* != 0 (A is stored [k,m] then).  Sequential K reduction per element: bit-reproducible. */
int me_gemm_f32(int32_t trans_a, int32_t trans_b, int32_t m, int32_t n, int32_t k, float alpha, const float* a,
                int64_t lda, const float* b, int64_t ldb, float beta, float* c, int64_t ldc, void* stream);
int me_colsum_f32(const float* x, int64_t ld, int32_t rows, int32_t cols, float* out, void* stream);
/* nn.BatchNorm2d in training mode over rows of x[rows, channels] (NHWC maps: rows = n*h*w):
 * batch mean / biased variance, y = act((x-mean)*rstd*gamma+beta), running stats updated in place
 * (momentum, unbiased variance) when running_mean != NULL.  workspace: me_bn_workspace_bytes(channels). */
int64_t me_bn_workspace_bytes(int32_t channels);
int me_bn_train_fwd_f32(const float* x, int64_t ldx, int32_t rows, int32_t channels, const float* gamma,
                        const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                        int32_t act, float* y, int64_t ldy, float* save_mean, float* save_var, float* save_rstd,
                        void* workspace, void* stream);
/* dy is the gradient w.r.t. the ACTIVATED output; dx / dgamma / dbeta may be NULL. */
int me_bn_train_bwd_f32(const float* x, int64_t ldx, const float* dy, int64_t lddy, int32_t rows, int32_t channels,
                        const float* gamma, const float* beta, const float* save_mean, const float* save_rstd,
                        int32_t act, float* dx, int64_t lddx, float* dgamma, float* dbeta, void* workspace,
                        void* stream);
/* ---- detector backward (Darknet.forward(x, targets) -> loss.backward(); reference yolov3/models.py:181-267 under
 * autograd, BatchNorm in eval mode = a per-channel affine) ---------------------------------------------------------
 * me_affine_act_bwd_f32: conv block y = act(scale*c + shift) with the stored output y [rows,channels] and dy:
 *   dc = dy * act'(y) * scale (scale NULL = 1), dshift[c] = sum dy*act'(y) (= d beta, or d bias when there is no BN),
 *   dgamma[c] = sum dy*act'(y) * xhat with xhat = (act^-1(y) - beta) / gamma (gamma/beta/dgamma NULL when no BN).
 *   act: linear or leaky.  workspace: me_affine_bwd_workspace_bytes(rows, channels).  Fixed-order reductions.
 * me_upsample2_bwd_f32: dx [n,h,w,c] += 2x2 block sums of dy [n,2h,2w,c] (nearest x2, models.py:82-92).
 * me_maxpool_bwd_f32: dx += dy routed to the first maximum of each window (MaxPool2d / ZeroPad2d+MaxPool2d, :43-49).
 * me_yolo_loss_bwd_f32: d(loss)/d(raw map) of one YOLOLayer (:196-214) from the build_targets tensors
 *   (obj / noobj masks [N,A,G,G] u8; tx,ty,tw,th,tconf [N,A,G,G]; tcls [N,A,G,G,C]), n_obj / n_noobj = mask counts,
 *   grad_scale = upstream gradient of the scalar loss; raw / draw are NHWC [N,G,G,A*(5+C)] with a pitch. */
int64_t me_affine_bwd_workspace_bytes(int32_t rows, int32_t channels);
int me_affine_act_bwd_f32(const float* y, int64_t ldy, const float* dy, int64_t lddy, int32_t rows, int32_t channels,
                          const float* scale, const float* gamma, const float* beta, int32_t act, float* dc, int64_t lddc,
                          float* dshift, float* dgamma, void* workspace, void* stream);
/* (ABI 11) me_affine_act_bwd_f32 with dshift == dgamma == NULL writes dc and leaves one row of partial sums per row chunk in
 * `workspace`; me_affine_bwd_sums_f32 adds them in chunk order (the same sums, the same bits) on any stream ordered behind the
 * first call.  vec4: non-zero when the first call's operands were 16-byte aligned with channels and pitches % 4 == 0. */
int me_affine_bwd_sums_f32(void* workspace, int32_t rows, int32_t channels, int32_t vec4, float* dshift, float* dgamma, void* stream);
/* me_affine_act_bwd_h16 (ABI 10): the same pass in a 16-bit storage mode (half_type 0 = bfloat16, 1 = IEEE half): y, dy and dc are
 * 16-bit [rows, channels] (channels and pitches % 8 == 0, 16-byte aligned), dc is rounded once (RNE); scale / gamma / beta and the
 * sums dshift / dgamma stay fp32 (double accumulation, fixed order).  The activation-gradient half of the mixed-precision detector
 * backward (millieye_amd/detector_train16.py; autograd semantics of yolov3/models.py:22-41,181-267).
 * workspace: me_affine_bwd_h16_workspace_bytes(rows, channels). */
int64_t me_affine_bwd_h16_workspace_bytes(int32_t rows, int32_t channels);
/* (ABI 11) me_affine_act_bwd_h16 with dshift == dgamma == NULL writes dc and leaves one row of partial sums per row chunk in
 * `workspace`; me_affine_bwd_h16_sums adds them in chunk order (the same sums, the same bits) - on any stream that is ordered
 * behind the first call: the detector backward runs it beside the data gradients (millieye_amd/detector_train16.py). */
int me_affine_bwd_h16_sums(const void* workspace, int32_t rows, int32_t channels, float* dshift, float* dgamma, void* stream);
int me_affine_act_bwd_h16(const void* y, int64_t ldy, const void* dy, int64_t lddy, int32_t rows, int32_t channels,
                          const float* scale, const float* gamma, const float* beta, int32_t act, void* dc, int64_t lddc,
                          float* dshift, float* dgamma, void* workspace, int32_t half_type, void* stream);
/* me_conv_wgrad_h16 (ABI 10): dW of y = conv(x, W) from 16-BIT x [n,h,w,cin] and dy [n,ho,wo,cout] (NHWC, pitched; channels and
 * pitches % 8 == 0, 16-byte aligned) on v_mfma_f32_32x32x16_bf16 / _f16 with fp32 accumulation; dw is FLOAT32, [cout][cin][k][k]
 * (oihw != 0: the parameter's layout) or [cout][k][k][cin].  Slices of the pixel reduction go through `workspace`
 * (me_conv_wgrad_workspace_bytes) and are added in a fixed order.  The weight-gradient half of the mixed-precision detector
 * backward (autograd semantics of yolov3/models.py:22-41,181-267). */
int me_conv_wgrad_h16(const void* x, int64_t x_pitch, const void* dy, int64_t dy_pitch, float* dw, int32_t n, int32_t h, int32_t w,
                      int32_t cin, int32_t cout, int32_t ksize, int32_t stride, int32_t pad, void* workspace, int64_t workspace_bytes,
                      int32_t oihw, int32_t half_type, void* stream);
int me_upsample2_bwd_f32(const float* dy, int64_t lddy, float* dx, int64_t lddx, int32_t n, int32_t h, int32_t w,
                         int32_t c, void* stream);
int me_maxpool_bwd_f32(const float* x, int64_t ldx, const float* dy, int64_t lddy, float* dx, int64_t lddx, int32_t n,
                       int32_t h, int32_t w, int32_t c, int32_t size, int32_t stride, int32_t pad, int32_t zero_ext,
                       void* stream);
int me_yolo_loss_bwd_f32(const float* raw, int64_t pitch, int32_t n, int32_t g, int32_t num_anchors, int32_t num_classes,
                         const uint8_t* obj_mask, const uint8_t* noobj_mask, const float* tx, const float* ty,
                         const float* tw, const float* th, const float* tcls, const float* tconf, float n_obj,
                         float n_noobj, float obj_scale, float noobj_scale, float grad_scale, float* draw,
                         int64_t dpitch, void* stream);
/* me_yolo_loss_bwd_dev_f32 (ABI 11): the same pass with n_obj / n_noobj taken from result_device[13] / [14] (the result[16] the
 * forward wrote) and the upstream gradient from the device float grad_scale_device (NULL = 1): no host value of the step is a
 * launch argument, so the call can be captured in a hipGraph (millieye_amd/detector_graph.py). */
int me_yolo_loss_bwd_dev_f32(const float* raw, int64_t pitch, int32_t n, int32_t g, int32_t num_anchors, int32_t num_classes,
                             const uint8_t* obj_mask, const uint8_t* noobj_mask, const float* tx, const float* ty,
                             const float* tw, const float* th, const float* tcls, const float* tconf,
                             const float* result_device, float obj_scale, float noobj_scale, const float* grad_scale_device,
                             float* draw, int64_t dpitch, void* stream);

/* dx = dy * act'(y) from the activation output y (sigmoid / leaky) */
int me_act_bwd_f32(const float* y, int64_t ldy, const float* dy, int64_t lddy, float* dx, int64_t lddx, int64_t rows,
                   int32_t channels, int32_t act, void* stream);
/* dW[cout][ky][kx][cin] = sum_pixels dy * x(shifted), NHWC operands (the packed-weight layout of me_conv2d_f32).
 * (The data gradient of a stride-1 conv is me_conv2d_f32 itself on the 180-degree rotated, transposed weights.) */
int me_conv_wgrad_f32(const float* x, int64_t x_pitch, const float* dy, int64_t dy_pitch, float* dw, int32_t n,
                      int32_t h, int32_t w, int32_t cin, int32_t cout, int32_t ksize, int32_t stride, int32_t pad,
                      void* stream);
/* the same weight gradient on the matrix pipe (v_mfma_f32_32x32x2_f32, reduction over the output pixels split into
 * slices whose slabs are added in a fixed order: deterministic).  workspace: me_conv_wgrad_workspace_bytes(n, ho, wo,
 * cin, cout, ksize) bytes (0 = no slicing needed); without it the kernel runs unsliced (correct, slower). */
int64_t me_conv_wgrad_workspace_bytes(int32_t n, int32_t ho, int32_t wo, int32_t cin, int32_t cout, int32_t ksize);
int me_conv_wgrad_mfma_f32(const float* x, int64_t x_pitch, const float* dy, int64_t dy_pitch, float* dw, int32_t n,
                           int32_t h, int32_t w, int32_t cin, int32_t cout, int32_t ksize, int32_t stride, int32_t pad,
                           void* workspace, int64_t workspace_bytes, void* stream);
/* the same, with dW written in the parameter's own layout [cout][cin][k][k] (what autograd hands to the optimizer): the
 * slab reduction does the transposition.  workspace: max(me_conv_wgrad_workspace_bytes(...), cout*k*k*cin*4) bytes for k > 1. */
int me_conv_wgrad_mfma_oihw_f32(const float* x, int64_t x_pitch, const float* dy, int64_t dy_pitch, float* dw, int32_t n,
                                int32_t h, int32_t w, int32_t cin, int32_t cout, int32_t ksize, int32_t stride, int32_t pad,
                                void* workspace, int64_t workspace_bytes, void* stream);
/* me_pack_conv_f32 - one launch from a conv block's parameters (nn.Conv2d weight [cout][cin][k][k] + bias, optional
 * nn.BatchNorm2d gamma / beta / running mean / running var, reference yolov3/models.py:22-41) to every packed copy the fp32
 * kernels read: ohwi [cout][k][k][cin] (me_conv_desc.wgt), tiled [k*k][cin/16][cout][16] (wgt_tiled; NULL or cin % 16 == 0),
 * rot [cin][k][k][cout] = the data gradient's weights (180-degree rotation, channels transposed; NULL = skip) with its tiled
 * copy rot_tiled [k*k][cout/16][cin][16] (NULL or cout % 16 == 0), and the folded (scale, shift) [cout] computed in double:
 * scale = gamma / sqrt(var + eps), shift = beta - mean * scale (+ bias * scale); gamma == NULL: scale = 1, shift = bias or 0.
 * Replaces ~25 torch launches per layer and training step (millieye_amd/engine.py:ConvWeights.refresh). */
int me_pack_conv_f32(const float* w_oihw, int32_t cout, int32_t cin, int32_t ksize, const float* bias, const float* gamma,
                     const float* beta, const float* mean, const float* var, float eps, float* ohwi, float* tiled,
                     float* rot, float* rot_tiled, float* scale, float* shift, void* stream);
/* The same for a whole network in ONE launch (a training step re-packs every layer: 75 launches of a few microseconds of
 * work each were ~1.4 ms of a Darknet-53 step).  The caller fills one me_pack_desc per conv block (same pointers and meaning as
 * the arguments of me_pack_conv_f32), me_pack_conv_plan validates them and fills first_block / blocks_x and returns the
 * grid size (< 0: error code), the table is copied to device memory once (the packed buffers and parameters of a module
 * tree are stable) and me_pack_conv_batch_f32 runs it. */
typedef struct me_pack_desc {
  const float* w;      /* [cout][cin][k][k] */
  const float* bias;   /* or NULL */
  const float* gamma;  /* BatchNorm weight or NULL (then beta / mean / var are ignored) */
  const float* beta;
  const float* mean;
  const float* var;
  float* ohwi;
  float* tiled;        /* or NULL */
  float* rot;          /* or NULL */
  float* rot_tiled;    /* or NULL */
  float* scale;
  float* shift;
  float* parity;       /* or NULL; 3x3 only: [4*cin][2][2][cout], the weights of the 2x2 convolution that computes the four
                        * output-parity classes of a stride-2 / pad-1 layer's data gradient at once (class = 2*py + px, channel
                        * class*cin + c; tap (i, j) of class (py, px) = W[ky(py,i)][kx(px,j)], ky(0,0)=1, ky(0,1)=none (zero),
                        * ky(1,0)=2, ky(1,1)=0) - millieye_amd/detector_train.py */
  int32_t cout, cin, ksize;
  float eps;
  int32_t first_block, blocks_x;  /* filled by me_pack_conv_plan */
  /* ABI 11: 16-bit copies written by the same launch (each NULL = not wanted): the values of ohwi / rot / parity rounded once
   * (RNE) to bfloat16 (half_type 0) or IEEE half (1) - the weights of the mixed-precision detector step
   * (millieye_amd/detector_train16.py), which then needs neither the fp32 rot / parity copies nor a conversion pass.  With
   * ohwi16 set, ohwi may be NULL. */
  void* ohwi16;
  void* rot16;
  void* parity16;
  int32_t half_type;
  int32_t reserved0;
} me_pack_desc;
int64_t me_pack_conv_plan(me_pack_desc* descs_host, int32_t count);
int me_pack_conv_batch_f32(const me_pack_desc* descs_device, int32_t count, int64_t total_blocks, int32_t max_ksize,
                           void* stream);
/* me_yolo_loss_fwd_f32 - the YOLO loss of one detection scale (module3_our_dataset/yolov3/models.py:181-232 with
 * utils/utils.py:381-440 build_targets) in three launches and one read-back: the dense build_targets tensors
 * (obj / noobj masks uint8 [N,A,G,G], tx / ty / tw / th / tconf / class_mask / iou_scores float [N,A,G,G], tcls
 * [N,A,G,G,C] - what me_yolo_loss_bwd_f32 takes) from targets [m,6] = (image, class, cx, cy, w, h) in [0,1] (device),
 * and result[16] (device floats): total loss, x, y, w, h, conf, cls, cls_acc, recall50, recall75, precision, conf_obj,
 * conf_noobj (the reference's metrics dict), n_obj, n_noobj, and a flag (1 = a target outside the batch / grid / class
 * range: the reference raises IndexError there).  scaled_anchors_host: 2 * num_anchors floats (anchor / stride), host
 * memory, num_anchors <= 16.  Two targets owning the same cell: the later one wins (the reference's CPU index_put_
 * order).  workspace: me_yolo_loss_workspace_bytes() bytes, 16-byte aligned, ZERO on first use (the call leaves its
 * ticket words zero again).  Sums in double, added in a fixed order: deterministic. */
int64_t me_yolo_loss_workspace_bytes(void);
int me_yolo_loss_fwd_f32(const float* raw, int64_t pitch, int32_t n, int32_t g, int32_t num_anchors, int32_t num_classes,
                         const float* scaled_anchors_host, const float* targets, int32_t m, float ignore_thres, float obj_scale,
                         float noobj_scale, uint8_t* obj_mask, uint8_t* noobj_mask, float* tx, float* ty, float* tw, float* th,
                         float* tcls, float* tconf, float* class_mask, float* iou_scores, void* workspace, float* result,
                         void* stream);
/* me_yolo_loss_fwd_counted_f32 (ABI 11): the same launches over a fixed-capacity target table: `capacity` rows of `targets` are
 * addressable, the device word *m_device (clamped to [0, capacity]) says how many of them are targets of this step. */
int me_yolo_loss_fwd_counted_f32(const float* raw, int64_t pitch, int32_t n, int32_t g, int32_t num_anchors, int32_t num_classes,
                                 const float* scaled_anchors_host, const float* targets, int32_t capacity,
                                 const int32_t* m_device, float ignore_thres, float obj_scale, float noobj_scale, uint8_t* obj_mask,
                                 uint8_t* noobj_mask, float* tx, float* ty, float* tw, float* th, float* tcls, float* tconf,
                                 float* class_mask, float* iou_scores, void* workspace, float* result, void* stream);
/* RoI pooling backward: grad_out [k,c_out,7,7] scattered (atomicAdd) into the zero-filled NHWC grad_map */
int me_roi_align_bwd_f32(const float* grad_out, const float* rois, int32_t k, int32_t n, int32_t h, int32_t w,
                         int32_t c, int32_t pooled, float spatial_scale, float* grad_map, int64_t pitch, void* stream);
int me_ps_roi_align_bwd_f32(const float* grad_out, const float* rois, int32_t k, int32_t n, int32_t h, int32_t w,
                            int32_t c, int32_t pooled, float spatial_scale, float* grad_map, int64_t pitch,
                            void* stream);
/* ---- row counts from device memory (ABI 12): the stage-3 backward as ONE captured hipGraph (millieye_amd/train_path.py) --------
 * The number of proposals changes from step to step (n_img from the NMS + the radar boxes), and every launch argument of a captured
 * graph is fixed; these forms take the buffers' CAPACITY as the launch size and the live row count from a device word: rows behind it
 * produce zeros (so the dense gradient products over the whole capacity add exact zeros) and cost nothing in the scatter kernels.
 *   me_heads_tail_bwd_dev_f32     me_heads_tail_bwd_f32 over `cap` rows, zeros behind *k_dev
 *   me_bn_train_bwd_dev_f32       me_bn_train_bwd_f32 with the statistics' row count = min(*rows_dev, rows_cap), dx = 0 behind it
 *   me_[ps_]roi_align_bwd_dev_f32 the RoI scatters over min(*k_dev, k_cap) RoIs */
int me_heads_tail_bwd_dev_f32(const me_heads_desc* d, const float* small, const float* refine, const float* mask1,
                              const float* seed_p, const float* seed_conf, int32_t cap, const int32_t* k_dev, float* g_o,
                              float* g_hpre, float* h_act, float* xin, float* g_z2, float* g_rl, float* rl, float* g_rlogit,
                              void* stream);
int me_bn_train_bwd_dev_f32(const float* x, int64_t ldx, const float* dy, int64_t lddy, int32_t rows_cap, const int32_t* rows_dev,
                            int32_t channels, const float* gamma, const float* beta, const float* save_mean,
                            const float* save_rstd, int32_t act, float* dx, int64_t lddx, float* dgamma, float* dbeta,
                            void* workspace, void* stream);
int me_roi_align_bwd_dev_f32(const float* grad_out, const float* rois, int32_t k_cap, const int32_t* k_dev, int32_t n, int32_t h,
                             int32_t w, int32_t c, int32_t pooled, float spatial_scale, float* grad_map, int64_t pitch,
                             void* stream);
int me_ps_roi_align_bwd_dev_f32(const float* grad_out, const float* rois, int32_t k_cap, const int32_t* k_dev, int32_t n, int32_t h,
                                int32_t w, int32_t c, int32_t pooled, float spatial_scale, float* grad_map, int64_t pitch,
                                void* stream);

/* stand-alone RoI ops (tests, training path): out [k, c_out, 7, 7] dense */
int me_roi_align_f32(const float* map, int64_t pitch, int32_t n, int32_t h, int32_t w, int32_t c,
                     const float* rois, int32_t k, int32_t pooled, float spatial_scale, float* out,
                     void* stream);
int me_ps_roi_align_f32(const float* map, int64_t pitch, int32_t n, int32_t h, int32_t w, int32_t c,
                        const float* rois, int32_t k, int32_t pooled, float spatial_scale, float* out,
                        void* stream);

/* ---- input producer (module3_our_dataset/utils/datasets.py) -------------------------------
 * me_image_pad_resize_u8_f32: one decoded frame, uint8 HWC RGB [h,w,3] on the device -> float32 CHW [3,size,size]:
 *   transforms.ToTensor() (datasets.py:203: /255), pad_to_square(img, 0) (:16-27, :211: the shorter side is padded
 *   with floor(diff/2) before and the rest after), resize(img, size) = F.interpolate(mode="nearest") (:30-32, :317).
 * me_radar_heatmap_f32: radar points -> the radar map of every frame of a batch.  points [total,4] float64 rows
 *   (u, v, depth, velocity) of all frames back to back, offsets [n+1] the row range of frame i, sizes [n,2] the
 *   original (w, h) of its image.  Per frame: plot_radar_heatmap(points.T, (w,h), radar_maps_size) (:59-106: count /
 *   mean depth / mean |velocity| histograms in float64, np.histogram2d bin rules, range normalisation + clip),
 *   ToTensor().float() (:267), pad_to_square (:270), F.interpolate(bilinear, align_corners=True) to
 *   [3,map_size,map_size] (collate_fn :318-321).  out [n,3,map_size,map_size]. */
int me_image_pad_resize_u8_f32(const uint8_t* src, int32_t h, int32_t w, float* dst, int32_t size, void* stream);
/* the same with the stage-2 augmentation (module2_mixed/utils/datasets.py:143-146, utils/augmentations.py:6-9): flip != 0
 * mirrors the PADDED square left-right (horisontal_flip runs between pad_to_square and collate_fn's resize). */
int me_image_pad_resize_flip_u8_f32(const uint8_t* src, int32_t h, int32_t w, float* dst, int32_t size, int32_t flip,
                                    void* stream);
int me_radar_heatmap_f32(const double* points, const int32_t* offsets, const int32_t* sizes, int32_t n,
                         int32_t radar_maps_size, float* out, int32_t map_size, void* stream);

/* ---- evaluation tail -----------------------------------------------------------------
 * get_batch_statistics (module3_our_dataset/utils/utils.py:185-236) for one batch, on the output rows of
 * Network.forward as they are: rows [m,cols] = (image_i, x1,y1,x2,y2, ..., class_pred last), targets [q,6] =
 * (image_i, class, x1,y1,x2,y2) already in pixels (test_fusion.py:95-96).  tp [m] <- 1.0 where the reference's greedy
 * scan marks a true positive, else 0.0 (rows of images >= n_images are left untouched).  One wave per image. */
int me_batch_statistics_f32(const float* rows, int32_t m, int32_t cols, const float* targets, int32_t q,
                            int32_t n_images, float iou_threshold, float* tp, void* stream);

/* ---- 16-bit storage modes (BASELINE configs[2] / [4]: "bf16 inference", "fp16 MFMA convs") ------------------------
 * The same conv block as me_conv2d_f32 (models.py:22-41) with 16-bit activations and weights - bfloat16 (half_type 0,
 * v_mfma_f32_32x32x16_bf16) or IEEE half (half_type 1, v_mfma_f32_32x32x16_f16) - and fp32 accumulation.  Opt-in; the fp32 entry points stay the default and the ones the 1e-3 parity bar is quoted on.
 *   x    bf16 NHWC (pitches in elements, %% 8);   stem only (cin == 3): float32 NCHW (x_nchw = 1) or NHWC
 *   wgt  bf16 [cout][k][k][cin], cin %% 32 == 0;  stem: float32 [cout][3][3][3]
 *   res  same type as y (or NULL);  y bf16, or float32 when y_f32 != 0 (the detection convs that feed me_yolo_decode_f32)
 * Rounding points: weights once on the host, every stored activation once (RNE, after activation + residual). */
typedef struct me_conv16_desc {
  const void* x;
  const void* wgt;
  const float* scale;
  const float* shift;
  const void* res;
  void* y;
  int64_t x_pitch, res_pitch, y_pitch;
  int32_t n, h, w, cin;
  int32_t cout, ksize, stride, pad;
  int32_t ho, wo;
  int32_t act;
  int32_t upsample;
  int32_t x_nchw;
  int32_t y_f32;
  int32_t half_type; /* 0 = bfloat16, 1 = IEEE half (x, wgt, res / y unless y_f32) */
  int32_t tile;    /* 0 = auto; 1..4 = 128x128 / 128x64 / 64x64 / 256x128; 11..14 = single sub-stage variants; 5 / 15 = 192x128; 40 / 41 = small-batch tiles 32x32 / 32x64 with the K split over the eight waves of a workgroup (one launch, no slabs; conv_kw_h16.hip); >= 100: patch-resident big tiles (conv_p8_h16.hip; need wgt_tiled) */
  int32_t split_k; /* as me_conv_desc */
  void* workspace;
  int64_t workspace_bytes;
  const void* wgt_tiled; /* tile ids >= 100 (patch-resident 3x3 kernels) read the weights from this second packing:
                            [ksize*ksize][cin/32][cout][32] - every (tap, 32-channel chunk) slab of cout rows x 64 bytes is
                            contiguous, so one LDS-DMA instruction moves 1 KiB of whole 128-byte lines.  May be NULL otherwise. */
  int32_t* tile_counters; /* as in me_conv_desc: arrival counters of the in-launch split-K reduction (zero on entry and exit) */
  int64_t tile_counters_len;
  /* ABI 13: per-column-class tap masks, as me_conv_desc.tap_mask (ABI 10) - tap_mask_cols > 0: the output channels form
     cout / tap_mask_cols (<= 4) classes of tap_mask_cols consecutive channels; bit t of tap_mask[class] clear = tap t
     (ky * ksize + kx) of that class's filters is all zero and is skipped (the stride-2 data gradient of the 16-bit training step:
     7 of the 16 (class, tap) pairs of its 2x2 parity convolution).  Per-tap tiles 1 / 2 / 3 / 11 / 12 / 13 only, whole tiles
     (split_k <= 1), the tile width must divide tap_mask_cols.  A zeroed tail keeps the previous behaviour. */
  uint32_t tap_mask[4];
  int32_t tap_mask_cols;
  int32_t reserved1;
} me_conv16_desc;
int me_conv2d_h16(const me_conv16_desc* d, void* stream);
int64_t me_conv2d_h16_workspace_bytes(const me_conv16_desc* d);
/* ---- one launch for a Darknet bottleneck in the 16-bit storage modes (ABI 12; csrc/bneck_h16.hip) ------------------
 * Replaces the launch PAIR of a [convolutional] 1x1 block and the [convolutional] 3x3 / stride-1 / pad-1 block behind it
 * (+ the [shortcut] that adds the pair's input; reference yolov3/models.py:22-41, 258-260 - the eight 52x52 and the two
 * 104x104 residual blocks of yolov3.cfg and the 1x1 -> 3x3 pairs of its 52x52 head):
 *     mid = act1(scale1 * conv1x1(x, w1) + shift1)   rounded to the storage type, kept in LDS, never written
 *     y   = act2(scale2 * conv3x3(mid, w2) + shift2) (+ res)
 * Same rounding points as the two me_conv2d_h16 launches (mid and y once each, RNE); results equal theirs bit for bit
 * whenever the two paths sum each dot product in the same order (they do for the whole-tile kernels; tests/test_gpu_h16.py).
 *   x, res, y  16-bit NHWC, pitches in elements (%% 8), 16-byte aligned;  res may be NULL
 *   w1_tiled   [cin / 32][cmid][32]        = me_conv16_desc.wgt_tiled of the 1x1 block
 *   w2_tiled   [9][cmid / 32][cout][32]    = me_conv16_desc.wgt_tiled of the 3x3 block
 *   tile       1: 192 positions x (cmid 128, cout 256)   3: 512 x (cmid 64, cout 128)   4: 256 x (64, 128)
 * me_bneck_h16_supported() says whether (tile, channels, map width) has an instance; me_bneck_h16 refuses loudly otherwise. */
typedef struct me_bneck16_desc {
  const void* x;
  const void* w1_tiled;
  const float* scale1;
  const float* shift1;
  const void* w2_tiled;
  const float* scale2;
  const float* shift2;
  const void* res;
  void* y;
  int64_t x_pitch, res_pitch, y_pitch;
  int32_t n, h, w, cin;
  int32_t cmid, cout, act1, act2;
  int32_t half_type; /* 0 = bfloat16, 1 = IEEE half */
  int32_t tile;
} me_bneck16_desc;
int me_bneck_h16(const me_bneck16_desc* d, void* stream);
int me_bneck_h16_supported(const me_bneck16_desc* d);
/* 16-bit NHWC twins of me_maxpool_f32 / me_upsample_f32 / me_add_f32 / me_copy_f32 (channels and pitches %% 8); upsample
 * and copy move bytes and serve both types */
int me_maxpool_h16(const void* x, int64_t x_pitch, void* y, int64_t y_pitch, int32_t n, int32_t h, int32_t w, int32_t c,
                   int32_t size, int32_t stride, int32_t pad, int32_t zero_ext, int32_t ho, int32_t wo, int32_t half_type,
                   void* stream);
int me_upsample_h16(const void* x, int64_t x_pitch, void* y, int64_t y_pitch, int32_t n, int32_t h, int32_t w, int32_t c,
                     int32_t factor, void* stream);
int me_add_h16(const void* a, int64_t a_pitch, const void* b, int64_t b_pitch, void* y, int64_t y_pitch, int64_t pixels,
               int32_t c, int32_t half_type, void* stream);
int me_copy_h16(const void* x, int64_t x_pitch, void* y, int64_t y_pitch, int64_t pixels, int32_t c, void* stream);

/* ---- optimizer step (ABI 13; csrc/optim.hip) -------------------------------------------------------------------------
 * optimizer.step() of the reference's training loops - torch.optim.Adam(lr=5e-4) in module3_our_dataset/train.py:161,196-197,
 * torch.optim.AdamW(lr=1e-4) in module2/train.py:122,150-151 - for up to ME_ADAM_MAX_TENSORS fp32 tensors in ONE launch, the
 * arithmetic of torch/optim/adam.py:_single_tensor_adam element by element (amsgrad / maximize off), every product, quotient and
 * sum rounded on its own like the sequence of tensor operations:
 *   decoupled != 0 (AdamW): p = p * decay;     decoupled == 0 and weight_decay != 0 (Adam): g = g + weight_decay * p
 *   m = m + one_minus_beta1 * (g - m)                       (lerp_)
 *   v = v * beta2;  v = v + (one_minus_beta2 * g) * g       (mul_, addcmul_)
 *   p = p + (neg_step_size * m) / (sqrt(v) / bias_correction2_sqrt + eps)      (addcdiv_)
 * The scalars are what the Python code of the torch class computes in double and hands to the tensor ops, rounded to fp32 once
 * by the caller: one_minus_beta1 = 1 - beta1 (< 0.5: lerp_'s first formula), one_minus_beta2 = 1 - beta2,
 * decay = 1 - lr * weight_decay, neg_step_size = -(lr / (1 - beta1^t)), bias_correction2_sqrt = sqrt(1 - beta2^t), t = the step
 * count after this step (>= 1).  first_chunk[i] = sum over j < i of ceil(numel[j] / me_adam_chunk()): the workgroups of tensor
 * i.  grad is read only; param, exp_avg, exp_avg_sq are updated in place. */
#define ME_ADAM_MAX_TENSORS 64
typedef struct me_adam_desc {
  float* param[ME_ADAM_MAX_TENSORS];
  const float* grad[ME_ADAM_MAX_TENSORS];
  float* exp_avg[ME_ADAM_MAX_TENSORS];
  float* exp_avg_sq[ME_ADAM_MAX_TENSORS];
  int64_t numel[ME_ADAM_MAX_TENSORS];
  int32_t first_chunk[ME_ADAM_MAX_TENSORS];
  int32_t count;
  int32_t decoupled;
  float beta2, eps, weight_decay, decay;
  float one_minus_beta1, one_minus_beta2;
  float neg_step_size, bias_correction2_sqrt;
} me_adam_desc;
int me_adam_step_f32(const me_adam_desc* d, void* stream);
int32_t me_adam_chunk(void);

/* sizes of the descriptor structs, so a binding can assert its mirror layout */
int32_t me_sizeof(int32_t which); /* 0 conv, 1 pool, 2 yolo, 3 nms, 4 heads, 5 heads_weights, 6 conv16, 7 pack, 8 bneck16, 9 adam */

#ifdef __cplusplus
}
#endif
#endif /* MILLIEYE_HIP_H */
