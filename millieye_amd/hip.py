"""ctypes binding of ``libmillieye_hip.so`` (C ABI: ``include/millieye_hip.h``).

This is the whole Python<->native boundary: struct mirrors, a lazy loader that fails
loudly when the library is missing, and thin tensor-level wrappers used by the host-side
modules.  PyTorch supplies device memory (``tensor.data_ptr()``) and the current HIP
stream; nothing here computes.

The library is built in-tree by ``__graft_entry__.build()`` (``hipcc --offload-arch=gfx950``)
as ``millieye_amd/libmillieye_hip.so``.
"""
import ctypes as C
import os

import torch

__all__ = ["lib", "available", "default_device", "MeError"]

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MILLIEYE_HIP_LIB") or os.path.join(_HERE, "libmillieye_hip.so")  # (the override: A/B runs of two builds)

ACT_LINEAR, ACT_LEAKY, ACT_SIGMOID = 0, 1, 2


class MeError(RuntimeError):
    pass


class ConvDesc(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("wgt", C.c_void_p), ("scale", C.c_void_p), ("shift", C.c_void_p),
        ("res", C.c_void_p), ("y", C.c_void_p),
        ("x_pitch", C.c_int64), ("res_pitch", C.c_int64), ("y_pitch", C.c_int64),
        ("n", C.c_int32), ("h", C.c_int32), ("w", C.c_int32), ("cin", C.c_int32),
        ("cout", C.c_int32), ("ksize", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
        ("ho", C.c_int32), ("wo", C.c_int32),
        ("act", C.c_int32), ("upsample", C.c_int32), ("x_nchw", C.c_int32), ("tile", C.c_int32),
        ("split_k", C.c_int32),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64),
        ("wgt_tiled", C.c_void_p),
        ("tile_counters", C.c_void_p), ("tile_counters_len", C.c_int64),
        ("tap_mask", C.c_uint32 * 4), ("tap_mask_cols", C.c_int32), ("reserved0", C.c_int32),
    ]


class Conv16Desc(C.Structure):
    """me_conv16_desc: the 16-bit-storage twin of ConvDesc (adds ``y_f32`` and ``half_type``: 0 bfloat16, 1 IEEE half)."""
    _fields_ = [
        ("x", C.c_void_p), ("wgt", C.c_void_p), ("scale", C.c_void_p), ("shift", C.c_void_p),
        ("res", C.c_void_p), ("y", C.c_void_p),
        ("x_pitch", C.c_int64), ("res_pitch", C.c_int64), ("y_pitch", C.c_int64),
        ("n", C.c_int32), ("h", C.c_int32), ("w", C.c_int32), ("cin", C.c_int32),
        ("cout", C.c_int32), ("ksize", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
        ("ho", C.c_int32), ("wo", C.c_int32),
        ("act", C.c_int32), ("upsample", C.c_int32), ("x_nchw", C.c_int32), ("y_f32", C.c_int32),
        ("half_type", C.c_int32), ("tile", C.c_int32), ("split_k", C.c_int32),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64),
        ("wgt_tiled", C.c_void_p),
        ("tile_counters", C.c_void_p), ("tile_counters_len", C.c_int64),
        ("tap_mask", C.c_uint32 * 4), ("tap_mask_cols", C.c_int32), ("reserved1", C.c_int32),
    ]


class PoolDesc(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("y", C.c_void_p),
        ("x_pitch", C.c_int64), ("y_pitch", C.c_int64),
        ("n", C.c_int32), ("h", C.c_int32), ("w", C.c_int32), ("c", C.c_int32),
        ("size", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32), ("zero_ext", C.c_int32),
        ("ho", C.c_int32), ("wo", C.c_int32),
    ]


class YoloDesc(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("out", C.c_void_p),
        ("x_pitch", C.c_int64),
        ("n", C.c_int32), ("g", C.c_int32), ("num_anchors", C.c_int32), ("num_classes", C.c_int32),
        ("rows_total", C.c_int32), ("row_offset", C.c_int32),
        ("stride", C.c_float),
        ("anchors", C.c_float * 16),
    ]


class NmsDesc(C.Structure):
    _fields_ = [
        ("pred", C.c_void_p), ("det", C.c_void_p), ("count", C.c_void_p), ("workspace", C.c_void_p),
        ("n", C.c_int32), ("rows", C.c_int32), ("num_classes", C.c_int32), ("max_det", C.c_int32),
        ("conf_thresh", C.c_float), ("iou_thresh", C.c_float),
        ("writeback_xyxy", C.c_int32),
    ]


class HeadsWeights(C.Structure):
    _fields_ = [(name, C.c_void_p) for name in (
        "w0t", "b0", "w1", "b1", "w2", "b2", "rw", "rscale", "rshift", "rw2", "rb2", "e1w", "e1b", "e2w", "e2b",
        "rb")]


class HeadsDesc(C.Structure):
    _fields_ = [
        ("img_map", C.c_void_p), ("radar_map", C.c_void_p),
        ("img_pitch", C.c_int64), ("radar_pitch", C.c_int64),
        ("n", C.c_int32), ("fh", C.c_int32), ("fw", C.c_int32), ("rh", C.c_int32), ("rw", C.c_int32),
        ("spatial_scale", C.c_float),
        ("img_boxes", C.c_void_p), ("n_img", C.c_void_p),
        ("n_img_cap", C.c_int32), ("box_cols", C.c_int32),
        ("radar_boxes", C.c_void_p),
        ("n_radar", C.c_int32),
        ("thr_img", C.c_float), ("thr_radar", C.c_float),
        ("regress", C.c_int32),
        ("wts", HeadsWeights),
        ("regress_out", C.c_void_p), ("refine_out", C.c_void_p), ("mask1_out", C.c_void_p),
        ("out_rows", C.c_void_p), ("keep", C.c_void_p), ("sort_key", C.c_void_p),
        ("save_feat_img", C.c_void_p), ("save_feat_rad", C.c_void_p), ("save_hidden", C.c_void_p),
        ("save_small", C.c_void_p),
        ("pool_scratch", C.c_void_p),
        ("img_bin_major", C.c_int32),
    ]


class PackDesc(C.Structure):
    """me_pack_desc: one conv block of a me_pack_conv_batch_f32 table."""
    _fields_ = [
        ("w", C.c_void_p), ("bias", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p), ("mean", C.c_void_p),
        ("var", C.c_void_p),
        ("ohwi", C.c_void_p), ("tiled", C.c_void_p), ("rot", C.c_void_p), ("rot_tiled", C.c_void_p), ("scale", C.c_void_p),
        ("shift", C.c_void_p), ("parity", C.c_void_p),
        ("cout", C.c_int32), ("cin", C.c_int32), ("ksize", C.c_int32), ("eps", C.c_float),
        ("first_block", C.c_int32), ("blocks_x", C.c_int32),
        ("ohwi16", C.c_void_p), ("rot16", C.c_void_p), ("parity16", C.c_void_p), ("half_type", C.c_int32), ("reserved0", C.c_int32),
    ]


class Bneck16Desc(C.Structure):
    """me_bneck16_desc: a 1x1 -> 3x3 (+ shortcut) bottleneck of the 16-bit storage modes as ONE launch (csrc/bneck_h16.hip)."""
    _fields_ = [
        ("x", C.c_void_p), ("w1_tiled", C.c_void_p), ("scale1", C.c_void_p), ("shift1", C.c_void_p),
        ("w2_tiled", C.c_void_p), ("scale2", C.c_void_p), ("shift2", C.c_void_p), ("res", C.c_void_p), ("y", C.c_void_p),
        ("x_pitch", C.c_int64), ("res_pitch", C.c_int64), ("y_pitch", C.c_int64),
        ("n", C.c_int32), ("h", C.c_int32), ("w", C.c_int32), ("cin", C.c_int32),
        ("cmid", C.c_int32), ("cout", C.c_int32), ("act1", C.c_int32), ("act2", C.c_int32),
        ("half_type", C.c_int32), ("tile", C.c_int32),
    ]


ADAM_MAX_TENSORS = 64


class AdamDesc(C.Structure):
    """me_adam_desc: one optimizer step over up to 64 fp32 tensors (csrc/optim.hip)."""
    _fields_ = [
        ("param", C.c_void_p * ADAM_MAX_TENSORS), ("grad", C.c_void_p * ADAM_MAX_TENSORS),
        ("exp_avg", C.c_void_p * ADAM_MAX_TENSORS), ("exp_avg_sq", C.c_void_p * ADAM_MAX_TENSORS),
        ("numel", C.c_int64 * ADAM_MAX_TENSORS), ("first_chunk", C.c_int32 * ADAM_MAX_TENSORS),
        ("count", C.c_int32), ("decoupled", C.c_int32),
        ("beta2", C.c_float), ("eps", C.c_float), ("weight_decay", C.c_float), ("decay", C.c_float),
        ("one_minus_beta1", C.c_float), ("one_minus_beta2", C.c_float),
        ("neg_step_size", C.c_float), ("bias_correction2_sqrt", C.c_float),
    ]


_STRUCTS = {0: ConvDesc, 1: PoolDesc, 2: YoloDesc, 3: NmsDesc, 4: HeadsDesc, 5: HeadsWeights, 6: Conv16Desc, 7: PackDesc,
            8: Bneck16Desc, 9: AdamDesc}

# name -> (restype, argtypes); every symbol include/millieye_hip.h declares
SIGNATURES = {
    "me_abi_version": (C.c_int, []),
    "me_last_error": (C.c_char_p, []),
    "me_device_query": (C.c_int, [C.POINTER(C.c_int32)] * 3),
    "me_sizeof": (C.c_int32, [C.c_int32]),
    "me_adam_step_f32": (C.c_int, [C.POINTER(AdamDesc), C.c_void_p]),
    "me_adam_chunk": (C.c_int32, []),
    "me_batch_statistics_f32": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_float,
                                          C.c_void_p, C.c_void_p]),
    "me_image_pad_resize_u8_f32": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]),
    "me_image_pad_resize_flip_u8_f32": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32,
                                                  C.c_void_p]),
    "me_radar_heatmap_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32,
                                       C.c_void_p]),
    "me_conv2d_f32": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p]),
    "me_conv2d_flops": (C.c_int64, [C.POINTER(ConvDesc)]),
    "me_conv2d_workspace_bytes": (C.c_int64, [C.POINTER(ConvDesc)]),
    "me_compact_sort_rows_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                           C.c_void_p]),
    "me_conv2d_h16": (C.c_int, [C.POINTER(Conv16Desc), C.c_void_p]),
    "me_bneck_h16": (C.c_int, [C.POINTER(Bneck16Desc), C.c_void_p]),
    "me_bneck_h16_supported": (C.c_int, [C.POINTER(Bneck16Desc)]),
    "me_conv2d_h16_workspace_bytes": (C.c_int64, [C.POINTER(Conv16Desc)]),
    "me_maxpool_h16": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64] + [C.c_int32] * 11 + [C.c_void_p]),
    "me_upsample_h16": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32,
                                   C.c_int32, C.c_int32, C.c_void_p]),
    "me_add_h16": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64,
                             C.c_int32, C.c_int32, C.c_void_p]),
    "me_copy_h16": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p]),
    "me_maxpool_f32": (C.c_int, [C.POINTER(PoolDesc), C.c_void_p]),
    "me_upsample_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32,
                                  C.c_int32, C.c_int32, C.c_void_p]),
    "me_add_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64,
                             C.c_int32, C.c_void_p]),
    "me_copy_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p]),
    "me_nhwc_to_nchw_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                      C.c_int32, C.c_void_p]),
    "me_yolo_decode_f32": (C.c_int, [C.POINTER(YoloDesc), C.c_void_p]),
    "me_nms_workspace_bytes": (C.c_int64, [C.c_int32, C.c_int32]),
    "me_nms_batched_f32": (C.c_int, [C.POINTER(NmsDesc), C.c_void_p]),
    "me_nms_boxes_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p]),
    "me_gather_class_boxes_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                            C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "me_roi_heads_f32": (C.c_int, [C.POINTER(HeadsDesc), C.c_void_p]),
    "me_linear_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                C.c_void_p, C.c_int64, C.c_void_p]),
    "me_mask_scale_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_int64, C.c_void_p, C.c_void_p]),
    "me_dropout_mask_u8": (C.c_int, [C.c_uint64, C.c_float, C.c_int64, C.c_void_p, C.c_void_p]),
    "me_m2_pairs_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "me_m2_rows_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_void_p]),
    "me_m2_loss_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "me_m2_heads_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p,
                                  C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(HeadsWeights), C.c_float,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p]),
    "me_heads_tail_f32": (C.c_int, [C.POINTER(HeadsDesc), C.c_void_p, C.c_int32, C.c_void_p]),
    "me_iou_labels_f32": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "me_heads_loss_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_float,
                                    C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "me_heads_tail_bwd_f32": (C.c_int, [C.POINTER(HeadsDesc)] + [C.c_void_p] * 5 + [C.c_int32] + [C.c_void_p] * 9),
    "me_heads_tail_bwd_dev_f32": (C.c_int, [C.POINTER(HeadsDesc)] + [C.c_void_p] * 5 + [C.c_int32] + [C.c_void_p] * 10),
    "me_bn_train_bwd_dev_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p]),
    "me_roi_align_bwd_dev_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                           C.c_int32, C.c_float, C.c_void_p, C.c_int64, C.c_void_p]),
    "me_ps_roi_align_bwd_dev_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                              C.c_int32, C.c_float, C.c_void_p, C.c_int64, C.c_void_p]),
    "me_gemm_f32": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_int64,
                              C.c_void_p, C.c_int64, C.c_float, C.c_void_p, C.c_int64, C.c_void_p]),
    "me_colsum_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "me_bn_workspace_bytes": (C.c_int64, [C.c_int32]),
    "me_bn_train_fwd_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_float,
                                      C.c_float, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "me_bn_train_bwd_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p]),
    "me_affine_bwd_workspace_bytes": (C.c_int64, [C.c_int32, C.c_int32]),
    "me_affine_act_bwd_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p]),
    "me_affine_bwd_sums_f32": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "me_affine_bwd_h16_workspace_bytes": (C.c_int64, [C.c_int32, C.c_int32]),
    "me_affine_bwd_h16_sums": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "me_conv_wgrad_h16": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p] + [C.c_int32] * 8
                          + [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p]),
    "me_affine_act_bwd_h16": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                        C.c_void_p]),
    "me_upsample2_bwd_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32,
                                       C.c_int32, C.c_void_p]),
    "me_maxpool_bwd_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int32,
                                     C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "me_yolo_loss_bwd_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32] + [C.c_void_p] * 8
                             + [C.c_float] * 5 + [C.c_void_p, C.c_int64, C.c_void_p]),
    "me_yolo_loss_bwd_dev_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32] + [C.c_void_p] * 9
                                 + [C.c_float] * 2 + [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "me_act_bwd_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64,
                                 C.c_int32, C.c_int32, C.c_void_p]),
    "me_conv_wgrad_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_int32,
                                    C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "me_conv_wgrad_workspace_bytes": (C.c_int64, [C.c_int32] * 6),
    "me_conv_wgrad_mfma_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p] + [C.c_int32] * 8
                               + [C.c_void_p, C.c_int64, C.c_void_p]),
    "me_yolo_decode_cand_f32": (C.c_int, [C.c_void_p, C.c_float, C.c_void_p, C.c_int32, C.c_void_p]),
    "me_yolo_decode_cand_multi_f32": (C.c_int, [C.c_void_p, C.c_int32, C.c_float, C.c_void_p, C.c_int32, C.c_void_p]),
    "me_nms_batched_prepped_f32": (C.c_int, [C.c_void_p, C.c_void_p]),
    "me_conv_wgrad_mfma_oihw_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p] + [C.c_int32] * 8
                                    + [C.c_void_p, C.c_int64, C.c_void_p]),
    "me_pack_conv_f32": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32] + [C.c_void_p] * 5 + [C.c_float]
                         + [C.c_void_p] * 7),
    "me_roi_align_bwd_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                       C.c_int32, C.c_float, C.c_void_p, C.c_int64, C.c_void_p]),
    "me_yolo_loss_workspace_bytes": (C.c_int64, []),
    "me_yolo_loss_fwd_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_float),
                                       C.c_void_p, C.c_int32, C.c_float, C.c_float, C.c_float] + [C.c_void_p] * 13),
    "me_yolo_loss_fwd_counted_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                               C.POINTER(C.c_float), C.c_void_p, C.c_int32, C.c_void_p, C.c_float, C.c_float,
                                               C.c_float] + [C.c_void_p] * 13),
    "me_pack_conv_plan": (C.c_int64, [C.POINTER(PackDesc), C.c_int32]),
    "me_pack_conv_batch_f32": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_void_p]),
    "me_ps_roi_align_bwd_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                          C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_int64, C.c_void_p]),
    "me_roi_align_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                   C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_void_p]),
    "me_ps_roi_align_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                      C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_void_p]),
}

_lib = None


def load(path=None):
    """dlopen the library, bind every declared symbol, verify the ABI.  Raises ``MeError``
    (never falls back) when the library is missing or does not match this binding."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise MeError(
            f"{path} not found: the HIP library is not built. Run `python -c 'import __graft_entry__ as g; "
            f"g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback for the product path.")
    try:
        lib_ = C.CDLL(path)
    except OSError as exc:  # e.g. no ROCm runtime
        raise MeError(f"cannot load {path}: {exc}") from exc
    for name, (restype, argtypes) in SIGNATURES.items():
        try:
            fn = getattr(lib_, name)
        except AttributeError as exc:
            raise MeError(f"{path} does not export {name}") from exc
        fn.restype = restype
        fn.argtypes = argtypes
    if lib_.me_abi_version() != 13:
        raise MeError(f"ABI version mismatch: library {lib_.me_abi_version()}, binding 13")
    for which, struct in _STRUCTS.items():
        if lib_.me_sizeof(which) != C.sizeof(struct):
            raise MeError(f"struct layout mismatch for {struct.__name__}: C {lib_.me_sizeof(which)} vs "
                          f"ctypes {C.sizeof(struct)}")
    _lib = lib_
    return _lib


def lib():
    return load()


def available():
    """True when the library is built AND a GPU is visible."""
    return os.path.exists(LIB_PATH) and torch.cuda.is_available()


def default_device():
    if not torch.cuda.is_available():
        raise MeError("no HIP device visible: millieye_amd runs its hot path on MI355X only "
                      "(the CPU restatement lives in oracle/ and is test infrastructure)")
    return torch.device("cuda", torch.cuda.current_device())


def check(rc, what):
    if rc != 0:
        msg = lib().me_last_error().decode("utf-8", "replace")
        raise MeError(f"{what} failed (rc={rc}): {msg}")


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr():
    """The current torch stream of the current device as a ``hipStream_t`` for the C ABI.  ``torch.cuda.current_stream()`` builds
    a Stream object (~6 us; a forward asks ~20 times), the raw query is a C call."""
    if _RAW_STREAM is not None:
        return C.c_void_p(_RAW_STREAM(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _require_cuda_f32(t, name):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32):
        raise MeError(f"{name} must be a CUDA float32 tensor (got {type(t).__name__}"
                      f"{'' if not isinstance(t, torch.Tensor) else f' {t.device} {t.dtype}'})")


# --------------------------------------------------------------------------------------
# tensor-level wrappers (stand-alone use + tests); the detector engine builds descriptors itself
# --------------------------------------------------------------------------------------
def pack_conv_weight(weight):
    """OIHW -> [cout][ky][kx][cin] contiguous fp32 (the layout me_conv2d_f32 reads)."""
    return weight.detach().to(torch.float32).permute(0, 2, 3, 1).contiguous()


def _nhwc_pitch(t, name):
    """Elements between consecutive pixels of an NHWC tensor that is contiguous or a channel slice of a contiguous one."""
    if t.is_contiguous():
        return t.shape[-1]
    n, h, w, _c = t.shape
    pitch = t.stride(2)
    if t.stride() != (h * w * pitch, w * pitch, pitch, 1):
        raise MeError(f"{name} must be NHWC-contiguous or a channel slice of an NHWC-contiguous tensor")
    return pitch


def conv2d(x_nhwc, wgt_packed, scale, shift, ksize, stride, pad, act, residual=None, upsample=1, out=None,
           x_nchw=False, tile=0, split_k=0, wgt_tiled=None, in_launch_reduce=False, tap_masks=None):
    """x_nhwc: [N,H,W,Cin] contiguous (or NCHW [N,Cin,H,W] when ``x_nchw``).  Returns NHWC
    [N,Ho*up,Wo*up,Cout].  ``in_launch_reduce=True`` hands the library arrival counters: split-K slabs are then summed by the last
    workgroup of each tile instead of a second launch (same bits; measured slower on MI355X - DESIGN.md section 5 - so off).
    ``tap_masks`` = ``(cols, (m0, m1, ...))``: the output channels are classes of ``cols`` consecutive channels whose filters are
    all zero on the taps with a clear bit in ``m_class`` - those taps are skipped (``me_conv_desc.tap_mask``, ABI 10)."""
    _require_cuda_f32(x_nhwc, "x")
    if x_nchw:
        n, cin, h, w = x_nhwc.shape
    else:
        n, h, w, cin = x_nhwc.shape
    cout = wgt_packed.shape[0]
    ho = (h + 2 * pad - ksize) // stride + 1
    wo = (w + 2 * pad - ksize) // stride + 1
    if out is None:
        out = torch.empty((n, ho * upsample, wo * upsample, cout), device=x_nhwc.device, dtype=torch.float32)
    d = ConvDesc()
    d.x, d.wgt, d.scale, d.shift = x_nhwc.data_ptr(), wgt_packed.data_ptr(), scale.data_ptr(), shift.data_ptr()
    d.res = residual.data_ptr() if residual is not None else None
    d.y = out.data_ptr()
    d.x_pitch = cin
    if not x_nchw and not x_nhwc.is_contiguous():
        # a channel slice of a wider NHWC buffer (what a darknet [route] concat hands to the next conv)
        pitch = x_nhwc.stride(2)
        if x_nhwc.stride() != (h * w * pitch, w * pitch, pitch, 1):
            raise MeError("x must be NHWC-contiguous or a channel slice of an NHWC-contiguous tensor")
        d.x_pitch = pitch
    d.res_pitch = _nhwc_pitch(residual, "residual") if residual is not None else 0
    d.y_pitch = _nhwc_pitch(out, "out")
    d.n, d.h, d.w, d.cin, d.cout = n, h, w, cin, cout
    d.ksize, d.stride, d.pad, d.ho, d.wo = ksize, stride, pad, ho, wo
    d.act, d.upsample, d.x_nchw, d.tile, d.split_k = act, upsample, 1 if x_nchw else 0, tile, split_k
    if tile >= 100 and wgt_tiled is None and wgt_packed.shape[3] % 16 == 0:
        wgt_tiled = tile_weights_f32(wgt_packed)  # callers that care about time pass their own copy
    d.wgt_tiled = wgt_tiled.data_ptr() if wgt_tiled is not None else None
    if tap_masks is not None:
        d.tap_mask_cols = int(tap_masks[0])
        for i, m in enumerate(tap_masks[1]):
            d.tap_mask[i] = int(m)
        d.split_k = 1
    need = lib().me_conv2d_workspace_bytes(C.byref(d)) if tap_masks is None else 0
    keep = None
    if need > 0:
        ws_ptr, keep = _workspace(need, x_nhwc.device, slot="conv")
        d.workspace, d.workspace_bytes = ws_ptr, need
        if in_launch_reduce:
            d.tile_counters, d.tile_counters_len, _keep2 = tile_counters(x_nhwc.device)
    check(lib().me_conv2d_f32(C.byref(d), stream_ptr()), "me_conv2d_f32")
    return out


_CONV_AUTO = {}
_CONV_AUTO_LOADED = [False]


def _conv_auto_file():
    base = os.environ.get("MILLIEYE_TUNE_CACHE")
    if base:
        return base + ".train"
    root = os.environ.get("XDG_CACHE_HOME", os.path.join(os.path.expanduser("~"), ".cache"))
    # the choices belong to one kernel generation (ABI version) on one GPU model: neither may leak into another
    try:
        gpu = torch.cuda.get_device_name(torch.cuda.current_device()).replace(" ", "_").replace("/", "_")
    except Exception:
        gpu = "unknown"
    return os.path.join(root, "millieye_amd", f"conv_auto_v2_abi{int(lib().me_abi_version())}_{gpu}.json")


def _conv_auto_load():
    if _CONV_AUTO_LOADED[0]:
        return
    _CONV_AUTO_LOADED[0] = True
    try:
        import json
        with open(_conv_auto_file()) as fh:
            for k, v in json.load(fh).items():
                _CONV_AUTO[k] = tuple(int(x) for x in v)   # (tile, split_k[, masked])
    except (OSError, ValueError):
        pass


def _conv_auto_save():
    try:
        import json
        path = _conv_auto_file()
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        tmp = f"{path}.{os.getpid()}.tmp"
        with open(tmp, "w") as fh:
            json.dump({k: list(v) for k, v in _CONV_AUTO.items()}, fh)
        os.replace(tmp, path)
    except OSError:
        pass  # an optimisation only


def conv2d_auto(x_nhwc, wgt_packed, scale, shift, ksize, stride, pad, act, residual=None, out=None, x_nchw=False,
                wgt_tiled=None, tap_masks=None):
    """:func:`conv2d` with the (tile, split_k) pair measured on this GPU the first time a layer shape is seen (the training
    path's convolutions - forward and data gradient - have no engine plan whose autotuner would do it; the library's cold-start
    guess took 128 x 64 tiles for every data gradient: 7.5 ms of a 42 ms Darknet-53 step where the tuned forward needs 5).
    The candidates are timed into a scratch output, so an ``out`` that aliases ``residual`` (gradient accumulation in place)
    is only written once, by the final call."""
    if x_nchw or wgt_packed.shape[3] <= 4:
        return conv2d(x_nhwc, wgt_packed, scale, shift, ksize, stride, pad, act, residual=residual, out=out, x_nchw=x_nchw,
                      wgt_tiled=wgt_tiled)
    _conv_auto_load()
    key = repr((tuple(x_nhwc.shape), x_nhwc.stride(2), wgt_packed.shape[0], ksize, stride, pad, residual is not None,
                wgt_tiled is not None) + ((tap_masks,) if tap_masks is not None else ()))
    hit = _CONV_AUTO.get(key)
    if hit is None:
        cin = wgt_packed.shape[3]
        cands = [(0, 0, 0)] + [(t, sp, 0) for t in (1, 2, 3, 4, 5) for sp in (1, 2, 4)]
        if cin % 16 == 0:
            cands += [(t, sp, 0) for t in (41, 42, 43) for sp in (2, 4)]
        if tap_masks is not None:
            # the masks are an optimisation (the skipped taps are zero either way): whole tiles of the buffer kernel WITH them
            # compete against every unmasked candidate - on the small maps one round of tiles lasts as long as its four-tap
            # class, and the unmasked K splits / tail splits win (tools/dgrad_bench.py, profiles/r05_micro_dgrad_masks.txt)
            cands += [(0, 1, 1)] + [(t, 1, 1) for t in (1, 2, 3, 4, 5)]
        scratch = None
        best = (float("inf"), 0, 0, 0)
        torch.cuda.synchronize()  # every stream idle (the weight-gradient stream too): the timings are the candidates' own
        for tile, split, masked in cands:
            tm = tap_masks if masked else None
            try:
                for _ in range(2):
                    scratch = conv2d(x_nhwc, wgt_packed, scale, shift, ksize, stride, pad, act, residual=residual, out=scratch,
                                     tile=tile, split_k=split, wgt_tiled=wgt_tiled, tap_masks=tm)
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(3):
                    conv2d(x_nhwc, wgt_packed, scale, shift, ksize, stride, pad, act, residual=residual, out=scratch, tile=tile,
                           split_k=split, wgt_tiled=wgt_tiled, tap_masks=tm)
                b.record()
                torch.cuda.synchronize()
                ms = a.elapsed_time(b) / 3
            except MeError:
                continue
            if ms < best[0]:
                best = (ms, tile, split, masked)
        hit = _CONV_AUTO[key] = (best[1], best[2], best[3])
        _conv_auto_save()
    masked = len(hit) > 2 and hit[2]
    try:
        return conv2d(x_nhwc, wgt_packed, scale, shift, ksize, stride, pad, act, residual=residual, out=out, tile=hit[0],
                      split_k=hit[1], wgt_tiled=wgt_tiled, tap_masks=tap_masks if masked else None)
    except MeError:
        if hit[0] == 0 and not masked:
            raise
        _CONV_AUTO[key] = (0, 0, 0)  # a stale entry (a tile this library refuses for the shape): the planner's own choice
        return conv2d(x_nhwc, wgt_packed, scale, shift, ksize, stride, pad, act, residual=residual, out=out, wgt_tiled=wgt_tiled)


def tile_weights_f32(wgt_packed):
    """Second packing of fp32 OHWI weights for the patch-resident kernels (tile ids >= 100): ``[k*k][cin/16][cout][16]`` -
    every (tap, 16-channel chunk) slab of cout rows x 64 bytes contiguous (``me_conv_desc.wgt_tiled``)."""
    cout, kh, kw, cin = wgt_packed.shape
    if cin % 16:
        raise MeError("tile_weights_f32: cin must be a multiple of 16")
    return wgt_packed.reshape(cout, kh * kw, cin // 16, 16).permute(1, 2, 0, 3).contiguous()


def tile_weights_h16(wgt_packed):
    """Second packing of 16-bit OHWI weights for the patch-resident kernels (tile ids >= 100): ``[k*k][cin/32][cout][32]``
    - every (tap, 32-channel chunk) slab of cout rows x 64 bytes contiguous (``me_conv16_desc.wgt_tiled``)."""
    cout, kh, kw, cin = wgt_packed.shape
    if cin % 32:
        raise MeError("tile_weights_h16: cin must be a multiple of 32")
    return wgt_packed.reshape(cout, kh * kw, cin // 32, 32).permute(1, 2, 0, 3).contiguous()


def conv2d_h16(x, wgt_packed, scale, shift, ksize, stride, pad, act, residual=None, upsample=1, out=None,
                x_nchw=False, y_f32=False, tile=0, split_k=0, half=torch.bfloat16, wgt_tiled=None, debug_ws=None, tap_masks=None):
    """16-bit-storage twin of :func:`conv2d`; ``half`` = ``torch.bfloat16`` (default) or ``torch.float16``.  ``x``: 16-bit
    NHWC [N,H,W,Cin] (or a channel slice), or - stem, Cin == 3 - float32 NCHW / NHWC; ``wgt_packed``: 16-bit [Cout,k,k,Cin]
    (float32 for the stem); ``residual``: the output's dtype.  Returns 16-bit NHWC (float32 when ``y_f32``).
    ``tap_masks`` = ``(cols, (m0, m1, ...))`` as in :func:`conv2d` (``me_conv16_desc.tap_mask``, ABI 13): per-tap tiles 1 / 2 / 3 / 11 /
    12 / 13, no K split."""
    if not (isinstance(x, torch.Tensor) and x.is_cuda):
        raise MeError("x must be a CUDA tensor")
    if x_nchw:
        n, cin, h, w = x.shape
    else:
        n, h, w, cin = x.shape
    if cin > 4:
        half = x.dtype
    if half not in HALF_TYPES:
        raise MeError(f"me_conv2d_h16: 16-bit type must be bfloat16 or float16 (got {half})")
    want = torch.float32 if cin <= 4 else half
    if x.dtype != want or wgt_packed.dtype != want:
        raise MeError(f"me_conv2d_h16: x / wgt must be {want} for cin={cin} (got {x.dtype} / {wgt_packed.dtype})")
    cout = wgt_packed.shape[0]
    ho = (h + 2 * pad - ksize) // stride + 1
    wo = (w + 2 * pad - ksize) // stride + 1
    odt = torch.float32 if y_f32 else half
    if out is None:
        out = torch.empty((n, ho * upsample, wo * upsample, cout), device=x.device, dtype=odt)
    if out.dtype != odt or (residual is not None and residual.dtype != odt):
        raise MeError("me_conv2d_h16: out / residual dtype does not match y_f32")
    d = Conv16Desc()
    d.x, d.wgt, d.scale, d.shift = x.data_ptr(), wgt_packed.data_ptr(), scale.data_ptr(), shift.data_ptr()
    d.res = residual.data_ptr() if residual is not None else None
    d.y = out.data_ptr()
    d.x_pitch = cin
    if not x_nchw and not x.is_contiguous():
        pitch = x.stride(2)
        if x.stride() != (h * w * pitch, w * pitch, pitch, 1):
            raise MeError("x must be NHWC-contiguous or a channel slice of an NHWC-contiguous tensor")
        d.x_pitch = pitch
    d.res_pitch = _nhwc_pitch(residual, "residual") if residual is not None else 0
    d.y_pitch = _nhwc_pitch(out, "out")
    d.n, d.h, d.w, d.cin, d.cout = n, h, w, cin, cout
    d.ksize, d.stride, d.pad, d.ho, d.wo = ksize, stride, pad, ho, wo
    d.act, d.upsample, d.x_nchw, d.y_f32, d.tile, d.split_k = act, upsample, 1 if x_nchw else 0, 1 if y_f32 else 0, \
        tile, split_k
    d.half_type = HALF_TYPES[half]
    if tap_masks is not None:
        d.tap_mask_cols = int(tap_masks[0])
        for i, m in enumerate(tap_masks[1]):
            d.tap_mask[i] = int(m)
    if tile >= 100 and wgt_tiled is None and cin % 32 == 0:
        wgt_tiled = tile_weights_h16(wgt_packed)  # callers that care about time pass their own copy
    d.wgt_tiled = wgt_tiled.data_ptr() if wgt_tiled is not None else None
    need = lib().me_conv2d_h16_workspace_bytes(C.byref(d)) if tap_masks is None else 0
    keep = None
    if need > 0:
        ws_ptr, keep = _workspace(need, x.device, slot="conv")
        d.workspace, d.workspace_bytes = ws_ptr, need
    if debug_ws is not None:  # instrumented tile ids (tools/p8_timeline.py): the kernel writes its time stamps here
        d.workspace, d.workspace_bytes = debug_ws.data_ptr(), debug_ws.numel() * debug_ws.element_size()
    check(lib().me_conv2d_h16(C.byref(d), stream_ptr()), "me_conv2d_h16")
    return out


HALF_TYPES = {torch.bfloat16: 0, torch.float16: 1}  # me_conv16_desc.half_type


BNECK_TILES = (1, 3, 4)  # me_bneck16_desc.tile


def bneck_h16(x, w1_packed, scale1, shift1, w2_packed, scale2, shift2, residual=None, out=None, tile=1, act1=1, act2=1,
              w1_tiled=None, w2_tiled=None):
    """One launch for ``conv1x1 -> conv3x3 (+ residual)`` in a 16-bit storage mode (``me_bneck_h16``): ``x`` 16-bit NHWC
    [N,H,W,Cin] (or a channel slice), ``w1_packed`` [Cmid,1,1,Cin], ``w2_packed`` [Cout,3,3,Cmid] (16-bit OHWI), scales /
    shifts fp32.  Equals ``conv2d_h16(conv2d_h16(x, w1...), w2..., residual=residual)`` without the mid tensor in HBM."""
    if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype in HALF_TYPES):
        raise MeError("me_bneck_h16: x must be a 16-bit CUDA tensor")
    n, h, w, cin = x.shape
    cmid, cout = w1_packed.shape[0], w2_packed.shape[0]
    if w1_packed.dtype != x.dtype or w2_packed.dtype != x.dtype or tuple(w1_packed.shape[1:]) != (1, 1, cin) \
            or tuple(w2_packed.shape[1:]) != (3, 3, cmid):
        raise MeError("me_bneck_h16: weights must be 16-bit OHWI [cmid,1,1,cin] and [cout,3,3,cmid] of x's type")
    if out is None:
        out = torch.empty((n, h, w, cout), device=x.device, dtype=x.dtype)
    if out.dtype != x.dtype or (residual is not None and residual.dtype != x.dtype):
        raise MeError("me_bneck_h16: out / residual must have x's type")
    w1_tiled = tile_weights_h16(w1_packed) if w1_tiled is None else w1_tiled
    w2_tiled = tile_weights_h16(w2_packed) if w2_tiled is None else w2_tiled
    d = Bneck16Desc()
    d.x, d.x_pitch = x.data_ptr(), _nhwc_pitch(x, "x")
    d.w1_tiled, d.scale1, d.shift1 = w1_tiled.data_ptr(), scale1.data_ptr(), shift1.data_ptr()
    d.w2_tiled, d.scale2, d.shift2 = w2_tiled.data_ptr(), scale2.data_ptr(), shift2.data_ptr()
    d.res = residual.data_ptr() if residual is not None else None
    d.res_pitch = _nhwc_pitch(residual, "residual") if residual is not None else 0
    d.y, d.y_pitch = out.data_ptr(), _nhwc_pitch(out, "out")
    d.n, d.h, d.w, d.cin, d.cmid, d.cout = n, h, w, cin, cmid, cout
    d.act1, d.act2, d.half_type, d.tile = act1, act2, HALF_TYPES[x.dtype], tile
    keep = (w1_tiled, w2_tiled)  # alive across the asynchronous launch
    check(lib().me_bneck_h16(C.byref(d), stream_ptr()), "me_bneck_h16")
    del keep
    return out


def _require_cuda_bf16(t, name):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype in HALF_TYPES and t.is_contiguous()):
        raise MeError(f"{name} must be a contiguous CUDA bfloat16 / float16 tensor")


def maxpool_h16(x_nhwc, size, stride, zero_ext=False):
    _require_cuda_bf16(x_nhwc, "x")
    n, h, w, c = x_nhwc.shape
    pad = 0 if zero_ext else (size - 1) // 2
    ext = 1 if zero_ext else 0
    ho = (h + ext + 2 * pad - size) // stride + 1
    wo = (w + ext + 2 * pad - size) // stride + 1
    out = torch.empty((n, ho, wo, c), device=x_nhwc.device, dtype=x_nhwc.dtype)
    check(lib().me_maxpool_h16(x_nhwc.data_ptr(), c, out.data_ptr(), c, n, h, w, c, size, stride, pad, ext, ho, wo,
                               HALF_TYPES[x_nhwc.dtype], stream_ptr()), "me_maxpool_h16")
    return out


def upsample_h16(x_nhwc, factor):
    _require_cuda_bf16(x_nhwc, "x")
    n, h, w, c = x_nhwc.shape
    out = torch.empty((n, h * factor, w * factor, c), device=x_nhwc.device, dtype=x_nhwc.dtype)
    check(lib().me_upsample_h16(x_nhwc.data_ptr(), c, out.data_ptr(), c, n, h, w, c, factor, stream_ptr()),
          "me_upsample_h16")
    return out


def add_h16(a, b):
    _require_cuda_bf16(a, "a")
    _require_cuda_bf16(b, "b")
    c = a.shape[-1]
    out = torch.empty_like(a)
    check(lib().me_add_h16(a.data_ptr(), c, b.data_ptr(), c, out.data_ptr(), c, a.numel() // c, c, HALF_TYPES[a.dtype],
                           stream_ptr()),
          "me_add_h16")
    return out


def maxpool(x_nhwc, size, stride, zero_ext=False):
    _require_cuda_f32(x_nhwc, "x")
    n, h, w, c = x_nhwc.shape
    pad = 0 if zero_ext else (size - 1) // 2
    ext = 1 if zero_ext else 0
    ho = (h + ext + 2 * pad - size) // stride + 1
    wo = (w + ext + 2 * pad - size) // stride + 1
    out = torch.empty((n, ho, wo, c), device=x_nhwc.device, dtype=torch.float32)
    d = PoolDesc()
    d.x, d.y, d.x_pitch, d.y_pitch = x_nhwc.data_ptr(), out.data_ptr(), c, c
    d.n, d.h, d.w, d.c = n, h, w, c
    d.size, d.stride, d.pad, d.zero_ext, d.ho, d.wo = size, stride, pad, ext, ho, wo
    check(lib().me_maxpool_f32(C.byref(d), stream_ptr()), "me_maxpool_f32")
    return out


def upsample(x_nhwc, factor):
    _require_cuda_f32(x_nhwc, "x")
    n, h, w, c = x_nhwc.shape
    out = torch.empty((n, h * factor, w * factor, c), device=x_nhwc.device, dtype=torch.float32)
    check(lib().me_upsample_f32(x_nhwc.data_ptr(), c, out.data_ptr(), c, n, h, w, c, factor, stream_ptr()),
          "me_upsample_f32")
    return out


def nhwc_to_nchw(x_nhwc):
    _require_cuda_f32(x_nhwc, "x")
    n, h, w, c = x_nhwc.shape
    out = torch.empty((n, c, h, w), device=x_nhwc.device, dtype=torch.float32)
    check(lib().me_nhwc_to_nchw_f32(x_nhwc.data_ptr(), c, out.data_ptr(), n, h, w, c, stream_ptr()),
          "me_nhwc_to_nchw_f32")
    return out


def yolo_decode(x_nhwc, anchors, num_classes, img_dim, out=None, rows_total=None, row_offset=0):
    """x_nhwc [N,G,G,A*(5+C)] -> rows of out [N,rows_total,5+C] (reference YOLOLayer decode)."""
    _require_cuda_f32(x_nhwc, "x")
    n, g, _, ch = x_nhwc.shape
    na = len(anchors)
    if rows_total is None:
        rows_total = na * g * g
    if out is None:
        out = torch.empty((n, rows_total, 5 + num_classes), device=x_nhwc.device, dtype=torch.float32)
    d = YoloDesc()
    d.x, d.out, d.x_pitch = x_nhwc.data_ptr(), out.data_ptr(), ch
    d.n, d.g, d.num_anchors, d.num_classes = n, g, na, num_classes
    d.rows_total, d.row_offset = rows_total, row_offset
    stride = img_dim / g
    d.stride = stride
    for k, (aw, ah) in enumerate(anchors):
        d.anchors[2 * k] = aw / stride
        d.anchors[2 * k + 1] = ah / stride
    check(lib().me_yolo_decode_f32(C.byref(d), stream_ptr()), "me_yolo_decode_f32")
    return out


_ws_cache = {}


_counter_cache = {}
TILE_COUNTERS = 1 << 16


def tile_counters(device, slot="conv"):
    """``(pointer, length, keep-alive)`` of a zeroed int32 array for ``me_conv_desc.tile_counters`` (arrival counters of the
    in-launch split-K reduction; the library leaves them zero).  One array per (slot, device, stream): launches that may run
    concurrently must not share one."""
    key = (slot, device.index if device.index is not None else torch.cuda.current_device(),
           torch.cuda.current_stream().cuda_stream)
    t = _counter_cache.get(key)
    if t is None:
        t = _counter_cache[key] = torch.zeros(TILE_COUNTERS, dtype=torch.int32, device=device)
    return t.data_ptr(), TILE_COUNTERS, t


def _workspace(nbytes, device, slot="nms"):
    """Cached scratch per (slot, device, STREAM): kernels on different streams never share a buffer (the weight gradients of
    the detector step run on a side stream while the main stream's convolutions use theirs), and a buffer that is replaced
    by a bigger one is handed back to the caching allocator only behind the work already queued on its stream."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    cur = torch.cuda.current_stream(idx)
    key = (slot, idx, cur.cuda_stream)
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() - ((-ws.data_ptr()) % 256) < nbytes:  # usable bytes behind the aligned start
        if ws is not None:
            ws.record_stream(cur)  # (allocated under this stream already; explicit for buffers adopted from older versions)
        with torch.cuda.device(idx):
            ws = torch.empty(int(nbytes) + 256, dtype=torch.uint8, device=device)
        _ws_cache[key] = ws
    off = (-ws.data_ptr()) % 256
    return ws.data_ptr() + off, ws


def nms_workspace(n, rows, device):
    """The cached NMS workspace for ``[n, rows]`` predictions (the buffer :func:`nms_batched` uses): ``(ptr, keepalive)``."""
    return _workspace(lib().me_nms_workspace_bytes(n, rows), device)


def nms_batched(pred, conf_thresh, iou_thresh, max_det, writeback_xyxy=True, prepped=False):
    """pred [N,R,5+C] (modified in place when ``writeback_xyxy``) -> (det [N,max_det,7+C],
    count int32 [N]) on the device; rows >= count are unspecified.  ``prepped``: the candidate lists are already in the
    workspace (the engine decoded with ``me_yolo_decode_cand_f32`` at this ``conf_thresh``): selection + emit only."""
    _require_cuda_f32(pred, "prediction")
    if not pred.is_contiguous():
        raise MeError("prediction must be contiguous")
    n, rows, per = pred.shape
    nc = per - 5
    det = torch.empty((n, max_det, 7 + nc), device=pred.device, dtype=torch.float32)
    count = torch.empty((n,), device=pred.device, dtype=torch.int32)
    if n == 0 or rows == 0:
        return det, count.zero_()
    nbytes = lib().me_nms_workspace_bytes(n, rows)
    ws_ptr, _keep = _workspace(nbytes, pred.device)
    d = NmsDesc()
    d.pred, d.det, d.count, d.workspace = pred.data_ptr(), det.data_ptr(), count.data_ptr(), ws_ptr
    d.n, d.rows, d.num_classes, d.max_det = n, rows, nc, max_det
    d.conf_thresh, d.iou_thresh, d.writeback_xyxy = conf_thresh, iou_thresh, 1 if writeback_xyxy else 0
    if prepped:
        check(lib().me_nms_batched_prepped_f32(C.byref(d), stream_ptr()), "me_nms_batched_prepped_f32")
    else:
        check(lib().me_nms_batched_f32(C.byref(d), stream_ptr()), "me_nms_batched_f32")
    return det, count


def nms_indices(boxes, scores, idxs, iou_threshold):
    """torchvision-style ``nms`` / ``batched_nms``: int64 kept indices, descending score."""
    src = boxes.device
    dev = boxes if boxes.is_cuda else boxes.to(default_device())
    m = dev.shape[0]
    if m == 0:
        return torch.empty((0,), dtype=torch.int64, device=src)
    b = dev.to(torch.float32).contiguous()
    s = scores.to(b.device, torch.float32).contiguous()
    lab = idxs.to(b.device, torch.float32).contiguous() if idxs is not None else None
    keep = torch.empty((m,), dtype=torch.int64, device=b.device)
    cnt = torch.empty((1,), dtype=torch.int32, device=b.device)
    ws_ptr, _keep = _workspace(lib().me_nms_workspace_bytes(1, m), b.device)
    check(lib().me_nms_boxes_f32(b.data_ptr(), s.data_ptr(), lab.data_ptr() if lab is not None else None, m,
                                 float(iou_threshold), keep.data_ptr(), cnt.data_ptr(), ws_ptr, stream_ptr()),
          "me_nms_boxes_f32")
    return keep[: int(cnt.item())].to(src)


def _roi(fn_name, map_nhwc, rois, pooled, spatial_scale, ps):
    _require_cuda_f32(map_nhwc, "map")
    n, h, w, c = map_nhwc.shape
    r = rois.to(map_nhwc.device, torch.float32).contiguous()
    k = r.shape[0]
    cout = c // (pooled * pooled) if ps else c
    out = torch.empty((k, cout, pooled, pooled), device=map_nhwc.device, dtype=torch.float32)
    fn = getattr(lib(), fn_name)
    check(fn(map_nhwc.data_ptr(), c, n, h, w, c, r.data_ptr() if k else None, k, pooled, float(spatial_scale),
             out.data_ptr(), stream_ptr()), fn_name)
    return out


def roi_align(map_nhwc, rois, pooled=7, spatial_scale=1.0 / 16):
    """torchvision.ops.roi_align(aligned=False, sampling_ratio=-1) on an NHWC map -> [K,C,7,7]."""
    return _roi("me_roi_align_f32", map_nhwc, rois, pooled, spatial_scale, False)


def ps_roi_align(map_nhwc, rois, pooled=7, spatial_scale=1.0 / 16):
    """torchvision.ops.ps_roi_align(sampling_ratio=-1) on an NHWC map -> [K,C/49,7,7]."""
    return _roi("me_ps_roi_align_f32", map_nhwc, rois, pooled, spatial_scale, True)


def conv_wgrad_h16(x_nhwc, dy_nhwc, ksize, stride, pad, oihw=True):
    """float32 dW from 16-bit ``x`` / ``dy`` (``me_conv_wgrad_h16``: 16-bit MFMA, fp32 accumulation, fixed-order slab sums)."""
    _require_cuda_bf16(x_nhwc, "x")
    _require_cuda_bf16(dy_nhwc, "dy")
    n, h, w, cin = x_nhwc.shape
    _, ho, wo, cout = dy_nhwc.shape
    shape = (cout, cin, ksize, ksize) if oihw else (cout, ksize, ksize, cin)
    dw = torch.empty(shape, device=x_nhwc.device, dtype=torch.float32)
    need = max(lib().me_conv_wgrad_workspace_bytes(n, ho, wo, cin, cout, ksize), 4 * cout * cin * ksize * ksize)
    ws_ptr, keep = _workspace(need, x_nhwc.device, slot="wgrad")
    check(lib().me_conv_wgrad_h16(x_nhwc.data_ptr(), cin, dy_nhwc.data_ptr(), cout, dw.data_ptr(), n, h, w, cin, cout, ksize, stride,
                                  pad, ws_ptr, need, 1 if oihw else 0, HALF_TYPES[x_nhwc.dtype], stream_ptr()), "me_conv_wgrad_h16")
    return dw


def conv_wgrad(x_nhwc, dy_nhwc, ksize, stride, pad, oihw=False):
    """dW of y = conv(x, W) given dy, on the matrix pipe (me_conv_wgrad_mfma_f32; slices of the pixel reduction go through a
    workspace and are added in a fixed order): [cout, k, k, cin] (the packed layout of me_conv2d_f32), or - ``oihw`` - the
    parameter's own [cout, cin, k, k] (me_conv_wgrad_mfma_oihw_f32: the slab reduction transposes, no extra launch)."""
    _require_cuda_f32(x_nhwc, "x")
    _require_cuda_f32(dy_nhwc, "dy")
    n, h, w, cin = x_nhwc.shape
    _, ho, wo, cout = dy_nhwc.shape
    shape = (cout, cin, ksize, ksize) if oihw else (cout, ksize, ksize, cin)
    dw = torch.empty(shape, device=x_nhwc.device, dtype=torch.float32)
    need = lib().me_conv_wgrad_workspace_bytes(n, ho, wo, cin, cout, ksize)
    if oihw and ksize > 1:
        need = max(need, 4 * cout * cin * ksize * ksize)
    ws_ptr, keep = (None, None)
    if need > 0:
        ws_ptr, keep = _workspace(need, x_nhwc.device, slot="wgrad")
    fn = lib().me_conv_wgrad_mfma_oihw_f32 if oihw else lib().me_conv_wgrad_mfma_f32
    check(fn(x_nhwc.data_ptr(), x_nhwc.stride(2), dy_nhwc.data_ptr(), dy_nhwc.stride(2), dw.data_ptr(), n, h, w, cin, cout,
             ksize, stride, pad, ws_ptr, need, stream_ptr()), "me_conv_wgrad_mfma_f32")
    return dw
