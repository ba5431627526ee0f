"""Box utilities, post-processing wrappers and detection metrics.

Host-side mirror of ``module3_our_dataset/utils/utils.py`` for the functions on the
hot path (SURVEY.md section 8a rows a8, a18 and the helpers they use).  Same names,
argument meaning and return shapes as the reference, so ``from utils.utils import *``
in the reference scripts keeps working (the star-import re-exports ``torch``, ``np``,
``nn``, ``F``, ``tqdm``, ``box_ops`` ... exactly like the reference module does).

Where the reference dispatches into torchvision's C++ (``box_ops.batched_nms``,
reference ``utils/utils.py:372``) this module dispatches into the HIP library
through :mod:`millieye_amd.hip`; there is no CPU fallback - without the GPU
library the call raises.

The metric functions (``get_batch_statistics``, ``ap_per_class``, ``compute_ap``,
``bbox_iou``) define what "mAP@0.5 equal to the reference" means and therefore stay
host-side float code with the reference's exact operation order.
"""
from __future__ import division

import math  # noqa: F401  (re-exported)
import time  # noqa: F401  (re-exported)

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F  # noqa: F401  (re-exported)
from torch.autograd import Variable  # noqa: F401  (re-exported)

try:  # progress bars are cosmetic
    import tqdm
except Exception:  # pragma: no cover
    tqdm = None
try:  # the reference re-exports plt / patches through its star import
    import matplotlib

    matplotlib.use("Agg")
    import matplotlib.pyplot as plt  # noqa: F401
    import matplotlib.patches as patches  # noqa: F401
except Exception:  # pragma: no cover
    plt = None
    patches = None

from .. import hip as _hip


# --------------------------------------------------------------------------------------
# small helpers (reference utils/utils.py:16-38)
# --------------------------------------------------------------------------------------
def to_cpu(tensor):
    return tensor.detach().cpu()


def load_classes(path):
    """One class name per line (last, empty, split element dropped: reference :21-27)."""
    with open(path, "r") as fh:
        return fh.read().split("\n")[:-1]


def weights_init_normal(m):
    """``model.apply`` hook, reference :29-38: Conv* N(0, .02); BatchNorm2d N(1, .02) / 0;
    Linear kaiming-normal.  Class-name matching is kept so the torch RNG stream is
    consumed in the same order as the reference."""
    name = type(m).__name__
    # the reference writes through ``.data`` (own version counter: the parameter's ``_version`` would stay put and the
    # engine's packed copies would not notice); ``nn.init.*`` on the parameter draws the same numbers and bumps it
    if "Conv" in name:
        torch.nn.init.normal_(m.weight, 0.0, 0.02)
    elif "BatchNorm2d" in name:
        torch.nn.init.normal_(m.weight, 1.0, 0.02)
        torch.nn.init.constant_(m.bias, 0.0)
    elif "Linear" in name:
        nn.init.kaiming_normal_(m.weight)


def rescale_boxes(boxes, current_dim, original_shape):
    """Undo pad-to-square + resize (reference :41-56); mutates and returns ``boxes`` (xyxy in the square frame ->
    xyxy in the original ``(h, w)`` frame).  Per axis: ``((v - pad // 2) / unpadded) * original``."""
    orig = {"y": original_shape[0], "x": original_shape[1]}
    ratio = current_dim / max(original_shape)
    pad = {"x": max(orig["y"] - orig["x"], 0) * ratio, "y": max(orig["x"] - orig["y"], 0) * ratio}
    for col, axis in ((0, "x"), (1, "y"), (2, "x"), (3, "y")):
        unpadded = current_dim - pad[axis]
        boxes[:, col] = ((boxes[:, col] - pad[axis] // 2) / unpadded) * orig[axis]
    return boxes


def _stack_last(parts, like):
    return torch.stack(parts, -1) if isinstance(like, torch.Tensor) else np.stack(parts, -1)


def xyxy2xywh(x):
    """corners -> centre / size, reference :59-66 (torch or numpy input; a new array of the same dtype)."""
    left, top, right, bottom = x[..., 0], x[..., 1], x[..., 2], x[..., 3]
    return _stack_last(((left + right) / 2, (top + bottom) / 2, right - left, bottom - top), x)


def xywh2xyxy(x):
    """centre / size -> corners, reference :68-74 (half extents computed as ``w / 2``)."""
    cx, cy, half_w, half_h = x[..., 0], x[..., 1], x[..., 2] / 2, x[..., 3] / 2
    return _stack_last((cx - half_w, cy - half_h, cx + half_w, cy + half_h), x)


# --------------------------------------------------------------------------------------
# metrics (reference utils/utils.py:77-236, 248-278) - host side by design (row a18)
# --------------------------------------------------------------------------------------
def compute_ap(recall, precision):
    """Area under the precision envelope (py-faster-rcnn style), reference :157-182."""
    mrec = np.concatenate(([0.0], recall, [1.0]))
    mpre = np.concatenate(([0.0], precision, [0.0]))
    for k in range(mpre.size - 1, 0, -1):
        mpre[k - 1] = np.maximum(mpre[k - 1], mpre[k])
    steps = np.where(mrec[1:] != mrec[:-1])[0]
    return np.sum((mrec[steps + 1] - mrec[steps]) * mpre[steps + 1])


def _progress(iterable, desc):
    if tqdm is None:
        return iterable
    return tqdm.tqdm(iterable, desc=desc)


def ap_per_class(tp, conf, pred_cls, target_cls, with_conf=False):
    """Per-class AP + pooled PR curve, reference :77-154.

    Returns ``(p, r, ap, f1, classes_int32, (precision_curve, recall_curve))``; ``with_conf`` = the stage-2 tree's variant
    (module2_mixed/utils/utils.py:219-295), whose curve tuple carries the sorted confidences as a third entry."""
    order = np.argsort(-conf)
    tp, conf, pred_cls = tp[order], conf[order], pred_cls[order]
    unique_classes = np.unique(target_cls)

    ap, p, r = [], [], []
    for c in _progress(unique_classes, "Computing AP"):
        sel = pred_cls == c
        n_p = sel.sum()
        n_gt = (target_cls == c).sum()
        if n_p == 0 and n_gt == 0:
            continue
        if n_p == 0 or n_gt == 0:
            ap.append(0)
            r.append(0)
            p.append(0)
            continue
        fpc = (1 - tp[sel]).cumsum()
        tpc = (tp[sel]).cumsum()
        recall_curve = tpc / (n_gt + 1e-16)
        r.append(recall_curve[-1])
        precision_curve = tpc / (tpc + fpc)
        p.append(precision_curve[-1])
        ap.append(compute_ap(recall_curve, precision_curve))

    p, r, ap = np.array(p), np.array(r), np.array(ap)
    f1 = 2 * p * r / (p + r + 1e-16)

    # pooled curve over every prediction whose class occurs in the targets
    keep = [bool(pred_cls[k] in unique_classes) for k in range(len(tp))]
    tp, pred_cls = tp[keep], pred_cls[keep]
    n_p, n_gt = len(tp), len(target_cls)
    if n_p == 0 or n_gt == 0:
        precision_curve, recall_curve = 0, 0
    else:
        fpc = (1 - tp).cumsum()
        tpc = (tp).cumsum()
        recall_curve = tpc / (n_gt + 1e-16)
        precision_curve = tpc / (tpc + fpc)
    if with_conf:
        return p, r, ap, f1, unique_classes.astype("int32"), (precision_curve, recall_curve, conf)
    return p, r, ap, f1, unique_classes.astype("int32"), (precision_curve, recall_curve)


def bbox_wh_iou(wh1, wh2):
    """IoU of one anchor shape ``wh1`` with target shapes ``wh2[n,2]``, both centred on the same point (reference
    :239-245): ``min(w) * min(h) / ((w1*h1 + 1e-16) + w2*h2 - inter)``."""
    anchor_w, anchor_h = wh1[0], wh1[1]
    tgt_w, tgt_h = wh2[:, 0], wh2[:, 1]
    overlap = torch.min(anchor_w, tgt_w) * torch.min(anchor_h, tgt_h)
    return overlap / ((anchor_w * anchor_h + 1e-16) + tgt_w * tgt_h - overlap)


def _as_corners(box, x1y1x2y2):
    if x1y1x2y2:
        return box[:, 0], box[:, 1], box[:, 2], box[:, 3]
    half_w, half_h = box[:, 2] / 2, box[:, 3] / 2
    return box[:, 0] - half_w, box[:, 1] - half_h, box[:, 0] + half_w, box[:, 1] + half_h


def bbox_iou(box1, box2, x1y1x2y2=True):
    """IoU with the reference's **+1 pixel** convention (reference :248-278): widths / heights are ``hi - lo + 1``
    for the intersection and for both areas; ``box1`` [1,4] or [n,4] broadcasts against ``box2`` [n,4]."""
    l1, t1, r1, b1 = _as_corners(box1, x1y1x2y2)
    l2, t2, r2, b2 = _as_corners(box2, x1y1x2y2)
    span_x = torch.clamp(torch.min(r1, r2) - torch.max(l1, l2) + 1, min=0)
    span_y = torch.clamp(torch.min(b1, b2) - torch.max(t1, t2) + 1, min=0)
    shared = span_x * span_y
    area1 = (r1 - l1 + 1) * (b1 - t1 + 1)
    area2 = (r2 - l2 + 1) * (b2 - t2 + 1)
    return shared / (area1 + area2 - shared + 1e-16)


def get_batch_statistics(outputs, targets, iou_threshold):
    """Greedy TP assignment per image, reference :185-236 (quirks q13: predictions whose
    label is absent from the image's targets are skipped; the loop stops once every GT
    is matched; a GT can be matched once).

    ``outputs``: list (len = batch) of ``[n,7]`` tensors or ``None``;
    ``targets``: ``[m,6]`` = (image_i, class, x1, y1, x2, y2) already in pixels.
    Returns a list of ``[true_positives(np), pred_scores, pred_labels]`` per non-None image."""
    batch_metrics = []
    for image_i in range(len(outputs)):
        if outputs[image_i] is None:
            continue
        output = outputs[image_i]
        pred_boxes = output[:, :4]
        pred_scores = output[:, 4]
        pred_labels = output[:, -1]
        true_positives = np.zeros(pred_boxes.shape[0])

        annotations = targets[targets[:, 0] == image_i][:, 1:]
        target_labels = annotations[:, 0] if len(annotations) else []
        if len(annotations):
            matched = []
            target_boxes = annotations[:, 1:]
            for pred_i, (pred_box, pred_label) in enumerate(zip(pred_boxes, pred_labels)):
                if len(matched) == len(annotations):
                    break
                if pred_label not in target_labels:
                    continue
                iou, box_index = bbox_iou(pred_box.unsqueeze(0), target_boxes).max(0)
                if iou >= iou_threshold and box_index not in matched:
                    true_positives[pred_i] = 1
                    matched += [box_index]
        batch_metrics.append([true_positives, pred_scores, pred_labels])
    return batch_metrics


# --------------------------------------------------------------------------------------
# NMS (reference utils/utils.py:337-378 + torchvision.ops.boxes.batched_nms)
# --------------------------------------------------------------------------------------
class _BoxOps:
    """Stand-in for ``torchvision.ops.boxes`` (the reference's ``box_ops`` re-export,
    used directly by ``run_sp.py:214`` / ``run_mp.py:320``): ``nms`` and ``batched_nms``
    executed by the HIP greedy-NMS kernel."""

    @staticmethod
    def nms(boxes, scores, iou_threshold):
        return _hip.nms_indices(boxes, scores, None, float(iou_threshold))

    @staticmethod
    def batched_nms(boxes, scores, idxs, iou_threshold):
        return _hip.nms_indices(boxes, scores, idxs, float(iou_threshold))


box_ops = _BoxOps()


def non_max_suppression_cpp(prediction, conf_thresh, nms_thresh=0.5, detections_per_img=200):
    """Confidence filter + per-class greedy NMS, at most ``detections_per_img`` per image.

    Mirror of reference ``utils/utils.py:337-378``: ``prediction`` ``[N,R,5+C]`` with
    (cx,cy,w,h,obj,cls...) rows; its first four columns are converted to xyxy **in place**
    (callers pass a clone); ranking is by objectness only (quirk q8).
    Returns a list of ``[n_i, 7+C]`` tensors (x1,y1,x2,y2,obj,cls_conf,cls_pred,C scores)
    on ``prediction.device`` or ``None`` for images without detections.

    The work runs in ``me_nms_*`` HIP kernels; a CPU ``prediction`` is staged to the GPU
    and the results copied back (the reference calls this with ``.cpu()`` tensors)."""
    src_device = prediction.device
    dev_pred = prediction if prediction.is_cuda else prediction.to(_hip.default_device())
    dense, counts = _hip.nms_batched(dev_pred, float(conf_thresh), float(nms_thresh), int(detections_per_img),
                                     writeback_xyxy=True)
    if dev_pred is not prediction:
        prediction[..., :4] = dev_pred[..., :4].to(src_device)
        dense = dense.to(src_device)
    counts_host = counts.tolist()  # the one host sync of this API (list lengths are data dependent)
    out = [None] * len(counts_host)
    for i, n in enumerate(counts_host):
        if n > 0:
            out[i] = dense[i, :n]
    return out


def non_max_suppression(prediction, conf_thresh=0.01, nms_thresh=0.5):
    """Legacy API of the reference (``utils/utils.py:281-334``: +1-pixel IoU, rows
    ``[n,7]``).  Not on the hot path (no m2/m3 script calls it); kept for name
    compatibility and implemented on top of the same device kernel is *not* possible
    (different IoU convention), so it is deliberately unsupported."""
    raise NotImplementedError(
        "non_max_suppression (python NMS with +1 IoU) is off the accelerated path; "
        "use non_max_suppression_cpp like module3_our_dataset/my_models.py:457 does")


# --------------------------------------------------------------------------------------
# YOLO training targets (reference utils/utils.py:381-440) - used by YOLOLayer's loss
# --------------------------------------------------------------------------------------
def build_targets(pred_boxes, pred_cls, target, anchors, ignore_thres):
    """Anchor matching for the YOLO loss (row a6).  Index bookkeeping on
    ``[nB,nA,nG,nG]`` masks; runs wherever the inputs live.

    Returns ``(iou_scores, class_mask, obj_mask, noobj_mask, tx, ty, tw, th, tcls, tconf)``."""
    dev = pred_boxes.device
    n_img, n_anchor, grid = pred_boxes.shape[0], pred_boxes.shape[1], pred_boxes.shape[2]
    cells = (n_img, n_anchor, grid, grid)
    f32 = dict(dtype=torch.float32, device=dev)
    obj_mask = torch.zeros(cells, dtype=torch.uint8, device=dev)
    noobj_mask = torch.ones(cells, dtype=torch.uint8, device=dev)
    class_mask, iou_scores = torch.zeros(cells, **f32), torch.zeros(cells, **f32)
    tx, ty, tw, th = (torch.zeros(cells, **f32) for _ in range(4))
    tcls = torch.zeros(cells + (pred_cls.shape[-1],), **f32)

    # targets in grid units; the anchor whose *shape* fits a target best owns it (position does not matter here)
    target_boxes = target[:, 2:6] * grid
    centre, extent = target_boxes[:, :2], target_boxes[:, 2:]
    shape_iou = torch.stack([bbox_wh_iou(anchor, extent) for anchor in anchors])  # [n_anchor, n_target]
    best_n = shape_iou.max(0)[1]
    image_of, label_of = target[:, 0].long(), target[:, 1].long()
    gi, gj = centre[:, 0].long(), centre[:, 1].long()
    owner = (image_of, best_n, gj, gi)

    obj_mask[owner] = 1
    noobj_mask[owner] = 0
    # other anchors that fit well are neither object nor background (reference: a python loop over the targets, one masked
    # assignment each; zeros are written, so one scatter over every (anchor, target) pair above the threshold is the same)
    a_fit, t_fit = torch.nonzero(shape_iou > ignore_thres, as_tuple=True)
    noobj_mask[image_of[t_fit], a_fit, gj[t_fit], gi[t_fit]] = 0

    tx[owner] = centre[:, 0] - centre[:, 0].floor()
    ty[owner] = centre[:, 1] - centre[:, 1].floor()
    own_anchor = anchors[best_n]
    tw[owner] = torch.log(extent[:, 0] / own_anchor[:, 0] + 1e-16)
    th[owner] = torch.log(extent[:, 1] / own_anchor[:, 1] + 1e-16)
    tcls[owner + (label_of,)] = 1
    class_mask[owner] = (pred_cls[owner].argmax(-1) == label_of).float()
    iou_scores[owner] = bbox_iou(pred_boxes[owner], target_boxes, x1y1x2y2=False)

    tconf = obj_mask.float()
    return iou_scores, class_mask, obj_mask, noobj_mask, tx, ty, tw, th, tcls, tconf
