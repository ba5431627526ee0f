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
    if "Conv" in name:
        torch.nn.init.normal_(m.weight.data, 0.0, 0.02)
    elif "BatchNorm2d" in name:
        torch.nn.init.normal_(m.weight.data, 1.0, 0.02)
        torch.nn.init.constant_(m.bias.data, 0.0)
    elif "Linear" in name:
        nn.init.kaiming_normal_(m.weight.data)


def rescale_boxes(boxes, current_dim, original_shape):
    """Undo pad-to-square + resize (reference :41-56); mutates and returns ``boxes``."""
    orig_h, orig_w = original_shape
    ratio = current_dim / max(original_shape)
    pad_x = max(orig_h - orig_w, 0) * ratio
    pad_y = max(orig_w - orig_h, 0) * ratio
    unpad_h = current_dim - pad_y
    unpad_w = current_dim - pad_x
    boxes[:, 0] = ((boxes[:, 0] - pad_x // 2) / unpad_w) * orig_w
    boxes[:, 1] = ((boxes[:, 1] - pad_y // 2) / unpad_h) * orig_h
    boxes[:, 2] = ((boxes[:, 2] - pad_x // 2) / unpad_w) * orig_w
    boxes[:, 3] = ((boxes[:, 3] - pad_y // 2) / unpad_h) * orig_h
    return boxes


def xyxy2xywh(x):
    """[x1,y1,x2,y2] -> [cx,cy,w,h] (torch or numpy), reference :59-66."""
    y = torch.zeros_like(x) if isinstance(x, torch.Tensor) else np.zeros_like(x)
    y[..., 0] = (x[..., 0] + x[..., 2]) / 2
    y[..., 1] = (x[..., 1] + x[..., 3]) / 2
    y[..., 2] = x[..., 2] - x[..., 0]
    y[..., 3] = x[..., 3] - x[..., 1]
    return y


def xywh2xyxy(x):
    """[cx,cy,w,h] -> [x1,y1,x2,y2], reference :68-74 (half extents computed as ``w / 2``)."""
    y = torch.empty_like(x)
    y[..., 0] = x[..., 0] - x[..., 2] / 2
    y[..., 1] = x[..., 1] - x[..., 3] / 2
    y[..., 2] = x[..., 0] + x[..., 2] / 2
    y[..., 3] = x[..., 1] + x[..., 3] / 2
    return y


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


def ap_per_class(tp, conf, pred_cls, target_cls):
    """Per-class AP + pooled PR curve, reference :77-154.

    Returns ``(p, r, ap, f1, classes_int32, (precision_curve, recall_curve))``."""
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
    return p, r, ap, f1, unique_classes.astype("int32"), (precision_curve, recall_curve)


def bbox_wh_iou(wh1, wh2):
    """IoU of anchor shape ``wh1`` with target shapes ``wh2[n,2]`` (reference :239-245)."""
    wh2 = wh2.t()
    w1, h1 = wh1[0], wh1[1]
    w2, h2 = wh2[0], wh2[1]
    inter_area = torch.min(w1, w2) * torch.min(h1, h2)
    union_area = (w1 * h1 + 1e-16) + w2 * h2 - inter_area
    return inter_area / union_area


def bbox_iou(box1, box2, x1y1x2y2=True):
    """IoU with the reference's **+1 pixel** convention (reference :248-278)."""
    if not x1y1x2y2:
        b1_x1, b1_x2 = box1[:, 0] - box1[:, 2] / 2, box1[:, 0] + box1[:, 2] / 2
        b1_y1, b1_y2 = box1[:, 1] - box1[:, 3] / 2, box1[:, 1] + box1[:, 3] / 2
        b2_x1, b2_x2 = box2[:, 0] - box2[:, 2] / 2, box2[:, 0] + box2[:, 2] / 2
        b2_y1, b2_y2 = box2[:, 1] - box2[:, 3] / 2, box2[:, 1] + box2[:, 3] / 2
    else:
        b1_x1, b1_y1, b1_x2, b1_y2 = box1[:, 0], box1[:, 1], box1[:, 2], box1[:, 3]
        b2_x1, b2_y1, b2_x2, b2_y2 = box2[:, 0], box2[:, 1], box2[:, 2], box2[:, 3]
    ix1 = torch.max(b1_x1, b2_x1)
    iy1 = torch.max(b1_y1, b2_y1)
    ix2 = torch.min(b1_x2, b2_x2)
    iy2 = torch.min(b1_y2, b2_y2)
    inter_area = torch.clamp(ix2 - ix1 + 1, min=0) * torch.clamp(iy2 - iy1 + 1, min=0)
    b1_area = (b1_x2 - b1_x1 + 1) * (b1_y2 - b1_y1 + 1)
    b2_area = (b2_x2 - b2_x1 + 1) * (b2_y2 - b2_y1 + 1)
    return inter_area / (b1_area + b2_area - inter_area + 1e-16)


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
    nB, nA, nG = pred_boxes.size(0), pred_boxes.size(1), pred_boxes.size(2)
    nC = pred_cls.size(-1)

    def zeros(*shape, dtype=torch.float32):
        return torch.zeros(*shape, dtype=dtype, device=dev)

    obj_mask = zeros(nB, nA, nG, nG, dtype=torch.uint8)
    noobj_mask = torch.ones(nB, nA, nG, nG, dtype=torch.uint8, device=dev)
    class_mask = zeros(nB, nA, nG, nG)
    iou_scores = zeros(nB, nA, nG, nG)
    tx, ty, tw, th = (zeros(nB, nA, nG, nG) for _ in range(4))
    tcls = zeros(nB, nA, nG, nG, nC)

    target_boxes = target[:, 2:6] * nG
    gxy = target_boxes[:, :2]
    gwh = target_boxes[:, 2:]
    ious = torch.stack([bbox_wh_iou(anchor, gwh) for anchor in anchors])
    _, best_n = ious.max(0)
    b, target_labels = target[:, :2].long().t()
    gx, gy = gxy.t()
    gw, gh = gwh.t()
    gi, gj = gxy.long().t()

    obj_mask[b, best_n, gj, gi] = 1
    noobj_mask[b, best_n, gj, gi] = 0
    for k, anchor_ious in enumerate(ious.t()):
        noobj_mask[b[k], anchor_ious > ignore_thres, gj[k], gi[k]] = 0

    tx[b, best_n, gj, gi] = gx - gx.floor()
    ty[b, best_n, gj, gi] = gy - gy.floor()
    tw[b, best_n, gj, gi] = torch.log(gw / anchors[best_n][:, 0] + 1e-16)
    th[b, best_n, gj, gi] = torch.log(gh / anchors[best_n][:, 1] + 1e-16)
    tcls[b, best_n, gj, gi, target_labels] = 1
    class_mask[b, best_n, gj, gi] = (pred_cls[b, best_n, gj, gi].argmax(-1) == target_labels).float()
    iou_scores[b, best_n, gj, gi] = bbox_iou(pred_boxes[b, best_n, gj, gi], target_boxes, x1y1x2y2=False)

    tconf = obj_mask.float()
    return iou_scores, class_mask, obj_mask, noobj_mask, tx, ty, tw, th, tcls, tconf
