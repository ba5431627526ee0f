"""ORACLE (test infrastructure only): CPU restatement of the stage-2 ``Network.forward`` inference branch,
``module2_mixed/my_models.py:299-364``, in stock torch CPU ops + oracle/tv_ops (``ps_roi_align`` / ``batched_nms``:
torchvision is absent, that boundary is PARITY-UNPINNED).  Everything else is pinned by
``tests/golden/network_m2_*.npz`` (outputs of the real module-2 reference, ``tests/golden/make_golden.py --module2``)."""
import torch
import torch.nn.functional as F

from . import darknet_ref, tv_ops
from .network_ref import box_regress, nms_cpp


def network_m2_forward(cfg_text, sd, images, conf_thresh=0.2, refine_threshold=0.0, class_num=12, tap_module=8,
                       return_internals=False):
    det_sd = {k[len("base_detector."):]: v for k, v in sd.items() if k.startswith("base_detector.")}
    with torch.no_grad():
        feature_map, output_tensor = darknet_ref.darknet_forward(cfg_text, det_sd, images, tap_module=tap_module)
        detections = nms_cpp(output_tensor, conf_thresh)                                   # :316
        boxes = []
        for image_i, det in enumerate(detections):                                         # :320-330
            if det is not None:
                b = torch.zeros((len(det), 8 + class_num))
                b[:, 0] = image_i
                b[:, 1:] = det[:, :7 + class_num]
                boxes.append(b)
        boxes = torch.cat(boxes, 0) if boxes else torch.empty((0, 8 + class_num))
        p = "fcn_layers.net."
        x = F.conv2d(feature_map, sd[p + "conv_0.weight"], sd[p + "conv_0.bias"])          # :337
        x = F.batch_norm(x, sd[p + "batch_norm_0.running_mean"], sd[p + "batch_norm_0.running_var"],
                         sd[p + "batch_norm_0.weight"], sd[p + "batch_norm_0.bias"], False, 0.1, 1e-5)
        roi_score_map = F.leaky_relu(x, 0.1)
        crop = tv_ops.ps_roi_align(roi_score_map, boxes[:, :5], (7, 7), spatial_scale=1. / 16)  # :340
        r = "refinement_head."
        t = F.leaky_relu(F.linear(crop.flatten(start_dim=1), sd[r + "net0.0.weight"], sd[r + "net0.0.bias"]), 0.1)
        regress_param = F.linear(t, sd[r + "net1.0.weight"], sd[r + "net1.0.bias"])
        refinement_vector = torch.sigmoid(F.linear(t, sd[r + "net2.0.weight"], sd[r + "net2.0.bias"]))
        yolo_vector = torch.cat((boxes[:, 5:6], boxes[:, 8:]), 1)                          # :344
        e = "ensemble_head."
        xx = torch.stack((refinement_vector, yolo_vector), -1)
        xx = F.leaky_relu(F.linear(xx, sd[e + "fc1.0.weight"], sd[e + "fc1.0.bias"]), 0.1)
        xx = F.leaky_relu(F.linear(xx.flatten(start_dim=1), sd[e + "fc2.0.weight"], sd[e + "fc2.0.bias"]), 0.1)
        masks = torch.softmax(xx, dim=1)
        positive = masks[:, 1] > refine_threshold                                          # :349
        output = torch.cat((boxes[positive, :1], box_regress(regress_param[positive], boxes[positive, 1:5]),
                            masks[positive, 1:], boxes[positive, 6:8]), -1)
        output = output[torch.sort(output[:, 5], descending=True, stable=True).indices]    # :357
    if return_internals:
        return output, dict(boxes=boxes, regress=regress_param, refine=refinement_vector, masks=masks, crop=crop)
    return output


# ---------------------------------------------------------------------------------------------------
# training step (module2_mixed/my_models.py:299-461 with targets; heads in train() mode, detector eval)
# ---------------------------------------------------------------------------------------------------
def network_m2_train_step(cfg_text, sd, images, targets, conf_thresh=0.2, class_num=12, tap_module=8,
                          iou_thresh=(0.3, 0.7), alpha=0.75, balance_fac=5, loss_lambda=(15, 5)):
    """One stage-2 training forward + backward on CPU autograd.  Seed ``torch`` (Dropout mask: one
    ``empty_like(hidden).bernoulli_(0.5)`` draw) and python's ``random`` (negative sampling) before calling.
    ``targets`` [q,6] (image_i, class, cx, cy, w, h in [0,1]) is NOT modified.  Returns dict(loss, terms, output, grads,
    buffers, n_pos, n_sampled)."""
    import random

    import numpy as np

    from .network_ref import obtain_iou_labels, xywh2xyxy, xyxy2xywh

    det_sd = {k[len("base_detector."):]: v for k, v in sd.items() if k.startswith("base_detector.")}
    P = {k: v.clone().requires_grad_(True) for k, v in sd.items()
         if not k.startswith("base_detector.") and v.dtype == torch.float32 and "running_" not in k}
    B = {k: v.clone() for k, v in sd.items() if "running_" in k and not k.startswith("base_detector.")}
    with torch.no_grad():
        feature_map, output_tensor = darknet_ref.darknet_forward(cfg_text, det_sd, images, tap_module=tap_module)
        detections = nms_cpp(output_tensor, conf_thresh)
        boxes = []
        for image_i, det in enumerate(detections):
            if det is not None:
                b = torch.zeros((len(det), 8 + class_num))
                b[:, 0] = image_i
                b[:, 1:] = det[:, :7 + class_num]
                boxes.append(b)
        boxes = torch.cat(boxes, 0) if boxes else torch.empty((0, 8 + class_num))
    p = "fcn_layers.net."
    x = F.conv2d(feature_map, P[p + "conv_0.weight"], P[p + "conv_0.bias"])
    x = F.batch_norm(x, B[p + "batch_norm_0.running_mean"], B[p + "batch_norm_0.running_var"], P[p + "batch_norm_0.weight"],
                     P[p + "batch_norm_0.bias"], True, 0.1, 1e-5)
    roi_score_map = F.leaky_relu(x, 0.1)
    crop = tv_ops.ps_roi_align(roi_score_map, boxes[:, :5], (7, 7), spatial_scale=1. / 16)
    r = "refinement_head."
    t = F.leaky_relu(F.linear(crop.flatten(start_dim=1), P[r + "net0.0.weight"], P[r + "net0.0.bias"]), 0.1)
    noise = torch.empty_like(t).bernoulli_(0.5)                       # nn.Dropout(0.5) in train mode (CPU path of aten)
    t = t * (noise / 0.5)
    regress_param = F.linear(t, P[r + "net1.0.weight"], P[r + "net1.0.bias"])
    refinement_vector = torch.sigmoid(F.linear(t, P[r + "net2.0.weight"], P[r + "net2.0.bias"]))
    yolo_vector = torch.cat((boxes[:, 5:6], boxes[:, 8:]), 1)
    e = "ensemble_head."
    xx = torch.stack((refinement_vector, yolo_vector), -1)
    xx = F.leaky_relu(F.linear(xx, P[e + "fc1.0.weight"], P[e + "fc1.0.bias"]), 0.1)
    xx = F.leaky_relu(F.linear(xx.flatten(start_dim=1), P[e + "fc2.0.weight"], P[e + "fc2.0.bias"]), 0.1)
    masks = torch.softmax(xx, dim=1)
    positive = masks[:, 1] > 0
    with torch.no_grad():
        output = torch.cat((boxes[positive, :1], box_regress(regress_param[positive], boxes[positive, 1:5]),
                            masks[positive, 1:], boxes[positive, 6:8]), -1)
        output = output[torch.sort(output[:, 5], descending=True, stable=True).indices]
    tg = targets.clone()
    tg[:, 2:] = xywh2xyxy(tg[:, 2:])
    tg[:, 2:] *= images.shape[-1]
    boxes_cpu = torch.cat((boxes[:, :1], boxes[:, 7:8], boxes[:, 1:5]), 1)
    iou_labels, target_location = obtain_iou_labels(boxes_cpu, tg)
    pos_filter = (iou_labels > iou_thresh[1]).flatten()
    neg_filter = (iou_labels < iou_thresh[0]).flatten()
    pos_idx, neg_idx = np.where(pos_filter)[0], np.where(neg_filter)[0]
    top_k = min(len(pos_idx) * balance_fac, len(neg_idx))
    label_onehot = torch.tensor([1.0, 0.0]).repeat(masks.shape[0], 1)
    for i in pos_idx:
        label_onehot[i] = torch.tensor([0.0, 1.0])
    sample_filter = pos_filter.clone()
    sample_filter[neg_idx[random.sample(range(len(neg_idx)), k=top_k)]] = True
    lab, m = label_onehot[sample_filter], masks[sample_filter]
    a = torch.where(lab[:, 1:2] == 1, torch.full((len(lab), 1), alpha), torch.full((len(lab), 1), 1 - alpha))
    probs = (m * lab).sum(1).view(-1, 1)
    masks_loss = (-a * torch.pow(1 - probs, 2) * probs.log()).sum()
    conf_label = torch.zeros(len(boxes_cpu))
    conf_label[pos_idx] = 1.0
    conf_loss = F.binary_cross_entropy(refinement_vector[sample_filter, 0], conf_label[sample_filter], reduction="sum")
    # regression_loss (:264-279): SmoothL1(sum)(computed targets, regress_param) - gradient flows into the 2nd argument
    roi = boxes[pos_filter, 1:5]
    xr, yr, wr, hr = xyxy2xywh(roi).t()
    xt, yt, wt, ht = xyxy2xywh(target_location[pos_filter]).t()
    p01 = torch.stack(((xt - xr) / (wr + 1e-16), (yt - yr) / (hr + 1e-16)), -1)
    p23 = torch.stack((torch.log(wt / wr + 1e-16), torch.log(ht / hr + 1e-16)), -1)
    rp = regress_param[pos_filter]
    loss_xy = F.smooth_l1_loss(p01, rp[:, :2], reduction="sum")
    loss_wh = F.smooth_l1_loss(p23, rp[:, 2:], reduction="sum")
    class_label = torch.zeros((len(boxes_cpu), class_num))
    for i, idx in enumerate(pos_idx):                                 # row i, not idx: the reference's quirk (:446-447)
        class_label[i, int(boxes_cpu[idx, 1])] = 1.0
    category_loss = F.binary_cross_entropy(refinement_vector[pos_filter, 1:], class_label[pos_filter], reduction="sum")
    loss = masks_loss + (conf_loss + category_loss) / loss_lambda[0] + (loss_xy + loss_wh) / loss_lambda[1]
    loss.backward()
    grads = {k: (v.grad.clone() if v.grad is not None else None) for k, v in P.items()}
    return dict(loss=loss.detach(), terms=[float(v.detach()) for v in (masks_loss, conf_loss, category_loss, loss_xy, loss_wh)],
                output=output, grads=grads, buffers=B, n_pos=int(pos_filter.sum()), n_sampled=int(sample_filter.sum()),
                refine=refinement_vector.detach(), masks=masks.detach(), boxes=boxes)
