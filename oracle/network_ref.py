"""ORACLE (test infrastructure - never imported by the product path).

CPU restatement of the reference's fusion network forward, ``Network.forward`` in
module3_our_dataset/my_models.py:433-641, as a function of a state dict with the reference's
parameter names:

  * non_max_suppression_cpp          utils/utils.py:337-378  (nms_cpp below)
  * proposal assembly                my_models.py:459-473
  * cnn_layers_1 / cnn_layers_3      my_models.py:47-77 / 130-157  (eval-mode BatchNorm)
  * ps_roi_align / roi_align         -> oracle/tv_ops (torchvision restatement, PARITY UNPINNED)
  * refinement_head / ensemble_head  my_models.py:260-284 / 202-210
  * masks, thresholds, box_regress, ordering   my_models.py:502-539, 378-391
  * training tail (labels, sampling, focal / BCE losses)   my_models.py:545-639

Pinned by tests/golden/network_*.npz, produced by running the imported reference (with its
torchvision calls routed to oracle/tv_ops) on the same deterministic weights and inputs.
Ties in the final ordering are resolved stably (lower row first); torch.sort leaves them
unspecified.
"""
import random

import numpy as np
import torch
import torch.nn.functional as F

from . import darknet_ref, tv_ops


def xywh2xyxy(x):
    y = torch.empty_like(x)
    y[..., 0] = x[..., 0] - x[..., 2] / 2
    y[..., 1] = x[..., 1] - x[..., 3] / 2
    y[..., 2] = x[..., 0] + x[..., 2] / 2
    y[..., 3] = x[..., 1] + x[..., 3] / 2
    return y


def xyxy2xywh(x):
    y = torch.zeros_like(x)
    y[..., 0] = (x[..., 0] + x[..., 2]) / 2
    y[..., 1] = (x[..., 1] + x[..., 3]) / 2
    y[..., 2] = x[..., 2] - x[..., 0]
    y[..., 3] = x[..., 3] - x[..., 1]
    return y


def bbox_iou_plus1(box1, box2):
    """utils/utils.py:248-278, x1y1x2y2=True branch (+1 pixel convention)."""
    ix1 = torch.max(box1[:, 0], box2[:, 0])
    iy1 = torch.max(box1[:, 1], box2[:, 1])
    ix2 = torch.min(box1[:, 2], box2[:, 2])
    iy2 = torch.min(box1[:, 3], box2[:, 3])
    inter = torch.clamp(ix2 - ix1 + 1, min=0) * torch.clamp(iy2 - iy1 + 1, min=0)
    a1 = (box1[:, 2] - box1[:, 0] + 1) * (box1[:, 3] - box1[:, 1] + 1)
    a2 = (box2[:, 2] - box2[:, 0] + 1) * (box2[:, 3] - box2[:, 1] + 1)
    return inter / (a1 + a2 - inter + 1e-16)


def nms_cpp(prediction, conf_thresh, nms_thresh=0.5, detections_per_img=200):
    """non_max_suppression_cpp (utils/utils.py:337-378); does not modify ``prediction``."""
    prediction = prediction.clone()
    prediction[..., :4] = xywh2xyxy(prediction[..., :4])
    output = [None for _ in range(len(prediction))]
    for image_i, image_pred in enumerate(prediction):
        image_pred = image_pred[image_pred[:, 4] >= conf_thresh]
        if not image_pred.size(0):
            continue
        class_confs, class_preds = image_pred[:, 5:].max(1, keepdim=True)
        detections = torch.cat((image_pred[:, :5], class_confs.float(), class_preds.float(), image_pred[:, 5:]), 1)
        keep = tv_ops.batched_nms(detections[:, :4], detections[:, 4], detections[:, 6], nms_thresh)
        keep = keep[:detections_per_img]
        if len(keep) > 0:
            output[image_i] = detections[keep]
    return output


def _bn_eval(x, sd, prefix, eps=1e-5):
    return F.batch_norm(x, sd[prefix + "running_mean"], sd[prefix + "running_var"], sd[prefix + "weight"],
                        sd[prefix + "bias"], False, 0.1, eps)


def img_cnn(sd, fm, storage="f32"):
    w = sd["img_cnn_layers.net.conv_0.weight"]
    if storage != "f32":  # 16-bit storage modes: the score-map convolution multiplies the 16-bit tap with 16-bit weights (fp32 sums, fp32 out)
        w = w.to(torch.float16 if storage == "f16" else torch.bfloat16).float()
    x = F.conv2d(fm, w, sd["img_cnn_layers.net.conv_0.bias"])
    return F.leaky_relu(_bn_eval(x, sd, "img_cnn_layers.net.batch_norm_0."), 0.1)


def radar_cnn(sd, maps):
    x = maps
    for name in ("conv1", "conv2", "conv3"):
        p = f"radar_cnn_layers.{name}."
        x = F.conv2d(x, sd[p + "0.weight"], sd[p + "0.bias"], padding=1)
        x = F.leaky_relu(_bn_eval(x, sd, p + "1."), 0.1)
    x = F.conv2d(x, sd["radar_cnn_layers.conv3.3.weight"], sd["radar_cnn_layers.conv3.3.bias"])
    return torch.sigmoid(x)


def refinement(sd, radar_feat, img_feat):
    p = "refinement_head."
    t = F.leaky_relu(F.linear(img_feat.flatten(start_dim=1), sd[p + "net0.0.weight"], sd[p + "net0.0.bias"]), 0.1)
    box_regression = F.linear(t, sd[p + "net1.0.weight"], sd[p + "net1.0.bias"])
    class_vector = torch.sigmoid(F.linear(t, sd[p + "net2.0.weight"], sd[p + "net2.0.bias"]))
    r = F.conv2d(radar_feat, sd[p + "radar_net.0.weight"], sd[p + "radar_net.0.bias"])
    r = F.leaky_relu(_bn_eval(r, sd, p + "radar_net.1."), 0.1)
    r = torch.sigmoid(F.conv2d(r, sd[p + "radar_net.3.weight"], sd[p + "radar_net.3.bias"]))
    radar_confidence = r.squeeze(-1).squeeze(-1)
    confidence = torch.sigmoid(radar_confidence + class_vector[:, :1])
    return box_regression, torch.cat((confidence, class_vector[:, 1:2]), -1)


def ensemble(sd, refinement_vector, yolo_vector):
    x = torch.stack((refinement_vector, yolo_vector), -1)
    x = F.leaky_relu(F.linear(x, sd["ensemble_head.fc1.0.weight"], sd["ensemble_head.fc1.0.bias"]), 0.1)
    x = F.linear(x.flatten(start_dim=1), sd["ensemble_head.fc2.0.weight"], sd["ensemble_head.fc2.0.bias"])
    return torch.softmax(x, dim=1)


def box_regress(regress_param, roi_location):
    x, y, w, h = xyxy2xywh(roi_location).t()
    xr = regress_param[:, 0] * w + x
    yr = regress_param[:, 1] * h + y
    wr = torch.exp(regress_param[:, 2]) * w
    hr = torch.exp(regress_param[:, 3]) * h
    return xywh2xyxy(torch.stack((xr, yr, wr, hr), 1))


def network_forward(cfg_text, sd, images, maps, radar_boxes, model_mode=0, conf_thresh=0.2, thr_img=0.0,
                    thr_radar=0.0, class_idx=0, class_num=1, tap_module=8, return_internals=False, storage="f32"):
    """Inference branch of Network.forward (targets=None).  ``sd``: state dict of the whole
    Network (detector keys prefixed ``base_detector.``).  ``radar_boxes`` [r,5] in [0,1] units is
    NOT modified (the reference scales it in place; the scaled copy is returned in internals).
    ``storage="bf16"`` / ``"f16"``: the build's 16-bit storage modes - the detector restated by ``darknet_forward(storage=...)``
    (16-bit activations and weights, fp32 detection maps), the score-map convolution on the 16-bit tap with 16-bit weights,
    everything behind it fp32 as in the HIP path.  No reference counterpart (the reference is fp32)."""
    det_sd = {k[len("base_detector."):]: v for k, v in sd.items() if k.startswith("base_detector.")}
    with torch.no_grad():
        feature_map, output_tensor = darknet_ref.darknet_forward(cfg_text, det_sd, images, tap_module=tap_module, storage=storage)
        detections = nms_cpp(output_tensor, conf_thresh)
        img_boxes = []
        for image_i, det in enumerate(detections):
            if det is not None:
                det = det[det[:, 6] == class_idx]
                if len(det) > 0:
                    b = torch.zeros((len(det), 8 + class_num))
                    b[:, 0] = image_i
                    b[:, 1:] = det[:, :7 + class_num]
                    img_boxes.append(b)
        img_boxes = torch.cat(img_boxes, 0) if img_boxes else torch.empty((0, 8 + class_num))
        num_img = len(img_boxes)
        if model_mode == 1:
            return img_boxes[:, :8]
        if model_mode == 2:
            thr_img = 1
        roi_score_map = img_cnn(sd, feature_map, storage)
        radar_score_map = radar_cnn(sd, maps)
        radar_boxes = radar_boxes.clone()
        if len(radar_boxes) > 0:
            radar_boxes[:, 1:] *= images.shape[-1]
        box_locations = torch.cat((img_boxes[:, :5], radar_boxes), 0)
        crop_img = tv_ops.ps_roi_align(roi_score_map, box_locations, (7, 7), spatial_scale=1. / 16)
        crop_radar = tv_ops.roi_align(radar_score_map, box_locations, (7, 7), spatial_scale=1. / 16)
        regress_param, refinement_vector = refinement(sd, crop_radar, crop_img)
        radar_rows = torch.cat((radar_boxes, refinement_vector[num_img:], torch.zeros((len(radar_boxes), 1)),
                                refinement_vector[num_img:, 1:]), -1)
        boxes = torch.cat((img_boxes, radar_rows), 0)
        yolo_vector = torch.cat((img_boxes[:, 5:6], img_boxes[:, 8:]), 1)
        masks_img = ensemble(sd, refinement_vector[:num_img], yolo_vector)
        masks = torch.cat((masks_img[:, :1], refinement_vector[num_img:, :1]), 0)
        masks = torch.cat((1 - masks, masks), -1)
        positive = torch.cat((masks[:num_img, 1] > thr_img, masks[num_img:, 1] > thr_radar), 0)
        if model_mode != 2:
            located = box_regress(regress_param[positive], boxes[positive, 1:5])
        else:
            located = boxes[positive, 1:5]
        output = torch.cat((boxes[positive, :1], located, masks[positive, 1:], boxes[positive, 6:8]), -1)
        masks_tmp = masks.clone()
        masks_tmp[num_img:, 1] /= 5
        order = torch.sort(masks_tmp[positive, 1], descending=True, stable=True).indices
        output = output[order]
    if return_internals:
        return output, dict(img_boxes=img_boxes, regress=regress_param, refine=refinement_vector, masks=masks,
                            roi_score_map=roi_score_map, radar_score_map=radar_score_map,
                            crop_img=crop_img, crop_radar=crop_radar, radar_boxes=radar_boxes, boxes=boxes)
    return output


# ---------------------------------------------------------------------------------------------------
# training step (reference my_models.py:433-641 with targets, train-mode heads, frozen eval detector)
# ---------------------------------------------------------------------------------------------------
def focal_loss(inputs, labels, alpha, gamma=2):
    """FocalLoss(alpha, gamma, reduction='sum'), my_models.py:287-314."""
    alpha_t = torch.where(labels[:, 1:2] == 1, torch.full((labels.shape[0], 1), alpha),
                          torch.full((labels.shape[0], 1), 1 - alpha))
    probs = (inputs * labels).sum(1).view(-1, 1)
    return (-alpha_t * torch.pow(1 - probs, gamma) * probs.log()).sum()


def obtain_iou_labels(boxes, targets, multi_boxes=True):
    """my_models.py:317-375 (multi_boxes is always truthy at the call site: quirk q5)."""
    image_index, pred_classes, pred_boxes = boxes[:, :1], boxes[:, 1:2], boxes[:, 2:]
    detected = []
    iou_labels = torch.zeros((len(image_index), 1))
    target_location = torch.zeros((len(image_index), 4))
    for i in range(len(boxes)):
        sel = (targets[:, 0] == image_index[i]) & (targets[:, 1] == pred_classes[i])
        if not bool(sel.any()):
            continue
        tb = targets[sel][:, 2:]
        ious = bbox_iou_plus1(pred_boxes[i].unsqueeze(0), tb)
        if len(ious) > 0:
            iou, ti = ious.max(0)
            if (ti not in detected) or multi_boxes:
                iou_labels[i] = iou
                target_location[i] = tb[ti]
                if iou > 0.7:
                    detected += [ti]
    return iou_labels, target_location


def _bn_train(x, P, B, prefix, training=True):
    return F.batch_norm(x, B[prefix + "running_mean"], B[prefix + "running_var"], P[prefix + "weight"],
                        P[prefix + "bias"], training, 0.1, 1e-5)


def network_train_step(cfg_text, sd, images, maps, radar_boxes, targets, conf_thresh=0.2, class_idx=0, class_num=1,
                       tap_module=8, iou_thresh=(0.3, 0.7), alpha=0.75, balance_factor=5, loss_lambda=(6, 1),
                       bn_training=True):
    """One training forward + backward on CPU autograd.  ``sd``: full Network state dict.  ``targets`` [q,6]
    (image_i, class, cx, cy, w, h in [0,1]) is NOT modified.  Python's ``random`` must be seeded by the
    caller (negative sampling, quirk q7).  Returns dict(loss, masks_loss, conf_loss, output, grads{name: tensor},
    buffers{name: updated running stat}, internals).  ``bn_training=False`` (not a reference mode; used by the
    data-parallel equivalence tests): the head BatchNorms use their running statistics, which makes every loss term and
    gradient a plain sum over frames."""
    det_sd = {k[len("base_detector."):]: v for k, v in sd.items() if k.startswith("base_detector.")}
    P = {k: v.clone().requires_grad_(True) for k, v in sd.items()
         if not k.startswith("base_detector.") and v.dtype == torch.float32 and "running_" not in k}
    B = {k: v.clone() for k, v in sd.items() if "running_" in k and not k.startswith("base_detector.")}
    with torch.no_grad():
        feature_map, output_tensor = darknet_ref.darknet_forward(cfg_text, det_sd, images, tap_module=tap_module)
        detections = nms_cpp(output_tensor, conf_thresh)
        img_boxes = []
        for image_i, det in enumerate(detections):
            if det is not None:
                det = det[det[:, 6] == class_idx]
                if len(det) > 0:
                    b = torch.zeros((len(det), 8 + class_num))
                    b[:, 0] = image_i
                    b[:, 1:] = det[:, :7 + class_num]
                    img_boxes.append(b)
        img_boxes = torch.cat(img_boxes, 0) if img_boxes else torch.empty((0, 8 + class_num))
    num_img = len(img_boxes)
    p = "img_cnn_layers.net."
    x = F.conv2d(feature_map, P[p + "conv_0.weight"], P[p + "conv_0.bias"])
    roi_score_map = F.leaky_relu(_bn_train(x, P, B, p + "batch_norm_0.", bn_training), 0.1)
    x = maps
    for name in ("conv1", "conv2", "conv3"):
        q = f"radar_cnn_layers.{name}."
        x = F.conv2d(x, P[q + "0.weight"], P[q + "0.bias"], padding=1)
        x = F.leaky_relu(_bn_train(x, P, B, q + "1.", bn_training), 0.1)
    radar_score_map = torch.sigmoid(F.conv2d(x, P["radar_cnn_layers.conv3.3.weight"], P["radar_cnn_layers.conv3.3.bias"]))
    radar_boxes = radar_boxes.clone()
    if len(radar_boxes) > 0:
        radar_boxes[:, 1:] *= images.shape[-1]
    box_locations = torch.cat((img_boxes[:, :5], radar_boxes), 0)
    crop_img = tv_ops.ps_roi_align(roi_score_map, box_locations, (7, 7), spatial_scale=1. / 16)
    crop_radar = tv_ops.roi_align(radar_score_map, box_locations, (7, 7), spatial_scale=1. / 16)
    r = "refinement_head."
    t = F.leaky_relu(F.linear(crop_img.flatten(start_dim=1), P[r + "net0.0.weight"], P[r + "net0.0.bias"]), 0.1)
    regress_param = F.linear(t, P[r + "net1.0.weight"], P[r + "net1.0.bias"])
    class_vector = torch.sigmoid(F.linear(t, P[r + "net2.0.weight"], P[r + "net2.0.bias"]))
    rr = F.conv2d(crop_radar, P[r + "radar_net.0.weight"], P[r + "radar_net.0.bias"])
    rr = F.leaky_relu(_bn_train(rr, P, B, r + "radar_net.1.", bn_training), 0.1)
    rr = torch.sigmoid(F.conv2d(rr, P[r + "radar_net.3.weight"], P[r + "radar_net.3.bias"]))
    confidence = torch.sigmoid(rr.squeeze(-1).squeeze(-1) + class_vector[:, :1])
    refinement_vector = torch.cat((confidence, class_vector[:, 1:2]), -1)
    radar_rows = torch.cat((radar_boxes, refinement_vector[num_img:], torch.zeros((len(radar_boxes), 1)),
                            refinement_vector[num_img:, 1:]), -1)
    boxes = torch.cat((img_boxes, radar_rows), 0)
    yolo_vector = torch.cat((img_boxes[:, 5:6], img_boxes[:, 8:]), 1).detach()
    xx = torch.stack((refinement_vector[:num_img], yolo_vector), -1)
    xx = F.leaky_relu(F.linear(xx, P["ensemble_head.fc1.0.weight"], P["ensemble_head.fc1.0.bias"]), 0.1)
    masks_img = torch.softmax(F.linear(xx.flatten(start_dim=1), P["ensemble_head.fc2.0.weight"],
                                       P["ensemble_head.fc2.0.bias"]), dim=1)
    masks = torch.cat((masks_img[:, :1], refinement_vector[num_img:, :1]), 0)
    masks = torch.cat((1 - masks, masks), -1)
    positive = torch.cat((masks[:num_img, 1] > 0, masks[num_img:, 1] > 0), 0)
    with torch.no_grad():
        output = torch.cat((boxes[positive, :1], box_regress(regress_param[positive], boxes[positive, 1:5]),
                            masks[positive, 1:], boxes[positive, 6:8]), -1)
        masks_tmp = masks.clone()
        masks_tmp[num_img:, 1] /= 5
        output = output[torch.sort(masks_tmp[positive, 1], descending=True, stable=True).indices]
    tg = targets.clone()
    tg[:, 2:] = xywh2xyxy(tg[:, 2:])
    tg[:, 2:] *= images.shape[3]
    boxes_cpu = torch.cat((boxes[:, :1], boxes[:, 7:8], boxes[:, 1:5]), 1).detach()
    iou_labels, _ = obtain_iou_labels(boxes_cpu, tg, iou_thresh)
    pos_filter = (iou_labels > iou_thresh[1]).flatten()
    neg_filter = (iou_labels < iou_thresh[0]).flatten()
    pos_idx = np.where(pos_filter)[0]
    neg_idx = np.where(neg_filter)[0]
    top_k = min(len(pos_idx) * balance_factor, len(neg_idx))
    label_onehot = torch.tensor([1.0, 0.0]).repeat(masks.shape[0], 1)
    for i in pos_idx:
        label_onehot[i] = torch.tensor([0.0, 1.0])
    sample_filter = pos_filter.clone()
    selected = neg_idx[random.sample(range(len(neg_idx)), k=top_k)]
    sample_filter[selected] = True
    lab = label_onehot[:num_img][sample_filter[:num_img]]
    oh = masks[:num_img][sample_filter[:num_img]]
    masks_loss = focal_loss(oh, lab, alpha)
    conf_label = torch.zeros(len(boxes_cpu))
    conf_label[pos_idx] = 1.0
    conf_loss = F.binary_cross_entropy(refinement_vector[sample_filter, 0], conf_label[sample_filter], reduction="sum")
    loss = masks_loss + conf_loss / loss_lambda[0]
    loss.backward()
    grads = {k: (v.grad.clone() if v.grad is not None else None) for k, v in P.items()}
    return dict(loss=loss.detach(), masks_loss=masks_loss.detach(), conf_loss=conf_loss.detach(), output=output,
                grads=grads, buffers=B, n_pos=int(pos_filter.sum()), n_sampled=int(sample_filter.sum()),
                num_img=num_img, refinement_vector=refinement_vector.detach(), masks=masks.detach())
