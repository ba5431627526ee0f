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


def img_cnn(sd, fm):
    x = F.conv2d(fm, sd["img_cnn_layers.net.conv_0.weight"], sd["img_cnn_layers.net.conv_0.bias"])
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
                    thr_radar=0.0, class_idx=0, class_num=1, tap_module=8, return_internals=False):
    """Inference branch of Network.forward (targets=None).  ``sd``: state dict of the whole
    Network (detector keys prefixed ``base_detector.``).  ``radar_boxes`` [r,5] in [0,1] units is
    NOT modified (the reference scales it in place; the scaled copy is returned in internals)."""
    det_sd = {k[len("base_detector."):]: v for k, v in sd.items() if k.startswith("base_detector.")}
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
        if model_mode == 1:
            return img_boxes[:, :8]
        if model_mode == 2:
            thr_img = 1
        roi_score_map = img_cnn(sd, feature_map)
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
