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
