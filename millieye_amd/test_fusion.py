"""Evaluation harness: ``evaluate(model, ...)`` -> (precision, recall, AP, f1, ap_class, box_stat, pr_curve).

Host-side mirror of ``module3_our_dataset/test_fusion.py:24-115``.  Like the reference it builds
``MyDataset(mode, illumination, augment=False, multiscale=False, test_list, dataset_folder)`` itself - note: *without*
``img_size``, so frames are always produced at the dataset default of 416 and ``img_size`` only scales the targets
(test_fusion.py:48,96) - from ``millieye_amd/utils/datasets.py`` (batches assembled on the GPU); ``dataset_folder``
(hard-wired to ``../data/our_dataset`` in the reference) and a ready ``dataloader`` can be passed instead.

Per batch: ``mode_selection`` -> ``model(imgs, radar_maps, radar_boxes, mode)`` (all device work:
detector, NMS, heads) -> regroup ``[m,8]`` rows per image -> ``get_batch_statistics``; finally
``ap_per_class``.  The metric code is host-side by design (row a18)."""
from __future__ import division

import numpy as np
import torch

from .utils.utils import ap_per_class, get_batch_statistics, xywh2xyxy

try:
    import tqdm
except Exception:  # pragma: no cover
    tqdm = None

__all__ = ["mode_selection", "evaluate", "regroup_outputs", "batch_statistics_device"]


def mode_selection(mode, img, paths):
    """mode in [millieye, yolo, radar, auto]; auto -> fusion iff the batch is dark (quirk q14)."""
    if mode in [0, 1, 2]:
        return mode
    if mode == 3:
        return 0 if img.mean() < 0.1 else 1


def regroup_outputs(outputs, batch_size):
    """``[m,8]`` rows (image_i first) -> list of ``[n_i,7]`` CPU tensors / ``None`` per image, row order
    kept (reference test_fusion.py:80-87, vectorised: one device->host copy instead of m)."""
    rows = outputs.to(torch.device("cpu"))
    grouped = [None for _ in range(batch_size)]
    if rows.shape[0] == 0:
        return grouped
    idx = rows[:, 0].int()
    for i in range(batch_size):
        sel = rows[idx == i]
        if sel.shape[0]:
            grouped[i] = sel[:, 1:]
    return grouped


def batch_statistics_device(outputs, targets, batch_size, iou_threshold):
    """``get_batch_statistics(regroup_outputs(outputs), targets, iou_threshold)`` with the greedy matching done by
    ``me_batch_statistics_f32`` on the device rows: same list of
    ``[true_positives, pred_scores, pred_labels]`` per image that has detections, one device->host copy per batch
    instead of a python loop per detection.  ``outputs`` [m,8] CUDA, ``targets`` [q,6] xyxy in pixels (any device)."""
    from . import hip

    m = outputs.shape[0]
    if m == 0:
        return []
    rows = outputs.contiguous()
    tg = targets.to(device=rows.device, dtype=torch.float32).contiguous()
    tp = torch.zeros(m, device=rows.device, dtype=torch.float32)
    hip.check(hip.lib().me_batch_statistics_f32(rows.data_ptr(), m, rows.shape[1], tg.data_ptr() if len(tg) else None,
                                                len(tg), batch_size, float(iou_threshold), tp.data_ptr(),
                                                hip.stream_ptr()), "me_batch_statistics_f32")
    host = torch.cat((rows, tp[:, None]), 1).cpu().numpy()
    idx = host[:, 0].astype(np.int32)
    metrics = []
    for i in range(batch_size):
        sel = host[idx == i]
        if len(sel):
            metrics.append([sel[:, -1].astype(np.float64), sel[:, 5], sel[:, -2]])
    return metrics


def evaluate(model, mode, model_mode, illumination, iou_thresh, nms_thresh, img_size, batch_size, test_list,
             dataloader=None, dataset_folder="../data/our_dataset", num_workers=4, device_statistics=True):
    model.eval()
    if dataloader is None:
        from .utils.datasets import MyDataset

        dataset = MyDataset(mode=mode, illumination=illumination, augment=False, multiscale=False,
                            test_list=test_list, dataset_folder=dataset_folder)
        dataloader = torch.utils.data.DataLoader(dataset, batch_size=batch_size, shuffle=False, num_workers=num_workers,
                                                 pin_memory=False, collate_fn=dataset.collate_fn)
    device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    labels = []
    sample_metrics = []
    box_stat = dict(before=[1], after=[1])
    it = dataloader if tqdm is None else tqdm.tqdm(dataloader, desc="Detecting objects")
    for (paths, imgs, targets, radar_boxes, radar_maps) in it:
        with torch.no_grad():
            imgs = imgs.to(device)
            radar_maps = radar_maps.to(device)
            radar_boxes = radar_boxes.to(device)
            mode_now = mode_selection(model_mode, imgs, paths)
            outputs = model(imgs, radar_maps, radar_boxes, mode_now)
            on_device = device_statistics and outputs.is_cuda
            if on_device:
                counts = torch.bincount(outputs[:, 0].to(torch.int64), minlength=len(imgs))[:len(imgs)].tolist()
                box_stat["after"] += [int(c) for c in counts]
            else:
                outputs_reshape = regroup_outputs(outputs, len(imgs))
                for image_pred in outputs_reshape:
                    box_stat["after"].append(len(image_pred) if image_pred is not None else 0)
        labels += targets[:, 1].tolist()
        targets[:, 2:] = xywh2xyxy(targets[:, 2:])
        targets[:, 2:] *= img_size
        if on_device:
            sample_metrics += batch_statistics_device(outputs, targets, len(imgs), iou_thresh)
        else:
            sample_metrics += get_batch_statistics(outputs_reshape, targets, iou_threshold=iou_thresh)

    if sample_metrics == []:
        true_positives, pred_scores, pred_labels, labels = np.array([0]), np.array([1]), np.array([1]), np.array([1])
    else:
        true_positives, pred_scores, pred_labels = [np.concatenate(x, 0) for x in list(zip(*sample_metrics))]
    precision, recall, AP, f1, ap_class, pr_curve = ap_per_class(true_positives, pred_scores, pred_labels, labels)
    return precision, recall, AP, f1, ap_class, box_stat, pr_curve
