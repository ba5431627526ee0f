"""Stage-2 evaluation harness: ``evaluate`` of ``module2_mixed/test_module2.py:25-96`` on the HIP ``Network``.

Same signature and return tuple ``(precision, recall, AP, f1, ap_class, box_stat, pr_curve)``; ``conf_thresh`` /
``nms_thresh`` are accepted and unused, as in the reference (the model carries its own thresholds).  The loop is the
reference's: batches of ``ListDataset(list_path, augment=False, multiscale=False)``, ``model(imgs)`` -> rows
``(image_i, x1, y1, x2, y2, conf, class_conf, class_pred)`` on the CPU, regrouped per image, ``box_stat["after"]`` counts,
targets rescaled to pixels, ``get_batch_statistics`` + ``ap_per_class`` (the stage-2 variant: the curve tuple carries the
confidences).  ``dataloader`` lets a caller (tests, the training loop) hand in its own iterable of
``(paths, imgs, targets)`` batches.
"""
import argparse

import numpy as np
import torch

from ..utils.parse_config import parse_data_config
from ..utils.utils import ap_per_class, get_batch_statistics, load_classes, xywh2xyxy
from .datasets import ListDataset
from .my_models import Network, define_yolo

__all__ = ["evaluate", "main"]


def evaluate(model, list_path, iou_thresh, conf_thresh, nms_thresh, img_size, batch_size, dataloader=None, n_cpu=0):
    model.eval()
    if dataloader is None:
        dataset = ListDataset(list_path, augment=False, multiscale=False)
        dataloader = torch.utils.data.DataLoader(dataset, batch_size=batch_size, shuffle=True, num_workers=n_cpu,
                                                 collate_fn=dataset.collate_fn)
    device = getattr(model, "device", torch.device("cuda"))
    labels = []
    sample_metrics = []
    box_stat = dict(before=[1], after=[1])
    for _, imgs, targets in dataloader:
        imgs = imgs.to(device)
        with torch.no_grad():
            outputs = model(imgs)  # [m,8] on the CPU
        outputs_reshape = [None for _ in range(len(imgs))]
        outputs = outputs.to(torch.device("cpu"))
        if outputs.shape[0]:
            idx = outputs[:, 0].int()
            for i in torch.unique(idx).tolist():  # rows keep their order inside an image, as the reference's row-by-row cat
                outputs_reshape[i] = outputs[idx == i][:, 1:]
        for image_pred in outputs_reshape:
            box_stat["after"].append(len(image_pred) if image_pred is not None else 0)
        labels += targets[:, 1].tolist()
        targets[:, 2:] = xywh2xyxy(targets[:, 2:])
        targets[:, 2:] *= img_size
        sample_metrics += get_batch_statistics(outputs_reshape, targets, iou_threshold=iou_thresh)
    if sample_metrics == []:
        true_positives, pred_scores, pred_labels, labels = np.array([0]), np.array([1]), np.array([1]), np.array([1])
    else:
        true_positives, pred_scores, pred_labels = [np.concatenate(x, 0) for x in list(zip(*sample_metrics))]
    precision, recall, AP, f1, ap_class, pr_curve = ap_per_class(true_positives, pred_scores, pred_labels, labels,
                                                                 with_conf=True)
    return precision, recall, AP, f1, ap_class, box_stat, pr_curve


def build_parser():
    p = argparse.ArgumentParser(description="stage-2 evaluation (module2_mixed/test_module2.py)")
    p.add_argument("--batch_size", type=int, default=32)
    p.add_argument("--data_config", type=str, default="config/exdark.data")
    p.add_argument("--classes_path", type=str, default="config/exdark.names")
    p.add_argument("--conf_thresh", type=float, default=0.01)
    p.add_argument("--iou_thresh", type=float, default=0.5)
    p.add_argument("--yolo_cfg", type=str, default="config/yolov3-tiny-12.cfg")
    p.add_argument("--img_size", type=int, default=416)
    p.add_argument("--checkpoint", type=str, default="./checkpoints/module2_best_mixed.pth")
    p.add_argument("--n_cpu", type=int, default=0)
    return p


def main(argv=None):
    opt = build_parser().parse_args(argv)
    print(opt)
    data_config = parse_data_config(opt.data_config)
    valid_path = data_config["valid"]
    class_names = load_classes(opt.classes_path)
    model = Network(define_yolo(opt.yolo_cfg), opt.conf_thresh)
    model = model.to(model.device)
    model.load_state_dict(torch.load(opt.checkpoint, map_location=model.device))
    print("Compute mAP...")
    precision, recall, ap, f1, ap_class, box_stat, pr_curve = evaluate(
        model, list_path=valid_path, iou_thresh=opt.iou_thresh, conf_thresh=0.01, nms_thresh=0.5, img_size=opt.img_size,
        batch_size=opt.batch_size, n_cpu=opt.n_cpu)
    print(f"img_number: {len(box_stat['after'])}, sample_number: {len(np.atleast_1d(pr_curve[0]))}")
    for i, c in enumerate(ap_class):
        print(f"+ Class {c} ({class_names[i]})".ljust(30)
              + f"-AP: {ap[i]:.3f} -Precision:{precision[i]:.3f} -Recall:{recall[i]:.3f}")
    print(f"mAP: {ap.mean()}")
    return precision, recall, ap, f1, ap_class, box_stat, pr_curve


if __name__ == "__main__":
    main()
