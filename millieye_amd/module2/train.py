"""Stage-2 training harness: the loop of ``module2_mixed/train.py``.

The reference's loop lives under ``if __name__ == "__main__"`` (train.py:23-203); here it is :func:`train_loop`, driven by
:func:`main` (the script's command line, ``ListDataset`` as the producer) or by any iterable of ``(paths, imgs, targets)``
batches.  Kept line by line:

* a checkpoint restores detector + stage-2 parameters (``--checkpoint``); otherwise ``weights_init_normal`` on every module
  and ``init_yolo`` of the detector from ``--yolo_weights`` (train.py:103-109);
* ``AdamW(model.parameters(), lr=1e-4)`` over *all* parameters (train.py:122) - the detector's never receive a gradient
  (its forward runs outside the autograd graph) and are skipped by the optimizer;
* per epoch ``model.train(); model.base_detector.eval()``; per batch ``output, loss, metric = model(imgs, targets)``,
  ``loss.backward()``, ``optimizer.step(); optimizer.zero_grad()`` when ``batches_done % gradient_accumulations == 0``
  (so the very first step sees one batch of gradients), ``model.seen += imgs.size(0)`` (train.py:128-167);
* ``evaluate(model, list_path=valid_path, iou_thresh=0.5, conf_thresh=0.01, nms_thresh=0.5, ...)`` every
  ``evaluation_interval`` epochs, *then* ``checkpoints/ckpt_{epoch}.pth`` every ``checkpoint_interval`` epochs
  (train.py:169-203; the order is the reverse of the stage-3 script).

The TensorBoard scalars (loss, "precesion" = tp / positive, recall = tp / true) go to an optional ``writer``; a zero
denominator skips the scalar instead of raising.  With ``torch.distributed`` initialised every rank runs the loop on its own
batches, the gradients are SUM-all-reduced in one bucket before the step (``millieye_amd/parallel.py``) and the head
BatchNorm's running statistics are averaged over the ranks once per epoch.
"""
import argparse
import datetime
import os
import time

import torch

from .. import parallel
from ..utils.parse_config import parse_data_config
from ..utils.utils import load_classes, weights_init_normal
from .datasets import ListDataset
from .my_models import Network, define_yolo, init_yolo
from .test_module2 import evaluate as _evaluate

__all__ = ["train_loop", "main"]


def train_loop(model, dataloader, *, epochs, gradient_accumulations=2, checkpoint_interval=1, evaluation_interval=1,
               valid_path=None, img_size=416, batch_size=24, class_names=None, optimizer=None, evaluate_fn=_evaluate,
               evaluate_kwargs=None, checkpoint_dir="checkpoints", writer=None, log=print):
    """Runs the loop; returns ``dict(losses, steps, checkpoints, evaluations)`` (the reference only prints)."""
    device = getattr(model, "device", torch.device("cuda"))
    os.makedirs(checkpoint_dir, exist_ok=True)
    if optimizer is None:
        params = list(model.parameters())   # AdamW(lr=1e-4) over all parameters (reference module2/train.py:122)
        if params and all(p.is_cuda and p.dtype == torch.float32 for p in params):
            from ..optim import AdamW   # one launch per step (millieye_amd/optim.py), torch.optim.AdamW's state and update rule
            optimizer = AdamW(params, lr=1e-4)
        else:
            optimizer = torch.optim.AdamW(params, lr=1e-4)
    distributed = torch.distributed.is_available() and torch.distributed.is_initialized()
    rank = torch.distributed.get_rank() if distributed else 0
    history = dict(losses=[], steps=[], checkpoints=[], evaluations=[])
    trainable = [p for n, p in model.named_parameters() if p.requires_grad and not n.startswith("base_detector.")]

    for epoch in range(epochs):
        model.train()
        model.base_detector.eval()
        start_time = time.time()
        parallel.begin_epoch(dataloader, epoch, device)  # equal batch counts on every rank + the sampler's epoch
        # one batch of look-ahead, as in the stage-3 loop (millieye_amd/train.py): the frozen detector's part of batch k + 1 runs
        # under the host-bound tail of batch k (Network.queue_detector_prefetch; MILLIEYE_DETECTOR_PREFETCH=0 = the plain loop).
        # Same batches, same order, same results.
        batches = iter(dataloader)
        upcoming = next(batches, None)
        imgs_ahead, batch_i = None, -1
        while upcoming is not None:
            batch_i += 1
            (_, imgs, targets), upcoming = upcoming, next(batches, None)
            batches_done = len(dataloader) * epoch + batch_i
            epoch_batches_left = len(dataloader) - (batch_i + 1)
            imgs = imgs_ahead if imgs_ahead is not None else imgs.to(device)
            imgs_ahead = None
            if upcoming is not None and os.environ.get("MILLIEYE_DETECTOR_PREFETCH", "1") != "0" and hasattr(model, "queue_detector_prefetch"):
                imgs_ahead = upcoming[1].to(device)
                model.queue_detector_prefetch(imgs_ahead)
            targets.requires_grad = False

            output, loss, metric = model(imgs, targets)  # imgs on the device, targets on the host
            loss.backward()

            if batches_done % gradient_accumulations == 0:
                if distributed:
                    parallel.allreduce_gradients(trainable, static_pattern=True)
                optimizer.step()
                optimizer.zero_grad()
                history["steps"].append(batches_done)

            loss_value = loss.item()
            history["losses"].append(loss_value)
            time_left = datetime.timedelta(seconds=epoch_batches_left * (time.time() - start_time) / (batch_i + 1))
            log("--- [Epoch %d/%d, Batch %d/%d] ---\nTotal loss %s\n---- ETA %s\n"
                % (epoch, epochs, batch_i, len(dataloader), loss_value, time_left))
            if writer is not None:
                writer.add_scalar("loss", loss, global_step=batches_done)
                if metric["positive"]:
                    writer.add_scalar("precesion", metric["tp"] / metric["positive"], global_step=batches_done)
                if metric["true"]:
                    writer.add_scalar("recall", metric["tp"] / metric["true"], global_step=batches_done)
            model.seen += imgs.size(0)

        if distributed:
            # identical parameters on every rank after the summed-gradient steps; the fcn_layers BatchNorm's running statistics
            # are not (each rank normalises its own batches): average them once per epoch, like the stage-3 loop does
            from ..train import sync_batchnorm_buffers
            sync_batchnorm_buffers(model)
        evaluating = epoch % evaluation_interval == 0 and evaluate_fn is not None
        if evaluating and rank == 0:
            log("\n---- Evaluating Model ----")
            precision, recall, AP, f1, ap_class, _, _ = result = evaluate_fn(
                model, list_path=valid_path, iou_thresh=0.5, conf_thresh=0.01, nms_thresh=0.5, img_size=img_size,
                batch_size=batch_size, **(evaluate_kwargs or {}))
            history["evaluations"].append(result)
            if writer is not None:
                writer.add_scalars("metrics", {"val_precision": precision.mean(), "val_recall": recall.mean(),
                                               "val_mAP": AP.mean(), "val_f1": f1.mean(),
                                               "val_(f1+mAP)": f1.mean() + AP.mean()}, global_step=epoch)
            rows = [["Index", "Class name", "AP"]]
            for i, c in enumerate(ap_class):
                rows.append([c, class_names[i] if class_names else str(c), "%.5f" % AP[i]])
            log("\n".join(" | ".join(str(v) for v in r) for r in rows))
            log(f"---- mAP {AP.mean()}")
        if distributed and evaluating:
            torch.distributed.barrier()
        if epoch % checkpoint_interval == 0 and rank == 0:
            path = os.path.join(checkpoint_dir, "ckpt_%d.pth" % epoch)
            torch.save(model.state_dict(), path)
            history["checkpoints"].append(path)
    return history


def build_parser():
    p = argparse.ArgumentParser(description="stage-2 training (module2_mixed/train.py)")
    p.add_argument("--epochs", type=int, default=400)
    p.add_argument("--batch_size", type=int, default=24)
    p.add_argument("--gradient_accumulations", type=int, default=2)
    p.add_argument("--n_cpu", type=int, default=24)
    p.add_argument("--img_size", type=int, default=416)
    p.add_argument("--multiscale_training", type=bool, default=True)
    p.add_argument("--checkpoint_interval", type=int, default=1)
    p.add_argument("--evaluation_interval", type=int, default=1)
    p.add_argument("--conf_thresh", type=float, default=0.01)
    p.add_argument("--classes_path", type=str, default="config/exdark.names")
    p.add_argument("--yolo_cfg", type=str, default="config/yolov3-tiny-12.cfg")
    p.add_argument("--yolo_weights", type=str, default="weights/best_mixed.pt")
    p.add_argument("--checkpoint", type=str)
    p.add_argument("--data_config", type=str, default="config/mixed.data")
    return p


def main(argv=None):
    opt = build_parser().parse_args(argv)
    data_config = parse_data_config(opt.data_config)
    train_path, valid_path = data_config["train"], data_config["valid"]
    class_names = load_classes(opt.classes_path)
    # data parallel (new in this build, as in millieye_amd/train.py main()): under ``python -m torch.distributed.run
    # --nproc-per-node N -m millieye_amd.module2.train ...`` every process takes the GPU of its LOCAL_RANK and a 1 / N shard
    # of each epoch; only rank 0 evaluates and writes checkpoints, the others wait in the loop's barrier
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local),
                                             timeout=datetime.timedelta(hours=4))  # rank 0's evaluation outlasts 10 min
    model = Network(define_yolo(opt.yolo_cfg), opt.conf_thresh)
    model = model.to(model.device)
    if opt.checkpoint:
        model.load_state_dict(torch.load(opt.checkpoint, map_location=model.device))
    else:
        model.apply(weights_init_normal)
        init_yolo(model=model.base_detector, weights_path=opt.yolo_weights)
    if world > 1:  # the random initialisation drew from each process's own RNG: replicas start from rank 0's state
        with torch.no_grad():
            for t in model.state_dict().values():
                torch.distributed.broadcast(t, 0)
        model.base_detector.invalidate_weights()
    dataset = ListDataset(train_path, augment=True, multiscale=opt.multiscale_training)
    sampler = torch.utils.data.distributed.DistributedSampler(dataset, shuffle=True) if world > 1 else None
    dataloader = torch.utils.data.DataLoader(dataset, batch_size=opt.batch_size, shuffle=sampler is None, sampler=sampler,
                                             num_workers=opt.n_cpu, pin_memory=False, collate_fn=dataset.collate_fn)
    writer = None
    try:
        from torch.utils.tensorboard import SummaryWriter
        writer = SummaryWriter()
    except Exception:  # tensorboard is optional here
        pass
    try:
        return train_loop(model, dataloader, epochs=opt.epochs, gradient_accumulations=opt.gradient_accumulations,
                          checkpoint_interval=opt.checkpoint_interval, evaluation_interval=opt.evaluation_interval,
                          valid_path=valid_path, img_size=opt.img_size, batch_size=opt.batch_size, class_names=class_names,
                          writer=writer, evaluate_kwargs=dict(n_cpu=opt.n_cpu))
    finally:
        if writer is not None:
            writer.close()


if __name__ == "__main__":
    main()
