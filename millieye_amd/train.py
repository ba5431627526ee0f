"""Stage-3 training harness: the loop of ``module3_our_dataset/train.py``.

The reference's loop lives under ``if __name__ == "__main__"`` (train.py:24-272); here it is a function
so that a caller (or :func:`main`, which mirrors the script's command line and feeds it from
``millieye_amd/utils/datasets.MyDataset``) can drive it with any iterable of
``(paths, imgs, targets, radar_boxes, radar_maps)`` batches.  What is kept, line by line:

* ``load_pretrained_module2`` - the stage-2 -> stage-3 tensor hand-over by *position* (train.py:113-141): the
  tensors of the stage-2 checkpoint whose names are in ``NAMES_M2`` are collected in the checkpoint's own order
  and assigned, in the model's ``state_dict`` order, to the names of ``NAMES_M3``; those parameters are then
  frozen (``requires_grad = False``, train.py:143-147).
* ``Adam(model.parameters(), lr=5e-4)`` created *after* freezing (train.py:161).
* per epoch ``model.train(); model.base_detector.eval()`` (train.py:167-168); per batch
  ``model(imgs, radar_maps, radar_boxes, targets.clone())`` - targets in the ``model_mode`` slot -
  ``loss.backward()``; ``optimizer.step(); optimizer.zero_grad()`` when ``batches_done % gradient_accumulations == 0``
  - so the very first step sees one batch of gradients, later ones two (train.py:185-191);
  ``model.seen += imgs.size(0)`` (train.py:236).
* ``checkpoints/{test_list}_ckpt_{epoch}.pth`` every ``checkpoint_interval`` epochs (train.py:238-239), then
  ``evaluate(model, mode="test", model_mode=0, illumination=["L"], iou_thresh=0.5, nms_thresh=0.5, ...)`` every
  ``evaluation_interval`` epochs (train.py:241-254).  ``evaluate`` leaves the model in ``eval()``; the next epoch
  switches back, as in the reference.

Added for MI355X: when ``torch.distributed`` is initialised every rank runs the loop on its shard of the batches
and the accumulated gradients are SUM-all-reduced in one flat bucket right before ``optimizer.step()``
(``millieye_amd/parallel.py``; losses are sums over RoIs, so SUM is the single-process gradient of the global batch).
TensorBoard image dumps (train.py:196-219) are optional: pass a ``SummaryWriter``-like ``writer``.
"""
import argparse
import datetime
import os
import time

import torch

from . import parallel
from .test_fusion import evaluate as _evaluate

NAMES_M3 = ["img_cnn_layers.net.conv_0.weight", "img_cnn_layers.net.conv_0.bias",
            "img_cnn_layers.net.batch_norm_0.weight", "img_cnn_layers.net.batch_norm_0.bias",
            "img_cnn_layers.net.batch_norm_0.running_mean", "img_cnn_layers.net.batch_norm_0.running_var",
            "img_cnn_layers.net.batch_norm_0.num_batches_tracked",
            "refinement_head.net0.0.weight", "refinement_head.net0.0.bias",
            "refinement_head.net1.0.weight", "refinement_head.net1.0.bias",
            "refinement_head.net2.0.weight", "refinement_head.net2.0.bias"]
NAMES_M2 = [n.replace("img_cnn_layers.", "fcn_layers.") for n in NAMES_M3]


def load_pretrained_module2(model, param, log=print):
    """train.py:113-147.  ``param`` is the stage-2 ``state_dict``.  Returns the names that were frozen."""
    module_list = model.state_dict()
    tmp = [param[name] for name in list(param) if name in NAMES_M2]
    for name in module_list:
        if name in NAMES_M3:
            module_list[name] = tmp.pop(0)
    model.load_state_dict(module_list)
    frozen = []
    for name, p in model.named_parameters():
        if name in NAMES_M3:
            log(name)
            p.requires_grad = False
            frozen.append(name)
    return frozen


def train_loop(model, dataloader, *, epochs, gradient_accumulations=2, checkpoint_interval=1, evaluation_interval=1,
               test_list=4, img_size=416, batch_size=16, class_names=None, optimizer=None, evaluate_fn=_evaluate,
               evaluate_kwargs=None, checkpoint_dir="checkpoints", writer=None, log=print):
    """Runs the loop; returns ``dict(losses, steps, checkpoints, evaluations)`` (the reference only prints)."""
    device = getattr(model, "device", torch.device("cuda"))
    os.makedirs(checkpoint_dir, exist_ok=True)
    if optimizer is None:
        # the reference's Adam(lr=5e-4) (train.py:161): on the GPU the one-launch step of millieye_amd/optim.py (same state, same
        # update rule element by element, checkpoints interchangeable with torch.optim.Adam); a model that is not on the GPU (the
        # host-logic tests) gets the torch class itself
        params = list(model.parameters())
        if params and all(p.is_cuda and p.dtype == torch.float32 for p in params):
            from .optim import Adam
            optimizer = Adam(params, lr=5e-4)
        else:
            optimizer = torch.optim.Adam(params, lr=5e-4)
    distributed = torch.distributed.is_available() and torch.distributed.is_initialized()
    rank = torch.distributed.get_rank() if distributed else 0
    history = dict(losses=[], steps=[], checkpoints=[], evaluations=[])
    trainable = [p for p in model.parameters() if p.requires_grad]

    for epoch in range(epochs):
        model.train()
        model.base_detector.eval()
        start_time = time.time()
        parallel.begin_epoch(dataloader, epoch, device)  # equal batch counts on every rank + the sampler's epoch
        # one batch of look-ahead: the frozen detector's part of batch k + 1 runs under the host-bound tail of batch k
        # (Network.queue_detector_prefetch; MILLIEYE_DETECTOR_PREFETCH=0 = the plain loop).  Same batches, same order, same results.
        batches = iter(dataloader)
        upcoming = next(batches, None)
        imgs_ahead, batch_i = None, -1
        while upcoming is not None:
            batch_i += 1
            (_, imgs, targets, radar_boxes, radar_maps), upcoming = upcoming, next(batches, None)
            batches_done = len(dataloader) * epoch + batch_i
            epoch_batches_left = len(dataloader) - (batch_i + 1)
            imgs = imgs_ahead if imgs_ahead is not None else imgs.to(device)
            imgs_ahead = None
            if upcoming is not None and os.environ.get("MILLIEYE_DETECTOR_PREFETCH", "1") != "0" and hasattr(model, "queue_detector_prefetch"):
                imgs_ahead = upcoming[1].to(device)
                model.queue_detector_prefetch(imgs_ahead)
            radar_maps = radar_maps.to(device)
            radar_boxes = radar_boxes.to(device)

            loss, outputs, metric, radar_attention = model(imgs, radar_maps, radar_boxes, targets.clone())
            loss.backward()

            if batches_done % gradient_accumulations == 0:
                if distributed:
                    parallel.allreduce_gradients(trainable, static_pattern=True)
                optimizer.step()
                optimizer.zero_grad()
                history["steps"].append(batches_done)

            if writer is not None:
                if batches_done % 50 == 0:
                    writer.add_images("image", imgs, global_step=batches_done)
                    writer.add_images("radar_attention", radar_attention[:, :3], global_step=batches_done)
                writer.add_scalar("loss", loss, global_step=batches_done)
                writer.add_scalar("precesion", metric["tp"] / metric["positive"], global_step=batches_done)
                writer.add_scalar("recall", metric["tp"] / metric["true"], global_step=batches_done)

            loss_value = loss.item()
            history["losses"].append(loss_value)
            time_left = datetime.timedelta(seconds=epoch_batches_left * (time.time() - start_time) / (batch_i + 1))
            log("--- [Epoch %d/%d, Batch %d/%d] ---\nTotal loss: %s\n---- ETA %s\n"
                % (epoch, epochs, batch_i, len(dataloader), loss_value, time_left))
            model.seen += imgs.size(0)

        if distributed:
            # parameters are identical on every rank after the summed-gradient step; the head BatchNorms' running statistics
            # are not (each rank normalises its own shard): average them once per epoch so that the replicas - and what
            # rank 0 checkpoints / evaluates - do not drift apart
            sync_batchnorm_buffers(model)
        if epoch % checkpoint_interval == 0 and rank == 0:
            path = os.path.join(checkpoint_dir, f"{test_list}_ckpt_{epoch}.pth")
            torch.save(model.state_dict(), path)
            history["checkpoints"].append(path)

        evaluating = epoch % evaluation_interval == 0 and evaluate_fn is not None
        if evaluating and rank == 0:
            log("\n---- Evaluating Model ----")
            precision, recall, AP, f1, ap_class, _, _ = result = evaluate_fn(
                model, mode="test", model_mode=0, illumination=["L"], iou_thresh=0.5, nms_thresh=0.5,
                img_size=img_size, batch_size=batch_size, test_list=test_list, **(evaluate_kwargs or {}))
            history["evaluations"].append(result)
            if writer is not None:
                writer.add_scalars("metrics", dict(val_precision=precision.mean(), val_recall=recall.mean(),
                                                   val_mAP=AP.mean(), val_f1=f1.mean()), global_step=epoch)
            rows = [["Index", "Class name", "AP"]]
            for i, c in enumerate(ap_class):
                rows.append([c, class_names[i] if class_names else str(c), "%.5f" % AP[i]])
            log("\n".join(" | ".join(str(v) for v in r) for r in rows))
            log(f"---- mAP {AP.mean()}")
        if distributed and evaluating:
            # rank 0 evaluated, the others wait HERE - behind the evaluation, not in the next epoch's first collective
            # (main() raises the process-group timeout above the default 10 min for exactly this wait)
            torch.distributed.barrier()
    return history


def sync_batchnorm_buffers(model):
    """Data-parallel runs: replace every non-detector BatchNorm running statistic by its mean over the ranks (one small
    all-reduce; ``num_batches_tracked`` is the same everywhere).  The packed eval-mode copies are re-folded at the next
    forward because the version counters move."""
    world = torch.distributed.get_world_size()
    bufs = [b for name, b in model.named_buffers()
            if not name.startswith("base_detector.") and b.is_floating_point() and "running_" in name]
    if not bufs or world == 1:
        return 0
    flat = torch.cat([b.reshape(-1).float() for b in bufs])
    if torch.distributed.get_backend() != "nccl":
        flat = flat.cpu()
    torch.distributed.all_reduce(flat, op=torch.distributed.ReduceOp.SUM)
    flat = (flat / world).to(bufs[0].device)
    off = 0
    with torch.no_grad():
        for b in bufs:
            b.copy_(flat[off:off + b.numel()].view_as(b))
            off += b.numel()
    return off


def build_parser():
    """The reference's command line (train.py:26-94), same names and defaults."""
    p = argparse.ArgumentParser()
    p.add_argument("--epochs", type=int, default=100)
    p.add_argument("--batch_size", type=int, default=16)
    p.add_argument("--gradient_accumulations", type=int, default=2)
    p.add_argument("--n_cpu", type=int, default=16)
    p.add_argument("--checkpoint_interval", type=int, default=1)
    p.add_argument("--evaluation_interval", type=int, default=1)
    p.add_argument("--multiscale_training", default=True)
    p.add_argument("--conf_thresh", type=float, default=0.01)
    p.add_argument("--img_size", type=int, default=416)
    p.add_argument("--classes_path", type=str, default="config/exdark.names")
    p.add_argument("--yolo_cfg", type=str, default="config/yolov3-tiny-12.cfg")
    p.add_argument("--yolo_weights", type=str, default="weights/best_mixed.pt")
    p.add_argument("--pretrained_module2", type=str, default="./weights/module2_best_mixed.pth")
    p.add_argument("--checkpoint", type=str)
    p.add_argument("--illumination", type=str, default=["H", "L"])
    p.add_argument("--test_list", type=int, default=4)
    return p


def main(argv=None):
    """``python -m millieye_amd.train`` - the reference script's command line (train.py:24-165)."""
    from .my_models import Network, define_yolo, init_yolo
    from .utils.datasets import MyDataset
    from .utils.utils import load_classes, weights_init_normal

    opt = build_parser().parse_args(argv)
    class_names = load_classes(opt.classes_path)
    # data parallel (new in this build): started under ``python -m torch.distributed.run --nproc-per-node N
    # -m millieye_amd.train ...`` every process takes the GPU of its LOCAL_RANK and a 1/N shard of each epoch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        # rank 0 evaluates alone at the end of an epoch while the others wait in a barrier: well past the default 10 min
        torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local),
                                             timeout=datetime.timedelta(hours=4))
    model = Network(define_yolo(opt.yolo_cfg), opt.conf_thresh)
    model = model.to(model.device)
    if opt.checkpoint:
        model.load_state_dict(torch.load(opt.checkpoint))
    else:
        model.apply(weights_init_normal)
        init_yolo(model=model.base_detector, weights_path=opt.yolo_weights)
    if opt.pretrained_module2:
        load_pretrained_module2(model, torch.load(opt.pretrained_module2))
    if world > 1:  # the random initialisation above drew from each process's own RNG: replicas start from rank 0's state
        with torch.no_grad():
            for t in model.state_dict().values():
                torch.distributed.broadcast(t, 0)
        model.base_detector.invalidate_weights()
    dataset = MyDataset(mode="train", illumination=opt.illumination, augment=False, multiscale=True)
    sampler = torch.utils.data.distributed.DistributedSampler(dataset, shuffle=True) if world > 1 else None
    dataloader = torch.utils.data.DataLoader(dataset, batch_size=opt.batch_size, shuffle=sampler is None, sampler=sampler,
                                             num_workers=opt.n_cpu, pin_memory=True, collate_fn=dataset.collate_fn)
    train_loop(model, dataloader, epochs=opt.epochs, gradient_accumulations=opt.gradient_accumulations,
               checkpoint_interval=opt.checkpoint_interval, evaluation_interval=opt.evaluation_interval,
               test_list=opt.test_list, img_size=opt.img_size, batch_size=opt.batch_size, class_names=class_names)


if __name__ == "__main__":
    main()
