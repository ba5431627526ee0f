"""Shared by the CPU and the GPU train-loop tests: drive ``millieye_amd.train.train_loop`` on the stand-in
batches of ``tests/golden/make_golden.LOOP_CASE`` and compare with what the REAL reference ``train.py`` did
(``tests/golden/trainloop_tiny12_s160.npz``)."""
import os
import random

import numpy as np
import torch

from millieye_amd import synth
from millieye_amd.train import load_pretrained_module2, train_loop
from tests.golden.make_golden import LOOP_CASE, loop_batches, loop_module2_params
from tests.parity_helpers import cfg_path

GOLD = os.path.join(os.path.dirname(__file__), "golden")


class Batches:
    """Re-creates the batches on every pass (``evaluate`` mutates ``targets`` in place, like the reference)."""

    def __init__(self, which, count, targets):
        self.which, self.count, self.targets = which, count, targets

    def __len__(self):
        return self.count

    def __iter__(self):
        c = LOOP_CASE
        return iter(loop_batches(c["name"], self.which, self.count, c["batch"], c["size"], self.targets))


def golden():
    return np.load(os.path.join(GOLD, LOOP_CASE["name"] + ".npz"))


def prepare(net_cls):
    """Model in the state train.py reaches before its loop: checkpoint weights, stage-2 hand-over, freezing."""
    from millieye_amd.my_models import define_yolo
    c = LOOP_CASE
    net = net_cls(define_yolo(cfg_path(c["cfg"])), c["conf"])
    synth.fill_network_(net, c["name"])
    frozen = load_pretrained_module2(net, loop_module2_params(c["name"]), log=lambda *_: None)
    return net, frozen


def run(net, tmpdir, optimizer_cls=None):
    c, g = LOOP_CASE, golden()
    targets = {k[len("targets/"):]: g[k] for k in g.files if k.startswith("targets/")}
    step_sums = []
    # (optimizer_cls: millieye_amd.optim.Adam in the GPU tests - the loop's default optimizer there - against the same goldens)
    opt = (optimizer_cls or torch.optim.Adam)(net.parameters(), lr=5e-4)
    real_step = opt.step

    def step(*a, **kw):
        r = real_step(*a, **kw)
        step_sums.append([float(p.detach().double().sum()) for grp in opt.param_groups for p in grp["params"]])
        return r

    opt.step = step
    random.seed(c["seed"])
    torch.manual_seed(c["seed"])
    hist = train_loop(net, Batches("train", c["train_batches"], targets), epochs=c["epochs"],
                      gradient_accumulations=c["grad_accum"], test_list=c["test_list"], img_size=c["size"],
                      batch_size=c["batch"], class_names=["person"], optimizer=opt,
                      evaluate_kwargs=dict(dataloader=Batches("test", c["test_batches"], targets)),
                      checkpoint_dir=os.path.join(str(tmpdir), "checkpoints"), log=lambda *_: None)
    hist["step_sums"] = np.asarray(step_sums)
    return hist


def check(net, frozen, hist, tmpdir, loss_tol, param_atol, sum_tol, ap_tol, late_ap_tol=None):
    c, g = LOOP_CASE, golden()
    # cadence: a step after batches 0, 2, 4 (train.py:188); seen; checkpoint names; what got frozen
    assert hist["steps"] == [0, 2, 4]
    assert net.seen == int(g["seen"]) == c["epochs"] * c["train_batches"] * c["batch"]
    assert sorted(os.listdir(os.path.join(str(tmpdir), "checkpoints"))) == list(g["checkpoints"])
    assert frozen == list(g["frozen"])
    assert sum(1 for _ in net.parameters()) == int(g["optimizer_param_count"])
    losses = np.asarray(hist["losses"])
    assert losses.shape == g["losses"].shape
    assert np.all(np.abs(losses - g["losses"]) <= loss_tol * np.maximum(1.0, np.abs(g["losses"]))), (losses, g["losses"])
    # parameter sums after each optimizer step (same order as Adam's param group)
    names = [k for k, _ in net.named_parameters()]
    numel = np.asarray([p.numel() for p in net.parameters()], dtype=np.float64)
    assert hist["step_sums"].shape == g["step_sums"].shape
    per_elem = np.abs(hist["step_sums"] - g["step_sums"]) / numel  # mean drift per element, per step and tensor
    # A conv bias that feeds a train-mode BatchNorm has a mathematically zero gradient: what autograd / our backward
    # return for it is rounding noise, and Adam turns noise into +-lr steps.  Those tensors can only be bounded by
    # lr * steps; every other tensor must track the reference closely on average.
    noise = np.asarray([k.endswith(".bias") and ("radar_cnn_layers.conv" in k and k.split(".")[-2] == "0"
                                                 or k == "refinement_head.radar_net.0.bias"
                                                 or k == "img_cnn_layers.net.conv_0.bias") for k in names])
    steps = np.arange(1, per_elem.shape[0] + 1)[:, None]
    bound = np.where(noise[None, :], 2 * 5e-4 * steps, sum_tol)
    bad = np.argwhere(per_elem > bound + 1e-4 / numel)
    assert bad.size == 0, [(names[j], int(i), float(per_elem[i, j])) for i, j in bad]
    # the last checkpoint, tensor by tensor
    final = torch.load(os.path.join(str(tmpdir), "checkpoints", f"{c['test_list']}_ckpt_{c['epochs'] - 1}.pth"),
                       map_location="cpu")
    n_checked = 0
    for k in g.files:
        if k.startswith("final/"):
            got, ref = final[k[6:]].numpy(), g[k]
            assert got.shape == ref.shape, k
            assert np.all(np.abs(got.astype(np.float64) - ref) <= param_atol + 1e-3 * np.abs(ref)), \
                (k, float(np.max(np.abs(got - ref))))
            n_checked += 1
    assert n_checked >= 40
    # evaluate() after each epoch
    assert len(hist["evaluations"]) == c["epochs"]
    for e, (precision, recall, AP, f1, ap_class, box_stat, _pr) in enumerate(hist["evaluations"]):
        assert list(ap_class) == list(g[f"eval{e}/ap_class"])
        assert list(box_stat["after"]) == list(g[f"eval{e}/after"])
        for name, got in (("precision", precision), ("recall", recall), ("AP", AP), ("f1", f1)):
            tol = late_ap_tol if (name == "AP" and e > 0 and late_ap_tol is not None) else ap_tol
            assert np.allclose(got, g[f"eval{e}/{name}"], rtol=0, atol=tol), (e, name, got, g[f"eval{e}/{name}"])
