"""Shared by the CPU and the GPU stage-2 loop tests: drive ``millieye_amd.module2.train.train_loop`` (+ its
``test_module2.evaluate``) on the stand-in batches of ``tests/golden/make_golden.M2_LOOP_CASE`` and compare with what the REAL
``module2_mixed/train.py`` did (``tests/golden/trainloop_m2_tiny12_s160.npz``)."""
import os
import random

import numpy as np
import torch

from millieye_amd import cfgs
from millieye_amd.module2.my_models import Network
from millieye_amd.module2.train import train_loop
from tests.golden.make_golden import M2_LOOP_CASE, m2_loop_batches, m2_train_fill_
from tests.parity_helpers import cfg_path

GOLD = os.path.join(os.path.dirname(__file__), "golden")


class Batches:
    """Re-creates the batches on every pass (``evaluate`` mutates ``targets`` in place, like the reference)."""

    def __init__(self, which, count, targets):
        self.which, self.count, self.targets = which, count, targets

    def __len__(self):
        return self.count

    def __iter__(self):
        c = M2_LOOP_CASE
        return iter(m2_loop_batches(c["name"], self.which, self.count, c["batch"], c["size"], self.targets))


def golden():
    return np.load(os.path.join(GOLD, M2_LOOP_CASE["name"] + ".npz"))


class _Inject(torch.autograd.Function):
    @staticmethod
    def forward(ctx, loss, grads, *params):
        ctx.grads = grads
        return loss.clone()

    @staticmethod
    def backward(ctx, grad_out):
        return (None, None) + tuple(None if g is None else g * grad_out for g in ctx.grads)


class OracleBackedM2Network(Network):
    """Same parameter container / attributes as the product, forward = oracle/network_m2_ref.py on the CPU."""

    def __init__(self, det, conf):
        super().__init__(det, conf)
        self.device = torch.device("cpu")
        self.cfg_text = cfgs.KNOWN[M2_LOOP_CASE["cfg"]]()

    def forward(self, images, targets=None):
        from oracle import network_m2_ref
        sd = self.state_dict()
        if targets is None:
            return network_m2_ref.network_m2_forward(self.cfg_text, sd, images, conf_thresh=self.conf_thresh)
        res = network_m2_ref.network_m2_train_step(self.cfg_text, sd, images, targets, conf_thresh=self.conf_thresh)
        own = dict(self.named_buffers())
        with torch.no_grad():
            for k, v in res["buffers"].items():
                own[k].copy_(v)
            for k, v in own.items():
                if k.endswith("num_batches_tracked") and not k.startswith("base_detector."):
                    v += 1
        named = [(k, p) for k, p in self.named_parameters() if not k.startswith("base_detector.")]
        grads = [res["grads"].get(k) for k, _ in named]
        loss = _Inject.apply(res["loss"], grads, *[p for _, p in named])
        return res["output"], loss, dict(true=res["n_pos"], total=len(res["masks"]), tp=0, positive=0)


def prepare(net_cls):
    """Model in the state train.py reaches before its loop: the checkpoint's weights."""
    from millieye_amd.module2.my_models import define_yolo
    c = M2_LOOP_CASE
    net = net_cls(define_yolo(cfg_path(c["cfg"])), c["conf"])
    m2_train_fill_(net, c["name"])
    return net


def run(net, tmpdir, optimizer_cls=None):
    c, g = M2_LOOP_CASE, golden()
    targets = {k[len("targets/"):]: g[k] for k in g.files if k.startswith("targets/")}
    step_sums = []
    # (optimizer_cls: millieye_amd.optim.AdamW in the GPU tests - the loop's default optimizer there - against the same goldens)
    opt = (optimizer_cls or torch.optim.AdamW)(net.parameters(), lr=1e-4)
    real_step = opt.step

    def step(*a, **kw):
        r = real_step(*a, **kw)
        step_sums.append([float(p.detach().double().sum()) for grp in opt.param_groups for p in grp["params"]])
        return r

    opt.step = step
    random.seed(c["seed"])
    torch.manual_seed(c["seed"])
    hist = train_loop(net, Batches("train", c["train_batches"], targets), epochs=c["epochs"],
                      gradient_accumulations=c["grad_accum"], valid_path="valid_list.txt", img_size=c["size"],
                      batch_size=c["batch"], class_names=[f"c{i}" for i in range(12)], optimizer=opt,
                      evaluate_kwargs=dict(dataloader=Batches("test", c["test_batches"], targets)),
                      checkpoint_dir=os.path.join(str(tmpdir), "checkpoints"), log=lambda *_: None)
    hist["step_sums"] = np.asarray(step_sums)
    return hist


def check(net, hist, tmpdir, loss_tol, param_atol, sum_tol, ap_tol, late_ap_tol=None, lr=1e-4):
    c, g = M2_LOOP_CASE, golden()
    assert hist["steps"] == [0, 2, 4]  # train.py:143: a step after batches 0, 2, 4
    assert net.seen == int(g["seen"]) == c["epochs"] * c["train_batches"] * c["batch"]
    assert sorted(os.listdir(os.path.join(str(tmpdir), "checkpoints"))) == list(g["checkpoints"])
    names = [k for k, _ in net.named_parameters()]
    assert names == list(g["step_names"])
    losses = np.asarray(hist["losses"])
    assert losses.shape == g["losses"].shape
    assert np.all(np.abs(losses - g["losses"]) <= loss_tol * np.maximum(1.0, np.abs(g["losses"]))), (losses, g["losses"])
    # parameter sums after each AdamW step, mean drift per element; a conv bias in front of a train-mode BatchNorm has a
    # mathematically zero gradient: rounding noise there becomes +-lr steps (see tests/train_loop_helpers.py)
    numel = np.asarray([p.numel() for p in net.parameters()], dtype=np.float64)
    assert hist["step_sums"].shape == g["step_sums"].shape
    per_elem = np.abs(hist["step_sums"] - g["step_sums"]) / numel
    noise = np.asarray([k == "fcn_layers.net.conv_0.bias" for k in names])
    steps = np.arange(1, per_elem.shape[0] + 1)[:, None]
    bound = np.where(noise[None, :], 2 * lr * steps, sum_tol)
    bad = np.argwhere(per_elem > bound + 1e-4 / numel)
    assert bad.size == 0, [(names[j], int(i), float(per_elem[i, j])) for i, j in bad]
    # the detector never moves (no gradient reaches it; AdamW skips parameters without one)
    det = [j for j, k in enumerate(names) if k.startswith("base_detector.")]
    assert np.array_equal(hist["step_sums"][0, det], hist["step_sums"][-1, det])
    final = torch.load(os.path.join(str(tmpdir), "checkpoints", f"ckpt_{c['epochs'] - 1}.pth"), map_location="cpu")
    n_checked = 0
    for k in g.files:
        if k.startswith("final/"):
            got, ref = final[k[6:]].numpy(), g[k]
            assert got.shape == ref.shape, k
            assert np.all(np.abs(got.astype(np.float64) - ref) <= param_atol + 1e-3 * np.abs(ref)), \
                (k, float(np.max(np.abs(got - ref))))
            n_checked += 1
    assert n_checked >= 15
    assert len(hist["evaluations"]) == c["epochs"]
    for e, (precision, recall, AP, f1, ap_class, box_stat, pr) in enumerate(hist["evaluations"]):
        assert list(ap_class) == list(g[f"eval{e}/ap_class"])
        assert list(box_stat["after"]) == list(g[f"eval{e}/after"])
        assert len(pr) == 3 and len(pr[2]) == len(g[f"eval{e}/pr_conf"])
        for name, got in (("precision", precision), ("recall", recall), ("AP", AP), ("f1", f1)):
            tol = late_ap_tol if (e > 0 and late_ap_tol is not None) else ap_tol
            assert np.allclose(got, g[f"eval{e}/{name}"], rtol=0, atol=tol), (e, name, got, g[f"eval{e}/{name}"])
