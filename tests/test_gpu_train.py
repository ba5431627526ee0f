"""-m gpu: the stage-3 training step.  Per-kernel parity of the training building blocks against torch
CPU autograd / the oracle's torchvision backward, then the whole step (loss, every parameter gradient,
BatchNorm running statistics, optimizer step) against oracle/network_ref.network_train_step and the
reference's own golden numbers (tests/golden/train_*.npz).  Tolerance 1e-3 relative to the tensor's scale."""
import os
import random

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from millieye_amd import cfgs, synth
from tests import parity_helpers as ph
from tests.golden.make_golden import TRAIN_CASE, train_inputs

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _t(tag, shape, lo=-1.0, hi=1.0):
    return torch.from_numpy(synth.uniform(tag, shape, lo, hi))


def _rel(got, ref):
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    return float((got - ref).abs().max() / max(1e-12, float(ref.abs().max())))


@pytest.mark.parametrize("ta,tb", [(0, 0), (1, 0), (0, 1), (1, 1)])
def test_gemm_and_colsum(hip_lib, ta, tb):
    from millieye_amd import hip, train_path as tp
    m, n, k = 70, 130, 45
    a = _t("ga", (k, m) if ta else (m, k))
    b = _t("gb", (n, k) if tb else (k, n))
    ref = (a.t() if ta else a) @ (b.t() if tb else b)
    c = torch.full((m, n), 7.0).cuda()
    tp._gemm(ta, tb, m, n, k, a.cuda(), a.shape[1], b.cuda(), b.shape[1], c, n)
    assert _rel(c, ref) < 1e-5
    tp._gemm(ta, tb, m, n, k, a.cuda(), a.shape[1], b.cuda(), b.shape[1], c, n, alpha=0.5, beta=2.0)
    assert _rel(c, 2.5 * ref) < 1e-5
    out = torch.empty(n).cuda()
    tp._colsum(c, n, m, n, out)
    assert _rel(out, (2.5 * ref).sum(0)) < 1e-5


def test_bn_train_forward_backward(hip_lib):
    from millieye_amd import hip, train_path as tp
    rows, c = 5408, 70
    x = _t("bnx", (rows, c), -2, 3)
    bn = torch.nn.BatchNorm2d(c, momentum=0.1)
    with torch.no_grad():
        bn.weight.copy_(_t("bng", (c,), 0.5, 1.5))
        bn.bias.copy_(_t("bnb", (c,), -0.5, 0.5))
        bn.running_mean.copy_(_t("bnm", (c,)))
        bn.running_var.copy_(_t("bnv", (c,), 0.5, 1.5))
    ref_bn = torch.nn.BatchNorm2d(c, momentum=0.1)
    ref_bn.load_state_dict(bn.state_dict())
    xr = x.clone().requires_grad_(True)
    yr = F.leaky_relu(ref_bn(xr.t().reshape(1, c, rows, 1)), 0.1)
    dy = _t("bndy", (rows, c))
    yr.backward(dy.t().reshape(1, c, rows, 1))
    bn = bn.cuda()
    ws_t = torch.empty(int(hip.lib().me_bn_workspace_bytes(c)) + 256, dtype=torch.uint8, device="cuda")
    ws = ws_t.data_ptr() + (-ws_t.data_ptr()) % 256
    xd, y = x.cuda(), torch.empty((rows, c)).cuda()
    st = tp._bn_fwd(xd, c, rows, c, bn, hip.ACT_LEAKY, y, c, ws)
    assert _rel(y, yr.reshape(c, rows).t()) < 1e-4
    assert _rel(bn.running_mean, ref_bn.running_mean) < 1e-5 and _rel(bn.running_var, ref_bn.running_var) < 1e-5
    assert int(bn.num_batches_tracked) == 1
    dx = torch.empty((rows, c)).cuda()
    dg, db = tp._bn_bwd(xd, c, dy.cuda(), c, rows, c, bn, st, hip.ACT_LEAKY, dx, c, ws)
    assert _rel(dx, xr.grad) < 1e-3 and _rel(dg, ref_bn.weight.grad) < 1e-3 and _rel(db, ref_bn.bias.grad) < 1e-3


@pytest.mark.parametrize("cin,cout,k", [(3, 32, 3), (32, 64, 3), (64, 128, 3), (128, 10, 1)])
def test_conv_wgrad_and_dgrad(hip_lib, cin, cout, k):
    from millieye_amd import hip, train_path as tp
    n, h, w = 2, 10, 12
    pad = (k - 1) // 2
    x = _t(f"wx{cin}", (n, cin, h, w)).requires_grad_(True)
    wt = _t(f"ww{cin}", (cout, cin, k, k), -0.2, 0.2).requires_grad_(True)
    y = F.conv2d(x, wt, None, 1, pad)
    dy = _t(f"wdy{cin}", tuple(y.shape))
    y.backward(dy)
    xd = x.detach().permute(0, 2, 3, 1).contiguous().cuda()
    dyd = dy.permute(0, 2, 3, 1).contiguous().cuda()
    dw = tp._wgrad(xd, cin, dyd, cout, n, h, w, cin, cout, k, pad)
    assert _rel(dw, wt.grad) < 1e-4
    if cout % 4 == 0 and cin > 4:  # data gradient through the forward kernel on rotated / transposed weights
        wd = wt.detach().flip(2, 3).permute(1, 2, 3, 0).contiguous().cuda()
        dx = torch.empty((n, h, w, cin)).cuda()
        tp._conv(dyd, cout, n, h, w, cout, wd, torch.ones(cin).cuda(), torch.zeros(cin).cuda(), k, pad, hip.ACT_LINEAR, dx)
        assert _rel(dx.permute(0, 3, 1, 2), x.grad) < 1e-4


@pytest.mark.parametrize("n,h,w,cin,cout,k,s", [(2, 10, 12, 32, 64, 3, 1), (3, 13, 13, 64, 255, 1, 1), (2, 16, 16, 3, 32, 3, 1),
                                                (2, 20, 20, 32, 64, 3, 2), (1, 7, 7, 10, 10, 7, 1), (4, 26, 26, 128, 256, 3, 1),
                                                (1, 5, 9, 48, 20, 3, 1), (3, 33, 41, 3, 32, 3, 2), (2, 24, 24, 3, 72, 3, 1),
                                                (1, 9, 9, 3, 30, 3, 1), (6, 53, 53, 128, 160, 3, 1), (8, 48, 48, 132, 128, 3, 1),
                                                # nine-tap kernel (csrc/wgrad9.hip: 3x3 / stride 1, cin and cout multiples of 64):
                                                # 13 x 13 (two images per slice), one tiny image (a slice shorter than its
                                                # halo), a wide map (ring of 16 slots), rectangular, several slices
                                                (5, 13, 13, 64, 64, 3, 1), (1, 4, 5, 64, 128, 3, 1), (2, 104, 104, 64, 64, 3, 1),
                                                (3, 20, 37, 128, 64, 3, 1), (8, 52, 52, 64, 192, 3, 1)])
def test_conv_wgrad_mfma(hip_lib, n, h, w, cin, cout, k, s):
    """me_conv_wgrad_mfma_f32 (matrix-pipe weight gradient, sliced pixel reduction + ordered slab sum) against torch
    CPU autograd: stride 2, ragged / unaligned channel counts (cin 3, cout 255), 1x1 and 7x7 (pad 0), many slices; the last
    two shapes (>= 16384 output pixels, both channel counts >= 128) run the 128x128-tile kernel, with ragged second tiles."""
    from millieye_amd import hip
    pad = 0 if k == 7 else (k - 1) // 2
    tag = f"wm{n}{h}{cin}{cout}{k}{s}"
    x = _t(tag + "x", (n, cin, h, w))
    wt = _t(tag + "w", (cout, cin, k, k), -0.2, 0.2).requires_grad_(True)
    y = F.conv2d(x, wt, None, s, pad)
    dy = _t(tag + "dy", tuple(y.shape))
    y.backward(dy)
    xd = x.permute(0, 2, 3, 1).contiguous().cuda()
    dyd = dy.permute(0, 2, 3, 1).contiguous().cuda()
    dw = hip.conv_wgrad(xd, dyd, k, s, pad)
    dw2 = hip.conv_wgrad(xd, dyd, k, s, pad)
    assert torch.equal(dw, dw2), "the sliced reduction must be deterministic"
    assert _rel(dw.permute(0, 3, 1, 2), wt.grad) < 1e-4
    # the parameter's own layout (the slab reduction transposes; cin = 3 with cout % 4 == 0 is the stem kernel)
    assert torch.equal(hip.conv_wgrad(xd, dyd, k, s, pad, oihw=True), dw.permute(0, 3, 1, 2).contiguous())


def test_roi_backward_vs_oracle(hip_lib):
    from millieye_amd import hip
    from oracle import tv_ops
    n, h, w, k = 2, 10, 10, 24
    size = 160.0
    c = synth.uniform("rbc", (k, 2), 0.1 * size, 0.9 * size)
    half = synth.uniform("rbh", (k, 2), 3.0, 0.4 * size)
    idx = np.floor(synth.uniform("rbi", (k, 1), 0, n))
    rois = torch.from_numpy(np.concatenate([idx, c - half, c + half], 1).astype(np.float32))
    for ps, ch in ((False, 10), (True, 490)):
        m = _t(f"rbm{ch}", (n, ch, h, w)).requires_grad_(True)
        out = (tv_ops.ps_roi_align if ps else tv_ops.roi_align)(m, rois, (7, 7), 1 / 16)
        g = _t(f"rbg{ch}", tuple(out.shape))
        out.backward(g)
        gmap = torch.zeros((n, h, w, ch)).cuda()
        gd, rd = g.cuda(), rois.cuda()  # keep the device copies alive across the asynchronous launch
        fn = hip.lib().me_ps_roi_align_bwd_f32 if ps else hip.lib().me_roi_align_bwd_f32
        hip.check(fn(gd.data_ptr(), rd.data_ptr(), k, n, h, w, ch, 7, 1.0 / 16, gmap.data_ptr(), ch,
                     hip.stream_ptr()), "roi bwd")
        torch.cuda.synchronize()
        assert _rel(gmap.permute(0, 3, 1, 2), m.grad) < 1e-4



@pytest.mark.parametrize("n,h,k,lds", [(8, 26, 2500, True), (1, 26, 300, True), (2, 40, 60, False)])
def test_roi_backward_through_lds_on_clustered_proposals(hip_lib, n, h, k, lds):
    """The stage-3 shape of the scatter: proposals of a frame piled on a few objects.  Maps whose per-frame slice fits LDS take
    roi_bwd_lds_kernel (one workgroup per bin and frame / per RoI slice and frame; 2500 RoIs = two rounds of its RoI list), larger
    ones the global-atomics kernel; both against oracle/tv_ops.c's backward, and the device-count entry points with a capacity
    above the live count against the host-count ones."""
    from millieye_amd import hip
    from oracle import tv_ops
    size = 16.0 * h
    g = np.random.RandomState(7)
    centres = g.uniform(0.15 * size, 0.85 * size, size=(n, 3, 2))
    rois = np.zeros((k, 5), dtype=np.float32)
    for i in range(k):
        b = i * n // k
        c = centres[b, g.randint(3)] + g.uniform(-6, 6, size=2)
        wh = g.uniform(40, 0.3 * size, size=2)
        rois[i] = [b, c[0] - wh[0] / 2, c[1] - wh[1] / 2, c[0] + wh[0] / 2, c[1] + wh[1] / 2]
    rois = torch.from_numpy(rois)
    assert ((h * h * 10 * 4) <= 48 * 1024) == lds
    for ps, ch in ((False, 10), (True, 490)):
        m = torch.zeros((n, ch, h, h), requires_grad=True)
        out = (tv_ops.ps_roi_align if ps else tv_ops.roi_align)(m, rois, (7, 7), 1 / 16)
        go = _t(f"rlg{ch}{k}", tuple(out.shape))
        out.backward(go)
        gd, rd = go.cuda(), rois.cuda()
        gmap = torch.zeros((n, h, h, ch)).cuda()
        fn = hip.lib().me_ps_roi_align_bwd_f32 if ps else hip.lib().me_roi_align_bwd_f32
        hip.check(fn(gd.data_ptr(), rd.data_ptr(), k, n, h, h, ch, 7, 1.0 / 16, gmap.data_ptr(), ch, hip.stream_ptr()), "roi bwd")
        torch.cuda.synchronize()
        assert _rel(gmap.permute(0, 3, 1, 2), m.grad) < 1e-4, (ps, _rel(gmap.permute(0, 3, 1, 2), m.grad))
        # capacity above the live count: rows behind the count hold garbage that must not be scattered
        cap = k + 100
        gd2 = torch.cat([gd, torch.full((100,) + tuple(gd.shape[1:]), 1e6, device="cuda")])
        rd2 = torch.cat([rd, rd[:100]])
        kd = torch.tensor([k], dtype=torch.int32, device="cuda")
        gmap2 = torch.zeros_like(gmap)
        fn2 = hip.lib().me_ps_roi_align_bwd_dev_f32 if ps else hip.lib().me_roi_align_bwd_dev_f32
        hip.check(fn2(gd2.data_ptr(), rd2.data_ptr(), cap, kd.data_ptr(), n, h, h, ch, 7, 1.0 / 16, gmap2.data_ptr(), ch,
                      hip.stream_ptr()), "roi bwd dev")
        torch.cuda.synchronize()
        assert _rel(gmap2, gmap) < 1e-5



def test_roi_backward_on_degenerate_boxes(hip_lib):
    """The boxes the reference's own edge cases produce (proposals on the border, empty and inverted boxes, boxes far
    outside the map, boxes much larger than it) through both scatter kernels against oracle/tv_ops.c's backward; k = 0 is a no-op."""
    from millieye_amd import hip
    from oracle import tv_ops
    n, h = 3, 26
    size = 16.0 * h
    rois = torch.tensor([
        [0, 10, 10, 10, 10],                     # empty box
        [0, 100, 120, 60, 40],                   # inverted box (x2 < x1, y2 < y1)
        [1, -200, -150, -20, -10],               # wholly outside (negative side)
        [1, size + 50, size + 60, size + 300, size + 200],   # wholly outside (far side)
        [2, -300, -300, 3 * size, 3 * size],     # much larger than the map
        [2, 0, 0, size, size],                   # the whole map exactly
        [0, size - 8, size - 8, size + 40, size + 40],   # straddles the far corner
        [1, -5, 30, 7, 33],                      # thin sliver across the near edge
        [2, 200.5, 100.25, 201.0, 100.75],       # sub-cell box
    ], dtype=torch.float32)
    k = rois.shape[0]
    for ps, ch in ((False, 10), (True, 490)):
        m = torch.zeros((n, ch, h, h), requires_grad=True)
        out = (tv_ops.ps_roi_align if ps else tv_ops.roi_align)(m, rois, (7, 7), 1 / 16)
        go = _t(f"rdg{ch}", tuple(out.shape))
        out.backward(go)
        gd, rd = go.cuda(), rois.cuda()
        gmap = torch.zeros((n, h, h, ch)).cuda()
        fn = hip.lib().me_ps_roi_align_bwd_f32 if ps else hip.lib().me_roi_align_bwd_f32
        hip.check(fn(gd.data_ptr(), rd.data_ptr(), k, n, h, h, ch, 7, 1.0 / 16, gmap.data_ptr(), ch, hip.stream_ptr()), "roi bwd")
        torch.cuda.synchronize()
        ref = m.grad.permute(0, 2, 3, 1)
        assert torch.isfinite(gmap).all()
        assert float((gmap.cpu() - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max())), (ps, float((gmap.cpu() - ref).abs().max()))
        before = gmap.clone()
        hip.check(fn(gd.data_ptr(), rd.data_ptr(), 0, n, h, h, ch, 7, 1.0 / 16, gmap.data_ptr(), ch, hip.stream_ptr()), "roi bwd k=0")
        torch.cuda.synchronize()
        assert torch.equal(gmap, before)


def _build(name, cfg, conf):
    from millieye_amd.my_models import Network, define_yolo
    net = Network(define_yolo(ph.cfg_path(cfg)), conf)
    synth.fill_network_(net, name)
    return net


def test_training_step_vs_oracle_and_reference_golden(hip_lib):
    from oracle import network_ref
    name, cfg, n, s, conf, seed = TRAIN_CASE
    g = np.load(os.path.join(GOLD, name + ".npz"))
    net = _build(name, cfg, conf)
    sd0 = {k: v.clone() for k, v in net.state_dict().items()}
    x, maps, rboxes = train_inputs(name, n, s)
    targets = torch.from_numpy(g["targets"])
    random.seed(seed)
    ref = network_ref.network_train_step(cfgs.KNOWN[cfg](), sd0, x, maps, rboxes, targets, conf_thresh=conf)

    net = net.cuda()
    net.train()
    net.base_detector.eval()
    random.seed(seed)
    tg = targets.clone()
    rb = rboxes.clone().cuda()
    loss, output, metric, att = net(x.cuda(), maps.cuda(), rb, tg)  # train.py:185 call form (targets in mode slot)
    assert torch.allclose(tg[:, 2:], network_ref.xywh2xyxy(targets[:, 2:]) * s, atol=1e-4), "targets mutate in place"
    assert loss.requires_grad and tuple(att.shape) == (n, 1, s // 16, s // 16)
    loss.backward()
    torch.cuda.synchronize()
    assert abs(float(loss) - float(ref["loss"])) <= 1e-3 * max(1.0, abs(float(ref["loss"]))), (float(loss), float(ref["loss"]))
    assert abs(float(loss) - float(g["loss"])) <= 1e-3 * max(1.0, abs(float(g["loss"])))
    assert int(metric["true"]) == int(g["n_pos"]) and metric["total"] == int(g["total"])
    ph.assert_close(output.detach().cpu(), torch.from_numpy(g["output"]), 1e-3, "training-mode output rows")
    checked = 0
    for k, p in net.named_parameters():
        if k.startswith("base_detector."):
            assert p.grad is None
            continue
        rg = ref["grads"][k]
        if rg is None:
            assert p.grad is None, f"{k} must not receive a gradient"
            continue
        assert p.grad is not None, k
        if float(rg.abs().max()) < 1e-5:
            # mathematically zero gradient (a conv bias in front of a train-mode BatchNorm): noise on both
            # sides - only require ours to be noise-sized as well
            assert float(p.grad.abs().max()) < 1e-4, k
            checked += 1
            continue
        err = _rel(p.grad, rg)
        assert err < 2e-3, f"grad {k}: rel err {err:.2e}"
        assert abs(float(p.grad.double().norm().cpu()) - float(g["gnorm/" + k])) <= 2e-3 * float(g["gnorm/" + k]) + 1e-9, k
        checked += 1
    assert checked >= 20
    for k, v in net.state_dict().items():
        if "running_" in k and not k.startswith("base_detector."):
            assert _rel(v, torch.from_numpy(g["buf/" + k])) < 1e-3, k
    # one optimizer step like train.py:163,188-191 must change exactly the tensors that got gradients
    opt = torch.optim.Adam([p for p in net.parameters() if p.requires_grad], lr=5e-4)
    before = {k: v.clone() for k, v in net.named_parameters()}
    opt.step()
    changed = [k for k, v in net.named_parameters() if not torch.equal(v, before[k])]
    assert "ensemble_head.fc1.0.weight" in changed and "radar_cnn_layers.conv1.0.weight" in changed
    assert not any(k.startswith("base_detector.") for k in changed)
    # the next forward picks the updated weights up (packed copies are refreshed by version counters)
    net.eval()
    with torch.no_grad():
        out2 = net(x.cuda(), maps.cuda(), rboxes.clone().cuda(), 0)
    assert out2.shape[1] == 8


def test_eval_mode_training_step_vs_oracle_and_reference_golden(hip_lib):
    """``Network.forward(..., targets)`` on a model left in eval() mode: the reference has no mode check
    (my_models.py:545-641) - the loss tuple comes from running-statistics BatchNorm, autograd reaches every head parameter
    (the conv biases in front of the BatchNorms included) and no running statistic moves.  HIP path (folded affine in the
    conv epilogue, me_affine_act_bwd_f32 backward) against the oracle and the real reference's eval-mode run."""
    from oracle import network_ref
    from tests.golden.make_golden import TRAIN_EVAL_NAME
    name, cfg, n, s, conf, seed = TRAIN_CASE
    g = np.load(os.path.join(GOLD, TRAIN_EVAL_NAME + ".npz"))
    net = _build(name, cfg, conf)
    sd0 = {k: v.clone() for k, v in net.state_dict().items()}
    x, maps, rboxes = train_inputs(name, n, s)
    targets = torch.from_numpy(g["targets"])
    random.seed(seed)
    ref = network_ref.network_train_step(cfgs.KNOWN[cfg](), sd0, x, maps, rboxes, targets, conf_thresh=conf,
                                         bn_training=False)
    net = net.cuda().eval()
    random.seed(seed)
    loss, output, metric, att = net(x.cuda(), maps.cuda(), rboxes.clone().cuda(), 0, targets.clone())
    loss.backward()
    torch.cuda.synchronize()
    assert abs(float(loss) - float(ref["loss"])) <= 1e-3 * max(1.0, abs(float(ref["loss"])))
    assert abs(float(loss) - float(g["loss"])) <= 1e-3 * max(1.0, abs(float(g["loss"])))
    assert int(metric["true"]) == int(g["n_pos"]) and metric["total"] == int(g["total"])
    ph.assert_close(output.detach().cpu(), torch.from_numpy(g["output"]), 1e-3, "eval-mode output rows")
    checked = 0
    for k, p in net.named_parameters():
        if k.startswith("base_detector."):
            assert p.grad is None
            continue
        rg = ref["grads"][k]
        if rg is None:
            assert p.grad is None, f"{k} must not receive a gradient"
            continue
        assert p.grad is not None, k
        err = _rel(p.grad, rg)
        assert err < 2e-3, f"grad {k}: rel err {err:.2e}"
        assert abs(float(p.grad.double().norm().cpu()) - float(g["gnorm/" + k])) <= 2e-3 * float(g["gnorm/" + k]) + 1e-9, k
        checked += 1
    assert checked >= 20
    for k, v in net.state_dict().items():
        if "running_" in k or "num_batches_tracked" in k:
            assert torch.equal(v.cpu(), sd0[k]), f"{k} moved in eval mode"


def test_frozen_parameters_get_no_gradient(hip_lib):
    """train.py:146-149 freezes the stage-2 tensors by requires_grad=False."""
    name, cfg, n, s, conf, seed = TRAIN_CASE
    g = np.load(os.path.join(GOLD, name + ".npz"))
    net = _build(name, cfg, conf).cuda()
    frozen = ["img_cnn_layers.net.conv_0.weight", "refinement_head.net0.0.weight", "refinement_head.net2.0.bias"]
    for k, p in net.named_parameters():
        if k in frozen:
            p.requires_grad = False
    net.train()
    net.base_detector.eval()
    x, maps, rboxes = train_inputs(name, n, s)
    random.seed(seed)
    loss, *_ = net(x.cuda(), maps.cuda(), rboxes.clone().cuda(), 0, torch.from_numpy(g["targets"]).clone())
    loss.backward()
    for k, p in net.named_parameters():
        if k in frozen:
            assert p.grad is None
    assert net.ensemble_head.fc2[0].weight.grad is not None


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_training_step_with_16bit_frozen_detector(hip_lib, dtype):
    """BASELINE configs[3] names "bf16 training": in stage 3 the detector is frozen (train.py:170), so the 16-bit part of
    the step is its inference pass (``Darknet.compute_dtype``); heads, loss, backward and the gradient bucket stay fp32.
    The step must be the fp32 step up to the detector's storage error: same RoI population (+-10 %), loss within 10 %,
    the concatenated head gradient pointing the same way (cosine >= 0.95; bf16 moves box corners by <= 2 px and the
    feature tap by its 8-bit mantissa), finite everywhere, run-to-run deterministic given the python RNG seed."""
    name, cfg, n, s, conf = "train16", "yolov3-tiny-12", 8, 416, 0.2
    from millieye_amd.my_models import Network, define_yolo
    net = Network(define_yolo(ph.cfg_path(cfg)), conf)
    synth.fill_network_(net, name, cls0_bias=3.0, cls_bias=-4.0)
    net = net.cuda().eval()
    x = torch.from_numpy(synth.uniform(name + "/x", (n, 3, s, s))).cuda()
    maps, rboxes = synth.radar_inputs(name + "/radar", n, s // 16)
    maps, rboxes = torch.from_numpy(maps).cuda(), torch.from_numpy(rboxes)
    with torch.no_grad():
        det = net(x, maps, rboxes.clone().cuda(), 1).cpu()
    tg = []
    for i in range(n):
        rows_i = det[det[:, 0] == i]
        for j in (0, 3):
            if j < len(rows_i):
                b = rows_i[j, 1:5] / s
                tg.append([i, 0, float((b[0] + b[2]) / 2), float((b[1] + b[3]) / 2), float(b[2] - b[0]) * 1.05,
                           float(b[3] - b[1]) * 0.95])
    targets = torch.tensor(tg, dtype=torch.float32).reshape(-1, 6)
    assert len(targets) >= n

    def step(mode):
        net.base_detector.compute_dtype = mode
        net.train()
        net.base_detector.eval()
        for p in net.parameters():
            p.grad = None
        random.seed(99)
        loss, rows, metric, _att = net(x, maps, rboxes.clone().cuda(), targets.clone())
        loss.backward()
        grads = torch.cat([p.grad.flatten() for k, p in net.named_parameters()
                           if not k.startswith("base_detector.") and p.grad is not None]).cpu()
        return float(loss), int(net._last_train["k"]), int(metric["true"]), grads, rows.detach().cpu()

    bn_state = {k: v.clone() for k, v in net.state_dict().items() if "running_" in k or "num_batches" in k}
    l32, k32, pos32, g32, _rows32 = step("f32")
    net.load_state_dict(bn_state, strict=False)   # the same running statistics in front of both steps
    l16, k16, pos16, g16, rows16 = step(dtype)
    net.load_state_dict(bn_state, strict=False)
    l16b, k16b, _p, g16b, rows16b = step(dtype)
    assert (l16, k16) == (l16b, k16b) and torch.equal(rows16, rows16b), "16-bit step must be deterministic"
    assert float((g16 - g16b).abs().max()) <= 1e-5 * float(g16.abs().max())  # RoI scatters use atomics (like torchvision)
    assert pos32 > 0 and k32 > 50
    assert abs(k16 - k32) <= 0.1 * k32, (k16, k32)
    assert abs(l16 - l32) <= 0.1 * abs(l32), (l16, l32)
    assert bool(torch.isfinite(g16).all()) and g16.shape == g32.shape
    cos = float(torch.dot(g16.double(), g32.double()) / (g16.double().norm() * g32.double().norm()))
    assert cos >= 0.95, f"gradient cosine {cos:.4f}"
    for k, p in net.named_parameters():
        if k.startswith("base_detector."):
            assert p.grad is None


def test_train_mode_forward_without_targets(hip_lib):
    """A model left in ``train()`` mode and called like at inference (reference my_models.py:433-539 under ``model.train()``):
    the same rows as the training call's ``output`` (batch-statistics BatchNorm over pixels / RoIs) and the same running
    statistics update - against the oracle's training forward; mode 1 never reaches a BatchNorm; a mix of modes is refused."""
    from oracle import network_ref
    name, cfg, n, s, conf, seed = TRAIN_CASE
    g = np.load(os.path.join(GOLD, name + ".npz"))
    net = _build(name, cfg, conf)
    sd0 = {k: v.clone() for k, v in net.state_dict().items()}
    x, maps, rboxes = train_inputs(name, n, s)
    random.seed(seed)
    ref = network_ref.network_train_step(cfgs.KNOWN[cfg](), sd0, x, maps, rboxes, torch.from_numpy(g["targets"]), conf_thresh=conf)
    net = net.cuda()
    net.eval()
    with torch.no_grad():
        out_eval = net(x.cuda(), maps.cuda(), rboxes.clone().cuda(), 0)
        out1_eval = net(x.cuda(), maps.cuda(), rboxes.clone().cuda(), 1)
    net.train()
    net.base_detector.eval()
    with torch.no_grad():
        out1 = net(x.cuda(), maps.cuda(), rboxes.clone().cuda(), 1)
        out = net(x.cuda(), maps.cuda(), rboxes.clone().cuda(), 0)
    assert torch.equal(out1, out1_eval)
    ph.assert_close(out.cpu(), ref["output"], 1e-3, "train()-mode forward rows vs the oracle's training forward")
    ph.assert_close(out.cpu(), torch.from_numpy(g["output"]), 1e-3, "train()-mode forward rows vs the reference golden")
    assert out.shape != out_eval.shape or not torch.allclose(out, out_eval, atol=1e-4), "batch statistics must matter"
    for k, v in net.state_dict().items():
        if "running_" in k and not k.startswith("base_detector."):
            assert _rel(v, ref["buffers"][k]) < 1e-3, k
    net.img_cnn_layers.eval()
    with pytest.raises(NotImplementedError):
        net(x.cuda(), maps.cuda(), rboxes.clone().cuda(), 0)


def test_training_step_with_a_radar_map_of_another_size(hip_lib):
    """utils/datasets.py resizes the radar map to S/16 for training, the demos feed the raw 32 x 32 map with the same
    spatial_scale (quirk q15): the training step must take a radar map whose side differs from the feature map's
    (12 x 12 maps on a 10 x 10 feature map here) - loss, rows and every gradient against the oracle."""
    from oracle import network_ref
    name, cfg, n, s, conf = "train_q15", "yolov3-tiny-12", 2, 160, 0.2
    net = _build(name, cfg, conf)
    sd0 = {k: v.clone() for k, v in net.state_dict().items()}
    x = torch.from_numpy(synth.uniform(name + "/x", (n, 3, s, s)))
    maps, rboxes = synth.radar_inputs(name + "/radar", n, 12)
    maps, rboxes = torch.from_numpy(maps), torch.from_numpy(rboxes)
    tg = []
    for row in rboxes.numpy():
        i, x1, y1, x2, y2 = row
        tg.append([i, 0, (x1 + x2) / 2, (y1 + y2) / 2, (x2 - x1) * 1.02, (y2 - y1) * 0.98])
    targets = torch.tensor(np.array(tg, dtype=np.float32))
    random.seed(5)
    ref = network_ref.network_train_step(cfgs.KNOWN[cfg](), sd0, x, maps, rboxes, targets, conf_thresh=conf)
    assert ref["n_pos"] > 0
    net = net.cuda()
    net.train()
    net.base_detector.eval()
    random.seed(5)
    loss, output, metric, att = net(x.cuda(), maps.cuda(), rboxes.clone().cuda(), 0, targets.clone())
    loss.backward()
    assert tuple(att.shape) == (n, 1, 12, 12)
    assert abs(float(loss.detach()) - float(ref["loss"])) <= 1e-3 * max(1.0, abs(float(ref["loss"])))
    ph.assert_close(output.detach().cpu(), ref["output"], 1e-3, "rows")
    checked = 0
    for k, p in net.named_parameters():
        if k.startswith("base_detector."):
            continue
        rg = ref["grads"][k]
        if rg is None:
            assert p.grad is None, k
            continue
        if float(rg.abs().max()) < 1e-5:
            assert float(p.grad.abs().max()) < 1e-4, k
        else:
            assert _rel(p.grad, rg) < 2e-3, f"grad {k}: {_rel(p.grad, rg):.2e}"
        checked += 1
    assert checked >= 20


def test_iou_labels_kernel_is_bit_identical_with_the_host_restatement(hip_lib):
    """me_iou_labels_f32 (device-side obtain_iou_labels, quirk q5) against train_path.iou_labels_vectorized - itself pinned
    bit for bit to the reference's per-box loop (tests/test_train_cpu.py): exact ties (first target wins), proposals with no
    target of their image / class, zero targets, zero proposals, and the packed side columns."""
    from millieye_amd import hip
    from millieye_amd.train_path import iou_labels_vectorized
    n_img, n_radar, q, cols = 700, 45, 23, 9
    xy = synth.uniform("ilk/xy", (n_img + n_radar, 2), 0, 300)
    wh = synth.uniform("ilk/wh", (n_img + n_radar, 2), 5, 120)
    img = np.floor(synth.uniform("ilk/i", (n_img + n_radar, 1), 0, 5))
    cls = np.floor(synth.uniform("ilk/c", (n_img, 1), 0, 2))
    ib = np.zeros((n_img, cols), dtype=np.float32)
    ib[:, 0:1], ib[:, 1:3], ib[:, 3:5], ib[:, 5], ib[:, 7:8] = img[:n_img], xy[:n_img], (xy + wh)[:n_img], 0.25, cls
    rb = np.concatenate([img[n_img:], xy[n_img:], (xy + wh)[n_img:]], 1).astype(np.float32)
    t_xy = synth.uniform("ilk/txy", (q, 2), 0, 300)
    t_wh = synth.uniform("ilk/twh", (q, 2), 5, 120)
    tg = np.concatenate([np.floor(synth.uniform("ilk/ti", (q, 1), 0, 4)), np.zeros((q, 1), np.float32), t_xy, t_xy + t_wh],
                        1).astype(np.float32)
    tg[5] = tg[4]                         # an exact tie: the first of the two wins either way (same label)
    ib[:10, 1:5], ib[:10, 0], ib[:10, 7] = tg[:10, 2:6], tg[:10, 0], tg[:10, 1]   # perfect matches
    rb[:5, 1:5], rb[:5, 0] = tg[10:15, 2:6] + 1.5, tg[10:15, 0]
    boxes = np.concatenate([np.concatenate([ib[:, :1], ib[:, 7:8], ib[:, 1:5]], 1),
                            np.concatenate([rb[:, :1], np.zeros((n_radar, 1), np.float32), rb[:, 1:5]], 1)], 0)
    want, _loc = iou_labels_vectorized(torch.from_numpy(boxes), torch.from_numpy(tg))
    k = n_img + n_radar
    refine = torch.rand((k, 2)).cuda()
    mask1 = torch.rand(k).cuda()
    keep = (torch.rand(k) > 0.5).to(torch.uint8).cuda()
    out = torch.full((k, 4), -7.0).cuda()
    ibd, rbd, tgd = torch.from_numpy(ib).cuda(), torch.from_numpy(rb).cuda(), torch.from_numpy(tg).cuda()
    hip.check(hip.lib().me_iou_labels_f32(ibd.data_ptr(), n_img, cols, rbd.data_ptr(), n_radar, tgd.data_ptr(), q,
                                          refine.data_ptr(), mask1.data_ptr(), keep.data_ptr(), out.data_ptr(),
                                          hip.stream_ptr()), "me_iou_labels_f32")
    got = out.cpu()
    assert torch.equal(got[:, 0:1], want), float((got[:, 0:1] - want).abs().max())
    assert (want > 0.7).sum() >= 10 and (want == 0).sum() > 0
    assert torch.equal(got[:, 1], keep.float().cpu()) and torch.equal(got[:, 3], mask1.cpu())
    assert torch.equal(got[:n_img, 2], torch.full((n_img,), 0.25)) and torch.equal(got[n_img:, 2], refine[n_img:, 0].cpu())
    out.fill_(-7.0)
    hip.check(hip.lib().me_iou_labels_f32(ibd.data_ptr(), n_img, cols, rbd.data_ptr(), n_radar, None, 0, refine.data_ptr(),
                                          mask1.data_ptr(), keep.data_ptr(), out.data_ptr(), hip.stream_ptr()), "no targets")
    assert float(out[:, 0].abs().max()) == 0
    hip.check(hip.lib().me_iou_labels_f32(None, 0, cols, None, 0, tgd.data_ptr(), q, refine.data_ptr(), mask1.data_ptr(),
                                          keep.data_ptr(), out.data_ptr(), hip.stream_ptr()), "no proposals")


@pytest.mark.parametrize("n,h,w,cin,cout,k", [(2, 18, 22, 3, 32, 3), (8, 48, 48, 128, 128, 3), (2, 12, 12, 64, 96, 1)])
def test_conv_wgrad_pitched_operands(hip_lib, n, h, w, cin, cout, k):
    """x and dy as channel slices of wider NHWC buffers (pitch > channels): the stem kernel (cin 3), the 128x128-tile kernel and
    the 64x64 one must honour the pitches (16-byte alignment kept by slicing at multiples of 4 channels)."""
    from millieye_amd import hip
    pad = (k - 1) // 2
    tag = f"wp{n}{h}{cin}{cout}{k}"
    x = _t(tag + "x", (n, cin, h, w))
    wt = _t(tag + "w", (cout, cin, k, k), -0.2, 0.2).requires_grad_(True)
    y = F.conv2d(x, wt, None, 1, pad)
    dy = _t(tag + "dy", tuple(y.shape))
    y.backward(dy)
    xw = torch.full((n, h, w, cin + 8), 3.0, device="cuda")
    xw[..., 4:4 + cin] = x.permute(0, 2, 3, 1).cuda()
    dw_ = torch.full((n, h, w, cout + 12), -2.0, device="cuda")
    dw_[..., 8:8 + cout] = dy.permute(0, 2, 3, 1).cuda()
    got = hip.conv_wgrad(xw[..., 4:4 + cin], dw_[..., 8:8 + cout], k, 1, pad, oihw=True)
    assert _rel(got, wt.grad) < 1e-4


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_detector_prefetch_is_only_an_overlap(hip_lib, dtype):
    """Network.queue_detector_prefetch (round 5): the frozen detector + NMS + proposal assembly of the NEXT batch issued on a second
    stream under the current batch's tail.  Every step must be the plain step: the batch that names a successor, the batch that
    consumes the prefetched part (loss, RoI count and output rows equal; gradients to the RoI scatters' atomics), and the cases where
    the prefetched part must NOT be used - other frames, detector weights changed in between."""
    name, cfg, n, s, conf = "prefetch", "yolov3-tiny-12", 4, 416, 0.2
    from millieye_amd.my_models import Network, define_yolo
    net = Network(define_yolo(ph.cfg_path(cfg)), conf)
    synth.fill_network_(net, name, cls0_bias=3.0, cls_bias=-4.0)
    net = net.cuda().eval()
    net.base_detector.compute_dtype = dtype
    xs = [torch.from_numpy(synth.uniform(f"{name}/x{i}", (n, 3, s, s))).cuda() for i in range(2)]
    maps, rboxes = synth.radar_inputs(name + "/radar", n, s // 16)
    maps, rboxes = torch.from_numpy(maps).cuda(), torch.from_numpy(rboxes)
    with torch.no_grad():
        det = net(xs[0], maps, rboxes.clone().cuda(), 1).cpu()
    tg = [[i, 0, 0.5, 0.5, 0.3, 0.3] for i in range(n)]
    for i in range(n):
        rows_i = det[det[:, 0] == i]
        if len(rows_i):
            b = rows_i[0, 1:5] / s
            tg.append([i, 0, float((b[0] + b[2]) / 2), float((b[1] + b[3]) / 2), float(b[2] - b[0]) * 1.05, float(b[3] - b[1]) * 0.95])
    targets = torch.tensor(tg, dtype=torch.float32)
    bn_state = {k: v.clone() for k, v in net.state_dict().items() if "running_" in k or "num_batches" in k}

    def step(x, ahead=None):
        net.load_state_dict(bn_state, strict=False)   # the same running statistics in front of every step
        net.train()
        net.base_detector.eval()
        for p in net.parameters():
            p.grad = None
        random.seed(7)
        if ahead is not None:
            net.queue_detector_prefetch(ahead)
        loss, rows, _metric, _att = net(x, maps, rboxes.clone().cuda(), targets.clone())
        loss.backward()
        torch.cuda.synchronize()
        grads = torch.cat([p.grad.flatten() for k, p in net.named_parameters()
                           if not k.startswith("base_detector.") and p.grad is not None]).cpu()
        return float(loss), int(net._last_train["k"]), rows.detach().cpu(), grads

    def same(a, b, what):
        assert a[0] == b[0] and a[1] == b[1] and torch.equal(a[2], b[2]), (what, a[0], b[0], a[1], b[1])
        assert float((a[3] - b[3]).abs().max()) <= 1e-5 * float(b[3].abs().max()), what

    plain = [step(xs[0]), step(xs[1])]
    assert plain[0][1] > 20 and plain[0][0] != plain[1][0]
    same(step(xs[0], ahead=xs[1]), plain[0], "the batch that names its successor")
    assert "_det_prefetch" in net.__dict__
    same(step(xs[1]), plain[1], "the batch that consumes the prefetched part")
    assert "_det_prefetch" not in net.__dict__
    # a longer run of look-ahead steps (from the third prefetch on the detector is one captured graph replay, engine.graph_replay)
    for i in range(6):
        same(step(xs[i % 2], ahead=xs[(i + 1) % 2]), plain[i % 2], f"look-ahead step {i}")
    net.__dict__.pop("_det_prefetch", None)
    # ... and that replay carries the NMS candidate pass (one capture per (threshold, workspace), engine._run_graph)
    plans = net.base_detector.engine_for(dtype)._plans.values()
    assert any(isinstance(p.graph, dict) and any(k is not None and k[0] == conf and r["graph"] is not None for k, r in p.graph.items())
               for p in plans), "the look-ahead detector never ran as a captured graph with the candidate decode"
    # other frames than the ones announced: computed again, behind the prefetch
    step(xs[0], ahead=xs[1])
    same(step(xs[0]), plain[0], "other frames than the prefetched ones")
    # detector weights changed between the prefetch and the call: the prefetched part is stale and must not be used
    step(xs[0], ahead=xs[1])
    w = net.base_detector.module_list[0][0].weight
    w0 = w.detach().clone()
    with torch.no_grad():
        w.mul_(1.5)
    changed = step(xs[1])
    assert "_det_prefetch" not in net.__dict__
    fresh = step(xs[1])            # the plain step on the changed weights
    same(changed, fresh, "detector weights changed after the prefetch")
    assert changed[0] != plain[1][0]
    with torch.no_grad():
        w.copy_(w0)
    # the inference path after an unconsumed prefetch waits for it (Darknet._run) and is the plain inference result
    step(xs[0], ahead=xs[1])
    net.load_state_dict(bn_state, strict=False)
    net.eval()
    with torch.no_grad():
        again = net(xs[0], maps, rboxes.clone().cuda(), 1).cpu()
    assert torch.equal(again, det)


def test_captured_backward_equals_the_eager_backward(hip_lib, monkeypatch):
    """The stage-3 backward as one captured hipGraph (train_path._graphed_backward: arena buffers of fixed capacity, launches over
    the capacity, the proposal count from a device word - me_heads_tail_bwd_dev_f32 / me_bn_train_bwd_dev_f32 / me_[ps_]roi_align_bwd_dev_f32)
    against the eager backward (MILLIEYE_TRAIN_GRAPH=0) over six steps whose proposal count CHANGES from step to step (other frames,
    other radar boxes - fewer and more rows than at capture time): the same loss, the same output rows, every gradient within 1e-5 of
    its largest entry (the dense products run over the capacity: exact zeros are added, the summation tree of the row-chunked sums
    differs), gradient accumulation over two batches without aliasing, and the Adam trajectories stay together."""
    import random
    name, cfg, n, s, conf, seed = TRAIN_CASE

    def run(graph):
        monkeypatch.setenv("MILLIEYE_TRAIN_GRAPH", "1" if graph else "0")
        net = _build(name, cfg, conf)
        net = net.to(net.device).train()
        net.base_detector.eval()
        heads = [p for k, p in net.named_parameters() if not k.startswith("base_detector.")]
        opt = torch.optim.Adam(heads, lr=1e-3)
        random.seed(seed)
        torch.manual_seed(seed)
        out = []
        for step in range(6):
            x = torch.from_numpy(synth.uniform(f"{name}/cap/x{step}", (n, 3, s, s))).cuda()
            maps, rboxes = synth.radar_inputs(f"{name}/cap/radar{step}", n, s // 16, boxes_per_image=(3, 1, 4, 1, 6, 2)[step])
            targets = torch.tensor([[0, 0, 0.3, 0.4, 0.3, 0.3], [1, 0, 0.6, 0.5, 0.4, 0.5]])
            loss, rows, metric, att = net(x, torch.from_numpy(maps).cuda(), torch.from_numpy(rboxes).cuda(), targets.clone())
            loss.backward()
            grads = {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}
            if step % 2 == 1:   # two batches per optimizer step, like the reference's loop (gradient accumulation)
                opt.step()
                opt.zero_grad()
            out.append((float(loss.detach()), rows.detach().cpu().clone(), int(metric["total"]), grads))
        arena = net.__dict__.get("_train_arena")
        if graph and moved:
            # a library workspace the capture points into is REPLACED (what a larger request on the same stream does): the next
            # backward must notice and capture again instead of replaying launches that point into the old buffer
            from millieye_amd import hip
            recs = [r for r in arena.graphs.values() if r.graph not in (None, False)]
            assert recs and recs[0].scratch
            old_graph = recs[0].graph
            for key in list(recs[0].scratch):
                t = hip._ws_cache[key]
                hip._ws_cache[key] = torch.empty(t.numel() + 4096, dtype=t.dtype, device=t.device)
            x = torch.from_numpy(synth.uniform(f"{name}/cap/x0", (n, 3, s, s))).cuda()
            maps, rboxes = synth.radar_inputs(f"{name}/cap/radar0", n, s // 16, boxes_per_image=3)
            targets = torch.tensor([[0, 0, 0.3, 0.4, 0.3, 0.3], [1, 0, 0.6, 0.5, 0.4, 0.5]])
            for p in net.parameters():
                p.grad = None
            loss, _rows, _metric, _att = net(x, torch.from_numpy(maps).cuda(), torch.from_numpy(rboxes).cuda(), targets.clone())
            loss.backward()
            torch.cuda.synchronize()
            assert recs[0].graph is not old_graph and recs[0].graph not in (None, False), "the moved workspace went unnoticed"
            assert all(hip._ws_cache[k].data_ptr() == v for k, v in recs[0].scratch.items())
            assert all(torch.isfinite(p.grad).all() for p in heads if p.grad is not None)
        return out, arena

    moved = False
    eager, arena_e = run(False)
    graphed, arena_g = run(True)
    moved = True
    run(True)
    assert arena_e is None and arena_g is not None
    recs = [r for r in arena_g.graphs.values()]
    assert any(r.graph not in (None, False) for r in recs), "the backward was never captured"
    totals = [e[2] for e in eager]
    print("proposal counts per step:", totals)
    assert min(totals[3:]) < totals[2] < max(totals[3:]), f"the replayed steps must have fewer AND more proposals than the captured one: {totals}"
    for step, (e, g) in enumerate(zip(eager, graphed)):
        assert abs(e[0] - g[0]) <= 1e-5 * max(1.0, abs(e[0])), (step, e[0], g[0])
        assert e[2] == g[2] and e[1].shape == g[1].shape, step
        assert torch.allclose(e[1], g[1], rtol=1e-4, atol=1e-4), step
        assert sorted(e[3]) == sorted(g[3])
        for k in e[3]:
            scale = max(float(e[3][k].abs().max()), 1e-8)
            err = float((e[3][k] - g[3][k]).abs().max()) / scale
            assert err <= 2e-4, f"step {step} {k}: {err:.2e}"
