"""-m gpu: the detector's training step as one captured hipGraph (millieye_amd/detector_graph.py) against the eager step
(Darknet.forward(x, targets) -> loss.backward(), which the other suites pin to the oracle / the reference): the same kernels in
the same order with the same arguments, so loss and every parameter gradient must be EQUAL, bit for bit - on the capture's inputs,
on other frames / target tables replayed through the static buffers, and after an optimizer step between replays."""
import pytest
import torch

from millieye_amd import synth
from tests import parity_helpers as ph

pytestmark = pytest.mark.gpu

T_A = [[0, 3, 0.30, 0.40, 0.20, 0.30], [1, 17, 0.70, 0.60, 0.50, 0.40], [1, 60, 0.52, 0.48, 0.10, 0.15]]
T_B = [[1, 5, 0.11, 0.81, 0.30, 0.22]]
T_C = [[0, 1, 0.25, 0.25, 0.4, 0.4], [0, 2, 0.26, 0.26, 0.41, 0.39], [1, 7, 0.9, 0.1, 0.05, 0.6], [1, 7, 0.5, 0.5, 0.9, 0.9],
       [0, 79, 0.6, 0.3, 0.2, 0.2]]


def _eager(model, x, t):
    model.zero_grad(set_to_none=True)
    loss, _fm, _yo = model(x, t)
    loss.backward()
    grads = {k: p.grad.clone() for k, p in model.named_parameters()}
    metrics = [dict(layer.metrics) for layer in model.yolo_layers]
    model.zero_grad(set_to_none=True)
    return loss.detach().clone(), grads, metrics


def _same(step_loss, model, loss, grads, what):
    assert float(step_loss) == float(loss), (what, float(step_loss), float(loss))
    diff = [k for k, p in model.named_parameters() if p.grad is None or not torch.equal(p.grad, grads[k])]
    assert not diff, f"{what}: {len(diff)} gradients differ from the eager step's, e.g. {diff[:4]}"


@pytest.mark.parametrize("dtype,cfg,n,s", [("bf16", "yolov3", 2, 128), ("f16", "yolov3", 2, 128), ("f32", "yolov3", 2, 128),
                                           ("f32", "yolov3-tiny-12", 2, 96)])
def test_captured_step_equals_the_eager_step(hip_lib, dtype, cfg, n, s):
    from millieye_amd.detector_graph import GraphedDetectorStep
    name = f"graph/{cfg}"
    classes = 80 if cfg == "yolov3" else 12
    tabs = [torch.tensor([[r[0], r[1] % classes] + r[2:] for r in t], dtype=torch.float32) for t in (T_A, T_B, T_C)]
    xs = [torch.from_numpy(synth.uniform(f"{name}/x{i}", (n, 3, s, s))).cuda() for i in range(3)]
    model = ph.make_darknet(cfg, tag=name).cuda().eval()
    model.compute_dtype = dtype
    step = GraphedDetectorStep(model, max_targets=8)
    # the capture's own inputs, then other frames with fewer / more target rows through the static buffers (host targets, then
    # device targets), then the first inputs again
    for i, (x, t) in enumerate([(xs[0], tabs[0]), (xs[1], tabs[1]), (xs[2], tabs[2].cuda()), (xs[0], tabs[0])]):
        loss, grads, metrics = _eager(model, x, t)
        got = step(x, t)
        _same(got, model, loss, grads, f"replay {i}")
        m = step.metrics()
        for a, b in zip(m, metrics):
            for k, v in b.items():
                assert a[k] == v or (a[k] != a[k] and v != v), (i, k, a[k], v)   # (equal, or NaN in both: a scale without objects)
        model.zero_grad(set_to_none=True)
    # an optimizer step between replays is seen (the weight packing is part of the graph)
    opt = torch.optim.SGD(model.parameters(), lr=1e-4)
    step(xs[0], tabs[0])
    opt.step()
    opt.zero_grad(set_to_none=True)
    loss, grads, _ = _eager(model, xs[0], tabs[0])
    got = step(xs[0], tabs[0])
    _same(got, model, loss, grads, "after an SGD step")
    # a .grad that is still there is added to (autograd's accumulation), whether it is the static tensor or another one
    g1 = {k: p.grad.clone() for k, p in model.named_parameters()}
    step(xs[1], tabs[1])
    acc = {k: p.grad.clone() for k, p in model.named_parameters()}
    model.zero_grad(set_to_none=True)
    _loss2, g2, _ = _eager(model, xs[1], tabs[1])
    for k in acc:
        assert torch.equal(acc[k], g1[k] + g2[k]), k
    # no targets at all: the eager step's NaN terms (0 / 0 means) and nothing worse
    model.zero_grad(set_to_none=True)
    got = step(xs[0], torch.zeros((0, 6)))
    loss0, _g0, _ = _eager(model, xs[0], torch.zeros((0, 6)))
    assert (float(got) != float(got)) == (float(loss0) != float(loss0))


def test_captured_step_reports_bad_targets_and_refuses_what_it_cannot_do(hip_lib):
    from millieye_amd.detector_graph import GraphedDetectorStep
    model = ph.make_darknet("yolov3-tiny-12", tag="graph/bad").cuda().eval()
    x = torch.from_numpy(synth.uniform("graph/bad/x", (2, 3, 96, 96))).cuda()
    step = GraphedDetectorStep(model, max_targets=4)
    step(x, torch.tensor([[0, 1, 0.5, 0.5, 0.2, 0.2]]))
    step.metrics()
    step(x, torch.tensor([[5, 1, 0.5, 0.5, 0.2, 0.2]]))   # image 5 of a batch of 2
    step(x, torch.tensor([[0, 1, 0.5, 0.5, 0.2, 0.2]]))   # (a good step afterwards does not hide it)
    with pytest.raises(IndexError):
        step.metrics()
    step(x, torch.tensor([[0, 1, 0.5, 0.5, 0.2, 0.2]]))
    step.metrics()                                          # reported once
    with pytest.raises(ValueError):                         # more rows than the static table holds
        step(x, torch.zeros((5, 6)))
    model.train()
    with pytest.raises(NotImplementedError):                # batch statistics: an eager path
        GraphedDetectorStep(model)(x, torch.tensor([[0, 1, 0.5, 0.5, 0.2, 0.2]]))
    cpu_model = ph.make_darknet("yolov3-tiny-12", tag="graph/cpu").eval()
    from millieye_amd import hip
    with pytest.raises(hip.MeError):                        # no CPU fallback
        GraphedDetectorStep(cpu_model)(x.cpu(), torch.tensor([[0, 1, 0.5, 0.5, 0.2, 0.2]]))


def test_counted_yolo_loss_entries_equal_the_host_counted_ones(hip_lib):
    """me_yolo_loss_fwd_counted_f32 / me_yolo_loss_bwd_dev_f32 against me_yolo_loss_fwd_f32 / _bwd_f32 (pinned to the oracle's
    torch restatement in test_gpu_train.py): a table of capacity 16 with 3 live rows (garbage behind them) == the 3-row table."""
    import ctypes as C
    from millieye_amd import hip
    from millieye_amd.yolov3.models import _yolo_loss_workspace
    lib, dev = hip.lib(), torch.device("cuda")
    n, g, na, nc = 2, 13, 3, 12
    per = nc + 5
    raw = torch.from_numpy(synth.uniform("graph/raw", (n, g, g, na * per), -2, 2)).to(dev)
    live = torch.tensor(T_A, dtype=torch.float32)
    live[:, 1] = torch.tensor([3.0, 7.0, 11.0])
    table = torch.full((16, 6), 123.0)
    table[:3] = live
    anchors = (C.c_float * 6)(1.25, 1.625, 2.0, 3.75, 4.125, 2.875)

    def run(counted):
        f32 = dict(device=dev, dtype=torch.float32)
        cells = (n, na, g, g)
        obj, noobj = torch.empty(cells, device=dev, dtype=torch.uint8), torch.empty(cells, device=dev, dtype=torch.uint8)
        tx, ty, tw, th, tconf, cm, iou = (torch.empty(cells, **f32) for _ in range(7))
        tcls = torch.empty(cells + (nc,), **f32)
        res = torch.zeros(16, **f32)
        ws = _yolo_loss_workspace(dev)
        draw = torch.empty_like(raw)
        tail = [obj, noobj, tx, ty, tw, th, tcls, tconf, cm, iou, ws, res]
        if counted:
            tg = table.to(dev)
            cnt = torch.tensor([3], dtype=torch.int32, device=dev)
            hip.check(lib.me_yolo_loss_fwd_counted_f32(raw.data_ptr(), na * per, n, g, na, nc, anchors, tg.data_ptr(), 16, cnt.data_ptr(),
                                                       0.5, 1.0, 100.0, *[t.data_ptr() for t in tail], hip.stream_ptr()), "counted")
            hip.check(lib.me_yolo_loss_bwd_dev_f32(raw.data_ptr(), na * per, n, g, na, nc, *[t.data_ptr() for t in (obj, noobj, tx, ty, tw, th, tcls, tconf)],
                                                   res.data_ptr(), 1.0, 100.0, None, draw.data_ptr(), na * per, hip.stream_ptr()), "bwd dev")
        else:
            tg = live.to(dev)
            hip.check(lib.me_yolo_loss_fwd_f32(raw.data_ptr(), na * per, n, g, na, nc, anchors, tg.data_ptr(), 3, 0.5, 1.0, 100.0,
                                               *[t.data_ptr() for t in tail], hip.stream_ptr()), "fwd")
            r = res.tolist()
            hip.check(lib.me_yolo_loss_bwd_f32(raw.data_ptr(), na * per, n, g, na, nc, *[t.data_ptr() for t in (obj, noobj, tx, ty, tw, th, tcls, tconf)],
                                               r[13], r[14], 1.0, 100.0, 1.0, draw.data_ptr(), na * per, hip.stream_ptr()), "bwd")
        torch.cuda.synchronize()
        return res, draw, obj, noobj, tcls

    a, b = run(False), run(True)
    for u, v in zip(a, b):
        assert torch.equal(u, v)
    assert float(a[0][13]) == 3.0 and float(a[0][15]) == 0.0
