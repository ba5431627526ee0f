"""-m gpu: whole-detector parity - Darknet(cfg).forward on the HIP engine vs the CPU oracle
(oracle/darknet_ref.py, itself pinned to the reference by tests/golden) on identical seeded
weights and frames.  Bar: 1e-3 (allclose rtol=atol) on featuremap and every yolo_outputs row."""
import pytest
import torch

from tests import parity_helpers as ph

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _run(name, n, s):
    from oracle import darknet_ref
    from millieye_amd.engine import pick_tap_module

    model = ph.make_darknet(name)
    x = ph.frames(f"{name}/{n}/{s}", n, s)
    tap = pick_tap_module(model.module_defs)
    ref_fm, ref_yolo = darknet_ref.darknet_forward(ph.cfg_text(name), model.state_dict(), x, tap_module=tap)
    model = model.cuda()
    with torch.no_grad():
        fm, yolo = model(x.cuda())
    torch.cuda.synchronize()
    assert tuple(fm.shape) == tuple(ref_fm.shape)
    e1 = ph.assert_close(fm.cpu(), ref_fm, TOL, f"{name} featuremap")
    e2 = ph.assert_close(yolo.cpu(), ref_yolo, TOL, f"{name} yolo_outputs")
    print(f"{name} n={n} s={s}: featuremap err {e1:.2e}, yolo err {e2:.2e}")
    return model, x


@pytest.mark.parametrize("name", ["yolov3-tiny-12", "yolov3-tiny-coco"])
@pytest.mark.parametrize("n,s", [(2, 96), (1, 416), (3, 160)])
def test_tiny(hip_lib, name, n, s):
    _run(name, n, s)


@pytest.mark.parametrize("n,s", [(2, 64), (1, 416), (2, 160)])
def test_darknet53(hip_lib, n, s):
    _run("yolov3", n, s)


def test_shapes_config0_and_repeatability(hip_lib):
    """BASELINE config 0 shapes (tiny-coco 416, batch 1) and run-to-run determinism + weight refresh."""
    model, x = _run("yolov3-tiny-coco", 1, 416)
    with torch.no_grad():
        fm, yolo = model(x.cuda())
        fm2, yolo2 = model(x.cuda())
    assert tuple(fm.shape) == (1, 256, 26, 26) and tuple(yolo.shape) == (1, 2535, 85)
    assert torch.equal(yolo, yolo2) and torch.equal(fm, fm2)
    # parameter update must be picked up by the packed device copies
    with torch.no_grad():
        model.module_list[0][0].weight.mul_(0.5)
        _, yolo3 = model(x.cuda())
    assert not torch.equal(yolo, yolo3)


def test_cpu_input_is_rejected_loudly(hip_lib):
    from millieye_amd import hip
    model = ph.make_darknet("yolov3-tiny-12").cuda()
    with pytest.raises(hip.MeError):
        model(ph.frames("cpu", 1, 96))


def test_yolo_loss_value_vs_reference_golden(hip_lib):
    """Darknet.forward(x, targets) -> (loss, featuremap, yolo_outputs): loss value and per-scale metrics
    against the reference's own numbers (row a6; the value only - the detector backward is not built)."""
    import os
    import numpy as np
    from millieye_amd import synth
    from tests.golden.make_golden import YOLO_LOSS_CASE
    name, cfg, n, s = YOLO_LOSS_CASE
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))
    model = ph.make_darknet(cfg, tag=name).cuda()
    x = torch.from_numpy(synth.uniform(name + "/x", (n, 3, s, s))).cuda()
    loss, fm, yolo = model(x, torch.from_numpy(g["targets"]))
    assert abs(float(loss) - float(g["loss"])) <= 1e-3 * abs(float(g["loss"]))
    for i, yl in enumerate(model.yolo_layers):
        for k, v in yl.metrics.items():
            ref = float(g[f"m{i}/{k}"])
            assert abs(float(v) - ref) <= 1e-3 * max(1.0, abs(ref)), (i, k, v, ref)
    assert tuple(fm.shape) == (n, 256, s // 16, s // 16) and yolo.shape[0] == n
