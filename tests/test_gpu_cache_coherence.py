"""GPU: the packed device copies of the parameters (engine.ConvWeights, my_models._HeadPack: OHWI weights, folded
BatchNorm scale/shift) must follow every writer of the module tree - including the ones that do not bump the autograd
version counter (HIP kernels writing running statistics through raw pointers; reference-style ``param.data.copy_()``)."""
import os
import random

import numpy as np
import pytest
import torch

from millieye_amd import cfgs, synth
from tests import parity_helpers as ph
from tests.golden.make_golden import TRAIN_CASE, train_inputs

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _rows_close(got, ref, what):
    got, ref = got.detach().cpu(), torch.as_tensor(ref)
    assert got.shape == ref.shape, f"{what}: {tuple(got.shape)} vs {tuple(ref.shape)}"
    assert torch.equal(got[:, 0], ref[:, 0]), what
    ph.assert_close(got, ref, 1e-3, what)


def test_eval_after_training_step_uses_the_updated_running_statistics(hip_lib):
    """Stage-3 default flow (train.py:146-149 freezes the stage-2 tensors): with conv_0 / batch_norm_0 frozen only the
    running statistics of the score-map BatchNorm change during a step - written by ``me_bn_train_fwd_f32`` through raw
    pointers.  The eval-mode forward that follows (train.py's per-epoch ``evaluate``) must fold the NEW statistics."""
    from oracle import network_ref
    from millieye_amd.my_models import Network, define_yolo
    name, cfg, n, s, conf, seed = TRAIN_CASE
    g = np.load(os.path.join(GOLD, name + ".npz"))
    net = Network(define_yolo(ph.cfg_path(cfg)), conf)
    synth.fill_network_(net, name)
    for k, p in net.named_parameters():
        if k.startswith("img_cnn_layers."):
            p.requires_grad = False
    net = net.cuda().eval()
    x, maps, rboxes = train_inputs(name, n, s)
    text = cfgs.KNOWN[cfg]()
    with torch.no_grad():
        out_before = net(x.cuda(), maps.cuda(), rboxes.clone().cuda(), 0)  # packs are built (and cached) here
    _rows_close(out_before, network_ref.network_forward(text, {k: v.cpu() for k, v in net.state_dict().items()}, x, maps,
                                                        rboxes, 0, conf_thresh=conf), "eval before the step")
    stats0 = net.img_cnn_layers.net[1].running_mean.clone()
    net.train()
    net.base_detector.eval()
    random.seed(seed)
    for _ in range(3):  # momentum 0.1: three steps move the statistics well away from the initial ones
        loss, *_ = net(x.cuda(), maps.cuda(), rboxes.clone().cuda(), torch.from_numpy(g["targets"]).clone())
        loss.backward()
    assert not torch.equal(stats0, net.img_cnn_layers.net[1].running_mean)
    net.eval()
    with torch.no_grad():
        out_after = net(x.cuda(), maps.cuda(), rboxes.clone().cuda(), 0)
    ref_after = network_ref.network_forward(text, {k: v.cpu() for k, v in net.state_dict().items()}, x, maps, rboxes, 0,
                                            conf_thresh=conf)
    _rows_close(out_after, ref_after, "eval after the step (updated running statistics)")
    assert out_after.shape != out_before.shape or not torch.allclose(out_after, out_before, atol=1e-4), \
        "the case must be sensitive to the statistics"


def test_weights_loaded_after_a_forward_are_picked_up(hip_lib, tmp_path):
    """``load_darknet_weights`` into a model that already ran (warm-up, autotune, a second weights file): the next
    forward must use the new weights.  Also the reference-style ``.data`` write + ``invalidate_weights()``."""
    from oracle import darknet_ref
    a = ph.make_darknet("yolov3-tiny-12", tag="coherence/a")
    b = ph.make_darknet("yolov3-tiny-12", tag="coherence/b")
    path = str(tmp_path / "b.weights")
    b.save_darknet_weights(path, cutoff=len(b.module_defs))
    x = ph.frames("coherence/x", 2, 96)
    text = ph.cfg_text("yolov3-tiny-12")
    ref_a = darknet_ref.darknet_forward(text, a.state_dict(), x)[1]
    ref_b = darknet_ref.darknet_forward(text, b.state_dict(), x)[1]
    assert not torch.allclose(ref_a, ref_b, atol=1e-3)
    a = a.cuda()
    with torch.no_grad():
        ph.assert_close(a(x.cuda())[1].cpu(), ref_a, 1e-3, "first weights")
        a.load_darknet_weights(path)
        ph.assert_close(a(x.cuda())[1].cpu(), ref_b, 1e-3, "after load_darknet_weights")
        # reference style: write through .data (no version bump) and announce it
        sd_a = ph.make_darknet("yolov3-tiny-12", tag="coherence/a").state_dict()
        for k, v in a.state_dict().items():
            v.data.copy_(sd_a[k])
        a.invalidate_weights()
        ph.assert_close(a(x.cuda())[1].cpu(), ref_a, 1e-3, "after .data writes + invalidate_weights()")


def test_reference_written_weights_file_through_the_hip_path(hip_lib, tmp_path):
    """Row a7 on the device: a ``.weights`` file written by the real reference (tests/golden/weights_io/) is loaded with
    ``init_yolo`` and run through the HIP detector; rows must equal the reference's own forward of that file.  The mini cfg
    also walks the narrow-channel kernels (cin 3/4/8/16, cout 4..21)."""
    from millieye_amd.my_models import init_yolo
    from millieye_amd.yolov3.models import Darknet
    from tests.golden.make_golden import WEIGHTS_DIR, write_mini17_cfg
    g = np.load(os.path.join(GOLD, "weights_io.npz"))
    m = Darknet(write_mini17_cfg(str(tmp_path))).eval().cuda()
    x = torch.from_numpy(synth.uniform("weights_io/x", (2, 3, 64, 64))).cuda()
    with torch.no_grad():
        m(x)                                                   # a forward BEFORE loading (default init)
        init_yolo(m, os.path.join(WEIGHTS_DIR, "mini17_full.weights"))
        _fm, yolo = m(x)
    ph.assert_close(yolo.cpu(), torch.from_numpy(g["yolo"]), 1e-3, "HIP forward of the reference-written weights file")
