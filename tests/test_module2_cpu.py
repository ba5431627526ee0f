"""CPU (-m "not gpu"): pin oracle/network_m2_ref.py (stage-2 Network.forward, module2_mixed/my_models.py:299-364) to the
outputs of the real module-2 reference (tests/golden/network_m2_*.npz), and check the product's module-2 parameter tree."""
import os

import numpy as np
import pytest
import torch

from millieye_amd import cfgs, synth
from oracle import network_m2_ref
from tests.golden.make_golden import M2_CASES, m2_fill_
from tests.parity_helpers import cfg_path

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("name,cfg,n,s,conf", M2_CASES)
def test_oracle_module2_matches_reference(name, cfg, n, s, conf):
    from millieye_amd.module2.my_models import Network, define_yolo
    g = np.load(os.path.join(GOLD, name + ".npz"))
    net = m2_fill_(Network(define_yolo(cfg_path(cfg)), conf), name)
    x = torch.from_numpy(synth.uniform(name + "/x", (n, 3, s, s)))
    out = network_m2_ref.network_m2_forward(cfgs.KNOWN[cfg](), net.state_dict(), x, conf_thresh=conf)
    assert out.shape == g["output"].shape
    assert np.allclose(out.numpy(), g["output"], rtol=1e-5, atol=1e-5)


def test_module2_parameter_names_match_reference():
    """The names stage-3 training picks out of a stage-2 checkpoint (train.py:119-130) exist with the right shapes."""
    from millieye_amd.module2.my_models import Network, define_yolo
    from millieye_amd.train import NAMES_M2
    net = Network(define_yolo(cfg_path("yolov3-tiny-12")), 0.2)
    sd = net.state_dict()
    assert all(k in sd for k in NAMES_M2)
    assert sd["refinement_head.net2.0.weight"].shape == (13, 256) and sd["ensemble_head.fc2.0.weight"].shape == (2, 416)
    assert sum(p.numel() for n_, p in net.named_parameters() if not n_.startswith("base_detector.")) == \
        (256 * 490 + 490 + 2 * 490) + (490 * 256 + 256) + (256 * 4 + 4) + (256 * 13 + 13) + (2 * 32 + 32) + (416 * 2 + 2)
    from millieye_amd import hip
    with pytest.raises(hip.MeError):  # no CPU path: training needs CUDA tensors too
        net.train()(torch.zeros(1, 3, 32, 32), torch.zeros(0, 6))


def test_oracle_module2_training_step_matches_reference():
    """oracle/network_m2_ref.network_m2_train_step (Dropout mask from the seeded CPU generator, python-random negative
    sampling, focal + confidence + category + SmoothL1 losses, the class-label row quirk) against the real module-2
    reference's training step (tests/golden/train_m2_tiny12_s160_n2.npz)."""
    import random
    from millieye_amd.module2.my_models import Network, define_yolo
    from tests.golden.make_golden import M2_TRAIN_CASE, m2_train_fill_
    name, cfg, n, s, conf, seed = M2_TRAIN_CASE
    g = np.load(os.path.join(GOLD, name + ".npz"))
    net = m2_train_fill_(Network(define_yolo(cfg_path(cfg)), conf), name)
    x = torch.from_numpy(synth.uniform(name + "/x", (n, 3, s, s)))
    random.seed(seed)
    torch.manual_seed(seed)
    res = network_m2_ref.network_m2_train_step(cfgs.KNOWN[cfg](), net.state_dict(), x, torch.from_numpy(g["targets"]),
                                               conf_thresh=conf)
    assert abs(float(res["loss"]) - float(g["loss"])) <= 1e-5 * abs(float(g["loss"])), (float(res["loss"]), float(g["loss"]))
    assert res["n_pos"] == int(g["n_pos"]) > 0
    assert np.allclose(res["output"].numpy(), g["output"], rtol=1e-5, atol=1e-5)
    seen = 0
    for key in g.files:
        if key.startswith("gnorm/"):
            k = key[6:]
            gr = res["grads"][k]
            assert abs(float(gr.double().norm()) - float(g[key])) <= 1e-4 * max(1e-6, float(g[key])), k
            assert np.allclose(gr.flatten()[::max(1, gr.numel() // 64)].numpy(), g["gsamp/" + k], rtol=1e-4, atol=1e-6), k
            seen += 1
        elif key.startswith("buf/"):
            assert np.allclose(res["buffers"][key[4:]].numpy(), g[key], rtol=1e-5, atol=1e-6), key
    assert seen == 14
