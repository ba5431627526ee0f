"""GPU: stage-2 Network.forward (millieye_amd/module2, me_m2_heads_f32 through the C ABI) against the oracle and the real
module-2 reference's golden rows.  Tolerance 1e-3 (north_star, fp32); rows are compared in order - the new confidences
differ by >1e-5 between neighbours in these fixtures, so the sort order is stable."""
import os

import numpy as np
import pytest
import torch

from millieye_amd import cfgs, synth
from tests.golden.make_golden import M2_CASES, m2_fill_
from tests.parity_helpers import assert_close, cfg_path

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("name,cfg,n,s,conf", M2_CASES)
def test_module2_forward_vs_oracle_and_reference(hip_lib, name, cfg, n, s, conf):
    from millieye_amd.module2.my_models import Network, define_yolo
    from oracle import network_m2_ref
    g = np.load(os.path.join(GOLD, name + ".npz"))
    net = m2_fill_(Network(define_yolo(cfg_path(cfg)), conf), name).eval()
    net = net.to(net.device)
    x = torch.from_numpy(synth.uniform(name + "/x", (n, 3, s, s)))
    out = net(x.cuda())
    assert out.device.type == "cpu" and out.shape[1] == 8          # the reference returns the rows on the CPU (:361)
    ref, internals = network_m2_ref.network_m2_forward(cfgs.KNOWN[cfg](), {k: v.cpu() for k, v in net.state_dict().items()},
                                                       x, conf_thresh=conf, return_internals=True)
    k = len(internals["boxes"])
    assert int(net._last["n_boxes"].item()) == k
    assert_close(net._last["boxes"][:k].cpu(), internals["boxes"], 1e-3, "boxes")
    assert_close(net._last["regress"][:k].cpu(), internals["regress"], 1e-3, "regress")
    assert_close(net._last["refine"][:k].cpu(), internals["refine"], 1e-3, "refine")
    assert_close(net._last["mask"][:k].cpu(), internals["masks"][:, 1], 1e-3, "mask")
    assert out.shape == ref.shape == g["output"].shape
    assert torch.equal(out[:, 0], ref[:, 0]) and torch.equal(out[:, 7], ref[:, 7]), "row order / classes differ"
    assert_close(out, ref, 1e-3, "output vs oracle")
    assert_close(out, torch.from_numpy(g["output"]), 1e-3, "output vs reference golden")
    # a higher refine_threshold drops rows, like masks[:, 1] > thr (:349)
    net.refine_threshold = float(np.median(g["output"][:, 5]))
    assert 0 < net(x.cuda()).shape[0] < out.shape[0]
