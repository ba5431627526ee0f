"""-m gpu: the stage-2 input producer and script loops on the HIP path (SURVEY.md section 8 f-3 remainder) against the REAL
reference: ``ListDataset`` batches bit-for-bit (tests/golden/m2_listdataset.npz) and the whole ``module2_mixed/train.py`` run -
losses, AdamW steps, checkpoints, per-epoch ``test_module2.evaluate`` (tests/golden/trainloop_m2_tiny12_s160.npz).

Tolerances of the loop as in tests/test_gpu_train_loop.py: losses 1e-3 relative; an AdamW step moves every element by about
``lr`` whatever its gradient's magnitude, so elements at rounding-noise level may step the other way: ``2 * lr * steps`` per
element (6e-4), 2e-4 on a tensor's mean; evaluation numbers 2e-3 after the first epoch, 2e-2 later (rank-order swaps of
near-equal scores move a 30-detection AP by one PR corner)."""
import os
import random

import numpy as np
import pytest
import torch

from tests import m2_loop_helpers as ml
from tests.golden import make_golden as mg

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_listdataset_batches_bitexact_vs_reference(hip_lib, tmp_path):
    from millieye_amd.module2.datasets import ListDataset
    g = np.load(os.path.join(GOLD, mg.M2_LIST_NAME + ".npz"))
    # (a) plain: every frame incl. the unlabelled one, the grey-scale file and both odd-padding orientations
    lp = mg.write_m2_list_dataset(str(tmp_path), with_unlabelled=True)
    ds = ListDataset(lp, img_size=64, augment=False, multiscale=False)
    _, staged, targets = ds.collate_fn([ds[i] for i in range(len(ds))])
    imgs = staged.to("cuda")
    assert np.array_equal(imgs.cpu().numpy(), g["plain/imgs"])
    assert np.array_equal(targets.numpy(), g["plain/targets"])
    with pytest.raises(Exception):
        staged.to("cpu")  # no CPU path
    # (b) flips + multiscale + shuffling, seeded like the reference run
    c = mg.M2_LIST_AUG
    lp = mg.write_m2_list_dataset(str(tmp_path), with_unlabelled=False)
    ds = ListDataset(lp, img_size=c["img_size"], augment=True, multiscale=True)
    random.seed(c["seed"])
    np.random.seed(c["seed"])
    torch.manual_seed(c["seed"])
    loader = torch.utils.data.DataLoader(ds, batch_size=c["batch"], shuffle=True, num_workers=0, collate_fn=ds.collate_fn)
    b, flips = 0, 0
    while b < c["batches"]:
        for paths, staged, targets in loader:
            imgs = staged.to("cuda").cpu()
            assert imgs.shape[-1] == int(g["aug/sizes"][b])
            assert [os.path.basename(p) for p in paths] == list(g[f"aug/b{b}/paths"]), b
            assert np.array_equal(targets.numpy(), g[f"aug/b{b}/targets"]), b
            assert np.array_equal(imgs[:, :, ::3, ::3].numpy(), g[f"aug/b{b}/imgs_sub"]), b
            assert float(imgs.double().sum()) == float(g[f"aug/b{b}/imgs_sum"]), b
            flips += sum(staged.flips)
            b += 1
            if b == c["batches"]:
                break
    assert 0 < flips < 2 * c["batches"]


@pytest.mark.parametrize("optimizer", ["torch", "hip"])
def test_m2_train_loop_on_gpu_matches_reference_script(hip_lib, tmp_path, optimizer):
    """optimizer = "hip": millieye_amd.optim.AdamW (the loop's default on the GPU) instead of torch.optim.AdamW - the same trajectory of
    the REAL reference's module2_mixed/train.py run."""
    from millieye_amd.module2.my_models import Network
    from millieye_amd import optim
    net = ml.prepare(Network)
    net = net.to(net.device)
    assert net.device.type == "cuda"
    hist = ml.run(net, tmp_path, optimizer_cls=optim.AdamW if optimizer == "hip" else None)
    ml.check(net, hist, tmp_path, loss_tol=1e-3, param_atol=6.5e-4, sum_tol=2e-4, ap_tol=2e-3, late_ap_tol=2e-2)


def test_m2_evaluate_with_listdataset_end_to_end(hip_lib, tmp_path):
    """``evaluate`` with its own ListDataset + DataLoader (no injected batches) runs on files and returns the reference's
    tuple shape; boxes-per-image bookkeeping covers every frame."""
    from millieye_amd.module2.my_models import Network, define_yolo
    from millieye_amd.module2.test_module2 import evaluate
    from tests.parity_helpers import cfg_path
    c = mg.M2_LOOP_CASE
    net = Network(define_yolo(cfg_path(c["cfg"])), c["conf"])
    mg.m2_train_fill_(net, c["name"])
    net = net.to(net.device)
    lp = mg.write_m2_list_dataset(str(tmp_path), with_unlabelled=False)
    torch.manual_seed(0)
    precision, recall, AP, f1, ap_class, box_stat, pr = evaluate(net, lp, 0.5, 0.01, 0.5, img_size=160, batch_size=2)
    n_frames = sum(1 for fr in mg.M2_LIST_FRAMES if fr[3] > 0)
    assert len(box_stat["after"]) == 1 + n_frames and len(pr) == 3
    assert len(precision) == len(recall) == len(AP) == len(f1) == len(ap_class) > 0
