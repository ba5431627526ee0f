"""CPU (-m "not gpu"): the stage-2 input producer and script loops (SURVEY.md section 8 f-3 remainder) against the REAL
reference: ``ListDataset`` (tests/golden/m2_listdataset.npz) and ``module2_mixed/train.py`` + ``test_module2.evaluate``
(tests/golden/trainloop_m2_tiny12_s160.npz).  The product's device kernels cannot run here: the oracle renders the frames
(``oracle/datasets_ref.py``) and does the network arithmetic (``oracle/network_m2_ref.py``); what is under test on the
product side is the host logic - label arithmetic, the random streams (flips / sizes / shuffles), collate, step cadence,
checkpoint names, the evaluate tail."""
import os
import random

import numpy as np
import torch

from millieye_amd.module2.datasets import ListDataset
from oracle import datasets_ref
from tests import m2_loop_helpers as ml
from tests.golden import make_golden as mg

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _label_of(path):
    return path.replace("images", "labels").replace(".png", ".txt")


def test_oracle_listdataset_plain_matches_reference(tmp_path):
    g = np.load(os.path.join(GOLD, mg.M2_LIST_NAME + ".npz"))
    lp = mg.write_m2_list_dataset(str(tmp_path), with_unlabelled=True)
    paths = [ln.rstrip() for ln in open(lp)]
    items = [datasets_ref.list_item(p, _label_of(p), False) for p in paths]
    for k, (img, tg) in enumerate(items):
        assert list(img.shape) == list(g[f"plain/item{k}/img_shape"])
        ref = g[f"plain/item{k}/targets"]
        assert (tg is None and ref.shape[0] == 0) or np.array_equal(tg.numpy(), ref), k
    imgs, targets = datasets_ref.list_collate(items, 64)
    assert np.array_equal(imgs.numpy(), g["plain/imgs"])
    assert np.array_equal(targets.numpy(), g["plain/targets"])
    # the product's host side on the same files: same targets, flags off, same collate re-numbering
    ds = ListDataset(lp, img_size=64, augment=False, multiscale=False)
    got = [ds[i] for i in range(len(ds))]
    for k, (_, (frame, flip), tg) in enumerate(got):
        assert frame.dtype == torch.uint8 and frame.shape[2] == 3 and flip is False
        ref = g[f"plain/item{k}/targets"]
        assert (tg is None and ref.shape[0] == 0) or np.array_equal(tg.numpy(), ref), k
    _, staged, tgs = ds.collate_fn(got)
    assert np.array_equal(tgs.numpy(), g["plain/targets"]) and tuple(staged.shape) == tuple(g["plain/imgs"].shape)


def test_listdataset_augmented_multiscale_streams_match_reference(tmp_path):
    """Seeded run through a shuffling DataLoader: batch composition (torch's sampler stream), flips (numpy's), sizes
    (python's ``random``) and targets are the reference's; the oracle renders what the device kernel will."""
    g = np.load(os.path.join(GOLD, mg.M2_LIST_NAME + ".npz"))
    c = mg.M2_LIST_AUG
    lp = mg.write_m2_list_dataset(str(tmp_path), with_unlabelled=False)
    ds = ListDataset(lp, img_size=c["img_size"], augment=True, multiscale=True)
    random.seed(c["seed"])
    np.random.seed(c["seed"])
    torch.manual_seed(c["seed"])
    loader = torch.utils.data.DataLoader(ds, batch_size=c["batch"], shuffle=True, num_workers=0, collate_fn=ds.collate_fn)
    b, sizes, flips = 0, [], 0
    while b < c["batches"]:
        for paths, staged, targets in loader:
            assert [os.path.basename(p) for p in paths] == list(g[f"aug/b{b}/paths"]), b
            assert np.array_equal(targets.numpy(), g[f"aug/b{b}/targets"]), b
            rendered = []
            for p, frame, flip in zip(paths, staged.frames, staged.flips):
                img, _ = datasets_ref.list_item(p, "/nonexistent", False)
                assert tuple(img.shape[1:]) == (max(frame.shape[:2]),) * 2
                if flip:
                    img = torch.flip(img, [-1])
                rendered.append(datasets_ref.resize(img, staged.size))
                flips += int(flip)
            imgs = torch.stack(rendered)
            assert np.array_equal(imgs[:, :, ::3, ::3].numpy(), g[f"aug/b{b}/imgs_sub"]), b
            assert float(imgs.double().sum()) == float(g[f"aug/b{b}/imgs_sum"]), b
            sizes.append(staged.size)
            b += 1
            if b == c["batches"]:
                break
    assert sizes == list(g["aug/sizes"]) and 0 < flips < 2 * c["batches"]


def test_m2_train_loop_harness_matches_reference_script(tmp_path):
    net = ml.prepare(ml.OracleBackedM2Network)
    hist = ml.run(net, tmp_path)
    ml.check(net, hist, tmp_path, loss_tol=1e-5, param_atol=2e-5, sum_tol=1e-6, ap_tol=1e-6)
