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


def _m2_dp_worker(rank, world, port, q, ckpt_dir):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from millieye_amd.module2.train import train_loop
    c, g = mg.M2_LOOP_CASE, ml.golden()
    targets = {k[len("targets/"):]: g[k] for k in g.files if k.startswith("targets/")}
    net = ml.prepare(ml.OracleBackedM2Network)

    class Shard:  # rank r trains on batches r, r + 1 of the stand-in set: different data, equal counts
        def __len__(self):
            return 2

        def __iter__(self):
            b = mg.m2_loop_batches(c["name"], "train", c["train_batches"], c["batch"], c["size"], targets)
            return iter(b[rank:rank + 2])

    random.seed(c["seed"] + rank)
    torch.manual_seed(c["seed"] + rank)
    hist = train_loop(net, Shard(), epochs=2, gradient_accumulations=1, evaluate_fn=None,
                      checkpoint_dir=os.path.join(ckpt_dir, f"r{rank}"), log=lambda *_: None)
    heads = {k: v.detach().clone() for k, v in net.named_parameters() if not k.startswith("base_detector.")}
    heads.update({"buf/" + k: v.detach().clone().float() for k, v in net.named_buffers()
                  if "running_" in k and not k.startswith("base_detector.")})
    q.put((rank, hist["steps"], hist["losses"], {k: v.numpy() for k, v in heads.items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_m2_train_loop_data_parallel_replicas_stay_identical(tmp_path):
    """``module2/train.py:train_loop`` under ``torch.distributed`` (gloo, world size 2): each rank trains on its own batches,
    the gradients are SUM-all-reduced in one bucket before every AdamW step - the replicas' head parameters stay bit-identical
    although their losses differ, they move away from the start, and only rank 0 writes checkpoints."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_m2_dp_worker, args=(r, 2, port, q, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(2):
        rank, steps, losses, heads = q.get(timeout=600)
        got[rank] = (steps, losses, heads)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert got[0][0] == got[1][0] == [0, 1, 2, 3]
    assert got[0][1] != got[1][1], "the ranks saw different batches"
    start = dict(ml.prepare(ml.OracleBackedM2Network).named_parameters())
    moved = 0
    for k, v in got[0][2].items():
        assert np.array_equal(v, got[1][2][k]), k   # parameters AND the averaged BatchNorm running statistics
        if not k.startswith("buf/"):
            moved += int(not np.array_equal(v, start[k].detach().numpy()))
    assert moved >= 10
    assert sorted(os.listdir(tmp_path / "r0" )) == ["ckpt_0.pth", "ckpt_1.pth"] and not os.path.exists(tmp_path / "r1" / "ckpt_0.pth")
