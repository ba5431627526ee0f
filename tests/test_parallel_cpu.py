"""CPU (-m "not gpu"): the N>1 path - frame sharding / output merging and the flat-bucket SUM all-reduce,
run as two real processes over gloo (127.0.0.1)."""
import os
import socket
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_and_merge_roundtrip():
    from millieye_amd import parallel as par
    n = 7
    images = torch.arange(n * 3 * 4 * 4, dtype=torch.float32).view(n, 3, 4, 4)
    maps = torch.arange(n * 3, dtype=torch.float32).view(n, 3, 1, 1)
    rboxes = torch.tensor([[0, .1, .1, .2, .2], [3, .3, .3, .4, .4], [6, .5, .5, .6, .6], [6, .1, .2, .3, .4]])
    targets = torch.tensor([[2, 0, .5, .5, .1, .1], [5, 0, .4, .4, .2, .2]])
    world = 3
    assert [par.shard_range(n, r, world) for r in range(world)] == [(0, 3), (3, 5), (5, 7)]
    seen, outs, frames = 0, [], []
    for r in range(world):
        im, mp_, rb, tg = par.shard_batch(images, maps, rboxes, targets, r, world)
        lo, hi = par.shard_range(n, r, world)
        assert torch.equal(im, images[lo:hi]) and torch.equal(mp_, maps[lo:hi])
        assert all(0 <= v < hi - lo for v in rb[:, 0].tolist() + tg[:, 0].tolist())
        seen += len(rb)
        rows = torch.zeros((len(rb), 8))
        rows[:, 0] = rb[:, 0]
        rows[:, 1:5] = rb[:, 1:]
        outs.append(rows)
        frames.append(hi - lo)
    assert seen == len(rboxes)
    merged = par.merge_outputs(outs, frames)
    assert sorted(merged[:, 0].tolist()) == sorted(rboxes[:, 0].tolist())
    assert torch.equal(merged[:, 1:5][merged[:, 0].argsort(stable=True)], rboxes[:, 1:][rboxes[:, 0].argsort(stable=True)])


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from millieye_amd import parallel as par
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(6, 4), torch.nn.Linear(4, 2), torch.nn.Linear(2, 1))
    model[2].weight.requires_grad = False  # frozen tensor: not part of the bucket
    x = torch.arange(12, dtype=torch.float32).view(2, 6) + rank
    model[1](model[0](x)).sum().backward() if rank == 0 else model[0](x).sum().backward()  # rank 1: layer 1 has no grad
    nbytes = par.allreduce_gradients(model.parameters())
    res = {k: (None if p.grad is None else p.grad.numpy().copy()) for k, p in model.named_parameters()}  # by value:
    # a tensor would travel as a shared-memory handle that dies with this process
    q.put((rank, nbytes, res))
    dist.barrier()
    dist.destroy_process_group()


def test_allreduce_gradients_gloo_world2():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(2):
        rank, nbytes, res = q.get(timeout=120)
        got[rank] = (nbytes, res)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference: sum of the two ranks' local gradients
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(6, 4), torch.nn.Linear(4, 2), torch.nn.Linear(2, 1))
    x0 = torch.arange(12, dtype=torch.float32).view(2, 6)
    model[1](model[0](x0)).sum().backward()
    model[0](x0 + 1).sum().backward()  # grads accumulate: rank 0 + rank 1
    for rank in (0, 1):
        nbytes, res = got[rank]
        assert nbytes == 4 * (6 * 4 + 4 + 4 * 2 + 2 + 1)  # trainable tensors only, one flat fp32 bucket
        for k, p in model.named_parameters():
            if not p.requires_grad or p.grad is None:
                assert res[k] is None  # frozen, or no gradient on any rank
            else:
                assert torch.allclose(torch.from_numpy(res[k]), p.grad, atol=1e-6), (rank, k)


def _pattern_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MILLIEYE_DP_CHECK="1")
    import torch.distributed as dist
    from millieye_amd import parallel as par
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(3, 2), torch.nn.Linear(2, 1))
    params = list(model.parameters())
    x = torch.ones(1, 3)
    outcome = []
    for step in range(3):
        for p in params:
            p.grad = None
        # steps 0, 1: both ranks produce layer-0 gradients only (the static set); step 2: rank 1 breaks the promise
        (model(x) if (step == 2 and rank == 1) else model[0](x)).sum().backward()
        try:
            par.allreduce_gradients(params, static_pattern=True)
            outcome.append("ok")
        except RuntimeError as exc:
            outcome.append("raised" if "static pattern" in str(exc) else "other:" + str(exc))
    # train.sync_batchnorm_buffers: head running statistics become the mean over the ranks, the detector's stay
    from millieye_amd.train import sync_batchnorm_buffers
    net = torch.nn.Module()
    net.base_detector = torch.nn.BatchNorm1d(2)
    net.head = torch.nn.BatchNorm1d(3)
    with torch.no_grad():
        net.head.running_mean.fill_(float(rank))
        net.head.running_var.fill_(1.0 + 2 * rank)
        net.base_detector.running_mean.fill_(float(rank))
    n_synced = sync_batchnorm_buffers(net)
    outcome.append((n_synced, net.head.running_mean.tolist(), net.head.running_var.tolist(),
                    net.base_detector.running_mean.tolist()))
    q.put((rank, outcome, getattr(params[0], "_me_dp_state").steps))
    dist.barrier()
    dist.destroy_process_group()


def test_static_pattern_promise_is_enforced():
    """with ``static_pattern=True`` a rank that sees another rank's gradient outside the assumed set must not
    diverge silently.  MILLIEYE_DP_CHECK=1 reads the reduced flags every step: the rank that kept its promise raises; the
    bookkeeping hangs on the Parameter object (no id()-keyed global)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_pattern_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict((r, (o, n)) for r, o, n in (q.get(timeout=120) for _ in range(2)))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0][0][:3] == ["ok", "ok", "raised"], got   # rank 0's static set lacks layer 1: refuses to go on
    assert got[1][0][:3] == ["ok", "ok", "ok"], got       # rank 1's own pattern changed: it re-reads the flags, as documented
    for r in (0, 1):
        assert got[r][0][3] == (6, [0.5] * 3, [2.0] * 3, [float(r)] * 2), got[r][0][3]
    assert got[0][1] == got[1][1] == 3


# ---------------------------------------------------------------------------------------------------------------------
# data parallelism of the REAL stage-3 step: shard_batch + one flat-bucket all-reduce must reproduce the 1-way step.
# The compute is the oracle's CPU training step (the HIP path needs a GPU; tests/test_gpu_parallel.py does the same on
# it): eval-mode head BatchNorm and balance_factor = "all negatives" make every loss term a plain sum over frames.
# ---------------------------------------------------------------------------------------------------------------------
DP_CASE = ("dp_tiny12_s160_n4", "yolov3-tiny-12", 4, 160, 0.2)


def _dp_problem():
    import numpy as np
    from millieye_amd import cfgs, synth
    from millieye_amd.my_models import Network, define_yolo
    from tests.parity_helpers import cfg_path
    from tests.golden.make_golden import train_inputs
    name, cfg, n, s, conf = DP_CASE
    net = Network(define_yolo(cfg_path(cfg)), conf)
    synth.fill_network_(net, name)
    x, maps, rboxes = train_inputs(name, n, s)
    # targets on top of two radar boxes per frame (radar rows are proposals of class 0, so IoU-positive samples exist)
    tg = []
    for row in rboxes.numpy():
        i, x1, y1, x2, y2 = row
        tg.append([i, 0, (x1 + x2) / 2, (y1 + y2) / 2, (x2 - x1) * 1.02, (y2 - y1) * 0.98])
    targets = torch.tensor(np.array(tg, dtype=np.float32))
    return cfgs.KNOWN[cfg](), net, x, maps, rboxes, targets, conf


def _dp_step(cfg_text, sd, x, maps, rboxes, targets, conf):
    import random
    from oracle import network_ref
    random.seed(0)
    return network_ref.network_train_step(cfg_text, sd, x, maps, rboxes, targets, conf_thresh=conf, bn_training=False,
                                          balance_factor=10 ** 9)


def _dp_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    import torch.distributed as dist
    from millieye_amd import parallel as par
    from millieye_amd.train_path import head_parameters
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg_text, net, x, maps, rboxes, targets, conf = _dp_problem()
    xs, ms, rb, tg = par.shard_batch(x, maps, rboxes, targets, rank, world)
    res = _dp_step(cfg_text, net.state_dict(), xs, ms, rb, tg, conf)
    heads = head_parameters(net)
    names = [k for k, _ in net.named_parameters() if not k.startswith("base_detector.")]
    for step in range(2):  # second call: the steady-state path (static pattern, no flag read-back)
        for name, p in zip(names, heads):  # (clone: the reduced values are written into p.grad in place)
            p.grad = None if res["grads"][name] is None else res["grads"][name].clone()
        nbytes = par.allreduce_gradients(heads, static_pattern=True)
    out = {name: (None if p.grad is None else p.grad.numpy().copy()) for name, p in zip(names, heads)}
    q.put((rank, nbytes, float(res["loss"]), res["num_img"], out))
    dist.barrier()
    dist.destroy_process_group()


def test_dp_stage3_step_equals_one_way():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(2):
        rank, nbytes, loss, num_img, grads = q.get(timeout=600)
        got[rank] = (nbytes, loss, num_img, grads)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    cfg_text, net, x, maps, rboxes, targets, conf = _dp_problem()
    ref = _dp_step(cfg_text, net.state_dict(), x, maps, rboxes, targets, conf)
    assert ref["n_pos"] > 0 and ref["num_img"] > 0
    assert got[0][1] + got[1][1] == __import__("pytest").approx(float(ref["loss"]), rel=1e-5)  # the loss is a sum over frames
    assert got[0][1] > 0 and got[1][1] > 0 and got[0][2] + got[1][2] == ref["num_img"]
    n_checked = 0
    for name, g in ref["grads"].items():
        for rank in (0, 1):
            mine = got[rank][3][name]
            if g is None:
                assert mine is None, name
                continue
            scale = max(float(g.abs().max()), 1e-6)
            assert float((torch.from_numpy(mine) - g).abs().max()) <= 2e-5 * scale, (name, rank)
        n_checked += g is not None
    assert n_checked >= 20
    assert got[0][0] == got[1][0] == 4 * sum(p.numel() for p in net.parameters()
                                            if p.requires_grad) - 4 * sum(p.numel() for p in net.base_detector.parameters())


# ---------------------------------------------------------------------------------------------------------------------
# GradChunkReducer (detector training, row a6 / e): gradients leave in production order, one collective per chunk
# ---------------------------------------------------------------------------------------------------------------------
def _chunk_worker(rank, world, port, q):
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    from millieye_amd import parallel as par
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(100 + rank)
    shapes = [(7,), (3, 5), (64, 3, 3, 3), (1,), (300,), (2, 2)]
    grads = [torch.randn(s, generator=g) for s in shapes]
    red = par.GradChunkReducer(chunk_bytes=1024)
    red.begin("cpu")
    for i, t in enumerate(grads):
        red.push(f"p{i}", t)
    done = red.finish()
    # by value (numpy): a tensor would travel as a shared-memory handle that dies with this process
    q.put((rank, red.chunks_last, red.bytes_last, {k: v.numpy().copy() for k, v in done.items()},
           [t.numpy().copy() for t in grads]))
    dist.barrier()
    dist.destroy_process_group()


def test_grad_chunk_reducer_two_ranks_gloo():
    """Two ranks push six gradients of uneven sizes with a 1 KB chunk limit: three collectives (the 6.9 KB conv weight closes
    its chunk alone), every reduced tensor = the sum of the two ranks' tensors bit for bit, returned in the producer's shape,
    names preserved, byte count = every gradient once."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_chunk_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(2):
        rank, chunks, nbytes, done, mine = q.get(timeout=300)
        got[rank] = (chunks, nbytes, {k: torch.from_numpy(v) for k, v in done.items()}, [torch.from_numpy(t) for t in mine])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    total = [a + b for a, b in zip(got[0][3], got[1][3])]
    for rank in (0, 1):
        chunks, nbytes, done, _mine = got[rank]
        assert chunks == 3 and nbytes == 4 * sum(t.numel() for t in total)
        assert sorted(done) == [f"p{i}" for i in range(6)]
        for i, t in enumerate(total):
            assert done[f"p{i}"].shape == t.shape and torch.equal(done[f"p{i}"], t), i


def test_overlap_detector_allreduce_needs_a_process_group():
    """Without a process group nothing is attached (bench.py then says "all-reduce skipped")."""
    from millieye_amd import parallel as par

    class M:
        pass
    m = M()
    assert par.overlap_detector_allreduce(m) is None and "_grad_reducer" not in m.__dict__


def _epoch_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from millieye_amd import parallel as par
    dist.init_process_group("gloo", rank=rank, world_size=world)
    data = torch.utils.data.TensorDataset(torch.arange(16))
    sampler = torch.utils.data.distributed.DistributedSampler(data, shuffle=True)
    loader = torch.utils.data.DataLoader(data, batch_size=2, sampler=sampler)
    orders = []
    for epoch in range(3):
        par.begin_epoch(loader, epoch)
        orders.append([int(v) for (b,) in loader for v in b])
    # the averaging form of the chunk reducer: mean over the ranks instead of the sum
    red = par.GradChunkReducer(1 << 10, average=True)
    red.begin("cpu")
    red.push("g", torch.full((5,), float(rank + 1)))
    mean = red.finish()["g"].tolist()
    # uneven batch counts are refused on every rank
    uneven = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(torch.arange(4 + 2 * rank)), batch_size=2)
    try:
        par.begin_epoch(uneven, 0)
        refused = False
    except RuntimeError:
        refused = True
    q.put((rank, orders, mean, refused))
    dist.barrier()
    dist.destroy_process_group()


def test_begin_epoch_reshuffles_the_distributed_sampler_two_ranks():
    """both training loops call ``parallel.begin_epoch`` at the top of every epoch - the sampler's
    permutation changes from epoch to epoch, the two ranks' shards stay disjoint and cover the dataset, an uneven shard is
    refused; and ``GradChunkReducer(average=True)`` returns the mean over the ranks."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_epoch_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(2):
        rank, orders, mean, refused = q.get(timeout=300)
        got[rank] = (orders, mean, refused)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for epoch in range(3):
        a, b = got[0][0][epoch], got[1][0][epoch]
        assert sorted(a + b) == list(range(16)) and not set(a) & set(b)
    for rank in (0, 1):
        orders, mean, refused = got[rank]
        assert orders[0] != orders[1] and orders[1] != orders[2]  # (without set_epoch all three are identical)
        assert mean == [1.5] * 5 and refused
    import inspect
    from millieye_amd import train as t3
    from millieye_amd.module2 import train as t2
    for mod in (t3, t2):
        assert "parallel.begin_epoch(dataloader, epoch" in inspect.getsource(mod.train_loop)
