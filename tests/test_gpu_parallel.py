"""GPU: the N-rank path on the hardware this suite gets (one GPU): ``bench.py`` starting its own ranks under
``torch.distributed.run``, RCCL initialised on the rank's GPU, and the stage-3 gradient bucket going through a real
``ncclAllReduce`` (world size 1 on a 1-GPU lease; the same code path as N > 1)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(extra, env_extra):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
           "--prewarm-seconds", "0.1", "--no-cpu-baseline", "--no-bf16-line", "--no-accuracy"] + extra
    res = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-4000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_starts_its_own_ranks_and_trains_through_rccl(hip_lib):
    """``BENCH_FORCE_SPAWN`` takes the self-launch branch that ``--gpus N`` (N > 1) takes without a launcher; the child
    rank initialises RCCL, all-reduces a probe and then the real stage-3 head bucket every step."""
    out = _bench(["--workload", "train", "--cfg", "yolov3-tiny-12", "--size", "160", "--batch", "4"],
                 {"BENCH_FORCE_SPAWN": "1"})
    assert out["n_gpus"] == 1 and out["config"]["rccl_ranks"] == 1
    from millieye_amd.my_models import Network, define_yolo
    from millieye_amd.train_path import head_parameters
    from tests import parity_helpers as ph
    heads = head_parameters(Network(define_yolo(ph.cfg_path("yolov3-tiny-12")), 0.2))
    assert out["config"]["grad_bucket_bytes"] == 4 * sum(p.numel() for p in heads)  # every head tensor, one bucket
    assert out["value"] > 0 and out["config"]["loss_last_step"] == out["config"]["loss_last_step"]  # not NaN


def test_bench_refuses_a_world_size_that_disagrees_with_gpus(hip_lib):
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1"], env=env,
                         cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert res.returncode != 0 and "must agree" in (res.stderr + res.stdout)


def test_bench_more_gpus_than_visible_fails_loudly(hip_lib):
    import torch
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1"], env=env,
                         cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert res.returncode != 0 and "visible" in (res.stderr + res.stdout)


# ---------------------------------------------------------------------------------------------------------------------
# data parallelism of the stage-3 step ON THE HIP PATH: two processes share the one leased GPU, each runs
# Network.forward(..., targets) + loss.backward() through libmillieye_hip on its shard_batch() of one global batch and the
# gradients go through allreduce_gradients (gloo over CUDA tensors: the lease has one GPU, and RCCL refuses two ranks on
# one device; the collective call site is the one the RCCL runs use).  eval()-mode model + balance_factor = "all
# negatives" make every loss term a plain sum over frames, so the reduced gradients must equal the 1-way HIP step - and the
# CPU oracle's (tests/test_parallel_cpu.py does the same with the oracle as the compute).
# ---------------------------------------------------------------------------------------------------------------------
def _hip_dp_step(net, x, maps, rboxes, targets):
    import random
    import torch
    random.seed(0)
    net.balance_factor = 10 ** 9
    loss, output, metric, _att = net(x.cuda(), maps.cuda(), rboxes.clone().cuda(), 0, targets.clone())
    loss.backward()
    torch.cuda.synchronize()
    return loss, metric


def _hip_dp_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    import torch.distributed as dist
    from millieye_amd import parallel as par
    from millieye_amd.train_path import head_parameters
    from tests.test_parallel_cpu import _dp_problem
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _text, net, x, maps, rboxes, targets, _conf = _dp_problem()
    net = net.cuda().eval()
    xs, ms, rb, tg = par.shard_batch(x, maps, rboxes, targets, rank, world)
    loss, metric = _hip_dp_step(net, xs, ms, rb, tg)
    heads = head_parameters(net)
    nbytes = par.allreduce_gradients(heads, static_pattern=True)
    names = [k for k, _ in net.named_parameters() if not k.startswith("base_detector.")]
    out = {name: (None if p.grad is None else p.grad.cpu().numpy().copy()) for name, p in zip(names, heads)}
    q.put((rank, nbytes, float(loss.detach()), int(metric["total"]), out))
    dist.barrier()
    dist.destroy_process_group()


def test_dp_stage3_hip_step_two_ranks_equals_one_way(hip_lib):
    import socket
    import torch
    import torch.multiprocessing as mp
    from millieye_amd.train_path import head_parameters
    from oracle import network_ref
    from tests.test_parallel_cpu import _dp_problem, _dp_step
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_hip_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(2):
        rank, nbytes, loss, total, grads = q.get(timeout=900)
        got[rank] = (nbytes, loss, total, grads)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    text, net, x, maps, rboxes, targets, conf = _dp_problem()
    ref = _dp_step(text, net.state_dict(), x, maps, rboxes, targets, conf)       # CPU oracle, 1-way
    net = net.cuda().eval()
    loss1, metric1 = _hip_dp_step(net, x, maps, rboxes, targets)                 # HIP path, 1-way
    assert int(metric1["total"]) == got[0][2] + got[1][2] > 0 and ref["n_pos"] > 0
    assert got[0][1] > 0 and got[1][1] > 0
    assert got[0][1] + got[1][1] == pytest.approx(float(loss1.detach()), rel=1e-4)
    assert got[0][1] + got[1][1] == pytest.approx(float(ref["loss"]), rel=1e-3)
    names = [k for k, _ in net.named_parameters() if not k.startswith("base_detector.")]
    checked = 0
    for name, p in zip(names, head_parameters(net)):
        rg = ref["grads"][name]
        for rank in (0, 1):
            mine = got[rank][3][name]
            if p.grad is None:
                assert mine is None and rg is None, name
                continue
            g1 = p.grad.cpu()
            scale = max(float(g1.abs().max()), 1e-6)
            # two-rank HIP vs one-way HIP: same kernels, sums regrouped by shard (RoI scatters are atomic): 1e-4
            assert float((torch.from_numpy(mine) - g1).abs().max()) <= 1e-4 * scale, (name, rank)
            assert float((torch.from_numpy(mine) - rg).abs().max()) <= 2e-3 * max(float(rg.abs().max()), 1e-6), (name, rank)
        checked += p.grad is not None
    assert checked >= 20
    assert got[0][0] == got[1][0] == 4 * sum(p.numel() for p in head_parameters(net))


# ---------------------------------------------------------------------------------------------------------------------
# detector training (row a6): the gradient stream of the HIP backward leaves in reverse-layer chunks through
# parallel.GradChunkReducer while the backward is still running.  Two processes on the one leased GPU (gloo over CUDA tensors,
# as above).  The YOLO loss terms are per-batch MEANS over the obj / noobj cells (reference yolov3/models.py:240-250), so
# the sum of two shards' gradients is the gradient of loss(shard 0) + loss(shard 1), not of the 1-way loss: the reference
# for the reduced gradients is therefore the sum of the two shards' steps computed one after the other in ONE process
# without any reducer - same kernels, same shards, only the exchange differs.
# ---------------------------------------------------------------------------------------------------------------------
def _det_problem():
    import torch
    from millieye_amd import synth
    from tests import parity_helpers as ph
    model = ph.make_darknet("yolov3-tiny-12", tag="dp-det", trained_like=True)
    x = torch.from_numpy(synth.uniform("dp-det/x", (4, 3, 96, 96)))
    targets = torch.tensor([[i, (3 * i) % 12, 0.3 + 0.1 * (i % 3), 0.4 + 0.05 * i, 0.25, 0.3] for i in range(4)],
                           dtype=torch.float32)
    return model, x, targets


def _det_shard_step(model, x, targets, lo, hi):
    import torch
    tg = targets[(targets[:, 0] >= lo) & (targets[:, 0] < hi)].clone()
    tg[:, 0] -= lo
    for p in model.parameters():
        p.grad = None
    loss, _fm, _yo = model(x[lo:hi].contiguous().cuda(), tg)
    loss.backward()
    torch.cuda.synchronize()
    return float(loss.detach()), {k: p.grad.detach().clone() for k, p in model.named_parameters()}


def _det_dp_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    import torch.distributed as dist
    from millieye_amd import parallel as par
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model, x, targets = _det_problem()
    model = model.cuda().eval()
    red = par.overlap_detector_allreduce(model, chunk_bytes=256 << 10)   # 8.7 M parameters -> dozens of chunks
    assert red is not None
    lo, hi = par.shard_range(x.shape[0], rank, world)
    loss, grads = _det_shard_step(model, x, targets, lo, hi)
    q.put((rank, loss, red.chunks_last, red.bytes_last, {k: v.cpu().numpy() for k, v in grads.items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_dp_detector_hip_step_chunked_allreduce_two_ranks(hip_lib):
    import socket
    import torch
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_det_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(2):
        rank, loss, chunks, nbytes, grads = q.get(timeout=900)
        got[rank] = (loss, chunks, nbytes, grads)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    model, x, targets = _det_problem()
    model = model.cuda().eval()
    l0, g0 = _det_shard_step(model, x, targets, 0, 2)
    l1, g1 = _det_shard_step(model, x, targets, 2, 4)
    assert got[0][0] == pytest.approx(l0, rel=1e-6) and got[1][0] == pytest.approx(l1, rel=1e-6)
    n_param_bytes = 4 * sum(p.numel() for p in model.parameters())
    checked = 0
    for rank in (0, 1):
        _loss, chunks, nbytes, grads = got[rank]
        assert nbytes == n_param_bytes and chunks >= 8, (chunks, nbytes, n_param_bytes)
        assert sorted(grads) == sorted(g0)
        for name, want in g0.items():
            want = (want + g1[name]).cpu()
            mine = torch.from_numpy(grads[name])
            scale = max(float(want.abs().max()), 1e-12)
            # same kernels on the same shards; only atomics inside a shard's own kernels may reorder sums
            assert float((mine - want).abs().max()) <= 1e-5 * scale, (name, rank)
            checked += 1
    assert checked >= 2 * 30


def _det_graph_dp_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    import torch.distributed as dist
    from millieye_amd import parallel as par
    from millieye_amd.detector_graph import GraphedDetectorStep
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model, x, targets = _det_problem()
    model = model.cuda().eval()
    lo, hi = par.shard_range(x.shape[0], rank, world)
    tg = targets[(targets[:, 0] >= lo) & (targets[:, 0] < hi)].clone()
    tg[:, 0] -= lo
    step = GraphedDetectorStep(model, max_targets=8)
    xs = x[lo:hi].contiguous().cuda()
    out = []
    for _ in range(2):   # the capture's step and a replay: the same exchanged gradients
        for p in model.parameters():
            p.grad = None
        loss = step(xs, tg)
        torch.cuda.synchronize()
        out.append((float(loss), {k: p.grad.cpu().numpy().copy() for k, p in model.named_parameters()}))
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_dp_detector_captured_step_exchanges_after_the_replay_two_ranks(hip_lib):
    """GraphedDetectorStep under a process group (two ranks on the one leased GPU, gloo over CUDA tensors): the replayed graph
    holds no collective; the gradients are SUM-all-reduced in one bucket after it.  Reference: the two shards' eager steps in
    one process, added - same kernels, so the sums agree to the shard kernels' own atomics (1e-5 of the tensor's maximum)."""
    import socket
    import torch
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_det_graph_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(2):
        rank, out = q.get(timeout=900)
        got[rank] = out
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    model, x, targets = _det_problem()
    model = model.cuda().eval()
    l0, g0 = _det_shard_step(model, x, targets, 0, 2)
    l1, g1 = _det_shard_step(model, x, targets, 2, 4)
    checked = 0
    for rank, want_loss in ((0, l0), (1, l1)):
        for loss, grads in got[rank]:
            assert loss == pytest.approx(want_loss, rel=1e-6)
            for name, a in g0.items():
                want = (a + g1[name]).cpu()
                mine = torch.from_numpy(grads[name])
                assert float((mine - want).abs().max()) <= 1e-5 * max(float(want.abs().max()), 1e-12), (name, rank)
                checked += 1
    assert checked >= 4 * 30


def test_bench_allreduce_microbenchmark_runs_through_rccl(hip_lib):
    """``bench.py --workload allreduce``: under a launcher RCCL exchanges the 247.8 MB bucket (world 1
    on this lease); without one the line says that nothing was exchanged."""
    out = _bench(["--workload", "allreduce", "--bytes", "33554432"], {"BENCH_FORCE_SPAWN": "1"})
    assert out["config"]["rccl_ranks"] == 1 and out["config"]["bytes"] == 33554432 and out["config"]["chunks"] == 1
    assert out["value"] > 0 and out["unit"] == "GB/s"
    out = _bench(["--workload", "allreduce", "--bytes", "33554432"], {})
    assert out["config"]["rccl_ranks"] == 0 and "nothing was exchanged" in out["config"]["workload"]


# ---------------------------------------------------------------------------------------------------------------------
# replicas agree on rank 0's tuned plan (engine._autotune, MILLIEYE_TUNE_SYNC=1 - what bench.py sets for N > 1)
# ---------------------------------------------------------------------------------------------------------------------
def _tune_sync_worker(rank, world, port, q, cache_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", MILLIEYE_TUNE_SYNC="1",
                      MILLIEYE_TUNE_CACHE=os.path.join(cache_dir, f"tune_rank{rank}.json"))
    import torch
    import torch.distributed as dist
    from millieye_amd import engine
    from tests import parity_helpers as ph
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model = ph.make_darknet("yolov3-tiny-12", tag="tunesync").cuda().eval()
    model.compute_dtype = "bf16"
    x = ph.frames("tunesync/x", 2, 96).cuda()
    if rank == 1:  # a stale private table: every layer on the smallest tile - rank 0's measurement must win
        os.environ["MILLIEYE_TUNE_SYNC"] = "0"   # (this rank plans alone here)
        eng = model.engine_for("bf16")
        with torch.no_grad():
            model(x)
        for k in list(engine._TUNE_CACHE):
            engine._TUNE_CACHE[k] = (3, 1)
        eng._plans.clear()
        os.environ["MILLIEYE_TUNE_SYNC"] = "1"
        dist.barrier()
    else:
        dist.barrier()
    with torch.no_grad():
        model(x)
    plan = model.engine_for("bf16").plan_for(x)
    q.put((rank, [(int(d.tile), int(d.split_k)) for _m, d in plan.conv_descs if d.cin > 4], engine._TUNE_STATS["synced"]))
    dist.barrier()
    dist.destroy_process_group()


def test_replicas_run_rank0s_tuned_plan(hip_lib, tmp_path):
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_tune_sync_worker, args=(r, 2, port, q, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(2):
        rank, tiles, synced = q.get(timeout=600)
        got[rank] = (tiles, synced)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert got[0][0] == got[1][0] and len(got[0][0]) >= 8, got
    assert got[0][1] >= 1 and got[1][1] >= 1


# ---------------------------------------------------------------------------------------------------------------------
# the whole N = 2 flow of bench.py on the one leased GPU (BENCH_TEST_BACKEND=gloo: both ranks on GPU 0): self-launch, lock-step
# pre-warm (the same number of untimed steps on every rank), rank 0's tuned plan on both replicas, barrier + max-over-ranks
# timing, and - for the detector training step - the gradients leaving in chunks through GradChunkReducer every step
# ---------------------------------------------------------------------------------------------------------------------
def _bench2(extra):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", BENCH_TEST_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--prewarm-seconds", "0.3",
           "--no-cpu-baseline", "--no-bf16-line", "--no-accuracy", "--no-batch-sweep"] + extra
    res = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-4000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_two_ranks_inference_and_detector_training(hip_lib):
    out = _bench2(["--workload", "full", "--cfg", "yolov3-tiny-12", "--size", "160", "--batch", "4"])
    assert out["n_gpus"] == 2 and out["config"]["rccl_ranks"] == 2 and out["config"]["global_batch"] == 8
    assert out["value"] > 0 and out["scaling"] == "weak"
    # both readings of "@batch N" in the one line: the weak value above (4 per GPU) and the same 4 frames
    # in total sharded 2 + 2
    ss = out["strong_scaling"]
    assert ss["scaling"] == "strong" and ss["global_batch"] == 4 and ss["frames_per_gpu"] == [2, 2] and ss["value"] > 0
    out = _bench2(["--workload", "full", "--cfg", "yolov3-tiny-12", "--size", "160", "--batch", "5", "--scaling", "strong"])
    assert out["scaling"] == "strong" and out["config"]["global_batch"] == 5 and out["config"]["batch_per_gpu"] == 3
    assert out["value"] > 0 and "strong_scaling" not in out and out["config"]["rois_last_step"] >= 6
    assert abs(out["value"] - 5 * out["steps"] / (out["ms_per_step"] * out["steps"] * 1e-3)) < 0.01 * out["value"]
    out = _bench2(["--workload", "detector_train", "--cfg", "yolov3-tiny-12", "--size", "96", "--batch", "2", "--chunk-mb", "1"])
    assert out["n_gpus"] == 2 and out["config"]["grad_chunks"] >= 4
    assert out["config"]["grad_bucket_bytes"] > 30e6 and "reverse-layer chunks" in out["config"]["workload"]
    assert out["config"]["loss_last_step"] == out["config"]["loss_last_step"] and out["roofline"]["by_pass"]["wgrad"]["ms"] > 0
    # the mixed-precision detector step (round 5) through the same N = 2 flow: float32 gradients leave in chunks like the fp32 step's
    out = _bench2(["--workload", "detector_train", "--dtype", "bf16", "--cfg", "yolov3", "--size", "64", "--batch", "2", "--chunk-mb", "32"])
    assert out["n_gpus"] == 2 and out["dtype"] == "bf16" and out["config"]["grad_chunks"] >= 4
    assert out["config"]["grad_bucket_bytes"] > 240e6 and "bf16 activations and activation gradients" in out["config"]["workload"]
    assert out["config"]["loss_last_step"] == out["config"]["loss_last_step"] and out["value"] > 0


def test_bench_two_ranks_captured_detector_step(hip_lib):
    """The mixed-precision detector step replayed from one captured hipGraph per rank (detector_graph.py) through the N = 2 flow
    of bench.py: the gradients leave in one bucket after the replay."""
    out = _bench2(["--workload", "detector_train", "--dtype", "bf16", "--cfg", "yolov3", "--size", "64", "--batch", "2", "--graph"])
    assert out["n_gpus"] == 2 and out["config"]["captured_graph"] is True and out["config"]["grad_bucket_bytes"] > 240e6
    assert "one captured hipGraph" in out["config"]["workload"] and "one bucket after the replay" in out["config"]["workload"]
    assert out["config"]["loss_last_step"] == out["config"]["loss_last_step"] and out["value"] > 0
