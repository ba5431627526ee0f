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
