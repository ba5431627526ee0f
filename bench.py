#!/usr/bin/env python
"""bench.py - frames/s of the milliEye detection(+fusion) hot path on N MI355X (one process per GPU).

Contract (driver): ``python bench.py --gpus N --steps K --warmup W`` prints ONE JSON line on rank 0.
A *step* = one pass of the hot path over one batch of synthetic frames already resident in HBM.
Weak scaling: every rank processes its own ``--batch`` frames (frames are independent units,
SURVEY.md section 8e) - no data-path collective; the barrier / max-over-ranks timing is the only
communication.

Workloads (``--workload``):
  detector  yolov3.cfg (Darknet-53) 416x416 fp32 forward -> (featuremap, yolo_outputs)   [BASELINE configs[1]]
  full      detector + NMS + R-CNN/radar-fusion heads (Network.forward, mode 0)          [metric's "+fusion"]
  module2   the stage-2 network (module2_mixed: detector + NMS + every-class proposals + PS-RoIAlign heads)  [BASELINE configs[2]]
  train     stage-3 training step: forward + loss + backward + one SUM all-reduce (RCCL) of the flat
            gradient bucket + Adam step, batch 8 per GPU                                  [BASELINE configs[3] shape]
  detector_train  Darknet.forward(x, targets) + HIP backward of every layer + full-gradient all-reduce + SGD
``--dtype bf16 | f16``: the opt-in 16-bit storage modes of the detector for the inference workloads and for the
frozen detector of ``train``; e.g. ``--dtype f16 --size 608 --batch 16`` is the per-GPU shape of BASELINE configs[4].  The default
fp32 line also carries the same steps re-run in bf16 storage as ``bf16_storage_mode`` (never as ``value``).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16, dense (no sparsity)
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None, help="frames per GPU per step")
    ap.add_argument("--size", type=int, default=416)
    ap.add_argument("--cfg", default="yolov3")
    ap.add_argument("--workload", default="full", choices=["detector", "full", "module2", "train", "detector_train", "allreduce"])
    ap.add_argument("--bytes", type=int, default=247796596, help="--workload allreduce: bucket size (default: the Darknet-53 fp32 gradient, SURVEY 8(d) config 4)")
    ap.add_argument("--chunk-mb", type=float, default=32.0, help="detector_train / allreduce: bytes per overlapped gradient chunk")
    ap.add_argument("--dtype", default="f32", choices=("f32", "bf16", "f16"),
                    help="storage of the detector activations / weights: f32 (default, the parity mode), bf16 or f16 "
                         "(BASELINE configs[2]/[4]: 16-bit operands, fp32 accumulate; inference workloads only)")
    ap.add_argument("--prewarm-seconds", type=float, default=1.0, help="untimed clock ramp-up before the warmup")
    ap.add_argument("--scaling", default="weak", choices=("weak", "strong"),
                    help="weak (default): --batch frames PER GPU; strong: --batch frames in total, sharded over the ranks "
                         "(parallel.shard_range).  An N > 1 weak run of an inference workload also times the strong reading "
                         "and reports it as `strong_scaling` in the same line")
    ap.add_argument("--graph", action="store_true",
                    help="detector_train: forward + YOLO losses + backward as one captured hipGraph per step "
                         "(millieye_amd/detector_graph.py; the gradients are exchanged after the replay when N > 1)")
    ap.add_argument("--no-prefetch", action="store_true",
                    help="train: do not run the frozen detector of the next batch under the current batch's tail "
                         "(Network.queue_detector_prefetch; A/B)")
    ap.add_argument("--no-strong-leg", action="store_true", help="N > 1 weak runs: skip the extra strong-scaling measurement")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-bf16-line", action="store_true", help="skip the extra bf16-storage-mode measurement")
    ap.add_argument("--no-batch-sweep", action="store_true", help="skip the extra batch 1 / 8 measurements")
    ap.add_argument("--no-accuracy", action="store_true", help="skip the mAP@0.5-vs-reference leg on the committed mini split")
    ap.add_argument("--cpu-seconds", type=float, default=30.0, help="budget of the CPU baseline leg (the benchmark's batch always gets a warm-up + >= 3 timed passes)")
    args = ap.parse_args()
    if args.graph and args.workload != "detector_train":
        ap.error("--graph applies to --workload detector_train (the inference engines have their own MILLIEYE_GRAPH switch)")
    return args


def spawn_ranks(args):
    """``python bench.py --gpus N`` outside a launcher: start the N ranks ourselves (one process per GPU) by re-running this
    very command line under ``torch.distributed.run`` on 127.0.0.1, and hand its exit code back.  Rank 0 of that job
    prints the JSON line to our stdout."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < args.gpus and not os.environ.get("BENCH_TEST_BACKEND"):
        raise SystemExit(f"[bench] --gpus {args.gpus} but only {have} GPU(s) are visible")
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "8"))
    raise SystemExit(subprocess.call(cmd, env=env))


def init_dist(args):
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or os.environ.get("BENCH_FORCE_SPAWN")):
        spawn_ranks(args)  # does not return (BENCH_FORCE_SPAWN: the 1-GPU test of this very path)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"[bench] launched with WORLD_SIZE={world} but --gpus {args.gpus}: the two must agree")
    test_backend = os.environ.get("BENCH_TEST_BACKEND")  # tests only ("gloo"): N ranks share GPU 0 - RCCL refuses two ranks on
    # one device, and the suite's lease has one GPU; every other line of the N-rank flow is the production one
    if "WORLD_SIZE" in os.environ and test_backend:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        local = 0
        torch.cuda.set_device(0)
        dist.init_process_group(test_backend, rank=rank, world_size=world)
        probe = torch.ones(1, device=torch.device("cuda", 0))
        dist.all_reduce(probe)
        if int(probe.item()) != world:
            raise SystemExit(f"[bench] all-reduce over {world} ranks returned {probe.item()}")
    elif "WORLD_SIZE" in os.environ:  # under a launcher (the driver's, or spawn_ranks above): RCCL even at world size 1
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        # n_gpus is what RCCL saw, not what the command line asked for
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"[bench] RCCL world size {dist.get_world_size()} != --gpus {args.gpus}")
        probe = torch.ones(1, device=torch.device("cuda", local))
        dist.all_reduce(probe)  # one real collective up front: every rank is there, on its own GPU
        if int(probe.item()) != world:
            raise SystemExit(f"[bench] all-reduce over {world} ranks returned {probe.item()}")
    else:
        torch.cuda.set_device(0)
    return world, rank, local


def conv_roofline(model, x, steps):
    """Per-launch HIP-event timing of every conv_igemm/stem launch of the plan, on the stream the
    kernels run on.  Returns (achieved TFLOP/s of the MFMA conv kernel, avg launch us, launches,
    algorithmic flops per launch)."""
    from millieye_amd import hip

    engine = model.engine_for(model.compute_dtype)
    plan = engine.plan_for(x)
    stream = hip.stream_ptr()
    engine.run(x)  # patches the input / output pointers of the plan
    torch.cuda.synchronize()
    lib = hip.lib()
    descs = {m: d for m, d in plan.conv_descs}
    timed = {}  # module -> list of (start, end) events; launches stay IN SEQUENCE (real cache state)
    for _ in range(steps):
        for fn, args, _k, name in plan.launches:
            mods = _launch_modules(lib, fn, name)   # () for a launch that is not a convolution
            mfma_conv = bool(mods) and descs[mods[0]].cin > 4
            if mfma_conv:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                fn(*args, stream)
                b.record()
                timed.setdefault(mods, []).append((a, b))
            else:
                fn(*args, stream)
    torch.cuda.synchronize()
    total_ms, total_flops, launches = 0.0, 0, 0
    per_layer, per_layer_descs = [], []
    for mods, evs in timed.items():
        ms = sum(a.elapsed_time(b) for a, b in evs) / len(evs)
        flops = sum(_conv_flops(descs[m]) for m in mods)   # a one-launch bottleneck: the ALGORITHMIC flops of its two layers (no halo term)
        mod = "+".join(str(m) for m in mods)
        per_layer.append((mod, flops, ms))
        per_layer_descs.append(([descs[m] for m in mods], flops, ms))
        total_ms += ms
        total_flops += flops
        launches += 1
    achieved = total_flops / (total_ms * 1e-3) / 1e12
    conv_roofline.by_class = _by_class(per_layer_descs, half=plan.dtype != "f32")
    return achieved, total_ms * 1e3 / launches, launches, total_flops / launches, per_layer


def _conv_bytes(d, esize):
    """SURVEY 8(d) algorithmic bytes of one conv launch: input + output activations once (+ the residual it adds), weights once."""
    out_es = 4 if getattr(d, "y_f32", 0) else esize
    act = d.n * (d.h * d.w * d.cin * esize + d.ho * d.wo * d.upsample * d.upsample * d.cout * out_es)
    if d.res:
        act += d.n * d.ho * d.wo * d.cout * out_es
    return act + d.cout * d.ksize * d.ksize * d.cin * esize


def _by_class(rows, half):
    """SURVEY 8(d)'s roofline classes for the conv launches of one step: a launch whose arithmetic intensity (algorithmic flops /
    algorithmic bytes) is above the ridge (matrix peak / 8 TB/s: 314 flop/B in the 16-bit modes, 20 in fp32) is priced against
    the MFMA peak, one below it against HBM - the single MFMA number of ``roofline`` hides that the 16-bit 1x1 bottlenecks (AI ~ 85)
    are traffic problems.  ``rows``: (descs of the launch, flops, ms)."""
    peak = BF16_MFMA_PEAK_TFLOPS if half else FP32_MFMA_PEAK_TFLOPS
    esize = 2 if half else 4
    ridge = peak * 1e12 / (HBM_PEAK_GBS * 1e9)
    acc = {}
    for descs, flops, ms in rows:
        nbytes = sum(_conv_bytes(d, esize) for d in descs)
        if len(descs) == 2:   # a one-launch bottleneck: the mid tensor is neither written nor read
            nbytes -= 2 * descs[0].n * descs[0].ho * descs[0].wo * descs[0].cout * esize
        ai = flops / nbytes
        k = max(d.ksize for d in descs)
        name = ("3x3" if k == 3 else "1x1") + (" (MFMA-bound: AI above the ridge)" if ai >= ridge else " (HBM-bound: AI below the ridge)")
        e = acc.setdefault(name, dict(ms=0.0, flops=0.0, bytes=0.0, launches=0, bound="mfma" if ai >= ridge else "hbm"))
        e["ms"] += ms
        e["flops"] += flops
        e["bytes"] += nbytes
        e["launches"] += 1
    out = []
    for name, e in sorted(acc.items(), key=lambda kv: -kv[1]["ms"]):
        row = {"class": name, "bound": e["bound"], "launches": e["launches"], "ms": round(e["ms"], 4),
               "arithmetic_intensity": round(e["flops"] / e["bytes"], 1)}
        if e["bound"] == "mfma":
            a = e["flops"] / (e["ms"] * 1e-3) / 1e12
            row.update(achieved=round(a, 2), peak=peak, unit="TFLOP/s", frac=round(a / peak, 4))
        else:
            a = e["bytes"] / (e["ms"] * 1e-3) / 1e9
            row.update(achieved=round(a, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(a / HBM_PEAK_GBS, 4), algorithmic_bytes=int(e["bytes"]))
        out.append(row)
    return {"ridge_flop_per_byte": round(ridge, 1), "classes": out}


def _launch_modules(lib, fn, name):
    """cfg modules a launch of the engine's list computes: ``(i,)`` for a convolution, ``(i, j)`` for a 1x1 -> 3x3 bottleneck that the
    16-bit plan runs as one launch (me_bneck_h16: "bneck13+14"), ``()`` for everything else."""
    if fn is lib.me_conv2d_f32 or fn is lib.me_conv2d_h16:
        return (int(name[4:]),)
    if fn is lib.me_bneck_h16:
        return tuple(int(t) for t in name[5:].split("+"))
    return ()


def _conv_flops(d):
    return 2 * d.n * d.ho * d.wo * d.cout * d.ksize * d.ksize * d.cin


HBM_PEAK_GBS = 8000.0   # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
MFMA_F32_PEAK_TF = 157.3


def roi_stage_bytes(rois, n, fh, fw, rh, rw, weight_bytes, box_cols):
    """SURVEY 8(d) algorithmic bytes of the "RoI pooling + heads" stage (two launches: roi_pool10_kernel, roi_heads_mfma_kernel):
    every operand ONCE - both score maps in (image [n,fh,fw,490], radar [n,rh,rw,12] fp32), the pooled features
    (980 floats per RoI) out of the pooling launch and back into the heads launch, the head weights once, the RoI rows in
    and the per-RoI results out (regress 4 + refine 2 + mask 1 + row 8 + key 1 floats + 1 keep byte).  No re-read term:
    what a kernel re-fetches through L2 (e.g. W0 per workgroup) is traffic, not algorithmic bytes."""
    maps = 4.0 * n * (fh * fw * 490 + rh * rw * 12)
    pooled = 2.0 * 4.0 * 980 * rois
    rows = 4.0 * box_cols * rois + (4.0 * (4 + 2 + 1 + 8 + 1) + 1.0) * rois
    return maps + pooled + float(weight_bytes) + rows


def stage_roofline(model, net, x, step, rois, reps=5):
    """Achieved fraction of the bounding roofline per stage of one step (north_star: "achieved-fraction-of-roofline
    per stage"), from HIP events recorded in sequence inside real steps.  Detector stages come from per-launch
    events over the engine's launch list; the post-detector stages from the marks Network.forward sets."""
    from millieye_amd import hip

    engine = model.engine_for(model.compute_dtype)
    plan = engine.plan_for(x)
    lib = hip.lib()
    stream = hip.stream_ptr()
    descs = {m: d for m, d in plan.conv_descs}
    n, size = x.shape[0], x.shape[-1]
    bf16 = model.compute_dtype != "f32"
    mfma_peak = BF16_MFMA_PEAK_TFLOPS if bf16 else MFMA_F32_PEAK_TF
    acc = {}

    def add(name, ms, **work):
        e = acc.setdefault(name, dict(ms=0.0, flops=0.0, bytes=0.0))
        e["ms"] += ms
        for k, v in work.items():
            e[k] += v

    _plan, _rows_alive = engine.run(x)  # (the decode descriptors point into this tensor)
    torch.cuda.synchronize()
    # Network.forward's detector run decodes all [yolo] scales in one launch that also fills the NMS candidate lists
    cand = None
    if net is not None and getattr(plan, "yolo_tail", None) is not None:
        cand = (float(net.conf_thresh), hip.nms_workspace(plan.n, plan.rows, x.device)[0])
    for _ in range(reps):
        evs = []
        decoded = False
        for fn, args, _k, name in plan.launches:
            if cand is not None and name.startswith("yolo") and decoded:
                continue
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            if cand is not None and name.startswith("yolo"):
                lib.me_yolo_decode_cand_multi_f32(plan.yolo_tail, len(plan.yolo_tail), cand[0], cand[1], 1, stream)
                decoded = True
            else:
                fn(*args, stream)
            b.record()
            evs.append((name, fn, a, b))
        torch.cuda.synchronize()
        for name, fn, a, b in evs:
            ms = a.elapsed_time(b)
            mods = _launch_modules(lib, fn, name)
            if mods:
                d = descs[mods[0]]
                if d.cin <= 4:
                    add("stem conv (cin 3, direct)", ms,
                        bytes=n * (4.0 * d.cin * d.h * d.w + (2.0 if bf16 else 4.0) * d.cout * d.ho * d.wo))
                else:
                    add("MFMA convs (3x3 / 1x1 + BN + leaky + shortcut/upsample/route epilogues)", ms,
                        flops=float(sum(_conv_flops(descs[m]) for m in mods)))
            elif fn is lib.me_yolo_decode_f32:
                rows_c = plan.rows * (5 + (plan.num_classes or 0))
                add("YOLO decode", ms, bytes=0.0)  # bytes added once below (the scales share the output tensor)
            else:
                add("pool / copy", ms)
    if "YOLO decode" in acc:
        acc["YOLO decode"]["bytes"] = reps * 2.0 * 4.0 * n * plan.rows * (5 + (plan.num_classes or 0))
        if cand is not None:  # the launch the full pipeline really issues
            acc["YOLO decode + NMS candidate lists (one launch for the three scales)"] = acc.pop("YOLO decode")
    if net is not None:
        marks = []
        net._stage_cb = lambda name: marks.append((name, _ev()))
        for _ in range(reps):
            marks.clear()
            step()
            torch.cuda.synchronize()
            for (n0, e0), (n1, e1) in zip(marks[:-1], marks[1:]):
                ms = e0.elapsed_time(e1)
                if n1 == "nms":
                    add("NMS (prep + rank sort + chip-wide IoU bit matrix + scan + emit)", ms, bytes=4.0 * n * plan.rows * (5 + (plan.num_classes or 0)))
                elif n1 == "proposals":
                    add("proposal assembly", ms)
                elif n1 == "score_maps":
                    add("score maps (1x1 256->490 + radar CNN)", ms, flops=n * (0.170e9 + 0.128e9) * (size / 416.0) ** 2)
                elif n1 == "roi_heads":
                    fh, fw, _fc = plan.tap_shape
                    hw = net._get_packs()["heads"].refresh(x.device)
                    add("RoI pooling + refinement / ensemble heads", ms, flops=rois * 0.27e6,
                        bytes=roi_stage_bytes(rois, n, fh, fw, fh, fw, sum(t.numel() * t.element_size() for t in hw.values()),
                                              8 + int(net.class_num)))
                elif n1 == "output":
                    add("compaction + sort of output rows", ms)
        net._stage_cb = None
    out = []
    for name, e in acc.items():
        ms = e["ms"] / reps
        row = {"stage": name, "ms": round(ms, 4)}
        if e["flops"] and name.startswith("MFMA"):
            tf = e["flops"] / reps / (ms * 1e-3) / 1e12
            row.update(bound="mfma", achieved=round(tf, 2), peak=mfma_peak, unit="TFLOP/s", frac=round(tf / mfma_peak, 4))
        elif e["bytes"]:
            gbs = e["bytes"] / reps / (ms * 1e-3) / 1e9
            row.update(bound="hbm", achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(gbs / HBM_PEAK_GBS, 4),
                       algorithmic_bytes=int(e["bytes"] / reps))
            if e["flops"]:
                row["gflops"] = round(e["flops"] / reps / (ms * 1e-3) / 1e9, 1)
        elif e["flops"]:
            tf = e["flops"] / reps / (ms * 1e-3) / 1e12
            row.update(bound="mfma", achieved=round(tf, 3), peak=MFMA_F32_PEAK_TF, unit="TFLOP/s", frac=round(tf / MFMA_F32_PEAK_TF, 4))
        else:
            row.update(bound="latency")
        out.append(row)
    return out


def _ev():
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


def kernel_generation(half=False):
    """Short hash of the kernel sources: a committed PMC measurement only speaks for the library it was taken on.  Every
    file under csrc/ that is linked into the conv path counts (stems, per-tap and patch kernels, shared headers, the
    dispatch in api.hip) - hashing the whole directory instead of a hand-kept list (``half`` kept for the call sites)."""
    import hashlib
    csrc = os.path.join(ROOT, "millieye_amd", "csrc")
    h = hashlib.sha1()
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".h")):
            with open(os.path.join(csrc, f), "rb") as fh:
                h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()[:10]


def conv_traffic(args, batch, half=False):
    """HBM bytes per conv launch from the rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE, two separate passes over
    this very command: tools/pmc_only.sh -> tools/pmc_traffic.py -> profiles/conv_traffic*.json).  PMC counters cannot
    be read from inside this process, so the committed measurement is reported ONLY when it was taken on this
    configuration (cfg, size, batch, workload, storage type) and this kernel generation (source hash); otherwise
    ``traffic`` is null and ``traffic_source`` says why.  Returns (bytes or None, source string)."""
    dt = ("bf16" if half else "f32") if args.dtype == "f32" else args.dtype
    # one file per measured line: conv_traffic_<workload>_<dtype>[_<size>][_graph].json (tools/profile_r06.sh), the two batch-32
    # inference lines under their old names
    names = [f"conv_traffic_{args.workload}_{dt}_{args.size}" + ("_graph" if getattr(args, "graph", False) else "") + ".json",
             f"conv_traffic_{args.workload}_{dt}" + ("_graph" if getattr(args, "graph", False) else "") + ".json",
             "conv_traffic_bf16.json" if half else "conv_traffic.json"]
    t = name = None
    for name in names:
        try:
            with open(os.path.join(ROOT, "profiles", name)) as fh:
                t = json.load(fh)
            break
        except (OSError, ValueError):
            continue
    if t is None:
        return None, f"no committed PMC measurement (profiles/{names[0]} missing)"
    want = dict(cfg=args.cfg, size=args.size, batch=batch, workload=args.workload,
                dtype=("bf16" if half else "f32") if args.dtype == "f32" else args.dtype,
                kernel_generation=kernel_generation(half))
    if half and args.dtype == "f32":
        want["dtype"] = "bf16"
    have = {k: t.get(k) for k in want}
    if have != want:
        diff = ", ".join(f"{k}: measured {have[k]!r}, running {want[k]!r}" for k in want if have[k] != want[k])
        return None, f"profiles/{name} does not match this run ({diff})"
    return int(t["hbm_bytes_per_launch"]), f"profiles/{name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, {t.get('date', 'n/a')})"


def allreduce_bench(args, world, rank, dev):
    """SURVEY 8(d) config 4 microbenchmark: the Darknet-53 full-gradient bucket (247.8 MB fp32 by default) SUM-all-reduced
    over RCCL, once as ONE flat call and once in the chunks the detector training step sends (``--chunk-mb``), with the
    bench protocol (warm-up, barrier + synchronize on both sides, max over ranks).  A "step" = one exchange of the whole
    bucket.  ``value`` = algorithmic GB/s of the chunked form (bytes / time); ``busbw`` = 2 (N - 1) / N of it.  Without a
    process group (plain ``python bench.py --workload allreduce`` on one GPU) nothing is exchanged and the line says so."""
    n = max(1, args.bytes // 4)
    flat = torch.ones(n, device=dev, dtype=torch.float32)
    chunk = max(1, int(args.chunk_mb * 2 ** 20) // 4)
    pieces = [flat[i:i + chunk] for i in range(0, n, chunk)]
    have_pg = dist.is_initialized()

    def one_flat():
        if have_pg:
            dist.all_reduce(flat)

    def chunked():
        if have_pg:
            for pc in pieces:
                dist.all_reduce(pc)

    def timed(fn):
        for _ in range(args.warmup):
            fn()
            flat.fill_(1.0)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            fn()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        e = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([e], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            e = float(t.item())
        flat.fill_(1.0)
        return e

    e_flat, e_chunk = timed(one_flat), timed(chunked)
    if rank == 0:
        nbytes = n * 4
        gbs = lambda e: round(nbytes * args.steps / e / 1e9, 2) if have_pg and e > 0 else None  # noqa: E731
        bus = 2.0 * (world - 1) / world
        out = {
            "metric": "all-reduce of the Darknet-53 fp32 gradient bucket (algorithmic GB/s)", "value": gbs(e_chunk) or 0.0,
            "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(e_chunk / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"RCCL SUM all-reduce of {nbytes} bytes fp32 in {len(pieces)} chunks of {args.chunk_mb:g} MB "
                                   f"(the form the detector training step sends) over {world} rank(s)"
                                   + ("" if have_pg else " - NO process group: nothing was exchanged (1 rank, no launcher)"),
                       "bytes": nbytes, "chunks": len(pieces), "rccl_ranks": dist.get_world_size() if have_pg else 0,
                       "one_flat_call": {"ms": round(e_flat / args.steps * 1e3, 4), "algbw_gbs": gbs(e_flat)},
                       "chunked": {"ms": round(e_chunk / args.steps * 1e3, 4), "algbw_gbs": gbs(e_chunk),
                                   "busbw_gbs": round(gbs(e_chunk) * bus, 2) if gbs(e_chunk) else None},
                       "parallelism": "one process per GPU, ring / tree chosen by RCCL over xGMI"},
            "roofline": {"bound": "xgmi", "achieved": round((gbs(e_chunk) or 0.0) * bus, 2), "peak": 7 * 153.0 if world > 1 else None,
                         "unit": "GB/s", "frac": round((gbs(e_chunk) or 0.0) * bus / (7 * 153.0), 4) if world > 1 else None,
                         "traffic": None,
                         "note": "bus bandwidth per GPU against 7 xGMI links x 153 GB/s; meaningful for N > 1 only"},
        }
        print(json.dumps(out))
    if dist.is_initialized():
        dist.barrier()


def cpu_baseline(args, frames_cpu, state_dict, cfg_text, tap, budget_s, radar=None):
    """The oracle (stock torch CPU ops = the reference's CPU path, pinned by tests/golden) timed on the host cores of
    this box, in this run: a bounded sample of the same workload at batch 1, 8 and the benchmark's own batch (north_star:
    "batch 1/8/32"), the same frames the GPU saw.  ``value`` is the rate at the benchmark's batch."""
    from oracle import darknet_ref, network_ref

    # torch's intra-op pool at os.cpu_count() threads is far from optimal on a many-core host (256 threads on 13x13
    # maps: 60 s per frame); pick the best thread count on a probe conv so the baseline is the CPU path at its best,
    # and report both the count used and what the host has.
    import torch.nn.functional as F
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    probe_x = torch.randn(1, 256, 26, 26)
    probe_w = torch.randn(512, 256, 3, 3)
    best = (float("inf"), 1)
    probe_s = {}
    for threads in sorted({t for t in (avail, 128, 64, 32, 16, 8) if 1 <= t <= avail}):
        torch.set_num_threads(threads)
        F.conv2d(probe_x, probe_w, padding=1)
        t0 = time.perf_counter()
        for _ in range(3):
            F.conv2d(probe_x, probe_w, padding=1)
        dt = time.perf_counter() - t0
        probe_s[threads] = dt
        if dt < best[0]:
            best = (dt, threads)
    cores = best[1]
    torch.set_num_threads(cores)
    n_all = frames_cpu.shape[0]

    def one_pass(b):
        xb = frames_cpu[:b]
        if radar is None:
            darknet_ref.darknet_forward(cfg_text, state_dict, xb, tap_module=tap)
        else:
            maps_b, boxes_b, conf = radar[0][:b], radar[1][radar[1][:, 0] < b], radar[2]
            network_ref.network_forward(cfg_text, state_dict, xb, maps_b, boxes_b, 0, conf_thresh=conf, tap_module=tap)

    what = ("detector forward (oracle/darknet_ref.py" if radar is None else
            "Network.forward mode 0: detector + NMS + RoI heads (oracle/network_ref.py + tv_ops.c")
    # Protocol = BASELINE.md section 3: TWO warm-up passes, then the MEDIAN of >= 5 timed passes, at every batch size
    # (1, 8 and the benchmark's own), frames/s = batch / median.  The budget (``--cpu-seconds``) only decides how many
    # passes ABOVE five a small batch gets; a pass at batch 32 is 4-8 s on these hosts, so the leg costs about a minute.
    # The thread count is the best of the probe above; one more measurement at os.cpu_count() threads (what BASELINE.md
    # names) is taken at batch 1 and reported beside it - at batch 32 it would not finish inside the run.
    import statistics
    WARMUPS, MIN_PASSES = 2, 5
    batches = sorted({b for b in (1, 8, n_all) if b <= n_all})
    t0 = time.perf_counter()
    one_pass(1)  # global warm-up (thread pool, allocator; also bounds one frame)
    per_frame = time.perf_counter() - t0
    by_batch, passes, spread = {}, {}, {}
    small = [b for b in batches if b != n_all]
    share = 0.4 * budget_s / max(1, len(small))
    for b in batches:
        for _ in range(WARMUPS):
            one_pass(b)
        if b == n_all:
            reps = MIN_PASSES
        else:
            reps = max(MIN_PASSES, min(9, int(share / max(per_frame * b, 1e-3))))
        times = []
        for _ in range(reps):
            t0 = time.perf_counter()
            one_pass(b)
            times.append(time.perf_counter() - t0)
        med = statistics.median(times)
        by_batch[str(b)] = round(b / med, 3)
        passes[str(b)] = reps
        spread[str(b)] = [round(b / max(times), 3), round(b / min(times), 3)]
        per_frame = min(per_frame, med / b)
    all_threads = None
    host = os.cpu_count() or cores
    slow = probe_s.get(host, float("inf")) / max(best[0], 1e-9)
    if host != cores and slow > 8.0:
        # torch's intra-op pool at this thread count is an order of magnitude slower on the probe convolution already (a whole
        # frame took 70 s with 256 threads in round 4's first run): report the probe instead of stalling the run for a minute
        all_threads = {"threads": host, "batch": 1, "value": None, "probe_conv_slowdown_vs_best": round(slow, 1),
                       "note": "not run on whole frames: the probe convolution (3x3 256->512 @26x26) is this much slower at this "
                               "thread count than at the best-of-probe count"}
    elif host != cores:
        torch.set_num_threads(host)
        t0 = time.perf_counter()
        one_pass(1)  # warm-up at this thread count; also decides whether timed passes fit
        first = time.perf_counter() - t0
        if first < 10.0:
            one_pass(1)
            times = []
            for _ in range(MIN_PASSES):
                t0 = time.perf_counter()
                one_pass(1)
                times.append(time.perf_counter() - t0)
            all_threads = {"threads": host, "batch": 1, "value": round(1.0 / statistics.median(times), 3),
                           "passes": MIN_PASSES, "warmups": WARMUPS}
        else:
            all_threads = {"threads": host, "batch": 1, "value": round(1.0 / first, 3), "passes": 1, "warmups": 0,
                           "note": "one pass only: a frame takes more than 10 s at this thread count"}
        torch.set_num_threads(cores)
    return {"value": by_batch[str(n_all)], "unit": "frames/s", "cores": cores, "host_cpus": os.cpu_count(),
            "kind": "port", "by_batch": by_batch, "min_max_by_batch": spread,
            "at_host_cpu_count_threads": all_threads,
            "protocol": f"BASELINE.md section 3: {WARMUPS} warm-up passes per batch size, median of >= {MIN_PASSES} timed passes",
            "sample": f"{args.cfg} {args.size}x{args.size} fp32 {what}, torch {torch.__version__} CPU, {cores} threads "
                      f"(best of a probe; the host has {os.cpu_count()}) ): "
                      + ", ".join(f"median of {passes[str(b)]} passes at batch {b}" for b in batches)}


def accuracy_leg(dev):
    """The metric's second half ("mAP@0.5 vs ref"): the whole evaluation chain of the product - jpg frames, labels and radar
    pickles of the committed mini split (tests/golden/dataset_small, synthetic), device-side batch assembly, Network.forward
    on the HIP path, get_batch_statistics, ap_per_class - against the numbers the REAL reference's `evaluate` produced on
    the same files with the same deterministic weights (tests/golden/evaluate_small.npz, recorded by make_golden.py).  The
    ExDark / our_dataset splits and trained checkpoints are not available here, so this is parity of the chain, not a
    dataset-level accuracy claim."""
    import numpy as np
    from millieye_amd import cfgs, synth
    from millieye_amd.my_models import Network, define_yolo
    from millieye_amd.test_fusion import evaluate
    gold = os.path.join(ROOT, "tests", "golden")
    ref = np.load(os.path.join(gold, "evaluate_small.npz"))
    cfg_path = cfgs.write_cfg("yolov3-tiny-12", os.path.join("/tmp", f"millieye_bench_cfg_{os.getuid()}_acc"))
    net = Network(define_yolo(cfg_path), 0.2)
    synth.fill_network_(net, "evaluate_small")       # the fixture's weights (tests/golden/make_golden.py:eval_small_weights_)
    with torch.no_grad():
        net.refinement_head.net1[0].weight.mul_(0.002)
        net.refinement_head.net1[0].bias.mul_(0.002)
    net = net.to(dev).eval()
    out = {"dataset": "tests/golden/dataset_small (committed synthetic mini split; frames of test scene 4)", "iou": 0.5}
    for mode, key in ((0, "fusion (mode 0)"), (3, "auto (mode 3)")):
        row = {"reference": round(float(ref[f"mode{mode}/AP"].mean()), 6)}
        for dtype in ("f32", "bf16"):
            net.base_detector.compute_dtype = dtype
            _p, _r, ap, _f1, _cls, _stat, _pr = evaluate(net, mode="test", model_mode=mode, illumination=["H", "L"],
                                                         iou_thresh=0.5, nms_thresh=0.5, img_size=416, batch_size=2, test_list=4,
                                                         dataset_folder=os.path.join(gold, "dataset_small"), num_workers=0)
            row["hip_" + dtype] = round(float(np.mean(ap)), 6)
        out[key] = row
    net.base_detector.compute_dtype = "f32"
    out["equal_fp32"] = all(out[k]["hip_f32"] == out[k]["reference"] for k in ("fusion (mode 0)", "auto (mode 3)"))
    return out


def main():
    args = parse()
    world, rank, local = init_dist(args)
    dev = torch.device("cuda", local if world > 1 else 0)

    if args.workload == "allreduce":
        return allreduce_bench(args, world, rank, dev)
    if world > 1:
        os.environ["MILLIEYE_TUNE_SYNC"] = "1"  # the replicas run rank 0's tuned (tile, split_k) table (engine._autotune)
    import __graft_entry__ as g
    if rank == 0:
        g.build()
    if world > 1:
        dist.barrier()
    from millieye_amd import cfgs, synth  # noqa: E402
    from millieye_amd.yolov3.models import Darknet

    batch = args.batch or (32 if args.workload in ("full", "module2") else 8)
    gbatch = batch  # frames per step over all ranks (strong) / per GPU (weak)
    strong = args.scaling == "strong"
    if strong and args.workload not in ("full", "detector", "module2"):
        raise SystemExit("--scaling strong: inference workloads (full, detector, module2) - the training steps keep a fixed "
                         "per-GPU batch")
    if strong and batch < world:
        raise SystemExit(f"--scaling strong: {batch} frames cannot be sharded over {world} ranks")
    conf_thresh = 0.2
    cfg_path = cfgs.write_cfg(args.cfg, os.path.join("/tmp", f"millieye_bench_cfg_{os.getuid()}_{rank}"))
    from millieye_amd import parallel as par
    if strong:  # ONE global batch (rank 0's frames), every rank takes its contiguous share
        lo, hi = par.shard_range(gbatch, rank, world)
        frames_cpu = torch.from_numpy(synth.uniform("bench/frames/0", (gbatch, 3, args.size, args.size)))[lo:hi].contiguous()
        batch = hi - lo
    else:
        frames_cpu = torch.from_numpy(synth.uniform(f"bench/frames/{rank}", (batch, 3, args.size, args.size)))
    x = frames_cpu.to(dev)
    radar = None
    if args.workload == "detector_train":
        # detector fine-tuning step (row a6): Darknet.forward(x, targets) under autograd -> loss.backward() (HIP backward,
        # eval-mode BatchNorm) -> SUM all-reduce of the full 61.9 M-parameter gradient bucket (247.8 MB) -> SGD step
        from millieye_amd import parallel as par
        model = Darknet(cfg_path).eval()
        synth.fill_darknet_(model, "bench/" + args.cfg)
        synth.trained_like_(model, "bench/" + args.cfg + "/trained")
        state_cpu = {}
        model = model.to(dev)
        net = None
        last = {}
        det_targets = torch.tensor([[i, (3 * i) % 80, 0.3 + 0.04 * (i % 8), 0.4 + 0.03 * (i % 5), 0.2, 0.3]
                                    for i in range(batch)], dtype=torch.float32)
        det_params = [p for p in model.parameters()]
        det_opt = torch.optim.SGD(det_params, lr=1e-5)
        # N > 1 (or any launcher-made process group): the gradients leave in reverse-layer chunks on a communication stream
        # while the backward of the shallower layers is still running (parallel.GradChunkReducer); without a process
        # group nothing is attached and nothing is exchanged - the workload string below says so
        reducer = None if args.graph else par.overlap_detector_allreduce(model, chunk_bytes=int(args.chunk_mb * 2 ** 20))

        def eager_step():
            loss, _fm, yo = model(x, det_targets)
            loss.backward()
            if not args.graph:   # (under --graph this eager form only serves the per-pass profile; it exchanges nothing)
                last["bucket_bytes"] = reducer.bytes_last if reducer is not None else 0
                last["chunks"] = reducer.chunks_last if reducer is not None else 0
            det_opt.step()
            det_opt.zero_grad(set_to_none=True)
            last["loss"] = loss.detach()
            return yo
        step = eager_step
        if args.graph:
            # the same launches as ONE captured hipGraph (forward, the three YOLO losses with their counts kept on the device,
            # backward of every layer, the weight packing); the step is then: copy frames + targets, replay, all-reduce, SGD
            from millieye_amd.detector_graph import GraphedDetectorStep
            graphed = GraphedDetectorStep(model, max_targets=max(64, batch))

            def step():
                loss = graphed(x, det_targets)
                det_opt.step()
                det_opt.zero_grad(set_to_none=True)
                last["loss"] = loss
                last["bucket_bytes"] = sum(p.numel() for p in det_params) * 4 if dist.is_initialized() else 0
                last["chunks"] = 1 if dist.is_initialized() else 0
                return loss
    elif args.workload == "module2":
        # BASELINE configs[2]: the stage-2 network (module2_mixed/my_models.py: detector + NMS + every-class proposals +
        # PS-RoIAlign + refinement / ensemble heads, no radar branch), batch 32; rows come back on the host like the reference's
        from millieye_amd.module2.my_models import Network as Network2
        net2 = Network2(Darknet(cfg_path), conf_thresh).eval()
        synth.fill_network_(net2, "bench/m2/" + args.cfg)
        state_cpu = {}
        net2 = net2.to(dev)
        model = net2.base_detector
        net = None   # no stage marks / RoI bookkeeping of the stage-3 network below
        last = {}

        def step():
            with torch.no_grad():
                last["out"] = net2(x)
            return last["out"]
    elif args.workload == "detector":
        model = Darknet(cfg_path).eval()
        synth.fill_darknet_(model, "bench/" + args.cfg)
        synth.trained_like_(model, "bench/" + args.cfg + "/trained")
        state_cpu = {k: v.clone() for k, v in model.state_dict().items()}
        model = model.to(dev)
        net = None

        def step():
            with torch.no_grad():
                return model(x)
    else:
        from millieye_amd.my_models import Network
        net = Network(Darknet(cfg_path), conf_thresh).eval()
        # class-0 logit clearly dominant (+3 vs -4) so that the class filter of Network.forward (class_idx 0,
        # my_models.py:462) keeps a realistic number of image proposals per frame with random backbone weights
        synth.fill_network_(net, "bench/" + args.cfg, cls0_bias=3.0, cls_bias=-4.0)
        state_cpu = {k: v.clone() for k, v in net.state_dict().items()}
        net = net.to(dev)
        model = net.base_detector
        if strong:
            maps_np, boxes_np = synth.radar_inputs("bench/radar/0", gbatch, args.size // 16, boxes_per_image=2)
            _x, maps_cpu, boxes_cpu, _t = par.shard_batch(torch.empty((gbatch, 0)), torch.from_numpy(maps_np),
                                                          torch.from_numpy(boxes_np), None, rank, world)
            maps_cpu = maps_cpu.contiguous()
        else:
            maps_np, boxes_np = synth.radar_inputs(f"bench/radar/{rank}", batch, args.size // 16, boxes_per_image=2)
            maps_cpu, boxes_cpu = torch.from_numpy(maps_np), torch.from_numpy(boxes_np)
        maps_d, boxes_d = maps_cpu.to(dev), boxes_cpu.to(dev)
        radar = (maps_cpu, boxes_cpu, conf_thresh)
        last = {}

        def step():
            with torch.no_grad():
                last["out"] = net(x, maps_d, boxes_d.clone(), 0)  # forward scales the radar boxes in place
            return last["out"]

        if args.workload == "train":
            import random
            from millieye_amd import parallel as par
            from millieye_amd.train_path import head_parameters
            random.seed(1234 + rank)
            with torch.no_grad():  # targets = a few of the model's own detections, so IoU-positive samples exist
                det = net(x, maps_d, boxes_d.clone(), 1).cpu()
            tg = []
            for i in range(batch):
                rows_i = det[det[:, 0] == i]
                for j in (0, 3):
                    if j < len(rows_i):
                        b = rows_i[j, 1:5] / args.size
                        tg.append([i, 0, float((b[0] + b[2]) / 2), float((b[1] + b[3]) / 2), float(b[2] - b[0]) * 1.05,
                                   float(b[3] - b[1]) * 0.95])
            targets = torch.tensor(tg, dtype=torch.float32).reshape(-1, 6)
            net.train()
            net.base_detector.eval()
            heads = head_parameters(net)
            if os.environ.get("MILLIEYE_TORCH_ADAM", "0") == "1":   # A/B: torch's fused implementation
                opt = torch.optim.Adam(heads, lr=5e-4, fused=True)
            else:
                from millieye_amd.optim import Adam   # the loop's default optimizer (millieye_amd/train.py): one launch per step
                opt = Adam(heads, lr=5e-4)

            def step():  # noqa: F811
                if not args.no_prefetch:   # the loop's look-ahead (millieye_amd/train.py): the next batch's frames are known here
                    net.queue_detector_prefetch(x)
                loss, out_rows, _metric, _att = net(x, maps_d, boxes_d.clone(), targets.clone())
                loss.backward()
                last["bucket_bytes"] = par.allreduce_gradients(heads, static_pattern=True)
                opt.step()
                opt.zero_grad(set_to_none=True)
                last["out"], last["loss"] = out_rows, loss.detach()
                return out_rows

    # untimed pre-warm: the GPU needs a few hundred ms of sustained load to reach its steady clocks (the first
    # ~100 ms run ~15 % slower, measured with tools/conv_bench.py); serving throughput is the steady state
    if args.dtype != "f32":
        if args.workload not in ("full", "detector", "module2", "train", "detector_train"):
            raise SystemExit("--dtype bf16 / f16 applies to inference (workloads full, detector, module2), to the frozen detector "
                             "of the stage-3 training step (workload train) and to the detector training step (detector_train: "
                             "16-bit activations and activation gradients, fp32 master weights and weight gradients)")
        model.compute_dtype = args.dtype
    t_pre = time.perf_counter() + args.prewarm_seconds
    def lockstep_until(deadline, step=None):
        """Untimed steps until ``deadline`` - the SAME number on every rank (the training steps hold collectives, plan builds
        broadcast rank 0's tuned table: a rank that ran one step more than its peers would wait for a partner that never
        comes).  Every rank votes after each step; all stop as soon as one has reached its deadline."""
        step = step or main_step
        while True:
            step()
            torch.cuda.synchronize()
            go = time.perf_counter() < deadline
            if world > 1:
                vote = torch.tensor([1.0 if go else 0.0], device=dev)
                dist.all_reduce(vote, op=dist.ReduceOp.MIN)
                go = bool(vote.item() > 0)
            if not go:
                break

    main_step = step

    def timed(fn, k):
        """K steps bracketed by a barrier + synchronize on both sides, max over the ranks (the contract's protocol)."""
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            fn()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        e = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([e], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            e = float(t.item())
        return e

    lockstep_until(t_pre)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # The same K steps once more in the opt-in bf16 storage mode (BASELINE configs[2] / [4] name bf16 / fp16 for their
    # shapes), reported beside - never instead of - the fp32 `value` above: same barrier + max-over-ranks protocol.
    alt = None
    if args.dtype == "f32" and args.workload in ("full", "detector", "module2") and not args.no_bf16_line:
        try:
            model.compute_dtype = "bf16"
            t_alt = time.perf_counter() + 0.5
            lockstep_until(t_alt)  # plans + autotunes the bf16 engine, untimed
            for _ in range(args.warmup):
                step()
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            e16 = time.perf_counter() - t0
            if world > 1:
                t = torch.tensor([e16], device=dev, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                e16 = float(t.item())
            alt = {"e": e16}
            if rank == 0:
                ach16, avg16, n16, _f16, _pl = conv_roofline(model, x, max(3, min(args.steps, 10)))
                alt.update(ach=ach16, avg_us=avg16, launches=n16, by_class=conv_roofline.by_class)
            model.compute_dtype = "f32"
        except Exception as exc:  # the extra line must never take the fp32 measurement down with it
            alt = {"error": f"{type(exc).__name__}: {exc}"}
            model.compute_dtype = "f32"

    # The other reading of "@batch32" on N GPUs: the metric's 32 frames in TOTAL, sharded over the ranks
    # (parallel.shard_range) - strong scaling - next to the weak line above (32 per GPU).  Same protocol, same K.
    strong_leg = None
    if world > 1 and not strong and not args.no_strong_leg and args.dtype == "f32" \
            and args.workload in ("full", "detector", "module2") and batch >= world:
        try:
            lo, hi = par.shard_range(batch, rank, world)
            xs = torch.from_numpy(synth.uniform("bench/frames/0", (batch, 3, args.size, args.size)))[lo:hi].contiguous().to(dev)
            if args.workload == "full":
                m_np, b_np = synth.radar_inputs("bench/radar/0", batch, args.size // 16, boxes_per_image=2)
                _x, ms_, bs_, _t = par.shard_batch(torch.empty((batch, 0)), torch.from_numpy(m_np), torch.from_numpy(b_np),
                                                   None, rank, world)
                ms_, bs_ = ms_.contiguous().to(dev), bs_.to(dev)

                def step_s():
                    with torch.no_grad():
                        return net(xs, ms_, bs_.clone(), 0)
            elif args.workload == "module2":
                def step_s():
                    with torch.no_grad():
                        return net2(xs)
            else:
                def step_s():
                    with torch.no_grad():
                        return model(xs)
            lockstep_until(time.perf_counter() + 0.5, step_s)  # plans (+ rank 0's tuned table) for the shard shape, untimed
            for _ in range(args.warmup):
                step_s()
            e_s = timed(step_s, args.steps)
            strong_leg = {"scaling": "strong", "global_batch": batch, "frames_per_gpu": hi - lo if rank else
                          [par.shard_range(batch, r, world)[1] - par.shard_range(batch, r, world)[0] for r in range(world)],
                          "value": round(batch * args.steps / e_s, 2), "unit": "frames/s",
                          "ms_per_step": round(e_s / args.steps * 1e3, 4),
                          "note": "the same metric read as 32 frames in total: one global batch sharded over the ranks, no "
                                  "data-path collective; barrier + max-over-ranks timing like `value`"}
        except Exception as exc:  # the extra leg must never take the measurement down with it (same exception on every rank)
            strong_leg = {"error": f"{type(exc).__name__}: {exc}"}

    # north_star: "frames/sec ... at batch 1/8/32": the same step at batch 1 and 8 (rank 0's frames; untimed plan building
    # and autotuning first), fp32 and bf16 storage, reported beside - never instead of - `value`
    os.environ["MILLIEYE_TUNE_SYNC"] = "0"  # from here on rank 0 plans alone (batch sweep, stage table, accuracy leg)
    sweep = []
    if rank == 0 and args.workload in ("full", "detector") and args.dtype == "f32" and not args.no_batch_sweep:
        for b in (1, 8):
            if b >= batch:
                continue
            xb = x[:b].contiguous()
            if args.workload == "full":
                mb = maps_d[:b].contiguous()
                bb = boxes_d[boxes_d[:, 0] < b].contiguous()

                def step_b():
                    with torch.no_grad():
                        return net(xb, mb, bb.clone(), 0)
            else:
                def step_b():
                    with torch.no_grad():
                        return model(xb)
            row = {"batch": b}
            for mode in ("f32", "bf16"):
                model.compute_dtype = mode
                # warm-up issued back to back like the timed loop (a sync after every step lets the GPU idle and its clocks fall between
                # steps), then at least 0.3 s of timed steps: 20 steps of a 1.6 ms forward (32 ms) read 1.64 - 1.97 ms from box to box
                # and run to run where 300-step runs read 1.61 - 1.65 (round 6)
                t_end, warm = time.perf_counter() + 0.3, 0
                while time.perf_counter() < t_end:
                    for _ in range(8):
                        step_b()
                    torch.cuda.synchronize()
                    warm += 8
                est = 0.3 / max(warm, 1)
                k = max(args.steps, 20, min(400, int(0.3 / est) + 1))
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(k):
                    step_b()
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / k
                ach_b = conv_roofline(model, xb, 5)[0]
                peak_b = BF16_MFMA_PEAK_TFLOPS if mode == "bf16" else FP32_MFMA_PEAK_TFLOPS
                row[mode] = {"value": round(b / dt, 2), "unit": "frames/s", "ms_per_step": round(dt * 1e3, 4),
                             "conv_roofline_frac": round(ach_b / peak_b, 4)}
            model.compute_dtype = "f32"
            sweep.append(row)
        step()  # back to the benchmark's own batch: the per-stage accounting below reads the last forward's RoI count
        torch.cuda.synchronize()

    det_passes = None
    if args.workload == "detector_train":  # every rank (the step holds collectives when there is a process group)
        from millieye_amd import detector_train as dtr
        det_passes = dtr.profile_step_passes(eager_step)   # (the passes' conv launches, sequential: an eager step also under --graph)
    if rank == 0:
        frames = (gbatch if strong else batch * world) * args.steps
        plan = model.engine_for(model.compute_dtype).plan_for(x)
        bf16 = args.dtype != "f32"
        peak = BF16_MFMA_PEAK_TFLOPS if bf16 else FP32_MFMA_PEAK_TFLOPS
        ach, avg_us, launches, flops_per_launch, per_layer = conv_roofline(model, x, max(3, min(args.steps, 10)))
        out = {
            "metric": "frames/sec YOLOv3-416+fusion @batch32, 1/2/4/8 MI355X; mAP@0.5 vs ref",
            "value": round(frames / elapsed, 2),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {
                "workload": f"{args.cfg}.cfg {args.size}x{args.size} "
                            + (f"{args.dtype}-storage ({args.dtype} operands, fp32 accumulate)" if bf16 else "fp32")
                            + (f" inference, batch={gbatch} in total sharded over the ranks, " if strong else
                               f" inference, batch={batch} per GPU, ")
                            + ("Darknet.forward -> featuremap + yolo_outputs" if args.workload == "detector" else
                               "full milliEye: Darknet.forward -> NMS -> Network.forward mode 0 (R-CNN head + radar "
                               "fusion, 2 radar boxes/frame) -> output rows" if args.workload == "full" else
                               "stage-2 network (module2_mixed): Darknet.forward -> NMS -> every-class proposals -> "
                               "PS-RoIAlign + refinement / ensemble heads -> output rows on the host"
                               if args.workload == "module2" else
                               "Darknet.forward(x, targets) -> loss.backward() -> all-reduce -> SGD"
                               if args.workload == "detector_train" else
                               "stage-3 training step: frozen detector + NMS + train-mode heads + focal/BCE loss + "
                               "backward + SUM all-reduce of the flat gradient bucket + Adam step")
                            + ", synthetic frames U[0,1), deterministic trained-like weights",
                "stage": args.workload,
                "batch_per_gpu": batch,
                "global_batch": gbatch if strong else batch * world,
                "img_size": args.size,
                "parallelism": f"frames sharded over {world} GPU(s), one process per GPU, "
                               + ("one SUM all-reduce of the gradient bucket per step (RCCL)" if args.workload == "train" else
                                  "SUM all-reduce of the gradients in reverse-layer chunks on a communication stream (RCCL)"
                                  if args.workload == "detector_train" else "no data-path collective"),
                "rccl_ranks": dist.get_world_size() if dist.is_initialized() else 0,
                "conv_gflop_per_frame": round(plan.conv_flops / batch / 1e9, 3),
                "arena_mb": round(plan.arena_bytes / 2 ** 20, 1),
            },
            "roofline": {
                "bound": "mfma",
                "kernel": (f"conv_igemm_buf_h16 (v_mfma_f32_32x32x16_{args.dtype} implicit-GEMM conv, all 3x3 / 1x1 layers)" if bf16
                           else "conv_igemm_buf_f32 (v_mfma_f32_32x32x2_f32 implicit-GEMM conv, all 3x3 / 1x1 layers)"),
                "achieved": round(ach, 2),
                "peak": peak,
                "unit": "TFLOP/s",
                "frac": round(ach / peak, 4),
                "traffic": conv_traffic(args, batch, half=bf16)[0],
                "traffic_source": conv_traffic(args, batch, half=bf16)[1],
                "launches_per_step": launches,
                "avg_launch_us": round(avg_us, 2),
                "gflop_per_launch": round(flops_per_launch / 1e9, 3),
                "by_class": conv_roofline.by_class,
            },
        }
        if sweep:
            out["batch_sweep"] = sweep
        if strong_leg is not None:
            out["strong_scaling"] = strong_leg
        if alt is not None and "error" in alt:
            out["bf16_storage_mode"] = alt
        elif alt is not None:
            out["bf16_storage_mode"] = {
                "note": "same workload, same K steps, detector activations / weights stored as bf16 (fp32 accumulate, fp32 "
                        "after the detector); opt-in mode with its own parity bar (DESIGN.md 5b) - not the headline value",
                "dtype": "bf16",
                "value": round(frames / alt["e"], 2),
                "unit": "frames/s",
                "ms_per_step": round(alt["e"] / args.steps * 1e3, 4),
                "roofline": {"bound": "mfma", "kernel": "conv_igemm_buf_h16", "achieved": round(alt["ach"], 2),
                             "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                             "frac": round(alt["ach"] / BF16_MFMA_PEAK_TFLOPS, 4),
                             "traffic": conv_traffic(args, batch, half=True)[0],
                             "traffic_source": conv_traffic(args, batch, half=True)[1],
                             "launches_per_step": alt["launches"], "avg_launch_us": round(alt["avg_us"], 2),
                             "by_class": alt.get("by_class")},
            }
        if not args.no_cpu_baseline and world == 1 and args.workload not in ("train", "detector_train", "module2"):
            from millieye_amd.engine import pick_tap_module
            out["cpu_baseline"] = cpu_baseline(args, frames_cpu, state_cpu, cfgs.KNOWN[args.cfg](),
                                               pick_tap_module(model.module_defs), args.cpu_seconds, radar)
        if net is not None:
            out["config"]["output_rows_last_step"] = int(last["out"].shape[0])
            out["config"]["rois_last_step"] = int(getattr(net, "_last", {}).get("n_img", torch.zeros(1)).sum().item()) \
                + batch * 2 if args.workload == "full" else int(net._last_train["k"])
        if args.workload == "module2":
            out["config"]["output_rows_last_step"] = int(last["out"].shape[0])
        if args.workload == "detector_train":
            out["config"]["grad_bucket_bytes"] = int(last.get("bucket_bytes", 0))
            out["config"]["loss_last_step"] = round(float(last["loss"]), 5)
            out["config"]["wgrad_stream"] = os.environ.get("MILLIEYE_WGRAD_STREAM", "1") != "0"  # weight gradients beside the data gradients
            ranks = dist.get_world_size() if dist.is_initialized() else 0
            out["config"]["grad_chunks"] = int(last.get("chunks", 0))
            out["config"]["captured_graph"] = bool(args.graph)
            if args.graph:
                graphed.metrics()   # (the deferred host read: raises if a target of any step was out of range)
            mixed = "" if args.dtype == "f32" else (f"{args.dtype} activations and activation gradients, fp32 master weights, weight "
                                                     "gradients on the fp32 matrix pipe, ")
            out["config"]["workload"] = out["config"]["workload"].replace(
                " inference, batch", " detector training step (forward + HIP backward of every layer, eval-mode BN, " + mixed
                + ("forward + losses + backward replayed as one captured hipGraph, " if args.graph else "")
                + (("full-gradient all-reduce in one bucket after the replay, " if args.graph else
                    f"full-gradient all-reduce in {int(last.get('chunks', 0))} reverse-layer chunks beside the backward, ")
                   if ranks else "all-reduce skipped (1 rank, no process group), ") + "SGD), batch", 1).replace(
                "Darknet.forward(x, targets) -> loss.backward() -> all-reduce -> SGD",
                "Darknet.forward(x, targets) -> loss.backward()" + (" (+ overlapped all-reduce)" if ranks else "") + " -> SGD")
            # this workload's own roofline: the convolution kernels of each pass, timed sequentially on one stream in one
            # extra (untimed) step, against the matrix peak of the step's arithmetic type - not the forward kernel's inference figure
            passes = det_passes
            total_flops = sum(v[1] for v in passes.values())
            # (the 16-bit step's three passes all run on the 16-bit matrix pipe since round 5: priced against ITS dense peak)
            train_peak = FP32_MFMA_PEAK_TFLOPS if args.dtype == "f32" else BF16_MFMA_PEAK_TFLOPS
            out["roofline"] = {
                "bound": "mfma", "unit": "TFLOP/s", "peak": train_peak,
                "kernel": "conv_igemm_buf_f32 (forward, data gradient) + conv_wgrad_* (weight gradient), fp32 MFMA" if args.dtype == "f32"
                          else f"me_conv2d_h16 (forward, data gradient) + conv_wgrad_tile_h16_kernel (weight gradient), {args.dtype} MFMA "
                               "(v_mfma_f32_32x32x16): batch-8 layers, 32 - 700 workgroups per launch",
                "achieved": round(total_flops / (elapsed / args.steps) / 1e12, 2),
                "frac": round(total_flops / (elapsed / args.steps) / 1e12 / train_peak, 4),
                "note": "achieved / frac: conv FLOPs of the three passes over the WHOLE timed step (affine, pack, loss, SGD "
                        "included); by_pass: each pass's conv launches alone, sequential",
                "by_pass": {k: {"ms": round(ms, 3), "gflop": round(fl / 1e9, 1), "launches": cnt,
                                "achieved": round(fl / ms / 1e9, 2), "frac": round(fl / ms / 1e9 / train_peak, 4)}
                            for k, (ms, fl, cnt) in sorted(passes.items())},
                # (per launch of the step's convolution kernels - forward, data gradient, weight gradient - from this workload's own
                #  PMC passes: profiles/conv_traffic_detector_train_<dtype>[_graph].json)
                "traffic": conv_traffic(args, batch, half=args.dtype != "f32")[0],
                "traffic_source": conv_traffic(args, batch, half=args.dtype != "f32")[1],
            }
        if args.workload == "train":
            out["config"]["detector_prefetch"] = not args.no_prefetch
            if not args.no_prefetch:
                out["config"]["workload"] += "; the frozen detector + NMS of batch k + 1 are issued on a second stream under the " \
                                             "host-bound tail of batch k (one detector forward per step, as without it)"
            out["config"]["grad_bucket_bytes"] = int(last.get("bucket_bytes", 0))
            out["config"]["loss_last_step"] = round(float(last["loss"]), 5)
            out["config"]["workload"] = out["config"]["workload"].replace("inference", "training (heads), inference (detector)")
        if args.workload in ("full", "detector"):
            rois = out["config"].get("rois_last_step", 0)
            out["stages"] = stage_roofline(model, net, x, step, rois)
        if args.workload == "full" and not args.no_accuracy:
            try:
                out["accuracy"] = accuracy_leg(dev)
            except Exception as exc:  # never take the measurement down
                out["accuracy"] = {"error": f"{type(exc).__name__}: {exc}"}
        if os.environ.get("BENCH_LAYERS"):
            for mod, flops, ms in per_layer:
                print(f"[layer] conv{mod}: {flops / 1e9:.3f} GF {ms * 1e3:.1f} us {flops / ms / 1e9:.1f} TF/s",
                      file=sys.stderr)
        print(json.dumps(out))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
