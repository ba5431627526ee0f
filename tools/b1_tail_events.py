"""Un-profiled timeline of one Network.forward at a small batch: at every trace point of forward() the host
clock and a HIP event on the stream the point belongs to; printed per point as host time and GPU time since entry (median of the
runs).  GPU time ~ host time => the GPU ran the work as soon as it was issued (host-bound there); GPU time >> host time => the host
is ahead (GPU-bound).  No profiler attached: the step keeps its real speed.
usage (GPU box): python tools/b1_tail_events.py [batch] [f32|bf16]"""
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

g.build()
from millieye_amd import cfgs, synth  # noqa: E402
from millieye_amd.my_models import Network  # noqa: E402
from millieye_amd.yolov3.models import Darknet  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dtype = sys.argv[2] if len(sys.argv) > 2 else "f32"
net = Network(Darknet(cfgs.write_cfg("yolov3", "/tmp/b1t_cfg")), 0.2).eval()
synth.fill_network_(net, "bench/yolov3", cls0_bias=3.0, cls_bias=-4.0)
net = net.cuda()
net.base_detector.compute_dtype = dtype
x = torch.from_numpy(synth.uniform("bench/frames/0", (batch, 3, 416, 416))).cuda()
maps_np, boxes_np = synth.radar_inputs("bench/radar/0", batch, 26, boxes_per_image=2)
maps_d, boxes_d = torch.from_numpy(maps_np).cuda(), torch.from_numpy(boxes_np).cuda()


def step():
    with torch.no_grad():
        return net(x, maps_d, boxes_d.clone(), 0)


for _ in range(30):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    step()
torch.cuda.synchronize()
print(f"batch {batch} {dtype}: {(time.perf_counter() - t0) * 5:.4f} ms/step untraced")

runs = []
for _ in range(40):
    marks = []

    def cb(name):
        e = torch.cuda.Event(enable_timing=True)
        e.record()  # (on the stream that is current at the trace point: the side stream for the "side:" points)
        marks.append((name, time.perf_counter(), e))
    torch.cuda.synchronize()
    net._trace_cb = cb
    step()
    net._trace_cb = None
    torch.cuda.synchronize()
    h0, e0 = marks[0][1], marks[0][2]
    runs.append([(name, (t - h0) * 1e6, e0.elapsed_time(e) * 1e3) for name, t, e in marks])
print(f"{'point':34s} {'host us':>9s} {'gpu us':>9s} {'gpu - host':>10s}")
for i, (name, _h, _g) in enumerate(runs[0]):
    h = statistics.median(r[i][1] for r in runs)
    gp = statistics.median(r[i][2] for r in runs)
    print(f"{name:34s} {h:9.1f} {gp:9.1f} {gp - h:10.1f}")
