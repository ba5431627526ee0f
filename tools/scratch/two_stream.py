"""Experiment: one batch-32 step as two batch-16 halves on two HIP streams (two model copies => separate arenas)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from millieye_amd import cfgs, synth
from millieye_amd.yolov3.models import Darknet

dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
cfg_path = cfgs.write_cfg("yolov3", "/tmp/ts_cfg")
dev = torch.device("cuda")
def make():
    m = Darknet(cfg_path).eval()
    synth.fill_darknet_(m, "bench/yolov3"); synth.trained_like_(m, "bench/yolov3/trained")
    m = m.to(dev); m.compute_dtype = dtype
    return m
m0, m1, m2 = make(), make(), make()
x = torch.from_numpy(synth.uniform("bench/frames/0", (32, 3, 416, 416))).to(dev)
xa, xb = x[:16].contiguous(), x[16:].contiguous()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def single():
    with torch.no_grad():
        m0(x)
def dual():
    with torch.no_grad():
        with torch.cuda.stream(s1):
            m1(xa)
        with torch.cuda.stream(s2):
            m2(xb)
def timeit(fn, n=30):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
# warm (autotune both shapes)
t_end = time.perf_counter() + 1.0
while time.perf_counter() < t_end:
    single(); dual(); torch.cuda.synchronize()
print(dtype, "single batch-32: %.3f ms" % timeit(single), " two streams x batch-16: %.3f ms" % timeit(dual))
