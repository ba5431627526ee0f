"""Fixed cost of one conv launch: tiny problems (one tile) back to back, fp32 and bf16 per-tap kernels."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from millieye_amd import hip
dev = "cuda"
def t(fn, reps=200):
    for _ in range(10): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for (n, h, cin, cout, k) in ((1, 8, 32, 64, 1), (1, 8, 1024, 64, 1), (32, 13, 1024, 512, 1), (32, 13, 64, 512, 1), (32, 26, 512, 256, 1), (32, 26, 64, 256, 1), (32, 52, 256, 128, 1), (32, 52, 32, 128, 1)):
    x = torch.randn((n, h, h, cin), device=dev); w = torch.randn((cout, k, k, cin), device=dev) * 0.05
    sc = torch.ones(cout, device=dev); sh = torch.zeros(cout, device=dev)
    out = torch.empty((n, h, h, cout), device=dev)
    f32 = t(lambda: hip.conv2d(x, w, sc, sh, k, 1, 0, 1, out=out, tile=3, split_k=1))
    xb, wb = x.bfloat16(), w.bfloat16(); ob = torch.empty((n, h, h, cout), device=dev, dtype=torch.bfloat16)
    b16 = t(lambda: hip.conv2d_h16(xb, wb, sc, sh, k, 1, 0, 1, out=ob, tile=3, split_k=1))
    print(f"n={n} {h}x{h} {cin}->{cout}: fp32 {f32:6.1f} us  bf16 {b16:6.1f} us (python call overhead included)", flush=True)
