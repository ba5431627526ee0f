import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from millieye_amd import hip
dev = torch.device("cuda")
x = torch.rand((32, 3, 416, 416), device=dev)
packed = (torch.randn((32, 3, 3, 3), device=dev) * 0.2).contiguous()
sc, sh = torch.ones(32, device=dev), torch.zeros(32, device=dev)
out = torch.empty((32, 416, 416, 32), device=dev)
def t(**kw):
    for _ in range(3): hip.conv2d(x, packed, sc, sh, 3, 1, 1, 1, x_nchw=True, out=out, **kw)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): hip.conv2d(x, packed, sc, sh, 3, 1, 1, 1, x_nchw=True, out=out, **kw)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / 10 * 1e3
print("fp32 VALU stem  %.1f us" % t(tile=92)); print("fp32 MFMA stem  %.1f us" % t())
print("bytes: %.0f MB" % ((x.numel() * 4 + out.numel() * 4) / 1e6))
