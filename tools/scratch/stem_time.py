import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch, torch.nn.functional as F
from millieye_amd import hip
dev = torch.device("cuda")
x = torch.rand((32, 3, 416, 416), device=dev)
w = (torch.randn((32, 3, 3, 3), device=dev) * 0.2)
packed = w.permute(0, 2, 3, 1).contiguous()
taps = F.pad(packed.reshape(32, 27), (0, 5)).to(torch.bfloat16).contiguous()
sc, sh = torch.ones(32, device=dev), torch.zeros(32, device=dev)
out = torch.empty((32, 416, 416, 32), device=dev, dtype=torch.bfloat16)
def t(**kw):
    for _ in range(3): hip.conv2d_h16(x, packed, sc, sh, 3, 1, 1, 1, x_nchw=True, out=out, **kw)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): hip.conv2d_h16(x, packed, sc, sh, 3, 1, 1, 1, x_nchw=True, out=out, **kw)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / 10 * 1e3
print("VALU stem  %.1f us" % t(wgt_tiled=taps, tile=1))
print("MFMA stem  %.1f us" % t(wgt_tiled=taps))
print("bytes: %.0f MB" % ((x.numel() * 4 + out.numel() * 2) / 1e6))
