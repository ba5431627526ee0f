"""1x1 conv layers: hot (same buffers re-used back to back) vs cold (a 1 GB buffer is streamed between launches) timing."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from millieye_amd import hip
dev = torch.device("cuda")
big = torch.empty(256 << 20, device=dev)  # 1 GiB
def run(h, cin, cout, tile, cold):
    n = 32
    x = torch.randn((n, h, h, cin), device=dev).to(torch.bfloat16)
    w = (torch.randn((cout, 1, 1, cin), device=dev) / cin ** 0.5).to(torch.bfloat16)
    sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    out = torch.empty((n, h, h, cout), device=dev, dtype=torch.bfloat16)
    for _ in range(3): hip.conv2d_h16(x, w, sc, sh, 1, 1, 0, 1, out=out, tile=tile, split_k=1)
    tot = 0.0
    reps = 10
    for _ in range(reps):
        if cold: big.add_(1.0)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); hip.conv2d_h16(x, w, sc, sh, 1, 1, 0, 1, out=out, tile=tile, split_k=1); b.record()
        torch.cuda.synchronize(); tot += a.elapsed_time(b)
    return tot / reps * 1e3
for h, cin, cout in ((52, 256, 128), (26, 512, 256), (13, 1024, 512)):
    line = f"{h:3d} {cin}->{cout}: "
    for tile in (2, 3, 4, 12, 13, 14):
        line += f" t{tile}: {run(h, cin, cout, tile, False):5.1f}/{run(h, cin, cout, tile, True):5.1f}"
    print(line + "   (hot/cold us, single launches between events)")
