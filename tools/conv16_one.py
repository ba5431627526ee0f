"""One Darknet-53 layer through me_conv2d_h16 (for PMC passes): 3x3 128->256 @52x52, batch 32, fused residual.
usage: python tools/conv16_one.py [tile] [reps] [h cin cout]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from millieye_amd import hip  # noqa: E402

tile = int(sys.argv[1]) if len(sys.argv) > 1 else 14
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n, h, cin, cout, k = 32, 52, 128, 256, 3
if len(sys.argv) > 5:
    h, cin, cout = int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
dev = torch.device("cuda")
x = torch.randn((n, h, h, cin), device=dev).to(torch.bfloat16)
if os.environ.get("P8_ZERO_X"):  # DVFS probe: same instructions and traffic, zero products
    x.zero_()
w = (torch.randn((cout, k, k, cin), device=dev) / (k * k * cin) ** 0.5).to(torch.bfloat16)
sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
r = torch.randn((n, h, h, cout), device=dev).to(torch.bfloat16)
out = torch.empty((n, h, h, cout), device=dev, dtype=torch.bfloat16)
for _ in range(reps):
    hip.conv2d_h16(x, w, sc, sh, k, 1, 1, 1, residual=r, out=out, tile=tile, split_k=1)
torch.cuda.synchronize()
print("done", tile, reps)
