"""One layer through me_conv2d_f32 (for rocprofv3 PMC passes): usage: python tools/conv32_one.py tile reps h cin cout k stride [res]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from millieye_amd import hip  # noqa: E402

tile, reps, h, cin, cout, k, s = (int(v) for v in sys.argv[1:8])
res = len(sys.argv) > 8 and sys.argv[8] == "res"
n = int(os.environ.get("CONV_BATCH", "32"))
dev = torch.device("cuda")
x = torch.randn((n, h, h, cin), device=dev)
if os.environ.get("CONV_ZERO_X"):
    x.zero_()
w = torch.randn((cout, k, k, cin), device=dev) / (k * k * cin) ** 0.5
sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
ho = (h + 2 * ((k - 1) // 2) - k) // s + 1
r = torch.randn((n, ho, ho, cout), device=dev) if res else None
out = torch.empty((n, ho, ho, cout), device=dev)
for _ in range(reps):
    hip.conv2d(x, w, sc, sh, k, s, (k - 1) // 2, 1, residual=r, out=out, tile=tile, split_k=1)
torch.cuda.synchronize()
print("done", tile, reps)
