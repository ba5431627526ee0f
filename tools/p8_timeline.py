"""Per-stage timeline of one workgroup of the patch-resident conv kernel (instrumented tile ids 299 = 221, 199 = 131):
s_memtime stamps of workgroup 0 / wave 0: [kernel start] then per stage (before wait, after wait, after barrier, stage end),
[main loop end], [kernel end].   usage: python tools/p8_timeline.py <h> <cin> <cout> [tile]   (GPU box)"""
import os
import sys

os.environ.setdefault("MILLIEYE_ABLATION", "1")  # the instrumented tile ids are refused without this opt-in

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from millieye_amd import hip  # noqa: E402


def main():
    h, cin, cout = (int(v) for v in sys.argv[1:4])
    tile = int(sys.argv[4]) if len(sys.argv) > 4 else 299
    n, dev = 32, torch.device("cuda")
    x = torch.randn((n, h, h, cin), device=dev).to(torch.bfloat16)
    w = (torch.randn((cout, 3, 3, cin), device=dev) / (9 * cin) ** 0.5).to(torch.bfloat16)
    sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    r = torch.randn((n, h, h, cout), device=dev).to(torch.bfloat16)
    out = torch.empty((n, h, h, cout), device=dev, dtype=torch.bfloat16)
    wt = hip.tile_weights_h16(w)
    dbg = torch.zeros(2048, dtype=torch.int64, device=dev)
    for _ in range(5):
        hip.conv2d_h16(x, w, sc, sh, 3, 1, 1, 1, residual=r, out=out, tile=tile, split_k=1, wgt_tiled=wt, debug_ws=dbg)
    torch.cuda.synchronize()
    t = dbg.cpu().numpy()
    cnt = int(t[2047])
    t = t[:cnt].astype(np.int64)
    t = t - t[0]
    stages = (cnt - 3) // 4
    print(f"{h}x{h} {cin}->{cout} tile {tile}: {stages} stages, main loop {t[1 + 4 * stages] - t[1]} ticks, "
          f"prologue {t[1]} ticks, epilogue {t[-1] - t[1 + 4 * stages]} ticks (s_memtime ticks = shader cycles per the guide)")
    body = t[1:1 + 4 * stages].reshape(stages, 4)
    wait = body[:, 1] - body[:, 0]
    barrier = body[:, 2] - body[:, 1]
    work = body[:, 3] - body[:, 2]
    total = np.diff(np.concatenate([body[:, 0], [t[1 + 4 * stages]]]))
    print("stage:  wait  barrier  issue+mfma  total")
    for s in range(min(stages, 40)):
        print(f"{s:4d}  {wait[s]:6d} {barrier[s]:7d} {work[s]:10d} {total[s]:7d}")
    print(f"mean over stages >= 9: wait {wait[9:].mean():.0f}  barrier {barrier[9:].mean():.0f}  issue+mfma {work[9:].mean():.0f}  "
          f"total {total[9:].mean():.0f}")


if __name__ == "__main__":
    main()
