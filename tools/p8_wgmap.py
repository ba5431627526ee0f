"""Chip-wide picture of one launch of the patch-resident conv kernel (instrumented tile ids 296 = 221, 196 = 131): every
workgroup's wave 0 stamps s_memrealtime (100 MHz, one clock for the whole chip) at kernel start, main-loop start, main-loop
end and end, plus its HW_ID / XCC_ID.  Prints the dispatch rounds, per-round phase durations and the per-CU schedule.
usage: python tools/p8_wgmap.py <h> <cin> <cout> [tile] [batch]   (GPU box)"""
import os
import sys

os.environ.setdefault("MILLIEYE_ABLATION", "1")

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from millieye_amd import hip  # noqa: E402


def main():
    h, cin, cout = (int(v) for v in sys.argv[1:4])
    tile = int(sys.argv[4]) if len(sys.argv) > 4 else 296
    n = int(sys.argv[5]) if len(sys.argv) > 5 else 32
    dev = torch.device("cuda")
    x = torch.randn((n, h, h, cin), device=dev).to(torch.bfloat16)
    w = (torch.randn((cout, 3, 3, cin), device=dev) / (9 * cin) ** 0.5).to(torch.bfloat16)
    sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    r = torch.randn((n, h, h, cout), device=dev).to(torch.bfloat16)
    out = torch.empty((n, h, h, cout), device=dev, dtype=torch.bfloat16)
    wt = hip.tile_weights_h16(w)
    bm = 384 if tile == 196 else 256
    nwg = -(-(n * (h + 1) * (h + 1)) // bm) * (cout // 128)
    dbg = torch.zeros(nwg * 6, dtype=torch.int64, device=dev)
    big = torch.randn((4096, 4096), device=dev)
    for _ in range(20):
        big @ big
    for _ in range(5):
        hip.conv2d_h16(x, w, sc, sh, 3, 1, 1, 1, residual=r, out=out, tile=tile, split_k=1, wgt_tiled=wt, debug_ws=dbg)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    hip.conv2d_h16(x, w, sc, sh, 3, 1, 1, 1, residual=r, out=out, tile=tile, split_k=1, wgt_tiled=wt, debug_ws=dbg)
    b.record()
    torch.cuda.synchronize()
    t = dbg.cpu().numpy().reshape(nwg, 6)
    t0 = t[:, 0].min()
    us = (t[:, :4] - t0) / 100.0
    hw, xcc = t[:, 4], t[:, 5] & 0xf
    cu = (hw >> 8) & 0xf
    sh_ = (hw >> 12) & 0x1
    se = (hw >> 13) & 0x7
    cuid = ((xcc * 8 + se) * 2 + sh_) * 16 + cu
    print(f"{h}x{h} {cin}->{cout} batch {n} tile {tile}: {nwg} workgroups, event time {a.elapsed_time(b) * 1e3:.1f} us, "
          f"first start -> last end {us[:, 3].max():.1f} us, distinct CU ids {len(set(cuid.tolist()))}")
    start = us[:, 0]
    late = start > 2.0
    for name, sel in (("first wave of workgroups (start < 2 us)", ~late), ("later workgroups", late)):
        if sel.sum() == 0:
            continue
        u = us[sel]
        print(f"  {name}: {int(sel.sum())}  start {u[:, 0].min():.1f}..{u[:, 0].max():.1f}  prologue {np.mean(u[:, 1] - u[:, 0]):.2f}  "
              f"main loop {np.mean(u[:, 2] - u[:, 1]):.1f} (min {np.min(u[:, 2] - u[:, 1]):.1f} max {np.max(u[:, 2] - u[:, 1]):.1f})  "
              f"epilogue {np.mean(u[:, 3] - u[:, 2]):.1f} (min {np.min(u[:, 3] - u[:, 2]):.1f} max {np.max(u[:, 3] - u[:, 2]):.1f})  "
              f"end {u[:, 3].min():.1f}..{u[:, 3].max():.1f}")
    per_cu = {}
    for i in range(nwg):
        per_cu.setdefault(int(cuid[i]), []).append(i)
    counts = np.bincount([len(v) for v in per_cu.values()])
    print("  workgroups per CU histogram:", {k: int(c) for k, c in enumerate(counts) if c})
    ends = sorted(us[:, 3])
    print("  end-time percentiles (us):", " ".join(f"p{q}={np.percentile(ends, q):.1f}" for q in (10, 50, 90, 100)))
    for c in sorted(per_cu)[:3]:
        rows = sorted(per_cu[c], key=lambda i: us[i, 0])
        print(f"  CU {c}: " + " | ".join(f"wg {i}: {us[i, 0]:.1f} {us[i, 1]:.1f} {us[i, 2]:.1f} {us[i, 3]:.1f}" for i in rows))


if __name__ == "__main__":
    main()
