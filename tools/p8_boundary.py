"""What a kernel boundary costs around the patch-resident conv kernel: the instrumented tile (296 = 221 + per-workgroup
s_memrealtime stamps, one chip-wide 100 MHz clock) is launched back to back into separate stamp buffers, so the idle time
between the last workgroup of launch i and the first workgroup of launch i + 1 can be read directly, beside the HIP-event
average of the uninstrumented tile.  MILLIEYE_P8_STORE = 0 / 1 / 2 selects plain / nt / sc1 epilogue stores (read once per
process: run this tool once per mode).   usage: python tools/p8_boundary.py <h> <cin> <cout> [batch]   (GPU box)"""
import os
import sys

os.environ.setdefault("MILLIEYE_ABLATION", "1")

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from millieye_amd import hip  # noqa: E402


def main():
    h, cin, cout = (int(v) for v in sys.argv[1:4])
    n = int(sys.argv[4]) if len(sys.argv) > 4 else 32
    dev = torch.device("cuda")
    x = torch.randn((n, h, h, cin), device=dev).to(torch.bfloat16)
    w = (torch.randn((cout, 3, 3, cin), device=dev) / (9 * cin) ** 0.5).to(torch.bfloat16)
    sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    r = torch.randn((n, h, h, cout), device=dev).to(torch.bfloat16)
    outs = [torch.empty((n, h, h, cout), device=dev, dtype=torch.bfloat16) for _ in range(2)]
    wt = hip.tile_weights_h16(w)
    nwg = -(-(n * (h + 1) * (h + 1)) // 256) * (cout // 128)
    reps = 8
    dbgs = [torch.zeros(nwg * 6, dtype=torch.int64, device=dev) for _ in range(reps)]
    big = torch.randn((4096, 4096), device=dev)
    for _ in range(20):
        big @ big

    def launch(tile, i, dbg=None):
        hip.conv2d_h16(x, w, sc, sh, 3, 1, 1, 1, residual=r, out=outs[i & 1], tile=tile, split_k=1, wgt_tiled=wt, debug_ws=dbg)

    for i in range(reps):
        launch(296, i, dbgs[i])
    torch.cuda.synchronize()
    for i in range(reps):
        launch(296, i, dbgs[i])
    torch.cuda.synchronize()
    t = [d.cpu().numpy().reshape(nwg, 6) for d in dbgs]
    first = [a[:, 0].min() for a in t]
    last = [a[:, 3].max() for a in t]
    win = [(b - a) / 100.0 for a, b in zip(first, last)]
    gap = [(first[i + 1] - last[i]) / 100.0 for i in range(reps - 1)]
    period = [(first[i + 1] - first[i]) / 100.0 for i in range(reps - 1)]
    for tile in (221, 296):
        for _ in range(3):
            launch(tile, 0, dbgs[0] if tile == 296 else None)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(20):
            launch(tile, i, dbgs[0] if tile == 296 else None)
        b.record()
        torch.cuda.synchronize()
        print(f"tile {tile}: HIP-event average over 20 back-to-back launches {a.elapsed_time(b) / 20 * 1e3:.1f} us")
    print(f"{h}x{h} {cin}->{cout} batch {n}, store mode {os.environ.get('MILLIEYE_P8_STORE', '0')}: "
          f"workgroup window {np.mean(win):.1f} us (min {min(win):.1f} max {max(win):.1f}), "
          f"idle between launches {np.mean(gap):.1f} us (min {min(gap):.1f} max {max(gap):.1f}), "
          f"launch period {np.mean(period):.1f} us")


if __name__ == "__main__":
    main()
