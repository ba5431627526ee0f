"""Timing of the five small-cin 3x3 layers of Darknet-53 (conv1, 3, 5, 7 / 10) in the 16-bit modes: every other tile vs the
weight-stationary 2-D-tiled kernel (tile 60, csrc/conv3x3_ws_h16.hip).  usage: python tools/ws3_bench.py [batch]   (GPU box)"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from millieye_amd import hip  # noqa: E402

LAYERS = [("conv1", 416, 32, 64, 2, False), ("conv3", 208, 32, 64, 1, True), ("conv5", 208, 64, 128, 2, False),
          ("conv7", 104, 64, 128, 1, True)]
OTHERS = (1, 2, 3, 4, 11, 12, 13, 14, 101, 121, 201, 221)


def time_tile(x, w, wt, sc, sh, r, out, stride, tile, reps=10):
    kw = dict(residual=r, out=out, tile=tile, split_k=1)
    if tile >= 100:
        kw["wgt_tiled"] = wt
    for _ in range(2):
        hip.conv2d_h16(x, w, sc, sh, 3, stride, 1, 1, **kw)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        hip.conv2d_h16(x, w, sc, sh, 3, stride, 1, 1, **kw)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    dev = torch.device("cuda")
    big = torch.randn((4096, 4096), device=dev)
    for _ in range(30):
        big @ big
    for name, h, cin, cout, stride, with_res in LAYERS:
        ho = (h + 2 - 3) // stride + 1
        x = torch.randn((n, h, h, cin), device=dev).to(torch.bfloat16)
        w = (torch.randn((cout, 3, 3, cin), device=dev) / (9 * cin) ** 0.5).to(torch.bfloat16)
        wt = hip.tile_weights_h16(w)
        sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
        r = torch.randn((n, ho, ho, cout), device=dev).to(torch.bfloat16) if with_res else None
        out = torch.empty((n, ho, ho, cout), device=dev, dtype=torch.bfloat16)
        mb = (x.numel() + out.numel() * (2 if with_res else 1)) * 2 / 1e6
        res = {}
        for tile in OTHERS + (60,):
            if 10 < tile < 20 and cin % 64:
                continue
            try:
                res[tile] = time_tile(x, w, wt, sc, sh, r, out, stride, tile)
            except hip.MeError:
                pass
        old = min((v, t) for t, v in res.items() if t != 60)
        print(f"{name} {h}^2 {cin}->{cout} s{stride} batch {n}: {mb:6.1f} MB = {mb / 8e3 * 1e3:5.1f} us at 8 TB/s | best other "
              f"{old[0]:6.1f} us (tile {old[1]}) | tile 60 {res.get(60, float('nan')):6.1f} us = {mb / res.get(60, 1e9):4.2f} TB/s")


if __name__ == "__main__":
    main()
