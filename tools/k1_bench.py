"""Timing of the 1x1 Darknet-53 layer shapes in the 16-bit modes: per-tap tiles vs the weight-stationary streaming kernel
(tile 50, csrc/conv1x1_ws_h16.hip).  usage: python tools/k1_bench.py [batch]   (GPU box)"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from millieye_amd import hip  # noqa: E402

LAYERS = [(208, 64, 32), (104, 128, 64), (52, 256, 128), (26, 512, 256), (26, 768, 256), (52, 384, 128)]
OLD = (1, 2, 3, 4, 11, 12, 13, 14)


def time_tile(x, w, sc, sh, out, tile, reps=20):
    for _ in range(3):
        hip.conv2d_h16(x, w, sc, sh, 1, 1, 0, 1, out=out, tile=tile, split_k=1)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        hip.conv2d_h16(x, w, sc, sh, 1, 1, 0, 1, out=out, tile=tile, split_k=1)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    dev = torch.device("cuda")
    big = torch.randn((4096, 4096), device=dev)
    for _ in range(30):
        big @ big
    torch.cuda.synchronize()
    for h, cin, cout in LAYERS:
        x = torch.randn((n, h, h, cin), device=dev).to(torch.bfloat16)
        w = (torch.randn((cout, 1, 1, cin), device=dev) / cin ** 0.5).to(torch.bfloat16)
        sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
        out = torch.empty((n, h, h, cout), device=dev, dtype=torch.bfloat16)
        mb = (x.numel() + out.numel()) * 2 / 1e6
        res = {}
        for tile in OLD + (50,):
            if 10 < tile < 20 and cin % 64:
                continue
            try:
                res[tile] = time_tile(x, w, sc, sh, out, tile)
            except hip.MeError:
                pass
        old = min((v, t) for t, v in res.items() if t != 50)
        ws = res.get(50)
        print(f"{h:4d}^2 {cin:4d}->{cout:4d} batch {n}: {mb:6.1f} MB in+out = {mb / 8e3 * 1e3:5.1f} us at 8 TB/s | per-tap best "
              f"{old[0]:6.1f} us (tile {old[1]}) | tile 50 {ws:6.1f} us = {mb / ws:4.2f} TB/s" if ws else f"{h} {cin}->{cout}: no ws")


if __name__ == "__main__":
    main()
