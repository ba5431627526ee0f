"""One launch for a 1x1 -> 3x3 (+ shortcut) bottleneck (me_bneck_h16) against the two tuned launches it replaces.
usage: python tools/bneck_bench.py [batch] [bf16|f16]   (GPU box)"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from millieye_amd import hip  # noqa: E402

LAYERS = [(52, 256, 128, 256, (1,), (221, 431, 131, 200, 110)), (104, 128, 64, 128, (3, 4), (131, 431, 121, 14))]
K1_TILES = (50, 1, 2, 3, 4, 11, 12, 13, 14)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    half = torch.float16 if len(sys.argv) > 2 and sys.argv[2] == "f16" else torch.bfloat16
    dev = torch.device("cuda")
    big = torch.randn((4096, 4096), device=dev)
    for _ in range(30):
        big @ big
    torch.cuda.synchronize()
    for h, cin, cmid, cout, tiles, t3 in LAYERS:
        x = torch.randn((n, h, h, cin), device=dev).to(half)
        w1 = (torch.randn((cmid, 1, 1, cin), device=dev) / cin ** 0.5).to(half)
        w2 = (torch.randn((cout, 3, 3, cmid), device=dev) / (9 * cmid) ** 0.5).to(half)
        s1, t1 = torch.ones(cmid, device=dev), torch.zeros(cmid, device=dev)
        s2, t2 = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
        w1t, w2t = hip.tile_weights_h16(w1), hip.tile_weights_h16(w2)
        mid = torch.empty((n, h, h, cmid), device=dev, dtype=half)
        out = torch.empty((n, h, h, cout), device=dev, dtype=half)
        best1 = best3 = (1e9, 0)
        for t in K1_TILES:
            try:
                us = timed(lambda: hip.conv2d_h16(x, w1, s1, t1, 1, 1, 0, 1, out=mid, tile=t, split_k=1))
                best1 = min(best1, (us, t))
            except hip.MeError:
                pass
        for t in t3:
            try:
                us = timed(lambda: hip.conv2d_h16(mid, w2, s2, t2, 3, 1, 1, 1, residual=x, out=out, tile=t, split_k=1, wgt_tiled=w2t))
                best3 = min(best3, (us, t))
            except hip.MeError:
                pass

        def pair():
            hip.conv2d_h16(x, w1, s1, t1, 1, 1, 0, 1, out=mid, tile=best1[1], split_k=1)
            hip.conv2d_h16(mid, w2, s2, t2, 3, 1, 1, 1, residual=x, out=out, tile=best3[1], split_k=1, wgt_tiled=w2t)
        us_pair = timed(pair)
        flops = 2 * n * h * h * (cin * cmid + 9 * cmid * cout)
        line = (f"{h:4d} {cin}->{cmid}->{cout}: 1x1 {best1[0]:6.1f} us (tile {best1[1]}) + 3x3 {best3[0]:6.1f} us (tile {best3[1]}) "
                f"= pair back to back {us_pair:6.1f} us |")
        for tile in tiles:
            try:
                us = timed(lambda: hip.bneck_h16(x, w1, s1, t1, w2, s2, t2, residual=x, out=out, tile=tile, w1_tiled=w1t, w2_tiled=w2t))
                line += f" one launch tile {tile}: {us:6.1f} us ({flops / us / 1e6:5.0f} TF, {us / us_pair:.2f}x)"
            except hip.MeError as exc:
                line += f" tile {tile}: refused ({exc})"
        print(line, flush=True)


if __name__ == "__main__":
    main()
