"""Timing of the 3x3 / stride-1 Darknet-53 layer shapes: per-tap tiles vs the patch-resident big tiles (ids >= 100).
usage: python tools/p8_bench.py [batch] [tile,tile,...]   (GPU box)
A tile given as "221/2" is timed with split_k = 2 (patch tiles cut along the 32-channel chunks, slabs + reduce launch)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from millieye_amd import hip  # noqa: E402

LAYERS = [(208, 32, 64), (104, 64, 128), (52, 128, 256), (26, 256, 512), (13, 512, 1024)]
if os.environ.get("P8_KSCAN"):   # fixed output shape, K swept: fixed cost (prologue + epilogue) vs per-stage cost
    LAYERS = [(26, 32, 512), (26, 64, 512), (26, 128, 512), (26, 256, 512), (26, 512, 512)]
OLD = (1, 2, 3, 4, 11, 12, 13, 14)
NEW = (110, 121, 131, 200, 201, 221, 301, 311, 321, 331, 421, 431, 441, 621, 721, 731, 810, 820, 821, 831, 841, 1210, 1221, 1231)


def time_tile(x, w, sc, sh, r, out, tile, reps=20, split=1):
    wt = hip.tile_weights_h16(w)
    for _ in range(3):
        hip.conv2d_h16(x, w, sc, sh, 3, 1, 1, 1, residual=r, out=out, tile=tile, split_k=split, wgt_tiled=wt)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        hip.conv2d_h16(x, w, sc, sh, 3, 1, 1, 1, residual=r, out=out, tile=tile, split_k=split, wgt_tiled=wt)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    new = tuple((int(t.split("/")[0]), int(t.split("/")[1])) if "/" in t else int(t) for t in sys.argv[2].split(",")) \
        if len(sys.argv) > 2 else NEW
    dev = torch.device("cuda")
    # clock pre-warm
    big = torch.randn((4096, 4096), device=dev)
    for _ in range(30):
        big @ big
    torch.cuda.synchronize()
    for h, cin, cout in LAYERS:
        x = torch.randn((n, h, h, cin), device=dev).to(torch.bfloat16)
        if os.environ.get("P8_ZERO_X"):   # DVFS probe: same traffic, zero products (the chip clocks to its power budget)
            x.zero_()
        if os.environ.get("P8_SMALL_X"):  # ... tiny but non-zero activations
            x.mul_(1e-30)
        w = (torch.randn((cout, 3, 3, cin), device=dev) / (9 * cin) ** 0.5).to(torch.bfloat16)
        sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
        r = torch.randn((n, h, h, cout), device=dev).to(torch.bfloat16)
        out = torch.empty((n, h, h, cout), device=dev, dtype=torch.bfloat16)
        flops = 2 * n * h * h * cout * 9 * cin
        res = {}
        for tile in OLD + new:
            tid, split = tile if isinstance(tile, tuple) else (tile, 1)
            if 10 < tid < 20 and cin % 64:
                continue
            try:
                res[tile] = time_tile(x, w, sc, sh, r, out, tid, split=split)
            except hip.MeError:
                pass
        old_best = min((v, t) for t, v in res.items() if not isinstance(t, tuple) and t < 100)
        line = f"{h:4d} {cin:4d}->{cout:4d}  per-tap best {old_best[0]:7.1f} us (tile {old_best[1]}, {flops / old_best[0] / 1e6:6.0f} TF) |"
        for t in new:
            if t in res:
                name = f"{t[0]}/{t[1]}" if isinstance(t, tuple) else str(t)
                line += f" {name}:{res[t]:6.1f}us/{flops / res[t] / 1e6:5.0f}TF"
        print(line, flush=True)


if __name__ == "__main__":
    main()
