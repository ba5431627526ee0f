"""fp32: per-tap tiles vs the patch-resident big tiles (ids >= 100) on the 3x3 / stride-1 Darknet-53 layer shapes.
usage: python tools/p8_bench_f32.py [batch]   (GPU box)"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from millieye_amd import hip  # noqa: E402

LAYERS = [(104, 64, 128), (52, 128, 256), (26, 256, 512), (13, 512, 1024)]
OLD = (1, 2, 3, 5)
NEW = (100, 110, 121, 131, 200, 201, 221, 311, 321)


def time_tile(x, w, wt, sc, sh, r, out, tile, reps=6):
    kw = dict(residual=r, out=out, tile=tile, split_k=1)
    if tile >= 100:
        kw["wgt_tiled"] = wt
    for _ in range(2):
        hip.conv2d(x, w, sc, sh, 3, 1, 1, 1, **kw)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        hip.conv2d(x, w, sc, sh, 3, 1, 1, 1, **kw)
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
        x = torch.randn((n, h, h, cin), device=dev)
        w = torch.randn((cout, 3, 3, cin), device=dev) / (9 * cin) ** 0.5
        wt = hip.tile_weights_f32(w)
        sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
        r = torch.randn((n, h, h, cout), device=dev)
        out = torch.empty((n, h, h, cout), device=dev)
        flops = 2 * n * h * h * cout * 9 * cin
        res = {}
        for tile in OLD + NEW:
            try:
                res[tile] = time_tile(x, w, wt, sc, sh, r, out, tile)
            except hip.MeError:
                pass
        old_best = min((v, t) for t, v in res.items() if t < 100)
        line = f"{h:4d} {cin:4d}->{cout:4d}  per-tap best {old_best[0]:7.1f} us (tile {old_best[1]}, {flops / old_best[0] / 1e6:5.1f} TF) |"
        for t in NEW:
            if t in res:
                line += f" {t}:{res[t]:6.1f}us/{flops / res[t] / 1e6:5.1f}TF"
        print(line, flush=True)


if __name__ == "__main__":
    main()
