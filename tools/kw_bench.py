"""Batch-1 layers of Darknet-53: the tuned (tile, split_k) of the implicit-GEMM kernels (main launch + slab reduce) against the
one-launch small-batch tiles 40 / 41 (K split over the waves of a workgroup).   usage: python tools/kw_bench.py [batch] [bf16|f16]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from millieye_amd import hip  # noqa: E402

LAYERS = [(13, 1024, 512, 1), (13, 512, 1024, 3), (26, 512, 256, 1), (26, 256, 512, 3), (52, 256, 128, 1), (52, 128, 256, 3),
          (104, 128, 64, 1), (104, 64, 128, 3), (13, 1024, 255, 1), (26, 768, 256, 1), (52, 384, 128, 1)]


def timed(fn, reps=30):
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
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    half = torch.float16 if len(sys.argv) > 2 and sys.argv[2] == "f16" else torch.bfloat16
    dev = torch.device("cuda")
    for h, cin, cout, k in LAYERS:
        x = torch.randn((n, h, h, cin), device=dev).to(half)
        w = (torch.randn((cout, k, k, cin), device=dev) / (k * k * cin) ** 0.5).to(half)
        sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
        y_f32 = cout == 255
        out = torch.empty((n, h, h, cout), device=dev, dtype=torch.float32 if y_f32 else half)
        best = (1e9, 0, 0)
        for tile in (1, 2, 3, 11, 12, 13):
            for split in (1, 2, 3, 4, 6, 8):
                try:
                    us = timed(lambda: hip.conv2d_h16(x, w, sc, sh, k, 1, (k - 1) // 2, 1, out=out, y_f32=y_f32, tile=tile, split_k=split))
                    best = min(best, (us, tile, split))
                except hip.MeError:
                    pass
        line = f"{h:4d} {cin:5d}->{cout:5d} k{k}: best implicit-GEMM {best[0]:6.1f} us (tile {best[1]}, split {best[2]})"
        for tile in (40, 41):
            us = timed(lambda: hip.conv2d_h16(x, w, sc, sh, k, 1, (k - 1) // 2, 1, out=out, y_f32=y_f32, tile=tile, split_k=1))
            line += f" | tile {tile}: {us:6.1f} us ({us / best[0]:.2f}x)"
        print(line, flush=True)


if __name__ == "__main__":
    main()
