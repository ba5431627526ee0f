"""Per-layer timing of me_conv2d_h16 on Darknet-53 layer shapes (HIP events, every tile / split-K candidate).
usage: python tools/conv16_bench.py [batch]   (run on the GPU box)"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from millieye_amd import hip  # noqa: E402

TILES = (1, 2, 3, 4, 11, 12, 13, 14)
LAYERS = [  # h, cin, cout, k, s, residual
    (416, 32, 64, 3, 2, False), (208, 64, 32, 1, 1, False), (208, 32, 64, 3, 1, True),
    (208, 64, 128, 3, 2, False), (104, 128, 64, 1, 1, False), (104, 64, 128, 3, 1, True),
    (104, 128, 256, 3, 2, False), (52, 256, 128, 1, 1, False), (52, 128, 256, 3, 1, True),
    (52, 256, 512, 3, 2, False), (26, 512, 256, 1, 1, False), (26, 256, 512, 3, 1, True),
    (26, 512, 1024, 3, 2, False), (13, 1024, 512, 1, 1, False), (13, 512, 1024, 3, 1, True),
]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    global TILES
    if len(sys.argv) > 2:
        TILES = tuple(int(t) for t in sys.argv[2].split(","))
    dev = torch.device("cuda")
    tot_best = 0.0
    tot_flops = 0
    for h, cin, cout, k, s, res in LAYERS:
        pad = (k - 1) // 2
        ho = (h + 2 * pad - k) // s + 1
        x = torch.randn((n, h, h, cin), device=dev).to(torch.bfloat16)
        w = (torch.randn((cout, k, k, cin), device=dev) / (k * k * cin) ** 0.5).to(torch.bfloat16)
        sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
        r = torch.randn((n, ho, ho, cout), device=dev).to(torch.bfloat16) if res else None
        out = torch.empty((n, ho, ho, cout), device=dev, dtype=torch.bfloat16)
        flops = 2 * n * ho * ho * cout * k * k * cin
        rows = []
        for tile in TILES:
            if 10 < tile < 20 and cin % 64:
                continue  # identical to tiles 1..4 for this shape
            for split in ((1, 2, 4) if len(TILES) > 6 else (1,)):
                try:
                    for _ in range(3):
                        hip.conv2d_h16(x, w, sc, sh, k, s, pad, 1, residual=r, out=out, tile=tile, split_k=split)
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    for _ in range(10):
                        hip.conv2d_h16(x, w, sc, sh, k, s, pad, 1, residual=r, out=out, tile=tile, split_k=split)
                    b.record()
                    torch.cuda.synchronize()
                    rows.append((a.elapsed_time(b) / 10, tile, split))
                except hip.MeError as exc:
                    rows.append((float("inf"), tile, split))
        rows = sorted(r for r in rows if r[0] != float('inf'))   # tiles that do not apply to this shape
        if not rows:
            continue
        best = rows[0]
        tot_best += best[0]
        tot_flops += flops
        byts = 2 * (x.numel() + out.numel() * (2 if res else 1) + w.numel())
        print(f"{h:4d} {cin:5d}->{cout:5d} k{k} s{s} res={int(res)}  best {best[0]*1e3:8.1f} us  tile {best[1]:2d} split {best[2]}"
              f"  {flops / best[0] / 1e9:7.1f} TF  {byts / best[0] / 1e6:7.1f} GB/s   next: "
              + " ".join(f"{t}/{sp}:{ms*1e3:.0f}" for ms, t, sp in rows[1:8]))
    print(f"sum of best: {tot_best:.3f} ms, {tot_flops / tot_best / 1e9:.1f} TF average")


if __name__ == "__main__":
    main()
