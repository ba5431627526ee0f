#!/usr/bin/env python
"""Micro-benchmark of me_conv2d_f32 on Darknet-53 layer shapes (tuning aid, GPU box only).

    python tools/conv_bench.py [--batch 8] [--tiles 0,1,2,3,4] [--reps 20] [--only 3x3]

Prints one line per (layer shape, tile id): microseconds, TFLOP/s, fraction of the 157.3 TF fp32 MFMA peak.
tile 0 = the library's own choice."""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

# (name, H(=W) of the input, cin, cout, ksize, stride) - the distinct conv shapes of yolov3.cfg @416
SHAPES = [
    ("s2 32->64 @416", 416, 32, 64, 3, 2),
    ("1x1 64->32 @208", 208, 64, 32, 1, 1),
    ("3x3 32->64 @208", 208, 32, 64, 3, 1),
    ("1x1 128->64 @104", 104, 128, 64, 1, 1),
    ("3x3 64->128 @104", 104, 64, 128, 3, 1),
    ("1x1 256->128 @52", 52, 256, 128, 1, 1),
    ("3x3 128->256 @52", 52, 128, 256, 3, 1),
    ("1x1 512->256 @26", 26, 512, 256, 1, 1),
    ("3x3 256->512 @26", 26, 256, 512, 3, 1),
    ("1x1 1024->512 @13", 13, 1024, 512, 1, 1),
    ("3x3 512->1024 @13", 13, 512, 1024, 3, 1),
    ("1x1 1024->255 @13", 13, 1024, 255, 1, 1),
    ("1x1 768->256 @26", 26, 768, 256, 1, 1),
    ("s2 64->128 @208", 208, 64, 128, 3, 2),
    ("1x1 384->128 @52", 52, 384, 128, 1, 1),
    ("1x1 256->255 @52", 52, 256, 255, 1, 1),
    ("1x1 512->255 @26", 26, 512, 255, 1, 1),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--tiles", default="0,1,2,3,4")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--only", default="")
    ap.add_argument("--splits", default="0", help="comma list of split_k values (0 = auto)")
    ap.add_argument("--prewarm", type=float, default=0.25, help="seconds of untimed load before each measurement")
    ap.add_argument("--res", action="store_true", help="fused residual add (the [shortcut] epilogue)")
    ap.add_argument("--zero", action="store_true", help="zero activations: same instructions and traffic, zero products (DVFS check)")
    ap.add_argument("--custom", default="", help="extra shape 'hw,cin,cout,k,stride' (square input)")
    args = ap.parse_args()
    if args.custom:
        hw_, ci_, co_, k_, s_ = (int(v) for v in args.custom.split(","))
        SHAPES.append((f"custom {k_}x{k_} {ci_}->{co_} @{hw_}", hw_, ci_, co_, k_, s_))
        args.only = args.only or "custom"
    import __graft_entry__ as g
    g.build()
    from millieye_amd import hip

    lib = hip.lib()
    tiles = [int(t) for t in args.tiles.split(",")]
    dev = "cuda"
    for name, hw, cin, cout, k, s in SHAPES:
        if args.only and args.only not in name:
            continue
        n = args.batch
        pad = (k - 1) // 2
        ho = (hw + 2 * pad - k) // s + 1
        x = torch.randn((n, hw, hw, cin), device=dev)
        if args.zero:
            x.zero_()
        w = torch.randn((cout, k, k, cin), device=dev) * 0.05
        sc = torch.ones(cout, device=dev)
        sh = torch.zeros(cout, device=dev)
        y = torch.empty((n, ho, ho, cout), device=dev)
        d = hip.ConvDesc()
        r = torch.randn_like(y) if args.res else None
        d.x, d.wgt, d.scale, d.shift, d.res, d.y = x.data_ptr(), w.data_ptr(), sc.data_ptr(), sh.data_ptr(), (r.data_ptr() if args.res else None), y.data_ptr()
        d.x_pitch, d.res_pitch, d.y_pitch = cin, (cout if args.res else 0), cout
        d.n, d.h, d.w, d.cin, d.cout, d.ksize, d.stride, d.pad, d.ho, d.wo = n, hw, hw, cin, cout, k, s, pad, ho, ho
        d.act, d.upsample, d.x_nchw = 1, 1, 0
        flops = lib.me_conv2d_flops(d)
        row = []
        ws = torch.empty(256 * 1024 * 1024, dtype=torch.uint8, device=dev)
        d.workspace, d.workspace_bytes = ws.data_ptr() + (-ws.data_ptr()) % 256, ws.numel() - 256
        for t, sk in [(t, sk) for t in tiles for sk in [int(v) for v in args.splits.split(",")]]:
            d.tile, d.split_k = t, sk
            if sk > 1 and not 41 <= t <= 45 and sk * n * ho * ho * cout * 4 > ws.numel() - 256:
                continue
            stream = hip.stream_ptr()
            if lib.me_conv2d_f32(C.byref(d), stream) != 0:   # a tile that refuses the shape (weight-stationary ids 50 / 60, patch tiles)
                row.append(f"t{t}/k{sk}: refused")
                continue
            torch.cuda.synchronize()
            # the GPU needs ~100 ms of sustained load to reach its steady clocks: pre-warm every variant for
            # the same wall time, otherwise the first one measured looks 15 % slower than it is
            import time
            t_end = time.perf_counter() + args.prewarm
            while time.perf_counter() < t_end:
                for _ in range(5):
                    lib.me_conv2d_f32(C.byref(d), stream)
                torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(args.reps):
                lib.me_conv2d_f32(C.byref(d), stream)
            b.record()
            torch.cuda.synchronize()
            us = a.elapsed_time(b) * 1e3 / args.reps
            tf = flops / us / 1e6
            row.append(f"t{t}/k{sk}: {us:7.1f} us {tf:5.1f} TF ({tf / 157.3:4.0%})")
        print(f"{name:22s} n={n:<3d} {flops / 1e9:7.2f} GF | " + " | ".join(row), flush=True)


if __name__ == "__main__":
    main()
