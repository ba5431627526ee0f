"""Timing of me_pack_conv_f32 (all four packed copies) over the weight shapes of yolov3.cfg.  usage: python tools/pack_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from millieye_amd import hip  # noqa: E402

# (count, cout, cin, k)
SHAPES = [(1, 32, 3, 3), (1, 64, 32, 3), (1, 32, 64, 1), (3, 128, 64, 3), (2, 64, 128, 1), (12, 256, 128, 3), (11, 128, 256, 1),
          (12, 512, 256, 3), (11, 256, 512, 1), (8, 1024, 512, 3), (7, 512, 1024, 1), (1, 256, 768, 1), (1, 128, 384, 1),
          (1, 51, 1024, 1), (1, 51, 512, 1), (1, 51, 256, 1)]


def main():
    dev = torch.device("cuda")
    lib = hip.lib()
    total = tot_bytes = 0.0
    for cnt, cout, cin, k in SHAPES:
        w = torch.randn((cout, cin, k, k), device=dev)
        g, b, m, v = (torch.rand(cout, device=dev) + 0.5 for _ in range(4))
        ohwi = torch.empty((cout, k, k, cin), device=dev)
        tiled = torch.empty((k * k, cin // 16, cout, 16), device=dev) if cin % 16 == 0 else None
        rot = torch.empty((cin, k, k, cout), device=dev)
        rott = torch.empty((k * k, cout // 16, cin, 16), device=dev) if cout % 16 == 0 else None
        sc, sh = torch.empty(cout, device=dev), torch.empty(cout, device=dev)
        ptr = lambda t: t.data_ptr() if t is not None else None  # noqa: E731

        def run():
            hip.check(lib.me_pack_conv_f32(w.data_ptr(), cout, cin, k, None, g.data_ptr(), b.data_ptr(), m.data_ptr(),
                                           v.data_ptr(), 1e-5, ohwi.data_ptr(), ptr(tiled), rot.data_ptr(), ptr(rott),
                                           sc.data_ptr(), sh.data_ptr(), hip.stream_ptr()), "pack")
        for _ in range(3):
            run()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            run()
        e.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(e) / 20 * 1e3
        nbytes = w.numel() * 4 * (1 + 1 + (tiled is not None) + 1 + (rott is not None))
        total += cnt * us
        tot_bytes += cnt * nbytes
        print(f"x{cnt:2d} {cout:4d}x{cin:4d}x{k}x{k}: {us:7.1f} us  {nbytes / us / 1e6:5.2f} TB/s")
    print(f"total {total / 1e3:.2f} ms for {tot_bytes / 1e6:.0f} MB = {tot_bytes / total / 1e6:.2f} TB/s")


if __name__ == "__main__":
    main()
