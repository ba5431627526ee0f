"""Per-layer timing of the weight gradient (me_conv_wgrad_mfma_oihw_f32) over the distinct conv shapes of yolov3.cfg at 416^2.
usage: python tools/wgrad_bench.py [batch] [f32|bf16|f16]   (GPU box; 16-bit: me_conv_wgrad_h16 on 16-bit operands)"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from millieye_amd import hip  # noqa: E402

# (count in the network, input h, cin, cout, k, stride)
SHAPES = [(1, 416, 3, 32, 3, 1), (1, 416, 32, 64, 3, 2), (1, 208, 64, 32, 1, 1), (1, 208, 32, 64, 3, 1), (1, 208, 64, 128, 3, 2),
          (2, 104, 128, 64, 1, 1), (2, 104, 64, 128, 3, 1), (1, 104, 128, 256, 3, 2), (8, 52, 256, 128, 1, 1),
          (8, 52, 128, 256, 3, 1), (1, 52, 256, 512, 3, 2), (8, 26, 512, 256, 1, 1), (8, 26, 256, 512, 3, 1),
          (1, 26, 512, 1024, 3, 2), (7, 13, 1024, 512, 1, 1), (7, 13, 512, 1024, 3, 1), (1, 13, 1024, 51, 1, 1),
          (1, 13, 512, 256, 1, 1), (1, 26, 768, 256, 1, 1), (2, 26, 512, 256, 1, 1), (3, 26, 256, 512, 3, 1),
          (1, 26, 512, 51, 1, 1), (1, 26, 256, 128, 1, 1), (1, 52, 384, 128, 1, 1), (2, 52, 256, 128, 1, 1),
          (3, 52, 128, 256, 3, 1), (1, 52, 256, 51, 1, 1)]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    mode = sys.argv[2] if len(sys.argv) > 2 else "f32"
    half = {"bf16": torch.bfloat16, "f16": torch.float16}.get(mode)
    dev = torch.device("cuda")
    big = torch.randn((4096, 4096), device=dev)
    for _ in range(30):
        big @ big
    total = flops_total = 0.0
    for cnt, h, cin, cout, k, s in SHAPES:
        pad = (k - 1) // 2
        ho = (h + 2 * pad - k) // s + 1
        x = torch.randn((n, h, h, cin), device=dev)
        dy = torch.randn((n, ho, ho, cout), device=dev)
        fn = hip.conv_wgrad
        if half is not None:
            if cin % 8 or cout % 8:
                continue   # (stem / detection convolutions: fp32 kernels in the mixed-precision step)
            x, dy, fn = x.to(half), dy.to(half), hip.conv_wgrad_h16
        for _ in range(3):
            fn(x, dy, k, s, pad, oihw=True)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            fn(x, dy, k, s, pad, oihw=True)
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) / 10 * 1e3
        fl = 2.0 * n * ho * ho * cout * cin * k * k
        total += cnt * us
        flops_total += cnt * fl
        print(f"x{cnt} {h:3d}^2 {cin:4d}->{cout:4d} k{k} s{s}: {us:7.1f} us  {fl / us / 1e6:6.1f} TF/s   (x{cnt} = {cnt * us:7.1f} us)")
    print(f"total {total / 1e3:.2f} ms, {flops_total / total / 1e6:.1f} TF/s")


if __name__ == "__main__":
    main()
