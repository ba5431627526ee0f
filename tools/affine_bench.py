"""Per-layer timing of the activation + BN-affine backward (me_affine_act_bwd_f32: dc = dy * act'(y) * scale, d gamma, d beta)
over the conv output shapes of yolov3.cfg at 416^2: us per layer, GB/s on the 12 algorithmic bytes per element (y and dy in, dc out).
usage: python tools/affine_bench.py [batch] [f32|bf16|f16]   (GPU box; 16-bit: me_affine_act_bwd_h16, 6 bytes per element)"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from millieye_amd import hip  # noqa: E402
from tools.wgrad_bench import SHAPES  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    dtype = sys.argv[2] if len(sys.argv) > 2 else "f32"
    half = {"bf16": torch.bfloat16, "f16": torch.float16}.get(dtype)
    dev = torch.device("cuda")
    lib = hip.lib()
    total = bytes_total = 0.0
    seen = {}
    for cnt, h, _cin, cout, k, s in SHAPES:
        pad = (k - 1) // 2
        ho = (h + 2 * pad - k) // s + 1
        seen[(ho, cout)] = seen.get((ho, cout), 0) + cnt
    for (ho, cout), cnt in sorted(seen.items(), reverse=True):
        if half is not None and cout % 8:
            continue   # (the detection convolutions: float32 maps, the fp32 kernel)
        rows = n * ho * ho
        y = torch.randn((rows, cout), device=dev)
        dy = torch.randn((rows, cout), device=dev)
        if half is not None:
            y, dy = y.to(half), dy.to(half)
        dc = torch.empty_like(y)
        scale, gam, bet = torch.rand(cout, device=dev) + 0.5, torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev)
        dshift, dgamma = torch.empty(cout, device=dev), torch.empty(cout, device=dev)
        ws = torch.empty(max(lib.me_affine_bwd_workspace_bytes(rows, cout), lib.me_affine_bwd_h16_workspace_bytes(rows, cout)),
                         dtype=torch.uint8, device=dev)

        def run16():
            hip.check(lib.me_affine_act_bwd_h16(y.data_ptr(), cout, dy.data_ptr(), cout, rows, cout, scale.data_ptr(), gam.data_ptr(),
                                                bet.data_ptr(), hip.ACT_LEAKY, dc.data_ptr(), cout, dshift.data_ptr(), dgamma.data_ptr(),
                                                ws.data_ptr(), hip.HALF_TYPES[half], hip.stream_ptr()), "me_affine_act_bwd_h16")

        def run():
            if half is not None:
                return run16()
            hip.check(lib.me_affine_act_bwd_f32(y.data_ptr(), cout, dy.data_ptr(), cout, rows, cout, scale.data_ptr(), gam.data_ptr(),
                                                bet.data_ptr(), hip.ACT_LEAKY, dc.data_ptr(), cout, dshift.data_ptr(), dgamma.data_ptr(),
                                                ws.data_ptr(), hip.stream_ptr()), "me_affine_act_bwd_f32")
        for _ in range(3):
            run()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            run()
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) / 20 * 1e3
        nbytes = (12.0 if half is None else 6.0) * rows * cout
        total += cnt * us
        bytes_total += cnt * nbytes
        print(f"x{cnt:2d} {ho:3d}^2 x {cout:4d}: {us:7.1f} us  {nbytes / us / 1e3:7.1f} GB/s  ({nbytes / 1e6:7.1f} MB)")
    print(f"total {total / 1e3:.3f} ms for {bytes_total / 1e9:.2f} GB = {bytes_total / total / 1e3:.0f} GB/s")


if __name__ == "__main__":
    main()
