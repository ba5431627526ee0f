"""Forward vs data-gradient timing (hip.conv2d_auto, the training path's tuned convolution) over the conv shapes of yolov3.cfg
at 416^2.  usage: python tools/dgrad_bench.py [batch]   (GPU box)"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("MILLIEYE_TUNE_CACHE", "/tmp/dgrad_bench_tune.json")
from millieye_amd import hip  # noqa: E402
from millieye_amd.detector_train import _parity_weights  # noqa: E402
from tools.wgrad_bench import SHAPES  # noqa: E402


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    dev = torch.device("cuda")
    tf = td = fl_all = 0.0
    for cnt, h, cin, cout, k, s in SHAPES:
        if cin < 16:
            continue
        pad = (k - 1) // 2
        ho = (h + 2 * pad - k) // s + 1
        x = torch.randn((n, h, h, cin), device=dev)
        dy = torch.randn((n, ho, ho, cout), device=dev)
        w = torch.randn((cout, k, k, cin), device=dev) / (k * k * cin) ** 0.5
        wt = hip.tile_weights_f32(w) if cin % 16 == 0 else None
        one_o, zero_o = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
        one_i, zero_i = torch.ones(cin, device=dev), torch.zeros(cin, device=dev)
        fwd = timed(lambda: hip.conv2d_auto(x, w, one_o, zero_o, k, s, pad, hip.ACT_LEAKY, wgt_tiled=wt))
        if cout % 4:
            dg = float("nan")
        elif s == 1:
            rot = w.flip(1, 2).permute(3, 1, 2, 0).contiguous()
            rt = hip.tile_weights_f32(rot) if cout % 16 == 0 else None
            dg = timed(lambda: hip.conv2d_auto(dy, rot, one_i, zero_i, k, 1, k - 1 - pad, hip.ACT_LINEAR, wgt_tiled=rt))
        else:
            pw = _parity_weights(w)
            o4, z4 = torch.ones(4 * cin, device=dev), torch.zeros(4 * cin, device=dev)
            from millieye_amd.detector_train import _PARITY_MASKS_ON, _PARITY_TAP_MASKS
            masks = (cin, _PARITY_TAP_MASKS) if _PARITY_MASKS_ON and cin % 32 == 0 and cout % 16 == 0 else None  # (what the step does)
            dg = timed(lambda: hip.conv2d_auto(dy, pw, o4, z4, 2, 1, 1, hip.ACT_LINEAR, tap_masks=masks))
        fl = 2.0 * n * ho * ho * cout * cin * k * k
        print(f"x{cnt} {h:3d}^2 {cin:4d}->{cout:4d} k{k} s{s}: fwd {fwd:7.1f} us {fl / fwd / 1e6:6.1f} TF/s | dgrad {dg:7.1f} us "
              f"{fl / dg / 1e6:6.1f} TF/s")
        tf += cnt * fwd
        if dg == dg:
            td += cnt * dg
        fl_all += cnt * fl
    print(f"total fwd {tf / 1e3:.2f} ms ({fl_all / tf / 1e6:.1f} TF/s), dgrad {td / 1e3:.2f} ms")


if __name__ == "__main__":
    main()
