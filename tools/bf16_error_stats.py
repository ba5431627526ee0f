"""Error statistics of the bf16 storage mode: HIP bf16 vs the oracle's bf16 restatement vs the fp32 oracle.
usage (GPU box): python tools/bf16_error_stats.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tests import parity_helpers as ph  # noqa: E402


def stats(a, b):
    d = (a - b).abs()
    scale = torch.clamp(b.abs(), min=1.0)
    rel = d / scale
    rms = float((d.pow(2).mean() / b.pow(2).mean()).sqrt())
    return f"max rel {float(rel.max()):.3e}  p99.9 {float(rel.flatten().kthvalue(max(1, int(rel.numel() * 0.999))).values):.3e}  " \
           f"mean {float(rel.mean()):.3e}  rms/rms {rms:.3e}"


DT = sys.argv[1] if len(sys.argv) > 1 else "bf16"   # bf16 | f16


def main():
    from oracle import darknet_ref
    from millieye_amd.engine import pick_tap_module
    for name, n, s in (("yolov3-tiny-12", 2, 96), ("yolov3-tiny-12", 1, 416), ("yolov3-tiny-coco", 3, 160),
                       ("yolov3", 2, 64), ("yolov3", 1, 416)):
        model = ph.make_darknet(name)
        x = ph.frames(f"{name}/{n}/{s}", n, s)
        tap = pick_tap_module(model.module_defs)
        f32_fm, f32_y = darknet_ref.darknet_forward(ph.cfg_text(name), model.state_dict(), x, tap_module=tap)
        b16_fm, b16_y = darknet_ref.darknet_forward(ph.cfg_text(name), model.state_dict(), x, tap_module=tap, storage=DT)
        model = model.cuda()
        model.compute_dtype = DT
        with torch.no_grad():
            fm, y = model(x.cuda())
        torch.cuda.synchronize()
        fm, y = fm.cpu(), y.cpu()
        print(f"== {name} n={n} s={s} storage={DT}")
        print("  hip16 vs oracle16  yolo:", stats(y, b16_y))
        print("  hip16 vs oracle16  fmap:", stats(fm, b16_fm))
        print("  hip16 vs oracle32  yolo:", stats(y, f32_y))
        print("  orac16 vs oracle32 yolo:", stats(b16_y, f32_y))
        print("  hip16 vs oracle32  fmap:", stats(fm, f32_fm))
        model.compute_dtype = "f32"
        with torch.no_grad():
            fm32, y32 = model(x.cuda())
        print("  hip32 vs oracle32  yolo:", stats(y32.cpu(), f32_y))


if __name__ == "__main__":
    main()
