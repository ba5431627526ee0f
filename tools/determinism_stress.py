"""Run-to-run determinism stress of the detector in every storage mode (catches LDS slot-reuse races): N forward passes of
the same frames must be bit-identical.  usage (GPU box): python tools/determinism_stress.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tests import parity_helpers as ph  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    bad = 0
    for name, n, s in (("yolov3", 1, 416), ("yolov3", 4, 416), ("yolov3-tiny-12", 2, 416)):
        model = ph.make_darknet(name).cuda()
        x = ph.frames(f"stress/{name}/{n}/{s}", n, s).cuda()
        for dtype in ("f32", "bf16", "f16"):
            model.compute_dtype = dtype
            with torch.no_grad():
                fm0, y0 = model(x)
                diff = 0
                for _ in range(reps):
                    fm, y = model(x)
                    diff += int(not (torch.equal(y, y0) and torch.equal(fm, fm0)))
            print(f"{name} n={n} {dtype}: {diff} of {reps} runs differ")
            bad += diff
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
