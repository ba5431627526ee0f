"""NMS stage in isolation: me_nms_batched_f32 on a synthetic prediction tensor (GPU box).
usage: python tools/nms_bench.py [n] [rows] [classes] [pass_fraction]     (rocprofv3 --kernel-trace --stats -- python ... for the
per-kernel split; MILLIEYE_NMS_LEGACY=1 = the single-workgroup select kernel)"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from millieye_amd import hip  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    rows = int(sys.argv[2]) if len(sys.argv) > 2 else 10647
    nc = int(sys.argv[3]) if len(sys.argv) > 3 else 80
    frac = float(sys.argv[4]) if len(sys.argv) > 4 else 0.25
    dev = torch.device("cuda")
    g = torch.Generator(device="cpu").manual_seed(1)
    cxcy = torch.rand((n, rows, 2), generator=g) * 416
    wh = torch.rand((n, rows, 2), generator=g) * 152 + 8
    u = torch.rand((n, rows, 1), generator=g)
    conf = torch.where(u < frac, 0.2 + 0.8 * u / frac, 0.19 * u)
    cls = torch.rand((n, rows, nc), generator=g) * 0.1
    cls[..., 0] += 0.5  # one class: every candidate competes with every other (the bench weights do the same)
    pred = torch.cat([cxcy, wh, conf, cls], -1).contiguous().to(dev)
    for _ in range(5):
        det, cnt = hip.nms_batched(pred, 0.2, 0.5, 200, writeback_xyxy=False)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    a.record()
    for _ in range(reps):
        det, cnt = hip.nms_batched(pred, 0.2, 0.5, 200, writeback_xyxy=False)
    b.record()
    torch.cuda.synchronize()
    print(f"n {n} rows {rows} classes {nc}: {a.elapsed_time(b) / reps * 1e3:.1f} us per call, kept {cnt.tolist()[:4]}..., "
          f"candidates ~{int(frac * rows)} per image, legacy={os.environ.get('MILLIEYE_NMS_LEGACY', '0')}")


if __name__ == "__main__":
    main()
