"""which rows of a frame's batch-1 run are missing from the batch-32 run in the bf16 storage mode, and why.

configs[2] (stage-2 network, Darknet-53 416x416, batch 32) under the pinned plan of tests/test_gpu_configs.py.  For the sampled
frames every batch-1 row is classified against the batch-32 rows of the same frame:
  matched        same class, every corner within 4 px, refined confidence within 0.1   (the test's criterion)
  moved          a same-class row exists but its nearest one is > 4 px away -> prints the distance
  conf           within 4 px but the confidence differs by > 0.1
  vanished       no same-class row at all -> prints the row's objectness x class score against the threshold 0.2
and the same the other way round (batch-32 rows without a batch-1 counterpart).  The detector's decoded rows of the frame
(before NMS) are compared as well: share of rows whose objectness differs by more than 1e-2 between the two runs.
usage (GPU box): python tools/bf16_row_diff.py [bf16|f16]"""
import os
import shutil
import sys
import tempfile

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from millieye_amd import synth  # noqa: E402
from tests.test_gpu_configs import PLAN, _frame_rows, _m2_net, _tie_share as tie_share  # noqa: E402

DT = sys.argv[1] if len(sys.argv) > 1 else "bf16"


def classify(ref, got, px=4.0, tol=0.1):
    out = dict(matched=0, moved=[], conf=[], vanished=[])
    for row in ref:
        cand = got[got[:, 7] == row[7]]
        if len(cand) == 0:
            out["vanished"].append(row)
            continue
        d = (cand[:, 1:5] - row[1:5]).abs().max(dim=1).values
        j = int(d.argmin())
        if float(d[j]) > px:
            out["moved"].append((row, float(d[j]), cand[j]))
        elif abs(float(cand[j, 5] - row[5])) > tol:
            out["conf"].append((row, cand[j]))
        else:
            out["matched"] += 1
    return out


def main():
    tmp = tempfile.mkdtemp()
    shutil.copy(PLAN, os.path.join(tmp, "plan.json"))
    os.environ["MILLIEYE_TUNE_CACHE"] = os.path.join(tmp, "plan.json")
    os.environ["MILLIEYE_AUTOTUNE"] = "1"
    name, n, s = "m2b32", 32, 416
    net = _m2_net(name)
    net = net.to(net.device)
    x = torch.from_numpy(synth.uniform(name + "/x", (n, 3, s, s))).cuda()
    net.base_detector.compute_dtype = DT
    with torch.no_grad():
        got = net(x)
        _fm, y32 = net.base_detector(x)
        for f in (0, 11, 19, 31):
            one = net(x[f:f + 1])
            _fm, y1 = net.base_detector(x[f:f + 1])
            mine = _frame_rows(got, f)
            c = classify(one, mine)
            back = classify(mine, one)
            print(f"== frame {f} [{DT}]: batch-1 rows {one.shape[0]}, batch-32 rows {mine.shape[0]}; matched {c['matched']} "
                  f"moved {len(c['moved'])} conf {len(c['conf'])} vanished {len(c['vanished'])} | the other way: matched "
                  f"{back['matched']} moved {len(back['moved'])} conf {len(back['conf'])} vanished {len(back['vanished'])}")
            print(f"   tie-aware share (any class within 4 px / 0.1, or IoU >= 0.45 with a kept neighbour): "
                  f"{tie_share(one, mine):.1%} of the batch-1 rows, {tie_share(mine, one):.1%} the other way")
            for row, d, near in c["moved"]:
                print(f"   moved    {d:7.1f} px  class {int(row[7])} conf {float(row[5]):.3f} (obj {float(row[6]):.3f})  box "
                      f"{[round(float(v), 1) for v in row[1:5]]} nearest {[round(float(v), 1) for v in near[1:5]]}")
            for row, near in c["conf"]:
                print(f"   conf     class {int(row[7])} {float(row[5]):.3f} vs {float(near[5]):.3f}")
            for row in c["vanished"]:
                print(f"   vanished class {int(row[7])} conf {float(row[5]):.3f} obj {float(row[6]):.3f} box "
                      f"{[round(float(v), 1) for v in row[1:5]]}")
            # decoded rows of the frame before NMS: how far apart are the two runs at the source?
            a, b = y32[f].float().cpu(), y1[0].float().cpu()
            dobj = (a[:, 4] - b[:, 4]).abs()
            near_thr = ((a[:, 4] - 0.2).abs() < 0.02) | ((b[:, 4] - 0.2).abs() < 0.02)
            flips = ((a[:, 4] >= 0.2) != (b[:, 4] >= 0.2))
            dbox = (a[:, :4] - b[:, :4]).abs().max(dim=1).values
            conf_rows = (a[:, 4] >= 0.2) | (b[:, 4] >= 0.2)
            print(f"   decode: {a.shape[0]} rows; objectness |d| max {float(dobj.max()):.3f}, > 1e-2 on {int((dobj > 1e-2).sum())}; "
                  f"rows within 0.02 of the threshold {int(near_thr.sum())}, threshold flips {int(flips.sum())}; "
                  f"rows over the threshold {int(conf_rows.sum())}, of them box |d| > 4 px {int((dbox[conf_rows] > 4).sum())} "
                  f"(max {float(dbox[conf_rows].max()) if conf_rows.any() else 0:.1f} px)")


if __name__ == "__main__":
    main()
