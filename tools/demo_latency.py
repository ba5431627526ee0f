"""Per-frame latency of the demo step (millieye_amd.demo.FrameFuser): python tools/demo_latency.py  (GPU box)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from millieye_amd import cfgs, synth  # noqa: E402
from millieye_amd.demo import FrameFuser  # noqa: E402
from millieye_amd.my_models import Network  # noqa: E402
from millieye_amd.yolov3.models import Darknet  # noqa: E402
from tests.golden.make_golden import RADAR_CALIB, radar_points  # noqa: E402

frame = (synth.uniform("demo/frame", (480, 640, 3)) * 25).astype(np.uint8)   # dark: fusion mode
for cfg in ("yolov3-tiny-12", "yolov3"):
    for dtype in ("f32", "bf16"):
        net = Network(Darknet(cfgs.write_cfg(cfg, f"/tmp/demo_lat_{cfg}")), 0.2).eval()
        synth.fill_network_(net, "demo/" + cfg, cls0_bias=3.0, cls_bias=-4.0)
        net = net.to(net.device)
        net.base_detector.compute_dtype = dtype
        fuser = FrameFuser(net, RADAR_CALIB, model_mode=3, min_hits=2)
        for f in range(10):
            fuser(frame, [radar_points(f % 6)])
        torch.cuda.synchronize()
        host = 0.0
        t0 = time.perf_counter()
        n = 100
        for f in range(n):
            t1 = time.perf_counter()
            fuser.generator([radar_points(f % 6)])   # proposal generator alone (host), measured on a second pass
            host += time.perf_counter() - t1
        t_host = host / n
        t0 = time.perf_counter()
        for f in range(n):
            rows, info = fuser(frame, [radar_points(f % 6)])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"{cfg:16s} {dtype}: {dt * 1e3:6.2f} ms / frame ({1 / dt:6.1f} frames/s), of which radar proposals (host) "
              f"{t_host * 1e3:.2f} ms; rows {len(rows)}, mode {info['mode']}, radar boxes {info['radar_boxes']}")
