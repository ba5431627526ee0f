"""RoI-pooling backward scatters of the stage-3 step (me_ps_roi_align_bwd_f32 / me_roi_align_bwd_f32) on clustered proposals.
usage: [MILLIEYE_ROI_BWD_XSPLIT=0|1] python tools/roi_bwd_bench.py [k] [n]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from millieye_amd import hip  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 320
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
g = np.random.RandomState(0)
centres = g.uniform(60, 360, size=(n, 3, 2))          # three objects per frame: the proposals pile up on them
rois = np.zeros((k, 5), dtype=np.float32)
for i in range(k):
    b = i * n // k
    c = centres[b, g.randint(3)] + g.uniform(-6, 6, size=2)
    wh = g.uniform(40, 120, size=2)
    rois[i] = [b, c[0] - wh[0] / 2, c[1] - wh[1] / 2, c[0] + wh[0] / 2, c[1] + wh[1] / 2]
rd = torch.from_numpy(rois).cuda()
lib = hip.lib()
for ps, ch, name in ((True, 490, "ps_roi_align_bwd (490 channels)"), (False, 10, "roi_align_bwd (10 channels)")):
    gout = torch.randn((k, 10, 7, 7), device="cuda")
    gmap = torch.zeros((n, 26, 26, ch), device="cuda")
    fn = lib.me_ps_roi_align_bwd_f32 if ps else lib.me_roi_align_bwd_f32
    def run():
        hip.check(fn(gout.data_ptr(), rd.data_ptr(), k, n, 26, 26, ch, 7, 1.0 / 16, gmap.data_ptr(), ch, hip.stream_ptr()), "roi bwd")
    for _ in range(3):
        run()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        run()
    b.record()
    torch.cuda.synchronize()
    print(f"{name}: k = {k}, n = {n}: {a.elapsed_time(b) / 20 * 1e3:.1f} us  (MILLIEYE_ROI_BWD_XSPLIT={os.environ.get('MILLIEYE_ROI_BWD_XSPLIT', '1')})")
