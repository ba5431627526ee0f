import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g; g.build()
from millieye_amd import cfgs, synth, hip
from millieye_amd.yolov3.models import Darknet
from millieye_amd.my_models import Network
cfg = cfgs.write_cfg("yolov3", "/tmp/nmsdbg_cfg")
net = Network(Darknet(cfg), 0.2).eval()
synth.fill_network_(net, "bench/yolov3", cls0_bias=3.0, cls_bias=-4.0)
net = net.cuda()
x = torch.from_numpy(synth.uniform("bench/frames/0", (32, 3, 416, 416))).cuda()
with torch.no_grad():
    plan, yo = net.base_detector._run(x)
    torch.cuda.synchronize()
    cand = (yo[..., 4] >= 0.2).sum(1)
    print("candidates per image: min/mean/max", int(cand.min()), float(cand.float().mean()), int(cand.max()))
    for _ in range(3):
        det, cnt = hip.nms_batched(yo, 0.2, 0.5, 200, writeback_xyxy=False)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        det, cnt = hip.nms_batched(yo, 0.2, 0.5, 200, writeback_xyxy=False)
    b.record(); torch.cuda.synchronize()
    print("nms_batched ms:", a.elapsed_time(b) / 20, "kept:", cnt.cpu().tolist()[:8])
    for md in (10, 50, 100, 200):
        for _ in range(3):
            hip.nms_batched(yo, 0.2, 0.5, md, writeback_xyxy=False)
        torch.cuda.synchronize()
        a.record()
        for _ in range(20):
            hip.nms_batched(yo, 0.2, 0.5, md, writeback_xyxy=False)
        b.record(); torch.cuda.synchronize()
        print("max_det", md, "ms:", a.elapsed_time(b) / 20)
