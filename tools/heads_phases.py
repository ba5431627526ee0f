"""Time of the RoI stage (pool + heads launches) of Network.forward at a batch size, from the forward's stage marks:
python tools/heads_phases.py [batch]   (MILLIEYE_HEADS_STOP=1..4 ends roi_heads_mfma_kernel after that phase: profiling only)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g; g.build()
from millieye_amd import cfgs, synth
from millieye_amd.my_models import Network
from millieye_amd.yolov3.models import Darknet
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 8
net = Network(Darknet(cfgs.write_cfg("yolov3", "/tmp/hp_cfg")), 0.2).eval()
synth.fill_network_(net, "bench/yolov3", cls0_bias=3.0, cls_bias=-4.0)
net = net.cuda()
x = torch.from_numpy(synth.uniform("bench/frames/0", (batch, 3, 416, 416))).cuda()
maps_np, boxes_np = synth.radar_inputs("bench/radar/0", batch, 26, boxes_per_image=2)
maps_d, boxes_d = torch.from_numpy(maps_np).cuda(), torch.from_numpy(boxes_np).cuda()
def step():
    with torch.no_grad():
        return net(x, maps_d, boxes_d.clone(), 0)
for _ in range(10): step()
marks = []
def cb(name):
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.append((name, e))
net._stage_cb = cb
acc = {}
for _ in range(30):
    marks.clear(); step(); torch.cuda.synchronize()
    for (n0, e0), (n1, e1) in zip(marks[:-1], marks[1:]):
        acc[n1] = acc.get(n1, 0.0) + e0.elapsed_time(e1) / 30
print("batch", batch, "stop", os.environ.get("MILLIEYE_HEADS_STOP", "0"), " ".join(f"{k} {v * 1e3:.1f} us" for k, v in acc.items() if k != "detector"))
