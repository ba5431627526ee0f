"""Host-side profile of Network.forward (tools; GPU box): python tools/forward_host_profile.py [batch]"""
import cProfile, os, pstats, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g; g.build()
from millieye_amd import cfgs, synth
from millieye_amd.my_models import Network
from millieye_amd.yolov3.models import Darknet
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1
net = Network(Darknet(cfgs.write_cfg("yolov3", "/tmp/fhp_cfg")), 0.2).eval()
synth.fill_network_(net, "bench/yolov3", cls0_bias=3.0, cls_bias=-4.0)
net = net.cuda()
x = torch.from_numpy(synth.uniform("bench/frames/0", (batch, 3, 416, 416))).cuda()
maps_np, boxes_np = synth.radar_inputs("bench/radar/0", batch, 26, boxes_per_image=2)
maps_d, boxes_d = torch.from_numpy(maps_np).cuda(), torch.from_numpy(boxes_np).cuda()
def step():
    with torch.no_grad():
        return net(x, maps_d, boxes_d.clone(), 0)
for _ in range(30): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(100): step()
torch.cuda.synchronize(); print("ms/step", (time.perf_counter() - t0) * 10)
pr = cProfile.Profile(); pr.enable()
for _ in range(100): step()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
