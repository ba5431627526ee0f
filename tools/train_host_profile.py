"""Where does the host time of a stage-3 training step go?  (tools; GPU box)  python tools/train_host_profile.py [f32|bf16|f16]"""
import cProfile, os, pstats, random, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g; g.build()
from millieye_amd import cfgs, synth, parallel as par
from millieye_amd.my_models import Network
from millieye_amd.train_path import head_parameters
from millieye_amd.yolov3.models import Darknet
batch = 8
net = Network(Darknet(cfgs.write_cfg("yolov3", "/tmp/thp_cfg")), 0.2).eval()
synth.fill_network_(net, "bench/yolov3", cls0_bias=3.0, cls_bias=-4.0)
net = net.cuda()
if len(sys.argv) > 1 and sys.argv[1] != "f32":
    net.base_detector.compute_dtype = sys.argv[1]
x = torch.from_numpy(synth.uniform("bench/frames/0", (batch, 3, 416, 416))).cuda()
maps_np, boxes_np = synth.radar_inputs("bench/radar/0", batch, 26, boxes_per_image=2)
maps_d, boxes_d = torch.from_numpy(maps_np).cuda(), torch.from_numpy(boxes_np).cuda()
with torch.no_grad():
    det = net(x, maps_d, boxes_d.clone(), 1).cpu()
tg = []
for i in range(batch):
    rows_i = det[det[:, 0] == i]
    for j in (0, 3):
        if j < len(rows_i):
            b = rows_i[j, 1:5] / 416
            tg.append([i, 0, float((b[0] + b[2]) / 2), float((b[1] + b[3]) / 2), float(b[2] - b[0]) * 1.05, float(b[3] - b[1]) * 0.95])
targets = torch.tensor(tg, dtype=torch.float32).reshape(-1, 6)
net.train(); net.base_detector.eval()
heads = head_parameters(net)
from millieye_amd.optim import Adam
opt = Adam(heads, lr=5e-4)   # the loop's optimizer (millieye_amd/train.py)
random.seed(1)
x2 = x.clone()
flip = [0]
def step():
    cur, nxt = (x, x2) if flip[0] == 0 else (x2, x)
    flip[0] ^= 1
    if os.environ.get("PREFETCH", "1") != "0":
        net.queue_detector_prefetch(nxt)
    loss, out_rows, _m, _a = net(cur, maps_d, boxes_d.clone(), targets.clone())
    loss.backward()
    par.allreduce_gradients(heads)
    opt.step(); opt.zero_grad(set_to_none=True)
for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t0) / 20 * 1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(20): step()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats(os.environ.get("SORT", "cumulative")).print_stats(int(os.environ.get("ROWS", "28")))
