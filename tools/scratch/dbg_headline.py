import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from millieye_amd import cfgs, synth
from millieye_amd.my_models import Network, define_yolo
from tests import parity_helpers as ph
name, cfg, n, s, conf = "headline", "yolov3", 32, 416, 0.2
net = Network(define_yolo(ph.cfg_path(cfg)), conf).eval()
synth.fill_network_(net, name, cls0_bias=3.0, cls_bias=-4.0)
x = torch.from_numpy(synth.uniform(name + "/x", (n, 3, s, s)))
maps, rboxes = synth.radar_inputs(name + "/radar", n, s // 16, boxes_per_image=2)
maps, rboxes = torch.from_numpy(maps), torch.from_numpy(rboxes)
net = net.cuda()
with torch.no_grad():
    out = net(x.cuda(), maps.cuda(), rboxes.clone().cuda(), 0).cpu()
    fm32, y32 = net.base_detector(x.cuda())
    for f in (0, 30, 31):
        fm1, y1 = net.base_detector(x[f:f + 1].cuda())
        print("frame", f, "yolo max abs diff b32 vs b1:", float((y32[f] - y1[0]).abs().max()), "fm:", float((fm32[f] - fm1[0]).abs().max()))
        rb = rboxes[rboxes[:, 0] == f].clone(); rb[:, 0] = 0
        o1 = net(x[f:f + 1].cuda(), maps[f:f + 1].cuda(), rb.cuda(), 0).cpu()
        g = out[out[:, 0] == f].clone(); g[:, 0] = 0
        print("  rows", g.shape, o1.shape)
        if g.shape == o1.shape:
            d = (g - o1).abs()
            bad = (d > 1e-3 * o1.abs().clamp_min(1)).any(1).nonzero().flatten()
            print("  bad rows:", bad.tolist()[:10])
            for r in bad.tolist()[:4]:
                print("   b32:", g[r].tolist()); print("   b1 :", o1[r].tolist())
from oracle import network_ref
sd = {k: v.cpu() for k, v in net.state_dict().items()}
f = 31
rb = rboxes[rboxes[:, 0] == f].clone(); rb[:, 0] = 0
ref, internals = network_ref.network_forward(cfgs.KNOWN[cfg](), sd, x[f:f + 1], maps[f:f + 1], rb, 0, conf_thresh=conf, tap_module=91, return_internals=True)
g = out[out[:, 0] == f].clone(); g[:, 0] = 0
print("oracle rows", ref.shape, "gpu rows", g.shape)
d = (g - ref).abs()
bad = (d > 1e-3 * ref.abs().clamp_min(1)).any(1).nonzero().flatten()
print("bad rows:", bad.tolist())
for r in bad.tolist()[:6]:
    print(" gpu:", [round(v, 4) for v in g[r].tolist()]); print(" ref:", [round(v, 4) for v in ref[r].tolist()])
