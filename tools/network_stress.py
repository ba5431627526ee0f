import sys, torch
sys.path.insert(0, "/root/repo")
from millieye_amd import cfgs, synth
from millieye_amd.my_models import Network
from millieye_amd.yolov3.models import Darknet
net = Network(Darknet(cfgs.write_cfg("yolov3", "/tmp/ns_cfg")), 0.2).eval()
synth.fill_network_(net, "bench/yolov3", cls0_bias=3.0, cls_bias=-4.0)
net = net.cuda()
for batch in (32, 3):
    x = torch.from_numpy(synth.uniform("bench/frames/0", (batch, 3, 416, 416))).cuda()
    maps_np, boxes_np = synth.radar_inputs("bench/radar/0", batch, 26, boxes_per_image=2)
    maps, boxes = torch.from_numpy(maps_np).cuda(), torch.from_numpy(boxes_np).cuda()
    for dt in ("f32", "bf16", "f16"):
        net.base_detector.compute_dtype = dt
        with torch.no_grad():
            ref = net(x, maps, boxes.clone(), 0).clone()
            bad = 0
            for _ in range(60):
                out = net(x, maps, boxes.clone(), 0)
                bad += int(out.shape != ref.shape or not torch.equal(out, ref))
        print(f"batch {batch} {dt}: rows {tuple(ref.shape)}, {bad} of 60 runs differ")
