"""Host-side profile of the mixed-precision detector step's forward and backward with the backward run in the MAIN thread (autograd
runs a custom Function's backward in its own thread, where cProfile does not look): where do the host's microseconds per layer go?
usage (GPU box): python tools/detector_train16_host_profile.py [bf16|f16] [batch]"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from millieye_amd import cfgs, hip, synth  # noqa: E402
from millieye_amd.detector_train16 import DetectorTrainer16  # noqa: E402
from millieye_amd.yolov3.models import Darknet  # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8
model = Darknet(cfgs.write_cfg("yolov3", "/tmp/dtp16_cfg")).eval()
synth.fill_darknet_(model, "bench/yolov3")
synth.trained_like_(model, "bench/yolov3/trained")
model = model.cuda()
model.compute_dtype = dtype
x = torch.from_numpy(synth.uniform("bench/frames/0", (batch, 3, 416, 416))).cuda()
tg = torch.tensor([[i, (3 * i) % 80, 0.3 + 0.04 * (i % 8), 0.4 + 0.03 * (i % 5), 0.2, 0.3] for i in range(batch)], dtype=torch.float32)
lib = hip.lib()


def step(timing=None):
    t0 = time.perf_counter()
    trainer = DetectorTrainer16(model)
    with torch.no_grad():
        st = trainer.forward(x)
        t1 = time.perf_counter()
        draws = {}
        for layer, (idx, raw) in zip(model.yolo_layers, sorted(st.raws.items())):
            layer.img_dim = x.shape[2]
            _value, bt = layer.loss_from_raw(raw, tg, return_targets=True)
            n, g, _, ch = raw.shape
            draw = torch.empty_like(raw)
            hip.check(lib.me_yolo_loss_bwd_f32(raw.data_ptr(), ch, n, g, layer.num_anchors, layer.num_classes, bt["obj"].data_ptr(),
                                               bt["noobj"].data_ptr(), bt["tx"].data_ptr(), bt["ty"].data_ptr(), bt["tw"].data_ptr(),
                                               bt["th"].data_ptr(), bt["tcls"].data_ptr(), bt["tconf"].data_ptr(), float(bt["n_obj"]),
                                               float(bt["n_noobj"]), float(layer.obj_scale), float(layer.noobj_scale), 1.0,
                                               draw.data_ptr(), ch, hip.stream_ptr()), "me_yolo_loss_bwd_f32")
            draws[idx] = draw
        t2 = time.perf_counter()
        grads = trainer.backward(st, draws, None)
        t3 = time.perf_counter()
    if timing is not None:
        timing.append((t1 - t0, t2 - t1, t3 - t2))
    return grads


for _ in range(4):
    step()
torch.cuda.synchronize()
tm = []
t0 = time.perf_counter()
for _ in range(10):
    step(tm)
host = (time.perf_counter() - t0) / 10
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / 10
print(f"{dtype} batch {batch}: host issue time {host * 1e3:.2f} ms / step (forward {sum(t[0] for t in tm) * 100:.2f}, loss {sum(t[1] for t in tm) * 100:.2f}, "
      f"backward {sum(t[2] for t in tm) * 100:.2f}); wall incl. the GPU {wall * 1e3:.2f} ms")
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    step()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
