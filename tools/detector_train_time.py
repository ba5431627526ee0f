"""Time Darknet-53 forward+backward (tools): python tools/detector_train_time.py [batch] [size]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g; g.build()
from millieye_amd import cfgs, synth
from millieye_amd.yolov3.models import Darknet
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 8
size = int(sys.argv[2]) if len(sys.argv) > 2 else 416
model = Darknet(cfgs.write_cfg("yolov3", "/tmp/dtt_cfg")).eval()
synth.fill_darknet_(model, "dtt")
model = model.cuda()
x = torch.from_numpy(synth.uniform("dtt/x", (batch, 3, size, size))).cuda()
tg = torch.tensor([[i, 3 + i, 0.3 + 0.05 * i, 0.4, 0.2, 0.3] for i in range(batch)], dtype=torch.float32)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    loss, _, _ = model(x, tg)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    loss.backward()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    model.zero_grad()
    print(f"iter {it}: forward {1e3*(t1-t0):.1f} ms, backward {1e3*(t2-t1):.1f} ms, loss {float(loss.detach()):.3f}, "
          f"peak mem {torch.cuda.max_memory_allocated()/2**30:.2f} GiB")
