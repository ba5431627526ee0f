import cProfile, os, pstats, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from millieye_amd import cfgs, synth, parallel as par
from millieye_amd.yolov3.models import Darknet
batch = 8
model = Darknet(cfgs.write_cfg("yolov3", "/tmp/dtp_cfg")).eval()
synth.fill_darknet_(model, "bench/yolov3"); synth.trained_like_(model, "bench/yolov3/trained")
model = model.cuda()
model.compute_dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"   # f32 | bf16 | f16 (the mixed-precision step)
x = torch.from_numpy(synth.uniform("bench/frames/0", (batch, 3, 416, 416))).cuda()
tg = torch.tensor([[i, (3 * i) % 80, 0.3 + 0.04 * (i % 8), 0.4 + 0.03 * (i % 5), 0.2, 0.3] for i in range(batch)], dtype=torch.float32)
params = [p for p in model.parameters()]
opt = torch.optim.SGD(params, lr=1e-5)
def step():
    loss, _fm, yo = model(x, tg)
    loss.backward()
    opt.step(); opt.zero_grad(set_to_none=True)
for _ in range(5): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): step()
torch.cuda.synchronize(); print("ms/step", (time.perf_counter() - t0) / 10 * 1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(10): step()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
# host-only time of a step: the same loop without waiting for the GPU between steps is what the profile above shows as
# wall time per call; a step whose host time exceeds its GPU time is host-bound
