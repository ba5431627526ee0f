"""The mixed-precision detector training step at the benchmark's shape (Darknet-53 416x416, batch 8) against the fp32 HIP step:
gradient cosines of the first step (all 222 tensors: minimum, the five lowest) and the loss trajectories of 25 SGD steps.
usage (GPU box): python tools/train16_check.py [bf16|f16] [batch] [size]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from millieye_amd import synth  # noqa: E402
from tests import parity_helpers as ph  # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8
size = int(sys.argv[3]) if len(sys.argv) > 3 else 416


def cos(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a @ b) / (a.norm() * b.norm() + 1e-300))


def make():
    m = ph.make_darknet("yolov3", tag="bench/yolov3", trained_like=True).cuda().eval()
    return m


x = torch.from_numpy(synth.uniform("bench/frames/0", (batch, 3, size, size))).cuda()
targets = torch.tensor([[i, (3 * i) % 80, 0.3 + 0.04 * (i % 8), 0.4 + 0.03 * (i % 5), 0.2, 0.3] for i in range(batch)], dtype=torch.float32)
ref, mix = make(), make()
mix.compute_dtype = dtype
l32, _, _ = ref(x, targets)
l32.backward()
l16, _, _ = mix(x, targets)
l16.backward()
rows = []
for (k, p16), (_k, p32) in zip(mix.named_parameters(), ref.named_parameters()):
    if float(p32.grad.norm()) > 1e-12:
        rows.append((cos(p16.grad, p32.grad), float(p16.grad.norm() / p32.grad.norm()), k))
rows.sort()
print(f"{dtype} batch {batch} {size}x{size}: loss {float(l16):.5f} vs fp32 {float(l32):.5f} ({abs(float(l16) - float(l32)) / float(l32):.2e}); "
      f"{len(rows)} gradients, cosine min {rows[0][0]:.5f}, norm ratio {min(r[1] for r in rows):.4f} .. {max(r[1] for r in rows):.4f}")
for c, nr, k in rows[:5]:
    print(f"   {c:.5f}  norm ratio {nr:.4f}  {k}")
opt32 = torch.optim.SGD(ref.parameters(), lr=1e-5)
opt16 = torch.optim.SGD(mix.parameters(), lr=1e-5)
opt32.step(); opt16.step(); opt32.zero_grad(); opt16.zero_grad()
for step in range(1, 26):
    a, _, _ = ref(x, targets)
    a.backward()
    opt32.step()
    opt32.zero_grad()
    b, _, _ = mix(x, targets)
    b.backward()
    opt16.step()
    opt16.zero_grad()
    if step % 5 == 0 or step == 1:
        print(f"   step {step:2d}: loss fp32 {float(a):.5f}  {dtype} {float(b):.5f}")
