"""Soak of the detector training step (stream-overlapped backward, device loss, whole-network pack): 30 SGD steps of
Darknet-53 at 416^2 / batch 8, run twice from the same state - every tensor of the final state_dict must have the same bits.
usage: python tools/soak_train.py [f32|bf16|f16]   (GPU box; prints the two last losses and the number of differing tensors)"""
import sys, torch, numpy as np
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from tests import parity_helpers as ph
from millieye_amd import synth
def run():
    torch.manual_seed(0)
    model = ph.make_darknet("yolov3", tag="soak").cuda().eval()
    model.compute_dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"   # 16-bit: the mixed-precision step (detector_train16.py)
    opt = torch.optim.SGD(model.parameters(), lr=1e-6)
    n, s = 8, 416
    rng = np.random.RandomState(3)
    for it in range(30):
        x = torch.from_numpy(synth.uniform(f"soak/x{it % 3}", (n, 3, s, s))).cuda()
        tg = np.zeros((16, 6), np.float32)
        tg[:, 0] = rng.randint(0, n, 16); tg[:, 1] = rng.randint(0, 80, 16)
        tg[:, 2:4] = rng.uniform(0.1, 0.9, (16, 2)); tg[:, 4:6] = rng.uniform(0.05, 0.4, (16, 2))
        loss, _, _ = model(x, torch.from_numpy(tg))
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
    torch.cuda.synchronize()
    return float(loss), {k: v.clone() for k, v in model.state_dict().items()}
l1, a = run(); l2, b = run()
bad = [k for k in a if not torch.equal(a[k], b[k])]
print("loss", l1, l2, "differing tensors:", len(bad), bad[:3])
