"""Stage-2 training steps (module2 Network.forward(images, targets) + backward + Adam) for profiling: which launches of a step are
still torch's (``at::native``) and the step's time with / without the one-batch look-ahead.
usage: python tools/m2_train_step.py [steps] [batch] [bf16|f32] [prefetch 0|1]   (GPU box; under rocprofv3 --kernel-trace --stats)"""
import os
import random
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from millieye_amd import synth  # noqa: E402
from millieye_amd.module2.my_models import Network, define_yolo  # noqa: E402
from tests import parity_helpers as ph  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    dtype = sys.argv[3] if len(sys.argv) > 3 else "bf16"
    prefetch = (sys.argv[4] if len(sys.argv) > 4 else "1") != "0"
    net = Network(define_yolo(ph.cfg_path("yolov3")), 0.2)
    synth.fill_network_(net, "m2step")
    net = net.cuda().train()
    net.base_detector.eval()
    net.base_detector.compute_dtype = dtype
    net.dropout_generator = os.environ.get("M2_DROPOUT", "philox")
    params = [p for k, p in net.named_parameters() if not k.startswith("base_detector.")]
    from millieye_amd.optim import AdamW   # the stage-2 loop's default optimizer (module2/train.py)
    opt = AdamW(params, lr=1e-4)
    frames = [torch.from_numpy(synth.uniform(f"m2step/x{i}", (n, 3, 416, 416))).cuda() for i in range(4)]
    targets = torch.tensor([[i % n, (3 * i) % 12, 0.3 + 0.04 * (i % 8), 0.4 + 0.03 * (i % 5), 0.3, 0.4] for i in range(2 * n)], dtype=torch.float32)
    random.seed(0)
    torch.manual_seed(0)

    def step(i):
        if prefetch:
            net.queue_detector_prefetch(frames[(i + 1) % len(frames)])
        out, loss, metric = net(frames[i % len(frames)], targets.clone())
        loss.backward()
        opt.step()
        opt.zero_grad()
        return loss

    for i in range(5):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(5, 5 + steps):
        loss = step(i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"stage-2 training step, Darknet-53 {dtype} detector (frozen), batch {n}, look-ahead {'on' if prefetch else 'off'}: "
          f"{dt * 1e3:.3f} ms / step = {n / dt:.1f} frames/s (loss {float(loss.detach()):.4f})")


if __name__ == "__main__":
    main()
