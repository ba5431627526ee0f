"""me_linear_f32 on the five layer shapes of a batch-8 stage-2 step (k = 1600 proposals).
usage: [MILLIEYE_M2_LINEAR_NAIVE=1] python tools/linear_bench.py [k]"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from millieye_amd import hip  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 1600
lib = hip.lib()
for name, rows, fin, fout, act in (("net0", k, 490, 256, 1), ("net1", k, 256, 4, 0), ("net2", k, 256, 13, 2), ("fc1", 13 * k, 2, 32, 1),
                                   ("fc2", k, 416, 2, 1)):
    x, w, b = torch.randn(rows, fin, device="cuda"), torch.randn(fout, fin, device="cuda"), torch.randn(fout, device="cuda")
    y = torch.empty(rows, fout, device="cuda")
    run = lambda: hip.check(lib.me_linear_f32(x.data_ptr(), fin, rows, fin, w.data_ptr(), b.data_ptr(), fout, act, y.data_ptr(), fout,
                                              hip.stream_ptr()), "me_linear_f32")
    for _ in range(5):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print(f"{name}: [{rows} x {fin}] -> {fout}: {us:.1f} us  ({2.0 * rows * fin * fout / us / 1e6:.2f} TFLOP/s)  naive={os.environ.get('MILLIEYE_M2_LINEAR_NAIVE', '0')}")
