"""Achievable HBM write / copy bandwidth on this GPU with plain torch kernels (context for the stem kernels, which write
10x what they read): python tools/scratch/write_bw.py"""
import torch
dev = "cuda"
for mb in (256, 775, 2048):
    n = mb * 2 ** 20 // 4
    y = torch.empty(n, device=dev)
    x = torch.randn(n, device=dev)
    for name, fn, nbytes in (("fill", lambda: y.fill_(1.0), n * 4), ("copy", lambda: y.copy_(x), n * 8),
                             ("add", lambda: torch.add(x, 1.0, out=y), n * 8)):
        for _ in range(3):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            fn()
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 20
        print(f"{mb:5d} MB {name:5s} {ms * 1e3:8.1f} us  {nbytes / ms / 1e9:7.2f} TB/s", flush=True)
