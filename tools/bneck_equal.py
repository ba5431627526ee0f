import torch, sys
sys.path.insert(0, "/root/repo")
from millieye_amd import hip
torch.manual_seed(0)
for half in (torch.bfloat16, torch.float16):
    for (n, h, cin, cmid, cout, tile) in ((4, 52, 256, 128, 256, 1), (2, 104, 128, 64, 128, 3), (2, 104, 128, 64, 128, 4)):
        x = torch.randn((n, h, h, cin), device="cuda").to(half)
        w1 = (torch.randn((cmid, 1, 1, cin), device="cuda") / cin ** 0.5).to(half)
        w2 = (torch.randn((cout, 3, 3, cmid), device="cuda") / (9 * cmid) ** 0.5).to(half)
        s1, t1 = torch.rand(cmid, device="cuda") + 0.5, torch.randn(cmid, device="cuda")
        s2, t2 = torch.rand(cout, device="cuda") + 0.5, torch.randn(cout, device="cuda")
        y = hip.bneck_h16(x, w1, s1, t1, w2, s2, t2, residual=x, tile=tile)
        for t1x in (1, 12, 50):
            for t3 in (1, 14, 221, 131):
                try:
                    mid = hip.conv2d_h16(x, w1, s1, t1, 1, 1, 0, 1, tile=t1x, split_k=1)
                    two = hip.conv2d_h16(mid, w2, s2, t2, 3, 1, 1, 1, residual=x, tile=t3, split_k=1)
                except hip.MeError as e:
                    continue
                print(half, h, "tile", tile, "vs 1x1 tile", t1x, "3x3 tile", t3, "differ:", int((y != two).sum()), "of", y.numel())
