"""YOLO decode + candidate lists (me_yolo_decode_cand_f32 per scale vs me_yolo_decode_cand_multi_f32, one launch) at the three
416^2 scales: python tools/decode_bench.py [batch]   - prints us per forward's decodes and checks the rows / NMS results agree."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g; g.build()
from millieye_amd import hip
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
nc, na, img = 80, 3, 416
anch = [[(116, 90), (156, 198), (373, 326)], [(30, 61), (62, 45), (59, 119)], [(10, 13), (16, 30), (33, 23)]]
gs = (13, 26, 52)
rows = sum(na * q * q for q in gs)
torch.manual_seed(0)
raws = [(torch.randn(n, q, q, 256, device="cuda") * 2 - 3).contiguous() for q in gs]  # pitch 256 >= 255 channels
descs, off = [], 0
for q, raw, an in zip(gs, raws, anch):
    d = hip.YoloDesc()
    d.x, d.x_pitch = raw.data_ptr(), 256
    d.n, d.g, d.num_anchors, d.num_classes = n, q, na, nc
    d.rows_total, d.row_offset, d.stride = rows, off, img / q
    for k, (aw, ah) in enumerate(an):
        d.anchors[2 * k], d.anchors[2 * k + 1] = aw / (img / q), ah / (img / q)
    off += na * q * q
    descs.append(d)
ptrs = (C.c_void_p * 3)(*[C.addressof(d) for d in descs])
lib, st = hip.lib(), hip.stream_ptr()
ws, _keep = hip.nms_workspace(n, rows, torch.device("cuda"))


def run(multi, out):
    for d in descs:
        d.out = out.data_ptr()
    if multi:
        hip.check(lib.me_yolo_decode_cand_multi_f32(ptrs, 3, 0.2, ws, 1, st), "multi")
    else:
        for i, d in enumerate(descs):
            hip.check(lib.me_yolo_decode_cand_f32(C.byref(d), 0.2, ws, int(i == 0), st), "single")


res = {}
for multi in (0, 1):
    out = torch.zeros(n, rows, 5 + nc, device="cuda")
    run(multi, out)
    det, cnt = hip.nms_batched(out, 0.2, 0.4, 200, writeback_xyxy=False, prepped=True)
    res[multi] = (out.clone(), cnt.clone(), torch.stack([det[i, :int(cnt[i])].sum() for i in range(n)]))
    for _ in range(20):
        run(multi, out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        run(multi, out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 5
    print(f"batch {n:3d} {'one launch ' if multi else 'three launches'}: {us:7.1f} us per forward "
          f"({2 * n * rows * (5 + nc) * 4 / us / 1e3:.0f} GB/s read + written)")
print("rows equal:", bool(torch.equal(res[0][0], res[1][0])), " counts equal:", bool(torch.equal(res[0][1], res[1][1])),
      " kept rows equal:", bool(torch.equal(res[0][2], res[1][2])), " candidates kept:", res[0][1].tolist()[:4])
