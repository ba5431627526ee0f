"""-m gpu: NMS parity (bit-exact: index work) of me_nms_batched_f32 / me_nms_boxes_f32 against the
oracle's C restatement of torchvision batched_nms (oracle/tv_ops.c - parity-unpinned boundary)."""
import numpy as np
import pytest
import torch

from millieye_amd import synth

pytestmark = pytest.mark.gpu


def _oracle_nms_cpp(pred, conf_thresh, nms_thresh=0.5, max_det=200):
    """Restatement of non_max_suppression_cpp (reference utils/utils.py:337-378) on CPU."""
    from oracle import tv_ops
    pred = pred.clone()
    xy = pred[..., :4].clone()
    pred[..., 0] = xy[..., 0] - xy[..., 2] / 2
    pred[..., 1] = xy[..., 1] - xy[..., 3] / 2
    pred[..., 2] = xy[..., 0] + xy[..., 2] / 2
    pred[..., 3] = xy[..., 1] + xy[..., 3] / 2
    out = []
    for img in pred:
        img = img[img[:, 4] >= conf_thresh]
        if not img.size(0):
            out.append(None)
            continue
        cc, cp = img[:, 5:].max(1, keepdim=True)
        det = torch.cat((img[:, :5], cc.float(), cp.float(), img[:, 5:]), 1)
        keep = tv_ops.batched_nms(det[:, :4], det[:, 4], det[:, 6], nms_thresh)[:max_det]
        out.append(det[keep] if len(keep) else None)
    return out, pred


def _pred(tag, n, rows, nc, frac_pass=0.3, size=416.0):
    cxcy = synth.uniform(tag + "c", (n, rows, 2), 0, size)
    wh = synth.uniform(tag + "s", (n, rows, 2), 8, 160)
    conf = synth.uniform(tag + "o", (n, rows, 1), 0, 1.0)
    conf = np.where(conf < frac_pass, conf / frac_pass, conf * 0.0099)  # ~frac_pass of rows above 0.01
    cls = synth.uniform(tag + "k", (n, rows, nc), 0, 1)
    return torch.from_numpy(np.concatenate([cxcy, wh, conf, cls], -1).astype(np.float32))


@pytest.mark.parametrize("n,rows,nc,thr", [(2, 300, 12, 0.2), (3, 2535, 12, 0.01), (1, 10647, 80, 0.5),
                                            (2, 64, 1, 0.0), (1, 5000, 3, 0.05), (1, 20000, 2, 0.0)])  # 1 / 4 / 16 waves per image
def test_nms_matches_oracle_bitexact(hip_lib, n, rows, nc, thr):
    from millieye_amd import hip
    pred = _pred(f"nms{n}{rows}", n, rows, nc)
    ref, ref_pred = _oracle_nms_cpp(pred, thr)
    dev = pred.cuda()
    det, cnt = hip.nms_batched(dev, thr, 0.5, 200, writeback_xyxy=True)
    torch.cuda.synchronize()
    assert torch.equal(dev.cpu(), ref_pred), "in-place xywh->xyxy writeback differs"
    cnt = cnt.cpu().tolist()
    for i in range(n):
        if ref[i] is None:
            assert cnt[i] == 0
        else:
            assert cnt[i] == ref[i].shape[0], f"image {i}: kept {cnt[i]} vs oracle {ref[i].shape[0]}"
            assert torch.equal(det[i, :cnt[i]].cpu(), ref[i]), f"image {i}: kept rows differ"


def test_nms_capped_matrix_falls_back_when_it_runs_out(hip_lib):
    """The multi-workgroup path looks at the 1024 best-scored candidates of an image; when those do not yield max_det
    winners and more candidates exist (here: 3000 candidates in 60 tight clusters - every cluster keeps one box, so the walk
    has to visit all of them), the image is redone by the single-workgroup kernel.  Both images of the batch must equal the
    oracle bit for bit: image 0 takes the fallback, image 1 (300 well separated boxes) does not."""
    from millieye_amd import hip
    n, rows, nc = 2, 3200, 4
    g = np.random.RandomState(5)
    pred = np.zeros((n, rows, 5 + nc), dtype=np.float32)
    centres = g.uniform(40, 380, size=(60, 2)).astype(np.float32)
    which = g.randint(0, 60, size=rows)
    pred[0, :, 0:2] = centres[which] + g.uniform(-1.5, 1.5, size=(rows, 2)).astype(np.float32)
    pred[0, :, 2:4] = 60.0
    pred[0, :, 4] = g.uniform(0.05, 1.0, size=rows)
    pred[0, :, 5] = 0.9
    pred[1, :, 0:2] = g.uniform(0, 416, size=(rows, 2))
    pred[1, :, 2:4] = g.uniform(4, 12, size=(rows, 2))
    pred[1, :, 4] = np.where(np.arange(rows) < 300, g.uniform(0.5, 1.0, size=rows), 0.01)
    pred[1, :, 5:] = g.uniform(0, 1, size=(rows, nc))
    pred = torch.from_numpy(pred)
    ref, _ = _oracle_nms_cpp(pred, 0.1)
    det, cnt = hip.nms_batched(pred.cuda(), 0.1, 0.5, 200, writeback_xyxy=False)
    cnt = cnt.cpu().tolist()
    assert ref[0].shape[0] < 200 and int((pred[0, :, 4] >= 0.1).sum()) > 1024, "image 0 must exhaust the capped list"
    for i in range(n):
        assert cnt[i] == ref[i].shape[0], f"image {i}: kept {cnt[i]} vs oracle {ref[i].shape[0]}"
        assert torch.equal(det[i, :cnt[i]].cpu(), ref[i]), f"image {i}: kept rows differ"


def test_nms_properties_and_edges(hip_lib):
    from millieye_amd import hip
    from millieye_amd.utils.utils import non_max_suppression_cpp, box_ops
    # nothing passes
    pred = _pred("none", 2, 100, 4)
    out = non_max_suppression_cpp(pred.clone().cuda(), conf_thresh=2.0)
    assert out == [None, None]
    # ties: identical scores -> lower row index wins; identical boxes suppress each other
    p = torch.zeros((1, 4, 6))
    p[0, :, :4] = torch.tensor([100.0, 100.0, 50.0, 50.0])
    p[0, :, 4] = 0.9
    p[0, :, 5] = 1.0
    out = non_max_suppression_cpp(p.cuda(), conf_thresh=0.5)
    assert out[0].shape[0] == 1
    # CPU tensors are staged through the GPU and come back on the CPU (reference passes .cpu())
    pred = _pred("cpu", 1, 400, 12)
    out = non_max_suppression_cpp(pred, conf_thresh=0.2)
    assert out[0] is not None and out[0].device.type == "cpu" and out[0].shape[1] == 7 + 12
    sc = out[0][:, 4]
    assert torch.all(sc[:-1] >= sc[1:]), "kept rows must be sorted by objectness"
    assert out[0].shape[0] <= 200
    # box_ops re-export: plain nms == oracle nms
    from oracle import tv_ops
    b = pred[0, :, :4].clone()  # already xyxy (in-place conversion above)
    keep = box_ops.nms(b.cuda(), pred[0, :, 4].cuda(), 0.3).cpu()
    assert torch.equal(keep, tv_ops.nms(b, pred[0, :, 4], 0.3))
    lab = (pred[0, :, 5] * 3).floor()
    keep = box_ops.batched_nms(b.cuda(), pred[0, :, 4].cuda(), lab.cuda(), 0.3).cpu()
    assert torch.equal(keep, tv_ops.batched_nms(b, pred[0, :, 4], lab, 0.3))
    assert box_ops.batched_nms(torch.zeros((0, 4)).cuda(), torch.zeros(0).cuda(), torch.zeros(0).cuda(), .5).numel() == 0


@pytest.mark.parametrize("n", [1, 6, 24])  # 1 / 4 / 16 rows per wave
def test_decode_with_candidate_lists_one_launch_for_three_scales(hip_lib, n):
    """me_yolo_decode_cand_multi_f32 (Network.forward's detector run: the three [yolo] scales in one launch that also fills the
    NMS candidate lists) against the plain decode (me_yolo_decode_f32 per scale) + me_nms_batched_f32, and against
    me_yolo_decode_cand_f32 per scale: the same prediction rows and the same kept detections, bit for bit."""
    import ctypes as C
    from millieye_amd import hip
    nc, na, img = 12, 3, 416
    anchors = [[(116, 90), (156, 198), (373, 326)], [(30, 61), (62, 45), (59, 119)], [(10, 13), (16, 30), (33, 23)]]
    grids = (13, 26, 52)
    rows = sum(na * g * g for g in grids)
    pitch = 64  # >= 3 * 17 channels
    raws = [torch.from_numpy(synth.normal(f"decode3/{n}/{g}", (n, g, g, pitch)) * 2.0 - 1.5).cuda().contiguous() for g in grids]
    raws[0][0, 0, 0, 4] = float("nan")  # a NaN objectness (never a candidate) and a NaN class score (wins the class max)
    raws[1][0, 1, 1, 5 + 3] = float("nan")
    raws[1][0, 1, 1, 4] = 5.0
    plain = torch.empty((n, rows, 5 + nc), device="cuda")
    off = 0
    descs = []
    for g, raw, an in zip(grids, raws, anchors):
        d = hip.YoloDesc()
        d.x, d.x_pitch = raw.data_ptr(), pitch
        d.n, d.g, d.num_anchors, d.num_classes = n, g, na, nc
        d.rows_total, d.row_offset, d.stride = rows, off, img / g
        for k, (aw, ah) in enumerate(an):
            d.anchors[2 * k], d.anchors[2 * k + 1] = aw / (img / g), ah / (img / g)
        d.out = plain.data_ptr()
        hip.check(hip.lib().me_yolo_decode_f32(C.byref(d), hip.stream_ptr()), "me_yolo_decode_f32")
        descs.append(d)
        off += na * g * g
    det0, cnt0 = hip.nms_batched(plain.clone(), 0.3, 0.4, 200, writeback_xyxy=False)
    ws, _keep = hip.nms_workspace(n, rows, torch.device("cuda"))
    results = []

    def same_bits(a, b):  # NaNs in the same places (any payload), everything else bit for bit
        an, bn = torch.isnan(a), torch.isnan(b)
        return bool(torch.equal(an, bn)) and bool(torch.equal(a[~an].view(torch.int32), b[~bn].view(torch.int32)))

    for multi in (True, False):
        out = torch.zeros_like(plain)
        for d in descs:
            d.out = out.data_ptr()
        if multi:
            ptrs = (C.c_void_p * 3)(*[C.addressof(d) for d in descs])
            hip.check(hip.lib().me_yolo_decode_cand_multi_f32(ptrs, 3, 0.3, ws, 1, hip.stream_ptr()), "multi")
        else:
            for i, d in enumerate(descs):
                hip.check(hip.lib().me_yolo_decode_cand_f32(C.byref(d), 0.3, ws, int(i == 0), hip.stream_ptr()), "single")
        det, cnt = hip.nms_batched(out, 0.3, 0.4, 200, writeback_xyxy=False, prepped=True)
        torch.cuda.synchronize()
        results.append((out, det, cnt))
    for out, det, cnt in results:
        assert same_bits(out, plain), "decoded rows differ from me_yolo_decode_f32"
        assert torch.equal(cnt, cnt0)
        for i in range(n):
            k = int(cnt0[i])
            assert k > 0
            assert same_bits(det[i, :k], det0[i, :k]), f"image {i}: kept rows differ"
    # refusals
    ptrs = (C.c_void_p * 3)(*[C.addressof(d) for d in descs])
    assert hip.lib().me_yolo_decode_cand_multi_f32(ptrs, 4, 0.3, ws, 1, hip.stream_ptr()) != 0
    descs[1].n = n + 1
    assert hip.lib().me_yolo_decode_cand_multi_f32(ptrs, 3, 0.3, ws, 1, hip.stream_ptr()) != 0
