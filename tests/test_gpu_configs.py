"""-m gpu: the BASELINE.json configurations at their named shapes (round-3 additions; VERDICT r02 "next round" item 1).

* configs[2]: the stage-2 network (module2_mixed) on Darknet-53, 416x416, **batch 32**, fp32 / bf16 / f16.
* configs[4] (per-GPU shape): the full ``Network.forward`` at **608x608, batch 16, IEEE half**, and NMS at that shape
  (16 images x 22 743 rows x 80 classes), bit-exact against the oracle.

Frames are independent units of the path (BASELINE north_star: "a batch of frames shards naturally"), which is the
size-independent property these tests lean on: the rows of frame f in a batch run are the rows of the batch-1 run of
frame f - exactly in fp32 up to accumulation-order rounding (1e-3, north_star), and up to the storage error in the 16-bit
modes (tile / split-K choices follow M, so a few roundings differ).  The fp32 runs are pinned to the CPU oracle on sampled
frames (the oracle needs seconds per Darknet-53 frame)."""
import os

import pytest
import torch

from millieye_amd import cfgs, synth
from tests import parity_helpers as ph

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _frame_rows(rows, f):
    r = rows[rows[:, 0] == f].clone()
    r[:, 0] = 0
    return r


def _m2_net(tag):
    from millieye_amd.module2.my_models import Network, define_yolo
    net = Network(define_yolo(ph.cfg_path("yolov3")), 0.2).eval()
    # objectness statistics tuned (on the CPU oracle) so that tens of proposals per frame, of several classes, reach the
    # heads with random backbone weights; a small box-regression layer keeps the refined boxes box-sized (random net1
    # weights give exp(tw) factors of 1e2, which would turn the pixel tolerances below into 1e-6 relative ones)
    synth.fill_network_(net, tag, obj_bias=-3.4, obj_std=3.0, cls0_bias=-2.0)
    with torch.no_grad():
        net.refinement_head.net1[0].weight.mul_(0.002)
        net.refinement_head.net1[0].bias.mul_(0.002)
    return net


def test_module2_batch32_fp32_vs_oracle(hip_lib):
    """configs[2] shape in the reference's arithmetic: yolov3.cfg, 416x416, batch 32 through the stage-2 network; the rows
    of three sampled frames against the CPU oracle run on those frames alone (tap = module 91, the documented Darknet-53
    extension), 1e-3, ties in the sort key matched as sets."""
    from oracle import network_m2_ref
    from tests.test_gpu_network import _cmp_rows_ties
    name, n, s = "m2b32", 32, 416
    net = _m2_net(name)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    x = torch.from_numpy(synth.uniform(name + "/x", (n, 3, s, s)))
    net = net.to(net.device)
    with torch.no_grad():
        out = net(x.cuda())
    assert out.device.type == "cpu" and out.shape[1] == 8 and out.shape[0] >= 32, out.shape
    text = cfgs.KNOWN["yolov3"]()
    total = 0
    for f in (0, 19, 31):
        ref = network_m2_ref.network_m2_forward(text, sd, x[f:f + 1], conf_thresh=0.2, tap_module=91)
        _cmp_rows_ties(_frame_rows(out, f), ref, f"module-2 batch-32 run, frame {f}")
        total += ref.shape[0]
    assert total >= 6, "the sampled frames must carry detections"


PLAN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "plan_16bit_configs.json")


def _pin_plan(monkeypatch, tmp_path):
    """The tuned (tile, split_k) table of these configurations as tools/dump_plan.py measured it (tests/golden/
    plan_16bit_configs.json) instead of whatever this box's autotuner would pick: the tests run the tiles the benchmark
    runs, and the same arithmetic on every GPU box.  Returns a function that asserts nothing had to be measured."""
    import shutil
    from millieye_amd import engine
    local = tmp_path / "plan.json"
    shutil.copy(PLAN, local)
    monkeypatch.setenv("MILLIEYE_TUNE_CACHE", str(local))
    monkeypatch.setenv("MILLIEYE_AUTOTUNE", "1")
    monkeypatch.setenv("MILLIEYE_BNECK", "0")   # (the one-launch bottleneck is a measured choice too: the pinned plan keeps the launch pairs)
    saved = dict(engine._TUNE_CACHE), engine._TUNE_FILE_LOADED[0]
    engine._TUNE_CACHE.clear()
    engine._TUNE_FILE_LOADED[0] = False
    before = engine._TUNE_STATS["measured"]

    def check():
        measured = engine._TUNE_STATS["measured"] - before
        engine._TUNE_CACHE.clear()
        engine._TUNE_CACHE.update(saved[0])
        engine._TUNE_FILE_LOADED[0] = saved[1]
        assert measured == 0, (f"{measured} layer shapes were not in {PLAN}: regenerate it with tools/dump_plan.py "
                               f"(the kernel generation or the planned shapes changed)")
    return check


def _match_share(ref, got, px, tol):
    """Share of ``ref`` rows with a ``got`` row of the same image and class within ``px`` pixels on every corner and
    ``tol`` on the refined confidence."""
    if ref.shape[0] == 0:
        return 1.0
    matched = 0
    for row in ref:
        cand = got[(got[:, 0] == row[0]) & (got[:, 7] == row[7])]
        if len(cand) == 0:
            continue
        d = (cand[:, 1:5] - row[1:5]).abs().max(dim=1).values
        j = int(d.argmin())
        matched += int(float(d[j]) <= px and abs(float(cand[j, 5] - row[5])) <= tol)
    return matched / ref.shape[0]


def _iou(a, b):
    """IoU of one xyxy box with many."""
    x1, y1 = torch.maximum(a[0], b[:, 0]), torch.maximum(a[1], b[:, 1])
    x2, y2 = torch.minimum(a[2], b[:, 2]), torch.minimum(a[3], b[:, 3])
    inter = (x2 - x1).clamp(min=0) * (y2 - y1).clamp(min=0)
    return inter / ((a[2] - a[0]) * (a[3] - a[1]) + (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]) - inter)


def _tie_share(ref, got, px=4.0, tol=0.1, iou_min=0.45):
    """Share of ``ref`` rows with a counterpart in ``got`` when the two tie-breaks of a random-weight detector are allowed for
    (round 5, tools/bf16_row_diff.py / profiles/r05_bf16_row_diff.txt: every batch-1 row the strict criterion misses in the
    batch-32 run is one of these - none vanishes at the confidence threshold, none moves by rounding): a counterpart is a row
    within ``px`` pixels and ``tol`` confidence of ANY class (the class arg-max among near-equal class scores flipped), or a row
    that overlaps it by IoU >= ``iou_min`` (NMS kept the neighbouring grid cell's near-equal candidate: the row "moves" by one
    cell = 16 / 32 px)."""
    if ref.shape[0] == 0 or got.shape[0] == 0:
        return 1.0 if ref.shape[0] == 0 else 0.0
    n = 0
    for row in ref:
        d = (got[:, 1:5] - row[1:5]).abs().max(dim=1).values
        near = (d <= px) & ((got[:, 5] - row[5]).abs() <= tol)
        n += int(bool(near.any()) or bool((_iou(row[1:5], got[:, 1:5]) >= iou_min).any()))
    return n / ref.shape[0]


@pytest.mark.parametrize("dtype,px,tol,bar,floor", [("bf16", 4.0, 0.1, 0.85, 0.70), ("f16", 2.0, 0.02, 0.97, 0.95)])
def test_module2_batch32_16bit(hip_lib, monkeypatch, tmp_path, dtype, px, tol, bar, floor):
    """(The per-layer tile choice is normally MEASURED on the GPU box, so the roundings - and with them a handful of rows near the
    confidence threshold - moved from box to box and this test's shares with them; here the plan is PINNED to the tuned table
    the benchmark configuration runs, ``_pin_plan``: the tiles under test are the ones bench.py times, and the arithmetic is
    the same on every box.)
    configs[2] literally ("module2 ... 416x416 bf16 inference, batch=32"), and the IEEE-half mode: the batch-32 run in a
    16-bit storage mode is deterministic, its frames agree with the batch-1 runs of the same frames in the same mode
    (``bar`` of the rows of the four sampled frames within the storage error, no frame below ``floor``.  Measured under the
    pinned plan, round 4: bf16 73.7 / 100 / 95 / 88 % per frame, 89.3 % overall; f16 100 % on every frame - the bars sit a
    few rows under that; the tile choice is measured per GPU box, so the share moves by a few rows from box to box: the tile choice
    follows M, so accumulation order - hence a few roundings of the 8-bit mantissa, amplified by 75 random-weight layers and a
    confidence threshold - differs; measured 89 - 100 % per frame in bf16), and it is as close to the fp32 batch-32 run as the
    batch-1 runs are (share of fp32 rows with a counterpart, -10 points)."""
    plan_check = _pin_plan(monkeypatch, tmp_path)
    name, n, s = "m2b32", 32, 416
    net = _m2_net(name)
    net = net.to(net.device)
    x = torch.from_numpy(synth.uniform(name + "/x", (n, 3, s, s))).cuda()
    picks = (0, 11, 19, 31)
    with torch.no_grad():
        ref32 = net(x)
        net.base_detector.compute_dtype = dtype
        got, again = net(x), net(x)
        ones = [net(x[f:f + 1]) for f in picks]
        # At the source - the decoded rows of the detector, before NMS - the batch-32 and batch-1 runs of a frame are close
        # (round 5, profiles/r05_bf16_row_diff.txt: objectness |d| <= 0.007, boxes of the rows over the threshold <= 0.5 px in
        # bf16): whatever differs in the output rows is decided by NMS / arg-max tie-breaks downstream, not by rounding
        _fm, y_all = net.base_detector(x)
        for f in picks:
            _fm, y_one = net.base_detector(x[f:f + 1])
            a, b = y_all[f].float(), y_one[0].float()
            over = (a[:, 4] >= 0.2) | (b[:, 4] >= 0.2)
            d_obj = float((a[:, 4] - b[:, 4]).abs().max())
            d_box = float((a[over, :4] - b[over, :4]).abs().max()) if bool(over.any()) else 0.0
            print(f"[m2b32 {dtype}] frame {f}: decoded rows batch-32 vs batch-1: objectness |d| {d_obj:.4f}, boxes over the threshold "
                  f"{d_box:.2f} px")
            assert d_obj <= (0.03 if dtype == "bf16" else 0.005) and d_box <= (2.0 if dtype == "bf16" else 0.5), (f, d_obj, d_box)
    assert torch.equal(got, again), "not deterministic"
    assert ref32.shape[0] >= 32 and abs(got.shape[0] - ref32.shape[0]) <= max(2, 0.1 * ref32.shape[0]), (got.shape, ref32.shape)
    found = rows_total = tie_found = 0
    shares = []
    for f, one in zip(picks, ones):
        mine = _frame_rows(got, f)
        assert abs(mine.shape[0] - one.shape[0]) <= max(3, 0.2 * one.shape[0]), (f, mine.shape, one.shape)
        share = _match_share(one, mine, px, tol)
        print(f"[m2b32 {dtype}] frame {f}: {share:.1%} of {one.shape[0]} batch-1 rows found in the batch-32 run")
        # ... and with the two tie-breaks of a random-weight detector allowed for (a neighbouring cell's near-equal candidate kept
        # by NMS, the class arg-max flipped among near-equal scores) every row has its counterpart, both ways: measured
        # 94.7 / 100 / 100 / 100 % (one row of 19 on frame 0), 100 % the other way; a row is 4 - 5 points of a frame
        t_fwd, t_back = _tie_share(one, mine, px, tol), _tie_share(mine, one, px, tol)
        print(f"[m2b32 {dtype}] frame {f}: tie-aware {t_fwd:.1%} / {t_back:.1%}")
        assert min(t_fwd, t_back) >= 0.85, f"{dtype}: frame {f}: tie-aware share {t_fwd:.0%} / {t_back:.0%}"
        tie_found += t_fwd * one.shape[0]
        shares.append((f, share))
        found += share * one.shape[0]
        rows_total += one.shape[0]
        r32 = _frame_rows(ref32, f)
        s_batch, s_one = _match_share(r32, mine, px, tol), _match_share(r32, one, px, tol)
        assert s_batch >= s_one - 0.1, f"{dtype}: frame {f}: batch-32 {s_batch:.0%} vs batch-1 {s_one:.0%} of the fp32 rows"
    print(f"[m2b32 {dtype}] overall {found / max(rows_total, 1):.1%}")
    for f, share in shares:
        assert share >= floor, f"{dtype}: frame {f}: {share:.0%} of the batch-1 rows found in the batch-32 run"
    assert found >= bar * rows_total, f"{dtype}: {found / max(rows_total, 1):.0%} of the batch-1 rows found in the batch-32 run"
    assert tie_found >= 0.93 * rows_total, f"{dtype}: tie-aware {tie_found / max(rows_total, 1):.0%} (measured 98.8 %)"
    plan_check()


def _net608(tag):
    from millieye_amd.my_models import Network, define_yolo
    net = Network(define_yolo(ph.cfg_path("yolov3")), 0.2).eval()
    synth.fill_network_(net, tag, cls0_bias=3.0, cls_bias=-4.0)
    with torch.no_grad():  # box-sized refined boxes (see _m2_net): the pixel tolerances below are absolute
        net.refinement_head.net1[0].weight.mul_(0.002)
        net.refinement_head.net1[0].bias.mul_(0.002)
    return net


def test_full_pipeline_608_batch16_fp32_and_f16(hip_lib, monkeypatch, tmp_path):
    """configs[4] per-GPU shape end to end: Darknet-53 at 608x608, batch 16 (128 frames over 8 GPUs), detector -> NMS over
    22 743 rows per frame -> proposals -> score maps (38x38) -> RoI heads -> ordered rows.  (1) fp32: two frames against the
    CPU oracle (1e-3); (2) IEEE half: deterministic, the frames of the batch-16 run agree with their batch-1 runs in the same
    mode, and the batch run is as close to the fp32 rows as the batch-1 runs are."""
    from oracle import network_ref
    from tests.test_gpu_network import _cmp_rows_ties
    plan_check = _pin_plan(monkeypatch, tmp_path)  # the benchmark's tuned tiles, the same on every GPU box (see the module-2 test)
    name, n, s = "full608", 16, 608
    net = _net608(name)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    x = torch.from_numpy(synth.uniform(name + "/x", (n, 3, s, s)))
    maps, rboxes = synth.radar_inputs(name + "/radar", n, s // 16, boxes_per_image=2)
    maps, rboxes = torch.from_numpy(maps), torch.from_numpy(rboxes)
    net = net.cuda()
    xd, md = x.cuda(), maps.cuda()

    def one_frame(f):
        rb = rboxes[rboxes[:, 0] == f].clone()
        rb[:, 0] = 0
        return net(xd[f:f + 1], md[f:f + 1], rb.cuda(), 0).cpu()

    picks = (0, 9, 15)
    with torch.no_grad():
        out32 = net(xd, md, rboxes.clone().cuda(), 0).cpu()
        net.base_detector.compute_dtype = "f16"
        out16 = net(xd, md, rboxes.clone().cuda(), 0).cpu()
        again = net(xd, md, rboxes.clone().cuda(), 0).cpu()
        ones = [one_frame(f) for f in picks]
    assert out32.shape[0] > 16 and out32.shape[1] == 8
    assert torch.equal(out16, again), "f16 run not deterministic"
    text = cfgs.KNOWN["yolov3"]()
    total = 0
    for f in (0, 15):
        rb = rboxes[rboxes[:, 0] == f].clone()
        rb[:, 0] = 0
        ref = network_ref.network_forward(text, sd, x[f:f + 1], maps[f:f + 1], rb, 0, conf_thresh=0.2, tap_module=91)
        _cmp_rows_ties(_frame_rows(out32, f), ref, f"608x608 fp32 batch-16 run, frame {f}")
        total += ref.shape[0]
    assert total >= 4, "the sampled frames must carry detections"
    found = rows_total = 0
    for f, one in zip(picks, ones):
        mine = _frame_rows(out16, f)
        assert abs(mine.shape[0] - one.shape[0]) <= max(1, 0.1 * one.shape[0]), (f, mine.shape, one.shape)
        # A row of one run has a counterpart (same image, class; corners within 2 px, confidence within 0.03) in the other.
        # Not every row can: the batch-16 and batch-1 plans use different tiles, a few half-precision roundings differ, and
        # where two overlapping candidates score within that noise NMS keeps the other one (measured 83 - 100 % per frame).
        share = _match_share(one, mine, 2.0, 0.03)
        print(f"[full608 f16] frame {f}: {share:.1%} of {one.shape[0]} batch-1 rows found in the batch-16 run")
        assert share >= 0.9, f"frame {f}: {share:.0%} of the f16 batch-1 rows found in the f16 batch-16 run"  # measured 100 / 94.2 / 100 %
        found += share * one.shape[0]
        rows_total += one.shape[0]
        r32 = _frame_rows(out32, f)
        assert _match_share(r32, mine, 2.0, 0.03) >= _match_share(r32, one, 2.0, 0.03) - 0.1, f"frame {f}: vs the fp32 rows"
    assert found >= 0.95 * rows_total, f"{found / max(rows_total, 1):.0%} of the f16 batch-1 rows found in the batch-16 run"
    plan_check()


@pytest.mark.parametrize("dtype,px,tol", [("bf16", 4.0, 0.1), ("f16", 2.0, 0.03)])
def test_full_pipeline_16bit_rows_vs_oracle_restatement(hip_lib, dtype, px, tol):
    """The OUTPUT ROWS of the full pipeline in a 16-bit storage mode against an independent check (the row bars
    above compare HIP 16-bit runs with each other): ``oracle.network_ref.network_forward(storage=...)`` = the detector restated with
    the mode's rounding points, the score-map convolution on the 16-bit tap with 16-bit weights, the fp32 tail.  Darknet-53, 416 px,
    trained-like weights, batch 8; three frames.  HIP and restatement round a few activations the other way (accumulation order), and
    with random weights neighbouring grid cells carry near-equal candidates, so rows are matched tie-aware both ways (a counterpart
    within ``px`` pixels / ``tol`` confidence of any class, or IoU >= 0.45 with the kept neighbour) at >= 0.95 (bf16) / 0.97 (f16) per
    frame (measured 98.0 - 99.0 % / 99.5 - 100 % of 202 rows); the strict share (same class, same tolerances) must reach 0.85 / 0.95
    over the three frames (measured 92.4 % / 99.3 %)."""
    from oracle import network_ref
    name, n, s = "rows16", 8, 416
    net = _net608(name)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    x = torch.from_numpy(synth.uniform(name + "/x", (n, 3, s, s)))
    maps, rboxes = synth.radar_inputs(name + "/radar", n, s // 16, boxes_per_image=2)
    maps, rboxes = torch.from_numpy(maps), torch.from_numpy(rboxes)
    net = net.cuda()
    net.base_detector.compute_dtype = dtype
    with torch.no_grad():
        out = net(x.cuda(), maps.cuda(), rboxes.clone().cuda(), 0).cpu()
    text = cfgs.KNOWN["yolov3"]()
    strict = rows_total = 0
    for f in (0, 3, 7):
        rb = rboxes[rboxes[:, 0] == f].clone()
        rb[:, 0] = 0
        ref = network_ref.network_forward(text, sd, x[f:f + 1], maps[f:f + 1], rb, 0, conf_thresh=0.2, tap_module=91, storage=dtype)
        mine = _frame_rows(out, f)
        assert ref.shape[0] >= 3 and abs(mine.shape[0] - ref.shape[0]) <= max(3, 0.25 * ref.shape[0]), (f, mine.shape, ref.shape)
        t_fwd, t_back = _tie_share(ref, mine, px, tol), _tie_share(mine, ref, px, tol)
        share = _match_share(ref, mine, px, tol)
        print(f"[rows16 {dtype}] frame {f}: {ref.shape[0]} restatement rows / {mine.shape[0]} HIP rows; strict {share:.1%}, tie-aware {t_fwd:.1%} / {t_back:.1%}")
        assert min(t_fwd, t_back) >= (0.95 if dtype == "bf16" else 0.97), f"{dtype} frame {f}: tie-aware share {t_fwd:.0%} / {t_back:.0%}"
        strict += share * ref.shape[0]
        rows_total += ref.shape[0]
    print(f"[rows16 {dtype}] strict share over the three frames {strict / rows_total:.1%}")
    assert strict >= (0.85 if dtype == "bf16" else 0.95) * rows_total


def test_nms_608_batch16_bitexact(hip_lib):
    """NMS at configs[4]'s shape: 16 images x 22 743 rows x (5 + 80) - 89 waves of candidates per image - bit-exact against
    the oracle (index work), in-place xywh -> xyxy write-back included."""
    from millieye_amd import hip
    from tests.test_gpu_nms import _oracle_nms_cpp, _pred
    n, rows, nc = 16, 22743, 80
    pred = _pred("nms608", n, rows, nc, frac_pass=0.05, size=608.0)
    ref, ref_pred = _oracle_nms_cpp(pred, 0.2)
    dev = pred.cuda()
    det, cnt = hip.nms_batched(dev, 0.2, 0.5, 200, writeback_xyxy=True)
    torch.cuda.synchronize()
    assert torch.equal(dev.cpu(), ref_pred), "in-place xywh->xyxy writeback differs"
    cnt = cnt.cpu().tolist()
    assert max(cnt) == 200, "the 200-detection cap must bind on this input"
    for i in range(n):
        assert cnt[i] == ref[i].shape[0], f"image {i}: kept {cnt[i]} vs oracle {ref[i].shape[0]}"
        assert torch.equal(det[i, :cnt[i]].cpu(), ref[i]), f"image {i}: kept rows differ"
