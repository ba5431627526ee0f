"""-m gpu: the fusion path - RoI pooling kernels (bit-exact vs oracle/tv_ops.c) and
Network.forward modes 0/1/2 (HIP path vs the CPU oracle AND vs the reference's own golden
outputs, 1e-3).  All product calls go through the C ABI."""
import os

import numpy as np
import pytest
import torch

from millieye_amd import cfgs, synth
from tests import parity_helpers as ph
from tests.golden.make_golden import NETWORK_CASES

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
TOL = 1e-3


def _rois(tag, k, n, size):
    c = synth.uniform(tag + "c", (k, 2), 0.1 * size, 0.9 * size)
    half = synth.uniform(tag + "h", (k, 2), 2.0, 0.45 * size)
    idx = np.floor(synth.uniform(tag + "i", (k, 1), 0, n))
    r = np.concatenate([idx, c - half, c + half], 1).astype(np.float32)
    r[0, 1:] = [10.0, 10.0, 10.0, 10.0]      # zero-area box (roi_align clamps to 1, ps_roi_align divides by 0)
    r[1, 1:] = [-50.0, -30.0, size + 80.0, size + 40.0]  # larger than the map: out-of-range samples
    r[2, 1:] = [100.0, 90.0, 60.0, 40.0]     # inverted box (negative extent)
    return torch.from_numpy(r)


@pytest.mark.parametrize("h,w,n", [(26, 26, 2), (10, 10, 3), (13, 20, 1)])
def test_roi_ops_bitexact_vs_oracle(hip_lib, h, w, n):
    from millieye_amd import hip
    from oracle import tv_ops
    size = 16.0 * max(h, w)
    rois = _rois(f"roi{h}", 40, n, size)
    m10 = torch.from_numpy(synth.uniform(f"m10{h}", (n, 10, h, w), -1, 1))
    m490 = torch.from_numpy(synth.uniform(f"m490{h}", (n, 490, h, w), -1, 1))
    ref_r = tv_ops.roi_align(m10, rois, (7, 7), 1 / 16)
    ref_p = tv_ops.ps_roi_align(m490, rois, (7, 7), 1 / 16)
    got_r = hip.roi_align(m10.permute(0, 2, 3, 1).contiguous().cuda(), rois).cpu()
    got_p = hip.ps_roi_align(m490.permute(0, 2, 3, 1).contiguous().cuda(), rois).cpu()
    # NaN from 0/0 (degenerate PS-RoI) must match too -> compare bit patterns
    assert torch.equal(got_r.view(torch.int32), ref_r.view(torch.int32)), "roi_align differs"
    same = (got_p.view(torch.int32) == ref_p.view(torch.int32)) | (got_p.isnan() & ref_p.isnan())
    assert bool(same.all()), "ps_roi_align differs"
    assert hip.roi_align(m10.permute(0, 2, 3, 1).contiguous().cuda(), torch.zeros((0, 5))).shape == (0, 10, 7, 7)


def _build(name, cfg, conf):
    from millieye_amd.my_models import Network, define_yolo
    net = Network(define_yolo(ph.cfg_path(cfg)), conf).eval()
    synth.fill_network_(net, name)
    return net


def _inputs(name, n, s):
    x = torch.from_numpy(synth.uniform(name + "/x", (n, 3, s, s)))
    maps, rboxes = synth.radar_inputs(name + "/radar", n, s // 16)
    return x, torch.from_numpy(maps), torch.from_numpy(rboxes)


def _cmp_rows(got, ref, what):
    got = got.detach().cpu()
    ref = torch.as_tensor(ref)
    assert got.shape == ref.shape, f"{what}: {tuple(got.shape)} vs {tuple(ref.shape)}"
    if got.numel():
        assert torch.equal(got[:, 0], ref[:, 0]), f"{what}: image index column differs (row order)"
        ph.assert_close(got, ref, TOL, what)


def _row_err(g, r):
    """Relative error of one output row ``(image, x1, y1, x2, y2, conf, cls_score, cls)``: 1e-3 of max(1, |ref|) per element
    as everywhere else - except that a box CORNER is judged against the box's own extent along its axis as well.  A corner is
    centre -+ extent / 2 with the extent scaled by exp(dw) (``box_regress``, my_models.py:378-391), so its rounding error
    scales with the extent, not with where the corner happens to land: y1 = -0.80 of a 98-pixel-tall box that differs by
    1.2e-3 pixels between two accumulation orders is a 1.2e-5 relative error, not a 1.2e-3 one (seen on one GPU box of the
    pool in round 4, where the autotuner picked other tiles for the batch-32 plan than for the batch-1 plan)."""
    g, r = g.double(), r.double()
    floor = torch.ones_like(r)
    wid, hgt = (r[3] - r[1]).abs(), (r[4] - r[2]).abs()
    floor[1] = floor[3] = torch.clamp(wid, min=1.0)
    floor[2] = floor[4] = torch.clamp(hgt, min=1.0)
    return float(((g - r).abs() / torch.maximum(r.abs(), floor)).max())


def _cmp_rows_ties(got, ref, what, tie=2e-6):
    """_cmp_rows for long row lists: rows whose sort keys (fused confidence, column 5) are closer than ``tie`` may come in
    either order (a 1e-7 difference between the device and the CPU arithmetic flips a stable sort) - they are matched as a
    set; everything else must agree position by position."""
    got, ref = got.detach().cpu(), torch.as_tensor(ref)
    assert got.shape == ref.shape, f"{what}: {tuple(got.shape)} vs {tuple(ref.shape)}"
    used = torch.zeros(len(ref), dtype=torch.bool)
    for i in range(len(got)):
        cands = [j for j in range(max(0, i - 4), min(len(ref), i + 5))
                 if not used[j] and (j == i or abs(float(ref[j, 5] - got[i, 5])) <= tie)]
        ok = [j for j in cands if _row_err(got[i], ref[j]) <= TOL]
        assert ok, f"{what}: row {i} {got[i].tolist()} has no counterpart (reference row {ref[i].tolist()})"
        used[ok[0]] = True


@pytest.mark.parametrize("name,cfg,n,s,conf", NETWORK_CASES)
def test_network_modes_vs_oracle_and_reference_golden(hip_lib, name, cfg, n, s, conf):
    from oracle import network_ref
    g = np.load(os.path.join(GOLD, name + ".npz"))
    net = _build(name, cfg, conf)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    x, maps, rboxes = _inputs(name, n, s)
    text = cfgs.KNOWN[cfg]()
    net = net.cuda()
    xd, md = x.cuda(), maps.cuda()
    with torch.no_grad():
        out1 = net(xd, md, rboxes.clone().cuda(), 1)
        rb = rboxes.clone().cuda()
        out0 = net(xd, md, rb, 0)
        assert torch.allclose(rb[:, 1:].cpu(), rboxes[:, 1:] * s), "radar boxes must be scaled in place (reference :491)"
        out0n = net(xd, md, torch.zeros((0, 5)).cuda(), 0)
        out2 = net(xd, md, rboxes.clone().cuda(), 2)
        assert net.refine_threshold_img == 1  # quirk q3: radar-only mode is sticky
        out0b = net(xd, md, rboxes.clone().cuda(), 0)
    torch.cuda.synchronize()
    ref0, internals = network_ref.network_forward(text, sd, x, maps, rboxes, 0, conf_thresh=conf, return_internals=True)
    _cmp_rows(out1, network_ref.network_forward(text, sd, x, maps, rboxes, 1, conf_thresh=conf), "mode1 vs oracle")
    _cmp_rows(out0, ref0, "mode0 vs oracle")
    _cmp_rows(out2, network_ref.network_forward(text, sd, x, maps, rboxes, 2, conf_thresh=conf), "mode2 vs oracle")
    for got, key in ((out1, "mode1"), (out0, "mode0"), (out0n, "mode0_noradar"), (out2, "mode2"),
                     (out0b, "mode0_after_mode2")):
        _cmp_rows(got, g[key], f"{key} vs reference golden")
    # ordering property: confidences (radar ones / 5) are non-increasing
    assert out0.shape[1] == 8 and out0.device.type == "cuda"


def test_network_darknet53_extension_tap(hip_lib):
    """Darknet-53 + heads (tap = module 91, the documented extension) vs the oracle."""
    from oracle import network_ref
    name, cfg, n, s, conf = "net53", "yolov3", 2, 160, 0.2
    net = _build(name, cfg, conf)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    x, maps, rboxes = _inputs(name, n, s)
    ref = network_ref.network_forward(cfgs.KNOWN[cfg](), sd, x, maps, rboxes, 0, conf_thresh=conf, tap_module=91)
    net = net.cuda()
    with torch.no_grad():
        out = net(x.cuda(), maps.cuda(), rboxes.clone().cuda(), 0)
    assert ref.shape[0] > 4
    _cmp_rows(out, ref, "darknet53 network mode0")


def test_train_py_call_form_is_recognised(hip_lib):
    """train.py:185 passes targets in the model_mode slot: must not be read as a mode.  The model is in
    eval() mode here: the reference has no mode check (my_models.py:545-641), so the call returns the training tuple
    computed with running-statistics BatchNorm (numerics: test_gpu_train.py::test_eval_mode_training_step_...)."""
    net = _build("callform", "yolov3-tiny-12", 0.2).cuda()
    x, maps, rboxes = _inputs("callform", 2, 96)
    tg = torch.tensor([[0, 0, 0.5, 0.5, 0.3, 0.3], [1, 0, 0.4, 0.6, 0.2, 0.3]])
    res = net(x.cuda(), maps.cuda(), rboxes.cuda(), tg)
    assert isinstance(res, tuple) and len(res) == 4
    loss, output, metric, att = res
    assert loss.requires_grad and output.shape[1] == 8 and metric["total"] >= 0 and tuple(att.shape) == (2, 1, 6, 6)
    # a mix of train- and eval-mode head BatchNorms is refused, loudly
    net.img_cnn_layers.train()
    with pytest.raises(NotImplementedError):
        net(x.cuda(), maps.cuda(), rboxes.cuda(), tg.clone())


def test_demo_radar_map_size_quirk(hip_lib):
    """run_sp.py / run_mp.py feed the raw 32x32 radar map with a 416 image (feature map 26x26) and the same
    spatial_scale (quirk q15): the radar branch pools from a map of its own size.  Here 12x12 radar maps with a
    160 px image (10x10 features), against the oracle."""
    from oracle import network_ref
    name, cfg, n, s, conf = "q15", "yolov3-tiny-12", 2, 160, 0.2
    net = _build(name, cfg, conf).eval()
    x = torch.from_numpy(synth.uniform(name + "/x", (n, 3, s, s)))
    maps, rboxes = synth.radar_inputs(name + "/radar", n, 12)
    maps, rboxes = torch.from_numpy(maps), torch.from_numpy(rboxes)
    ref = network_ref.network_forward(cfgs.KNOWN[cfg](), net.state_dict(), x, maps, rboxes, 0, conf)
    net = net.to(net.device)
    out = net(x.cuda(), maps.cuda(), rboxes.clone().cuda(), 0)
    _cmp_rows(out.cpu(), ref, "mode 0 with 12x12 radar maps on a 10x10 feature map")
    assert ref.shape[0] > 4


def test_network_forward_is_repeatable(hip_lib):
    """Same inputs, same rows - bit for bit - across runs (the score maps run on a second stream beside NMS; split-K sums
    are ordered; nothing in the inference path uses atomics)."""
    name, cfg, n, s, conf = NETWORK_CASES[0]
    net = _build(name, cfg, conf).eval()
    net = net.to(net.device)
    x, maps, rboxes = _inputs(name, n, s)
    first = net(x.cuda(), maps.cuda(), rboxes.clone().cuda(), 0).clone()
    for _ in range(20):
        again = net(x.cuda(), maps.cuda(), rboxes.clone().cuda(), 0)
        assert torch.equal(first, again)


@pytest.mark.parametrize("case", [0, 1])
def test_two_launch_heads_equal_the_fused_launch(hip_lib, case):
    """Per-bin RoI pooling + the heads on the fp32 matrix pipe (32 RoIs per workgroup, v_mfma_f32_32x32x2_f32 for net0) against
    the single fused VALU launch (a thread per pooled value, a hidden unit per thread): the same rows.  The pooled values are
    the same operations in the same order, the MFMA accumulates k in order with fp32 FMAs like the VALU chain - measured
    bit-identical; the bar is 2e-6 relative so that a different FMA grouping inside the matrix pipe would not be a failure."""
    name, cfg, n, s, conf = NETWORK_CASES[case]
    net = _build(name, cfg, conf).eval()
    net = net.to(net.device)
    x, maps, rboxes = _inputs(name, n, s)
    split = net(x.cuda(), maps.cuda(), rboxes.clone().cuda(), 0).clone()
    net._fused_heads = True
    fused = net(x.cuda(), maps.cuda(), rboxes.clone().cuda(), 0).clone()
    net._fused_heads = False
    assert split.shape == fused.shape and split.shape[0] > 0
    torch.testing.assert_close(split, fused, rtol=2e-6, atol=2e-6)
    print(f"two-launch heads vs fused launch ({name}): bit-identical = {bool(torch.equal(split, fused))}")


def test_pooled_tail_scratch_and_polled_row_count(hip_lib, monkeypatch):
    """Round 5: the tail's scratch tensors come from a per-network, per-stream pool - the rows a call returned must survive the
    next calls (they are never pool memory), a forward issued on another stream gets its own scratch (forwards of ONE network are
    still one at a time: the detector's plan arena is per network), and the A/B form of the row count
    (the kernel stores it into a pinned host word that the host polls, MILLIEYE_COUNT_SPIN=1) returns the same rows."""
    from millieye_amd import my_models
    name, cfg, n, s, conf = NETWORK_CASES[0]
    net = _build(name, cfg, conf).eval()
    net = net.to(net.device)
    x, maps, rboxes = _inputs(name, n, s)
    first = net(x.cuda(), maps.cuda(), rboxes.clone().cuda(), 0)
    keep = first.clone()
    yolo_only = net(x.cuda(), maps.cuda(), rboxes.clone().cuda(), 1)
    keep1 = yolo_only.clone()
    other = net(x.flip(0).contiguous().cuda(), maps.flip(0).contiguous().cuda(), rboxes.clone().cuda(), 0)
    assert torch.equal(first, keep) and torch.equal(yolo_only, keep1), "a later forward overwrote rows an earlier one returned"
    assert other.shape[1] == 8
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        again = net(x.cuda(), maps.cuda(), rboxes.clone().cuda(), 0)
    side.synchronize()
    assert torch.equal(again, keep)
    assert len({k[-1] for k in net._tail_bufs}) >= 3, "scratch is keyed by stream (two caller streams + the network's own side stream)"
    monkeypatch.setattr(my_models, "_COUNT_SPIN", True)
    polled = net(x.cuda(), maps.cuda(), rboxes.clone().cuda(), 0)
    assert torch.equal(polled, keep)


def test_bin_major_score_map_follows_weight_updates(hip_lib):
    """The 1x1 score-map convolution runs on row-permuted copies of its packed weights (bin-major channels for the pooling
    launch): an in-place update of the module's parameters must reach them - the rows after the update equal the oracle's on the
    updated state and differ from the rows before."""
    from oracle import network_ref
    name, cfg, n, s, conf = NETWORK_CASES[0]
    net = _build(name, cfg, conf).eval()
    net = net.to(net.device)
    x, maps, rboxes = _inputs(name, n, s)
    before = net(x.cuda(), maps.cuda(), rboxes.clone().cuda(), 0).clone()
    with torch.no_grad():
        net.img_cnn_layers.net[0].weight.mul_(1.25)
        net.img_cnn_layers.net[1].bias.add_(0.05)
    after = net(x.cuda(), maps.cuda(), rboxes.clone().cuda(), 0).clone()
    net._fused_heads = True
    fused = net(x.cuda(), maps.cuda(), rboxes.clone().cuda(), 0).clone()
    assert torch.equal(after, fused)
    assert after.shape != before.shape or not torch.equal(after, before)
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    ref = network_ref.network_forward(cfgs.KNOWN[cfg](), sd, x, maps, rboxes, 0, conf_thresh=conf)
    _cmp_rows(after, ref, "after an in-place weight update")


def test_no_rois_at_all(hip_lib):
    """Nothing passes the confidence threshold and there is no radar proposal: the reference's empty [0,8] result (its
    torch ops run on empty tensors), for modes 0 and 1, and with radar proposals only."""
    from oracle import network_ref
    name, cfg, n, s, conf = "empty", "yolov3-tiny-12", 2, 96, 2.0   # sigmoid outputs never reach 2.0
    net = _build(name, cfg, conf).eval()
    x = torch.from_numpy(synth.uniform(name + "/x", (n, 3, s, s)))
    maps, rboxes = synth.radar_inputs(name + "/radar", n, s // 16)
    maps, rboxes = torch.from_numpy(maps), torch.from_numpy(rboxes)
    sd = net.state_dict()
    net = net.to(net.device)
    for mode in (0, 1):
        out = net(x.cuda(), maps.cuda(), torch.zeros((0, 5)).cuda(), mode)
        assert tuple(out.shape) == (0, 8)
    out = net(x.cuda(), maps.cuda(), rboxes.clone().cuda(), 0)      # radar proposals only
    ref = network_ref.network_forward(cfgs.KNOWN[cfg](), sd, x, maps, rboxes, 0, conf)
    _cmp_rows(out.cpu(), ref, "radar-only RoIs")
    assert out.shape[0] == ref.shape[0] > 0


def test_compact_sort_rows_matches_torch(hip_lib):
    """me_compact_sort_rows_f32 against the torch ops it replaces (nonzero -> stable descending sort -> gather), with
    many equal keys (ties must stay in ascending row order), sizes across the 2048-key LDS tile boundary, and cap 0."""
    import ctypes as C
    from millieye_amd import hip
    lib = hip.lib()
    g = torch.Generator().manual_seed(3)
    for cap in (0, 1, 7, 255, 2048, 2049, 6464):
        rows = torch.randn((cap, 8), generator=g).cuda()
        keep = (torch.rand((cap,), generator=g) < 0.6).to(torch.uint8).cuda()
        key = (torch.randint(0, 40, (cap,), generator=g).float() / 40).cuda()  # lots of ties
        out = torch.full((cap, 8), float("nan")).cuda()
        cnt = torch.full((1,), -1, dtype=torch.int32).cuda()
        hip.check(lib.me_compact_sort_rows_f32(rows.data_ptr(), keep.data_ptr(), key.data_ptr(), cap, 8, out.data_ptr(),
                                               cnt.data_ptr(), hip.stream_ptr()), "me_compact_sort_rows_f32")
        idx = torch.nonzero(keep, as_tuple=False).flatten()
        order = torch.sort(key[idx], descending=True, stable=True).indices
        ref = rows[idx[order]]
        m = int(cnt.item())
        assert m == ref.shape[0], (cap, m, ref.shape)
        assert torch.equal(out[:m], ref), f"cap {cap}"


def test_demo_frame_step(hip_lib):
    """millieye_amd.demo.FrameFuser: seeded radar frames -> proposals -> staged frame / raw 32x32 radar map -> Network.forward
    -> extra NMS 0.3 -> boxes in frame pixels (run_mp's per-frame chain without its I/O).  Checks the plumbing: shapes, the
    radar proposals reach the network (mode 0), the auto mode follows the brightness rule, rows stay inside the frame and
    survive a second NMS unchanged, repeatability."""
    from millieye_amd.demo import FrameFuser, mode_selection, radar_boxes_for_network
    from millieye_amd import hip, radar_proposals as rp
    from tests.golden.make_golden import RADAR_CALIB, radar_points
    net = _build("demo", "yolov3-tiny-12", 0.1).eval()
    synth.fill_network_(net, "demo", cls0_bias=3.0, cls_bias=-4.0)
    net = net.to(net.device)
    frame = (synth.uniform("demo/frame", (480, 640, 3)) * 255).astype(np.uint8)
    dark = (frame.astype(np.float32) * 0.1).astype(np.uint8)
    rp.KalmanClusterTracker.count = 0
    fuser = FrameFuser(net, RADAR_CALIB, model_mode=3, min_hits=2)
    outs = []
    for f in range(4):
        rows, info = fuser(dark, [radar_points(f)])
        outs.append((rows, info))
        assert info["mode"] == 0, "a dark frame selects the fusion mode"
        assert rows.shape[1] == 7 and info["points"] > 10
        if len(rows):   # random head weights regress loosely: only gross plumbing errors (wrong scale / padding) are caught here
            assert float(rows[:, :4].min()) > -640 and float(rows[:, :4].max()) < 1280
            assert bool(torch.isfinite(rows).all())
            keep = hip.nms_indices(rows[:, :4], rows[:, 4], rows[:, 6], 0.3)   # already suppressed at 0.3: nothing more goes
            assert len(keep) == len(rows)
    assert outs[-1][1]["radar_boxes"] >= 2 and len(outs[-1][0]) > 0
    rows_b, info_b = fuser(frame, [radar_points(4)])
    assert info_b["mode"] == 1, "a bright frame selects the camera-only mode"
    assert mode_selection(2, None) == 2 and mode_selection(7, None) is None
    rb = radar_boxes_for_network([[10, 20, 110, 220], [700, 10, 650, 50], [-50, -50, 30, 30]], (480, 640))
    assert rb.shape == (2, 5) and torch.allclose(rb[0], torch.tensor([0, 10 / 640, 100 / 640, 110 / 640, 300 / 640]))
    assert float(rb[1, 1]) == 0.0 and float(rb[1, 2]) == 30 / 640   # clamped at 0; the reversed box was dropped
    # same inputs, fresh tracker -> same rows
    rp.KalmanClusterTracker.count = 0
    fuser2 = FrameFuser(net, RADAR_CALIB, model_mode=3, min_hits=2)
    for f in range(4):
        rows2, _ = fuser2(dark, [radar_points(f)])
        assert torch.equal(rows2, outs[f][0])


def test_headline_workload_batch32_vs_oracle(hip_lib):
    """The configuration the metric is quoted on, end to end: yolov3.cfg (Darknet-53) 416x416, batch 32, full
    ``Network.forward`` mode 0 with two radar boxes per frame.  The CPU oracle needs seconds per frame, so it runs on
    eight frames of the batch one at a time (frames are independent units; row order inside a frame is the batch run's
    order because the descending-confidence sort is stable); all 32 frames are compared with their batch-1 HIP runs."""
    from oracle import network_ref
    name, cfg, n, s, conf = "headline", "yolov3", 32, 416, 0.2
    from millieye_amd.my_models import Network, define_yolo
    net = Network(define_yolo(ph.cfg_path(cfg)), conf).eval()
    synth.fill_network_(net, name, cls0_bias=3.0, cls_bias=-4.0)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    x = torch.from_numpy(synth.uniform(name + "/x", (n, 3, s, s)))
    maps, rboxes = synth.radar_inputs(name + "/radar", n, s // 16, boxes_per_image=2)
    maps, rboxes = torch.from_numpy(maps), torch.from_numpy(rboxes)
    net = net.cuda()
    with torch.no_grad():
        out = net(x.cuda(), maps.cuda(), rboxes.clone().cuda(), 0).cpu()
        # every one of the 32 frames: the rows of frame f in the batch run are the rows of the batch-1 run of frame f (1e-3;
        # tile and split-K choices follow the batch, so the accumulation order differs) - the eight oracle frames below then
        # pin the batch-1 arithmetic, this loop pins the batching
        for f in range(n):
            rb = rboxes[rboxes[:, 0] == f].clone()
            rb[:, 0] = 0
            one = net(x[f:f + 1].cuda(), maps[f:f + 1].cuda(), rb.cuda(), 0).cpu()
            got = out[out[:, 0] == f].clone()
            got[:, 0] = 0
            _cmp_rows_ties(got, one, f"headline batch-32 run vs batch-1 run, frame {f}")
    assert out.shape[0] > 32 and out.shape[1] == 8
    text = cfgs.KNOWN[cfg]()
    total = 0
    for f in (0, 4, 9, 13, 18, 22, 27, 31):  # (eight oracle frames, every fourth of the batch)
        rb = rboxes[rboxes[:, 0] == f].clone()
        rb[:, 0] = 0
        ref = network_ref.network_forward(text, sd, x[f:f + 1], maps[f:f + 1], rb, 0, conf_thresh=conf, tap_module=91)
        got = out[out[:, 0] == f].clone()
        got[:, 0] = 0
        _cmp_rows_ties(got, ref, f"headline batch-32 run, frame {f}")
        total += ref.shape[0]
    assert total >= 16, "the sampled frames must carry detections"


def test_two_process_pipeline_equals_the_sequential_fuser(hip_lib):
    """millieye_amd/pipeline.py (run_mp.py's producer / consumer split) with the real device half: the rows of every frame
    equal those of the single-process FrameFuser on the same frames (back-pressure mode: no frame dropped)."""
    from millieye_amd import radar_proposals as rp
    from millieye_amd.demo import FrameFuser
    from millieye_amd.pipeline import FusionPipeline
    from tests.golden.make_golden import RADAR_CALIB
    from tests.pipeline_helpers import SyntheticSource
    net = _build("demo", "yolov3-tiny-12", 0.1).eval()
    synth.fill_network_(net, "demo", cls0_bias=3.0, cls_bias=-4.0)
    net = net.to(net.device)
    n = 6
    rp.KalmanClusterTracker.count = 0
    seq = FrameFuser(net, RADAR_CALIB, model_mode=3, min_hits=2)
    want = [seq(frame, radar) for frame, radar in SyntheticSource(n)()]
    rp.KalmanClusterTracker.count = 0
    pipe = FusionPipeline(FrameFuser(net, RADAR_CALIB, model_mode=3, min_hits=2), SyntheticSource(n), drop_oldest=False)
    got = list(pipe)
    assert [info["frame_idx"] for _r, info in got] == list(range(n)) and pipe.stats["dropped"] == 0
    for (rows, info), (rows_w, info_w) in zip(got, want):
        assert torch.equal(rows, rows_w) and info["mode"] == info_w["mode"] == 0
    assert len(got[-1][0]) > 0


def test_roi_heads_direct_2400_rois_vs_oracle(hip_lib):
    """``me_roi_heads_f32`` (roi_pool10_kernel + roi_heads_mfma_kernel, the product's two-launch path)
    called directly through the C ABI on synthetic score maps and 2 400 image proposals + 64 radar boxes over 8 frames
    (degenerate / out-of-range / inverted boxes included), then ``me_compact_sort_rows_f32`` - against the oracle's
    tv_ops RoI pooling + refinement / ensemble heads + box_regress + stable sort.  The new heads no longer lean on the four
    headline frames."""
    import ctypes as C

    from millieye_amd import hip
    from oracle import network_ref as nr
    from oracle import tv_ops
    name, n, fh, fw, k_img, k_rad = "headsdirect", 8, 26, 26, 2400, 64
    size = 16.0 * fh
    net = _build(name, "yolov3-tiny-12", 0.2).eval()
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    net = net.to(net.device)
    dev = torch.device("cuda")
    m490 = torch.from_numpy(synth.uniform(name + "/m490", (n, 490, fh, fw), -1, 1))
    m10 = torch.from_numpy(synth.uniform(name + "/m10", (n, 10, fh, fw), 0, 1))
    rois = _rois(name + "/img", k_img, n, size)
    rois = rois[torch.sort(rois[:, 0], stable=True).indices]  # image order, like the proposal assembly
    cols = 8 + net.class_num
    img_boxes = torch.zeros((k_img, cols))
    img_boxes[:, :5] = rois
    img_boxes[:, 5:] = torch.from_numpy(synth.uniform(name + "/conf", (k_img, cols - 5), 0.05, 0.95))
    img_boxes[:, 7] = 0.0
    rad = _rois(name + "/rad", k_rad, n, size)[3:]  # (the degenerate boxes stay among the image proposals)
    k_rad = rad.shape[0]
    cap_img = k_img + 100  # slots behind the last proposal, like n * detections_per_img
    cap = cap_img + k_rad

    # oracle
    boxes_all = torch.cat((img_boxes[:, :5], rad), 0)
    crop_img = tv_ops.ps_roi_align(m490, boxes_all, (7, 7), spatial_scale=1. / 16)
    crop_rad = tv_ops.roi_align(m10, boxes_all, (7, 7), spatial_scale=1. / 16)
    regress_ref, refine_ref = nr.refinement(sd, crop_rad, crop_img)
    yolo_vec = torch.cat((img_boxes[:, 5:6], img_boxes[:, 8:]), 1)
    masks_img = nr.ensemble(sd, refine_ref[:k_img], yolo_vec)
    mask1_ref = torch.cat((masks_img[:, 0], refine_ref[k_img:, 0]), 0)

    # product: bin-major image map (channel (ph*7+pw)*10 + c holds the reference's channel (c*7+ph)*7+pw), radar map pitch 12
    q = torch.arange(490)
    img_map = m490.permute(0, 2, 3, 1)[..., (q % 10) * 49 + q // 10].contiguous().to(dev)
    rad_map = torch.zeros((n, fh, fw, 12))
    rad_map[..., :10] = m10.permute(0, 2, 3, 1)
    rad_map = rad_map.to(dev)
    boxes_d = torch.zeros((cap_img, cols))
    boxes_d[:k_img] = img_boxes
    boxes_d = boxes_d.to(dev)
    n_img_d = torch.tensor([k_img], dtype=torch.int32, device=dev)
    rad_d = rad.contiguous().to(dev)
    f32 = dict(device=dev, dtype=torch.float32)
    regress, refine, mask1 = torch.empty((cap, 4), **f32), torch.empty((cap, 2), **f32), torch.empty((cap,), **f32)
    rows, key = torch.empty((cap, 8), **f32), torch.empty((cap,), **f32)
    keep = torch.full((cap,), 7, device=dev, dtype=torch.uint8)
    pooled = torch.empty((cap, 980), **f32)
    hw = net._get_packs()["heads"].refresh(dev)
    d = hip.HeadsDesc()
    d.img_map, d.radar_map = img_map.data_ptr(), rad_map.data_ptr()
    d.img_pitch, d.radar_pitch, d.img_bin_major = 490, 12, 1
    d.n, d.fh, d.fw, d.spatial_scale, d.rh, d.rw = n, fh, fw, 1.0 / 16, fh, fw
    d.img_boxes, d.n_img, d.n_img_cap, d.box_cols = boxes_d.data_ptr(), n_img_d.data_ptr(), cap_img, cols
    d.radar_boxes, d.n_radar = rad_d.data_ptr(), k_rad
    d.thr_img, d.thr_radar, d.regress = 0.0, 0.0, 1
    for nm in ("w0t", "b0", "w1", "b1", "w2", "b2", "rw", "rscale", "rshift", "rw2", "rb2", "e1w", "e1b", "e2w", "e2b"):
        setattr(d.wts, nm, hw[nm].data_ptr())
    d.regress_out, d.refine_out, d.mask1_out = regress.data_ptr(), refine.data_ptr(), mask1.data_ptr()
    d.out_rows, d.keep, d.sort_key = rows.data_ptr(), keep.data_ptr(), key.data_ptr()
    d.pool_scratch = pooled.data_ptr()
    lib = hip.lib()
    hip.check(lib.me_roi_heads_f32(C.byref(d), hip.stream_ptr()), "me_roi_heads_f32")
    ordered = torch.empty((cap, 8), **f32)
    n_out = torch.empty((1,), device=dev, dtype=torch.int32)
    hip.check(lib.me_compact_sort_rows_f32(rows.data_ptr(), keep.data_ptr(), key.data_ptr(), cap, 8, ordered.data_ptr(),
                                           n_out.data_ptr(), hip.stream_ptr()), "me_compact_sort_rows_f32")
    torch.cuda.synchronize()

    # RoIs are indexed compactly (image proposals, then radar boxes).  The pooled features are bit-exact (same operations in
    # the same order); the NaN row of the zero-area PS-RoI included
    k = k_img + k_rad
    got_pool = pooled[:k].cpu()
    want_img, want_rad = crop_img.reshape(k, 490), crop_rad.reshape(k, 490)
    same = (got_pool[:, :490].view(torch.int32) == want_img.view(torch.int32)) | (got_pool[:, :490].isnan() & want_img.isnan())
    assert bool(same.all()), "image-branch pooled features differ from tv_ops.ps_roi_align"
    assert torch.equal(got_pool[:, 490:].view(torch.int32), want_rad.view(torch.int32)), "radar-branch pooled features differ"
    finite = torch.isfinite(regress_ref).all(1) & torch.isfinite(refine_ref).all(1) & torch.isfinite(mask1_ref)
    assert int(finite.sum()) >= k - 2  # (only the zero-area PS-RoI is NaN)
    torch.testing.assert_close(regress[:k].cpu()[finite], regress_ref[finite], rtol=TOL, atol=TOL)
    torch.testing.assert_close(refine[:k].cpu()[finite], refine_ref[finite], rtol=TOL, atol=TOL)
    torch.testing.assert_close(mask1[:k].cpu()[finite], mask1_ref[finite], rtol=TOL, atol=TOL)
    assert bool((keep[k:cap] == 0).all()), "slots behind the last RoI must keep nothing"

    # rows: kept = mask > threshold; the oracle's assembly (network_ref.network_forward) on the same proposals
    radar_rows = torch.cat((rad, refine_ref[k_img:], torch.zeros((k_rad, 1)), refine_ref[k_img:, 1:]), -1)
    boxes = torch.cat((img_boxes[:, :8], radar_rows[:, :8]), 0)
    positive = mask1_ref > 0.0
    located = nr.box_regress(regress_ref[positive], boxes[positive, 1:5])
    out_ref = torch.cat((boxes[positive, :1], located, mask1_ref[positive, None], boxes[positive, 6:8]), -1)
    tmp = mask1_ref.clone()
    tmp[k_img:] /= 5
    out_ref = out_ref[torch.sort(tmp[positive], descending=True, stable=True).indices]
    got = ordered[: int(n_out.item())].cpu()
    assert got.shape[0] >= 2000
    _cmp_rows_ties(got, out_ref, "direct heads call, 2400 + 61 RoIs")
