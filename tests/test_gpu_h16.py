"""16-bit storage modes (bfloat16 / IEEE half; BASELINE configs[2] / [4]) through the C ABI: ``me_conv2d_h16`` and the
16-bit NHWC helpers against torch CPU fp32 ops applied to the same rounded operands.  "ulp" below is one unit in the last
place of the storage type (2^-7 relative for bfloat16, 2^-10 for half).

Parity bar for this mode (written here): with identical bf16 inputs and weights the fp32-output form
(``y_f32``) must match the fp32 CPU convolution to 1e-3 (only the accumulation order differs); the bf16-output form must
equal the RNE rounding of that fp32 result except where the two fp32 values straddle a rounding boundary - there the
difference is one bf16 ulp (<= 2^-7 relative; + 1e-5 absolute where a sum cancels to ~0) - and such elements must stay
below 0.5 % of the tensor.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


HALVES = {"bf16": torch.bfloat16, "f16": torch.float16}
ULP = {torch.bfloat16: 2.0 ** -7, torch.float16: 2.0 ** -10}


def _bf(t, half=torch.bfloat16):
    return t.to(half)


def _ref(x_bf, w_bf, scale, shift, k, s, pad, act, res=None, ups=1):
    y = F.conv2d(x_bf.float().permute(0, 3, 1, 2), w_bf.float(), stride=s, padding=pad)
    y = y * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    if act == 1:
        y = torch.where(y > 0, y, 0.1 * y)
    y = y.permute(0, 2, 3, 1)
    if res is not None:
        y = y + res.float()
    if ups == 2:
        y = y.repeat_interleave(2, 1).repeat_interleave(2, 2)
    return y.contiguous()


def _check_bf16(got_bf, ref_f32, what):
    half = got_bf.dtype
    got = got_bf.float().cpu()
    want = ref_f32.to(half).float()
    diff = (got - want).abs()
    # one ulp is at most 2^-7 (bf16) / 2^-10 (half) relative; 1e-5 absolute covers sums that cancel to ~0
    ulp = torch.maximum(want.abs(), ref_f32.abs()) * ULP[half] + 1e-5
    assert bool((diff <= ulp).all()), f"{what}: max {float((diff / ulp).max()):.2f} ulp"
    frac = float((diff > 0).float().mean())
    # a rounding flips when the two fp32 sums straddle a boundary: ~8x more often at half's 8x finer grid
    assert frac < (5e-3 if half == torch.bfloat16 else 4e-2), f"{what}: {frac:.4%} of the elements differ from the rounded fp32 result"


CASES = [
    # name, n, h, w, cin, cout, k, s, act, res, ups
    ("3x3 residual", 2, 26, 26, 64, 128, 3, 1, 1, True, 1),
    ("3x3 stride 2", 2, 52, 52, 64, 128, 3, 2, 1, False, 1),
    ("1x1 cin32", 3, 20, 20, 32, 64, 1, 1, 1, False, 1),
    ("1x1 upsample", 2, 13, 13, 256, 128, 1, 1, 1, False, 2),
    ("3x3 ragged", 1, 13, 13, 96, 72, 3, 1, 0, False, 1),
    ("3x3 wide", 1, 13, 13, 512, 320, 3, 1, 1, True, 1),
]


@pytest.mark.parametrize("half", ["bf16", "f16"])
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_conv_bf16_tiles(hip_lib, case, half):
    from millieye_amd import hip
    half = HALVES[half]
    name, n, h, w, cin, cout, k, s, act, with_res, ups = case
    g = torch.Generator().manual_seed(len(name) * 7 + cin)
    x = _bf(torch.randn((n, h, w, cin), generator=g), half)
    wgt = _bf(torch.randn((cout, cin, k, k), generator=g) / (k * k * cin) ** 0.5, half)
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.1
    pad = (k - 1) // 2
    ho = (h + 2 * pad - k) // s + 1
    res = _bf(torch.randn((n, ho, ho, cout), generator=g), half) if with_res else None
    ref = _ref(x, wgt, scale, shift, k, s, pad, act, res, ups)
    packed = wgt.permute(0, 2, 3, 1).contiguous().cuda()
    xs, sc, sh = x.cuda(), scale.cuda(), shift.cuda()
    rs = res.cuda() if res is not None else None
    tiles = (1, 2, 3, 4, 5, 11, 12, 13, 14) if cin % 64 == 0 else (1, 2, 3, 4, 5)
    for tile in tiles:
        for split in (1, 2, 3):
            if split > k * k * (cin // 64 if cin % 64 == 0 and tile < 10 else cin // 32):
                continue
            y = hip.conv2d_h16(xs, packed, sc, sh, k, s, pad, act, residual=rs, upsample=ups, tile=tile, split_k=split)
            _check_bf16(y, ref, f"{name} tile {tile} split {split}")
    if k == 3 and s == 1 and ups == 1 and cout % 128 == 0:   # a patch-resident tile on the same case (more in the p8 tests)
        y = hip.conv2d_h16(xs, packed, sc, sh, k, s, pad, act, residual=rs, upsample=ups, tile=221, split_k=1)
        _check_bf16(y, ref, f"{name} tile 221 (patch-resident)")
    # fp32 output form (detection convs): accumulation order is the only difference from the CPU convolution
    rs32 = res.float().cuda() if res is not None else None
    y32 = hip.conv2d_h16(xs, packed, sc, sh, k, s, pad, act, residual=rs32, upsample=ups, y_f32=True)
    ref32 = _ref(x, wgt, scale, shift, k, s, pad, act, res, ups)
    err = (y32.cpu() - ref32).abs()
    assert bool((err <= 1e-3 * torch.clamp(ref32.abs(), min=1.0)).all()), f"{name} fp32 out: {float(err.max())}"


def test_conv_bf16_detection_and_slices(hip_lib):
    """cout = 255 (ragged weight rows -> zero-filled by the buffer range check), fp32 output, input given as a channel
    slice of a wider NHWC buffer ([route]), output written into a slice (pitched)."""
    from millieye_amd import hip
    g = torch.Generator().manual_seed(5)
    n, h, cin, cout = 2, 13, 128, 255
    wide = _bf(torch.randn((n, h, h, 320), generator=g))
    x = wide[..., 64:64 + cin]
    wgt = _bf(torch.randn((cout, cin, 1, 1), generator=g) / cin ** 0.5)
    scale, shift = torch.ones(cout), torch.randn(cout, generator=g)
    ref = _ref(x.contiguous(), wgt, scale, shift, 1, 1, 0, 0)
    packed = wgt.permute(0, 2, 3, 1).contiguous().cuda()
    y = hip.conv2d_h16(wide.cuda()[..., 64:64 + cin], packed, scale.cuda(), shift.cuda(), 1, 1, 0, 0, y_f32=True)
    err = (y.cpu() - ref).abs()
    assert bool((err <= 1e-3 * torch.clamp(ref.abs(), min=1.0)).all()), float(err.max())


def test_stem_and_helpers_bf16(hip_lib):
    """cin = 3 stem in the 16-bit modes: frame and weights are rounded once to the storage type, accumulation is fp32.  Both
    implementations - the MFMA stem (cout 32, csrc/stem_mfma_h16.hip; needs the [32][32] tap-padded weight copy) and the VALU
    fallback - against the fp32 CPU convolution of the rounded operands; NCHW and NHWC frames, an M that is not a multiple of
    32, output written into a channel slice."""
    from millieye_amd import hip
    g = torch.Generator().manual_seed(9)
    for shape in ((2, 3, 40, 40), (3, 3, 13, 21)):
        x = torch.rand(shape, generator=g)
        for cout, half in ((16, torch.bfloat16), (32, torch.bfloat16), (64, torch.bfloat16), (32, torch.float16)):
            w = torch.randn((cout, 3, 3, 3), generator=g) * 0.2
            scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
            ref = F.conv2d(x.to(half).float(), w.to(half).float(), padding=1) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
            ref = torch.where(ref > 0, ref, 0.1 * ref).permute(0, 2, 3, 1).contiguous()
            packed = w.to(half).float().permute(0, 2, 3, 1).contiguous().cuda()  # fp32 container, values rounded by the host
            y = hip.conv2d_h16(x.cuda(), packed, scale.cuda(), shift.cuda(), 3, 1, 1, 1, x_nchw=True, half=half)
            assert y.dtype == half
            _check_bf16(y, ref, f"VALU stem cout {cout} {shape}")
            if cout == 32:
                taps = F.pad(packed.reshape(32, 27), (0, 5)).to(half).contiguous()
                ym = hip.conv2d_h16(x.cuda(), packed, scale.cuda(), shift.cuda(), 3, 1, 1, 1, x_nchw=True, half=half,
                                    wgt_tiled=taps)
                _check_bf16(ym, ref, f"MFMA stem {half} {shape} (NCHW frame)")
                xh = x.permute(0, 2, 3, 1).contiguous().cuda()
                yh = hip.conv2d_h16(xh, packed, scale.cuda(), shift.cuda(), 3, 1, 1, 1, x_nchw=False, half=half, wgt_tiled=taps)
                assert torch.equal(yh, ym), "NHWC frame"
                wide = torch.zeros(ym.shape[:3] + (48,), dtype=half).cuda()
                hip.conv2d_h16(x.cuda(), packed, scale.cuda(), shift.cuda(), 3, 1, 1, 1, x_nchw=True, half=half, wgt_tiled=taps,
                               out=wide[..., 8:40])
                assert torch.equal(wide[..., 8:40], ym) and float(wide[..., :8].abs().max()) == 0
                yv = hip.conv2d_h16(x.cuda(), packed, scale.cuda(), shift.cuda(), 3, 1, 1, 1, x_nchw=True, half=half,
                                    wgt_tiled=taps, tile=1)  # tile 1 forces the VALU stem: same rounding points
                assert float((yv.float() - ym.float()).abs().max()) <= 2 * float(ULP[half]) * float(ref.abs().max())
    for half in (torch.bfloat16, torch.float16):
        _helpers(hip, g, half)


def _helpers(hip, g, half):
    a = _bf(torch.randn((2, 13, 13, 64), generator=g), half)
    b = _bf(torch.randn((2, 13, 13, 64), generator=g), half)
    ac, bc = a.cuda(), b.cuda()
    # pools are exact on bf16 values
    ref = F.max_pool2d(a.float().permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1)
    assert torch.equal(hip.maxpool_h16(ac, 2, 2).float().cpu(), ref)
    padded = F.pad(a.float().permute(0, 3, 1, 2), (0, 1, 0, 1))
    ref = F.max_pool2d(padded, 2, 1).permute(0, 2, 3, 1)
    assert torch.equal(hip.maxpool_h16(ac, 2, 1, zero_ext=True).float().cpu(), ref)
    ref = a.float().repeat_interleave(2, 1).repeat_interleave(2, 2)
    assert torch.equal(hip.upsample_h16(ac, 2).float().cpu(), ref)
    ref = (a.float() + b.float()).to(half)
    assert torch.equal(hip.add_h16(ac, bc).cpu(), ref)


# ---------------------------------------------------------------------------------------------------------------------
# whole detector / whole pipeline in bf16 storage mode
# ---------------------------------------------------------------------------------------------------------------------
def _err(a, b):
    d = (a.double() - b.double()).abs() / b.double().abs().clamp_min(1.0)
    return float(d.mean()), float(d.max())


DETECTOR_CASES = [("yolov3-tiny-12", 2, 96), ("yolov3-tiny-12", 1, 416), ("yolov3-tiny-coco", 3, 160), ("yolov3", 2, 64),
                  ("yolov3", 1, 416),
                  ("yolov3-tiny-12", 1, 608), ("yolov3", 1, 608)]  # BASELINE configs[4]: 608x608, 16-bit MFMA convolutions


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("name,n,s", DETECTOR_CASES)
def test_detector_bf16(hip_lib, name, n, s, dtype):
    """``Darknet.compute_dtype = "bf16"`` against the oracle's restatement of the same rounding points
    (oracle/darknet_ref.py ``storage="bf16"``) and against the fp32 oracle.

    Bars (bf16 carries 8 significant bits, so the fp32 mode's 1e-3 cannot apply to a 13..75-layer network):
      * shallow cfgs (tiny, 13 convolutions): HIP vs the bf16 restatement - featuremap and yolo_outputs mean relative
        error <= 1e-3, max <= 2e-2 (the two differ only where fp32 accumulation order flips a bf16 rounding);
      * every cfg: the HIP path is no further from the fp32 oracle than the bf16 restatement itself is (mean error within
        1.25x + 1e-4) - the storage format, not the kernels, sets the error;
      * deep cfg (Darknet-53, random weights): rounding flips decorrelate element-wise, so only the second bar applies.
    """
    from oracle import darknet_ref
    from millieye_amd.engine import pick_tap_module
    from tests import parity_helpers as ph

    model = ph.make_darknet(name)
    x = ph.frames(f"{name}/{n}/{s}", n, s)
    tap = pick_tap_module(model.module_defs)
    text, sd = ph.cfg_text(name), model.state_dict()
    f32_fm, f32_y = darknet_ref.darknet_forward(text, sd, x, tap_module=tap)
    b16_fm, b16_y = darknet_ref.darknet_forward(text, sd, x, tap_module=tap, storage=dtype)
    model = model.cuda()
    model.compute_dtype = dtype
    with torch.no_grad():
        fm, y = model(x.cuda())
        fm2, y2 = model(x.cuda())
    assert fm.dtype == torch.float32 and y.dtype == torch.float32 and tuple(fm.shape) == tuple(f32_fm.shape)
    assert torch.equal(y, y2) and torch.equal(fm, fm2), "bf16 mode must be run-to-run deterministic"
    fm, y = fm.cpu(), y.cpu()
    assert bool(torch.isfinite(y).all()) and bool(torch.isfinite(fm).all())
    if "tiny" in name:
        for got, ref, what in ((fm, b16_fm, "featuremap"), (y, b16_y, "yolo_outputs")):
            mean, mx = _err(got, ref)
            assert mean <= 1e-3 and mx <= 2e-2, f"{name} {what} vs bf16 restatement: mean {mean:.2e} max {mx:.2e}"
    for got, b16, ref, what in ((fm, b16_fm, f32_fm, "featuremap"), (y, b16_y, f32_y, "yolo_outputs")):
        hip_mean, _ = _err(got, ref)
        fmt_mean, _ = _err(b16, ref)
        assert hip_mean <= 1.25 * fmt_mean + 1e-4, f"{name} {what}: HIP bf16 {hip_mean:.2e} vs format error {fmt_mean:.2e}"
    # the fp32 engine of the same model is untouched by the mode switch
    model.compute_dtype = "f32"
    with torch.no_grad():
        _fm32, y32 = model(x.cuda())
    ph.assert_close(y32.cpu(), f32_y, 1e-3, f"{name} fp32 after bf16")


def test_network_bf16_close_to_fp32(hip_lib):
    """Full pipeline (detector -> NMS -> RoI heads -> fusion -> output rows) with the detector in bf16 storage mode: the
    same detections as the fp32 run up to the storage error - at least 90 % of the fp32 output rows have a bf16-mode row
    of the same image within 2 px on every corner and 0.03 on the fused probability, and the row counts agree within 10 %."""
    from millieye_amd import synth
    from millieye_amd.my_models import Network, define_yolo
    from tests import parity_helpers as ph
    name, cfg, n, s, conf = "bf16net", "yolov3-tiny-12", 4, 416, 0.1
    net = Network(define_yolo(ph.cfg_path(cfg)), conf).eval()
    synth.fill_network_(net, name, cls0_bias=3.0, cls_bias=-4.0)  # class 0 dominant: Network keeps class-0 proposals
    net = net.cuda()
    x = torch.from_numpy(synth.uniform(name + "/x", (n, 3, s, s))).cuda()
    maps, rboxes = synth.radar_inputs(name + "/radar", n, s // 16)
    maps, rboxes = torch.from_numpy(maps).cuda(), torch.from_numpy(rboxes)
    with torch.no_grad():
        ref = net(x, maps, rboxes.clone().cuda(), 0).cpu()
        net.base_detector.compute_dtype = "bf16"
        got = net(x, maps, rboxes.clone().cuda(), 0).cpu()
        got2 = net(x, maps, rboxes.clone().cuda(), 0).cpu()
    assert torch.equal(got, got2)
    assert ref.shape[0] > 20, "the case must produce detections"
    assert abs(got.shape[0] - ref.shape[0]) <= 0.1 * ref.shape[0], (got.shape, ref.shape)
    matched = 0
    for row in ref:
        cand = got[got[:, 0] == row[0]]
        if cand.numel() == 0:
            continue
        d = (cand[:, 1:5] - row[1:5]).abs().max(dim=1).values
        k = int(d.argmin())
        if float(d[k]) <= 2.0 and abs(float(cand[k, 5] - row[5])) <= 0.03:
            matched += 1
    assert matched >= 0.9 * ref.shape[0], f"{matched} of {ref.shape[0]} fp32 rows have a bf16-mode counterpart"


def test_bf16_mode_is_inference_only(hip_lib):
    from tests import parity_helpers as ph
    model = ph.make_darknet("yolov3-tiny-12").cuda()
    model.compute_dtype = "bf16"
    x = ph.frames("bf16/loss", 1, 96).cuda()
    with torch.no_grad():  # the loss-value path keeps the raw maps: it runs on the fp32 engine whatever the mode
        loss, _fm, _y = model(x, torch.tensor([[0, 1, 0.5, 0.5, 0.2, 0.3]]))
    assert torch.isfinite(torch.as_tensor(loss))
    with pytest.raises(NotImplementedError):
        model.engine_for("bf16").run(x, keep_raw=True)


@pytest.mark.parametrize("dtype", ["f32", "bf16", "f16"])
def test_detector_determinism_stress(hip_lib, dtype):
    """40 forward passes of the same frames are bit-identical (Darknet-53, batch 1 and 4).  Regression test for an LDS
    slot-reuse race: the compiler sinks the tail of a stage's MFMAs and the waits of the ds_reads feeding them below the
    next barrier, so without `s_waitcnt lgkmcnt(0)` in front of the barrier another wave's refill DMA could overtake reads
    still in flight (10 % of batch-1 runs differed in the 16-bit modes; tools/determinism_stress.py)."""
    from tests import parity_helpers as ph
    for n in (1, 4):
        model = ph.make_darknet("yolov3").cuda()
        model.compute_dtype = dtype
        x = ph.frames(f"stress/{n}", n, 416).cuda()
        with torch.no_grad():
            fm0, y0 = model(x)
            for rep in range(40):
                fm, y = model(x)
                assert torch.equal(y, y0) and torch.equal(fm, fm0), f"{dtype} batch {n}: run {rep} differs"


@pytest.mark.parametrize("half", ["bf16", "f16"])
def test_conv_h16_random_shapes(hip_lib, half):
    """Seeded sweep over shapes the fixed cases do not hit: odd spatial sizes, M not a multiple of any tile, couts that are
    not multiples of 8 (scalar epilogue) next to ones that are (16-byte epilogue), ragged last tiles in both directions,
    stride 2, fused upsample, automatic tile / split choice, output written into a channel slice of a wider buffer."""
    from millieye_amd import hip
    half = HALVES[half]
    rng = np.random.RandomState(7)
    for case in range(14):
        n = int(rng.randint(1, 4))
        h, w = int(rng.randint(5, 40)), int(rng.randint(5, 40))
        cin = int(rng.choice([32, 64, 96, 160]))
        cout = int(rng.choice([8, 24, 33, 64, 100, 136, 255, 272]))
        k = int(rng.choice([1, 3]))
        s = int(rng.choice([1, 2])) if k == 3 else 1
        ups = 2 if (k == 1 and rng.rand() < 0.3) else 1
        act = int(rng.choice([0, 1]))
        pad = (k - 1) // 2
        ho, wo = (h + 2 * pad - k) // s + 1, (w + 2 * pad - k) // s + 1
        with_res = ups == 1 and rng.rand() < 0.5
        g = torch.Generator().manual_seed(1000 + case)
        x = _bf(torch.randn((n, h, w, cin), generator=g), half)
        wgt = _bf(torch.randn((cout, cin, k, k), generator=g) / (k * k * cin) ** 0.5, half)
        scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
        res = _bf(torch.randn((n, ho, wo, cout), generator=g), half) if with_res else None
        ref = _ref(x, wgt, scale, shift, k, s, pad, act, res, ups)
        packed = wgt.permute(0, 2, 3, 1).contiguous().cuda()
        y = hip.conv2d_h16(x.cuda(), packed, scale.cuda(), shift.cuda(), k, s, pad, act,
                           residual=res.cuda() if res is not None else None, upsample=ups)
        what = f"case {case}: n{n} {h}x{w} {cin}->{cout} k{k} s{s} ups{ups} res{int(with_res)}"
        _check_bf16(y, ref, what)
        if ups == 1 and cout % 8 == 0:   # the same layer writing into a slice of a wider (route) buffer: pitched 16-byte stores
            wide = torch.zeros((n, ho, wo, cout + 40), dtype=half).cuda()
            hip.conv2d_h16(x.cuda(), packed, scale.cuda(), shift.cuda(), k, s, pad, act,
                           residual=res.cuda() if res is not None else None, out=wide[..., 8:8 + cout])
            assert torch.equal(wide[..., 8:8 + cout], y), what + " (pitched)"
            assert float(wide[..., :8].abs().max()) == 0 and float(wide[..., 8 + cout:].abs().max()) == 0


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
def test_detector_16bit_608_batch16(hip_lib, dtype):
    """BASELINE configs[4] per-GPU shape: Darknet-53 at 608x608, batch 16 (128 frames over 8 GPUs), 16-bit MFMA
    convolutions.  Frames are independent units: frame f of the batch-16 run must equal the batch-1 run of that frame up
    to the storage error (tile / split-K choices differ with M, so accumulation order - hence a few roundings - may
    differ): both are compared with the fp32 HIP run of the same frames (itself pinned to the oracle at 608 in
    test_gpu_darknet.py), and the batch-16 rows may not be further from it than the batch-1 rows (x1.25 + 1e-4)."""
    from tests import parity_helpers as ph
    model = ph.make_darknet("yolov3").cuda()
    x = ph.frames("h16/608/b16", 16, 608).cuda()
    picks = (0, 7, 15)
    with torch.no_grad():
        model.compute_dtype = "f32"
        y32 = torch.cat([model(x[f:f + 1])[1] for f in picks]).cpu()
        model.compute_dtype = dtype
        fm16, y16 = model(x)
        fm16b, y16b = model(x)
        assert torch.equal(y16, y16b) and torch.equal(fm16, fm16b)
        assert tuple(y16.shape) == (16, 22743, 85) and tuple(fm16.shape) == (16, 256, 38, 38)
        y1 = torch.cat([model(x[f:f + 1])[1] for f in picks]).cpu()
    y16 = y16[list(picks)].cpu()
    assert bool(torch.isfinite(y16).all())
    e_batch, _ = _err(y16, y32)
    e_one, _ = _err(y1, y32)
    assert e_batch <= 1.25 * e_one + 1e-4, f"{dtype}: batch-16 error {e_batch:.2e} vs batch-1 error {e_one:.2e}"
    bound = 2e-2 if dtype == "bf16" else 3e-3   # measured format error of a 75-layer random-weight network x ~6
    assert e_batch <= bound, f"{dtype}: mean relative error vs fp32 {e_batch:.2e}"


P8_TILES_256 = (100, 110, 120, 200, 810, 820, 1210)
P8_TILES_128 = (101, 121, 131, 141, 201, 221, 301, 311, 321, 331, 421, 431, 441,
                621, 721, 731, 821, 831, 841, 1221, 1231)   # 6xx / 7xx: DMA duty split, 8xx: ping-pong halves (round 4)
P8_CASES = [
    # name, n, h, w, cin, cout, act, res
    ("13x13 two images per tile", 5, 13, 13, 64, 256, 1, True),
    ("26x26 residual", 3, 26, 26, 128, 256, 1, True),
    ("rectangular 9x31 linear", 2, 9, 31, 32, 128, 0, False),
    ("52x52 cout 512", 2, 52, 52, 96, 512, 1, True),
    ("one tiny image", 1, 5, 7, 64, 128, 1, False),
    ("104x104 wide patch", 1, 104, 104, 32, 128, 1, True),
]


@pytest.mark.parametrize("half", ["bf16", "f16"])
@pytest.mark.parametrize("case", P8_CASES, ids=[c[0] for c in P8_CASES])
def test_conv_p8_patch_resident_tiles(hip_lib, case, half):
    """The patch-resident big-tile generation (csrc/conv_p8_h16.hip, tile ids >= 100): tiles are BM consecutive positions
    of a padded-linear index space, so they start anywhere and cross rows and images; every tile shape against the fp32 CPU
    convolution under the same bar as the per-tap tiles (equal to the RNE rounding of the fp32 result up to rounding flips),
    also when the output is a channel slice of a wider buffer, and run-to-run deterministic."""
    from millieye_amd import hip
    half = HALVES[half]
    name, n, h, w, cin, cout, act, with_res = case
    g = torch.Generator().manual_seed(len(name) * 11 + cin)
    x = _bf(torch.randn((n, h, w, cin), generator=g), half)
    wgt = _bf(torch.randn((cout, cin, 3, 3), generator=g) / (9 * cin) ** 0.5, half)
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.1
    res = _bf(torch.randn((n, h, w, cout), generator=g), half) if with_res else None
    ref = _ref(x, wgt, scale, shift, 3, 1, 1, act, res, 1)
    packed = wgt.permute(0, 2, 3, 1).contiguous().cuda()
    xs, sc, sh = x.cuda(), scale.cuda(), shift.cuda()
    rs = res.cuda() if res is not None else None
    tiles = []
    for tile in P8_TILES_128 + (P8_TILES_256 if cout % 256 == 0 else ()):
        try:
            y = hip.conv2d_h16(xs, packed, sc, sh, 3, 1, 1, act, residual=rs, tile=tile, split_k=1)
        except hip.MeError as exc:  # a wide map's patch does not fit the LDS budget of this tile: refused, never wrong
            assert "LDS" in str(exc) or "does not fit" in str(exc), str(exc)
            continue
        tiles.append(tile)
        _check_bf16(y, ref, f"{name} tile {tile}")
        y2 = hip.conv2d_h16(xs, packed, sc, sh, 3, 1, 1, act, residual=rs, tile=tile, split_k=1)
        assert torch.equal(y, y2), f"{name} tile {tile}: not deterministic"
    assert len(tiles) >= 4, tiles
    wide = torch.zeros((n, h, w, cout + 48), dtype=half).cuda()
    hip.conv2d_h16(xs, packed, sc, sh, 3, 1, 1, act, residual=rs, out=wide[..., 16:16 + cout], tile=tiles[-1], split_k=1)
    assert torch.equal(wide[..., 16:16 + cout], y) and float(wide[..., :16].abs().max()) == 0 \
        and float(wide[..., 16 + cout:].abs().max()) == 0, f"{name}: pitched output"
    # input read from a channel slice of a wider tensor (route buffers)
    xw = torch.zeros((n, h, w, cin + 32), dtype=half).cuda()
    xw[..., 32:] = xs
    y3 = hip.conv2d_h16(xw[..., 32:], packed, sc, sh, 3, 1, 1, act, residual=rs, tile=tiles[0], split_k=1)
    y0 = hip.conv2d_h16(xs, packed, sc, sh, 3, 1, 1, act, residual=rs, tile=tiles[0], split_k=1)
    assert torch.equal(y3, y0), f"{name}: pitched input"


@pytest.mark.parametrize("half", ["bf16", "f16"])
@pytest.mark.parametrize("tile", [121, 221, 201, 100, 131, 431, 621, 731])
def test_conv_p8_k_split(hip_lib, tile, half):
    """Patch tiles with split_k > 1: every tile is cut along the 32-channel chunks into split_k workgroups (compact fp32
    slabs), conv3x3_p8_reduce_h16 sums them in a fixed order and applies the epilogue - same bar against the fp32 CPU
    convolution as the one-pass tiles, residual and pad positions through the second pass, uneven chunk counts (5 chunks cut
    2 / 3 / 4 ways), more splits than chunks (clamped), deterministic."""
    from millieye_amd import hip
    half = HALVES[half]
    for name, n, h, w, cin, cout, with_res, splits in (("a", 3, 13, 13, 160, 256, True, (2, 3, 4)),
                                                        ("b", 2, 26, 26, 64, 512, False, (2, 5)),
                                                        ("c", 1, 9, 7, 96, 256, True, (3,))):
        g = torch.Generator().manual_seed(tile * 7 + cin)
        x = _bf(torch.randn((n, h, w, cin), generator=g), half)
        wgt = _bf(torch.randn((cout, cin, 3, 3), generator=g) / (9 * cin) ** 0.5, half)
        scale = torch.rand(cout, generator=g) + 0.5
        shift = torch.randn(cout, generator=g) * 0.1
        res = _bf(torch.randn((n, h, w, cout), generator=g), half) if with_res else None
        ref = _ref(x, wgt, scale, shift, 3, 1, 1, 1, res, 1)
        packed = wgt.permute(0, 2, 3, 1).contiguous().cuda()
        xs, sc, sh = x.cuda(), scale.cuda(), shift.cuda()
        rs = res.cuda() if res is not None else None
        for split in splits:
            y = hip.conv2d_h16(xs, packed, sc, sh, 3, 1, 1, 1, residual=rs, tile=tile, split_k=split)
            _check_bf16(y, ref, f"{name} tile {tile} split {split}")
            y2 = hip.conv2d_h16(xs, packed, sc, sh, 3, 1, 1, 1, residual=rs, tile=tile, split_k=split)
            assert torch.equal(y, y2), f"{name} tile {tile} split {split}: not deterministic"
        wide = torch.zeros((n, h, w, cout + 48), dtype=half).cuda()
        hip.conv2d_h16(xs, packed, sc, sh, 3, 1, 1, 1, residual=rs, out=wide[..., 16:16 + cout], tile=tile, split_k=splits[0])
        y0 = hip.conv2d_h16(xs, packed, sc, sh, 3, 1, 1, 1, residual=rs, tile=tile, split_k=splits[0])
        assert torch.equal(wide[..., 16:16 + cout], y0) and float(wide[..., :16].abs().max()) == 0 \
            and float(wide[..., 16 + cout:].abs().max()) == 0, f"{name}: pitched output through the reduce pass"


WS_PAIRS = [(64, 32), (128, 64), (128, 128), (256, 128), (256, 256), (384, 128), (512, 256), (512, 512), (768, 256)]


@pytest.mark.parametrize("half", ["bf16", "f16"])
@pytest.mark.parametrize("cin,cout", WS_PAIRS)
def test_conv1x1_weight_stationary_tile(hip_lib, cin, cout, half):
    """Tile id 50 (csrc/conv1x1_ws_h16.hip): the streaming 1x1 kernel - weights in registers, activations through an LDS ring
    filled by swizzled-source LDS-DMA, persistent grid - on every built (cin, cout) pair: against the fp32 CPU convolution
    under the bar of the other 16-bit tiles; row counts that are not a multiple of the tile (1, 31, 33 rows and a few thousand:
    zero fill by the descriptor range, guarded stores), more tiles than workgroups (the ring wraps), input read from a channel
    slice of a wider buffer, output written into a slice, leaky and linear, bit-identical to the per-tap kernel's fp32-exact
    cases being out of scope (accumulation order differs), deterministic."""
    from millieye_amd import hip
    half = HALVES[half]
    g = torch.Generator().manual_seed(cin * 7 + cout)
    wgt = _bf(torch.randn((cout, cin, 1, 1), generator=g) / cin ** 0.5, half)
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    packed = wgt.permute(0, 2, 3, 1).contiguous().cuda()
    sc, sh = scale.cuda(), shift.cuda()
    shapes = [(1, 1, 1), (1, 1, 31), (1, 3, 11), (2, 13, 13), (3, 52, 52)]
    if cin <= 128:
        shapes.append((2, 104, 208))   # 43 264 rows: several tiles per workgroup of the 256-workgroup grid
    for case, (n, h, w) in enumerate(shapes):
        act = case % 2
        x = _bf(torch.randn((n, h, w, cin), generator=g), half)
        ref = _ref(x, wgt, scale, shift, 1, 1, 0, act)
        xs = x.cuda()
        y = hip.conv2d_h16(xs, packed, sc, sh, 1, 1, 0, act, tile=50, split_k=1)
        _check_bf16(y, ref, f"{cin}->{cout} {n}x{h}x{w} act {act} tile 50")
        assert torch.equal(y, hip.conv2d_h16(xs, packed, sc, sh, 1, 1, 0, act, tile=50, split_k=1)), "not deterministic"
        if case in (3, 4):
            xw = torch.zeros((n, h, w, cin + 40), dtype=half).cuda()
            xw[..., 8:8 + cin] = xs
            wide = torch.zeros((n, h, w, cout + 48), dtype=half).cuda()
            hip.conv2d_h16(xw[..., 8:8 + cin], packed, sc, sh, 1, 1, 0, act, out=wide[..., 16:16 + cout], tile=50, split_k=1)
            assert torch.equal(wide[..., 16:16 + cout], y), "pitched input / output"
            assert float(wide[..., :16].abs().max()) == 0 and float(wide[..., 16 + cout:].abs().max()) == 0


WS3_CASES = [
    # cin, cout, stride, n, h, w, res
    (32, 64, 1, 2, 16, 16, True), (32, 64, 1, 1, 37, 21, False), (32, 64, 1, 3, 48, 40, True),
    (32, 64, 2, 2, 32, 32, False), (32, 64, 2, 1, 45, 27, False), (32, 64, 2, 2, 64, 96, False),
    (64, 128, 1, 2, 8, 16, True), (64, 128, 1, 1, 29, 19, True), (64, 128, 1, 2, 40, 56, False),
    (64, 128, 2, 2, 16, 32, False), (64, 128, 2, 1, 37, 53, False), (64, 128, 2, 2, 48, 80, False),
]


@pytest.mark.parametrize("half", ["bf16", "f16"])
@pytest.mark.parametrize("cin,cout,stride,n,h,w,with_res", WS3_CASES)
def test_conv3x3_weight_stationary_tile(hip_lib, cin, cout, stride, n, h, w, with_res, half):
    """Tile id 60 (csrc/conv3x3_ws_h16.hip): 3x3 layers with 32 / 64 input channels - the whole filter in registers, 2-D input
    patches through an LDS ring, zero padding and ragged tiles by out-of-range DMA lanes - stride 1 and 2, with and without the
    fused shortcut, odd image sizes (tiles that hang over every border), more tiles than workgroups, pitched input / output /
    residual, against the fp32 CPU convolution under the bar of the other 16-bit tiles; deterministic."""
    from millieye_amd import hip
    half = HALVES[half]
    g = torch.Generator().manual_seed(cin + 3 * h + w + stride)
    x = _bf(torch.randn((n, h, w, cin), generator=g), half)
    wgt = _bf(torch.randn((cout, cin, 3, 3), generator=g) / (9 * cin) ** 0.5, half)
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    ho, wo = (h + 2 - 3) // stride + 1, (w + 2 - 3) // stride + 1
    res = _bf(torch.randn((n, ho, wo, cout), generator=g), half) if with_res else None
    act = 1 if (h + w) % 2 == 0 else 0
    ref = _ref(x, wgt, scale, shift, 3, stride, 1, act, res, 1)
    packed = wgt.permute(0, 2, 3, 1).contiguous().cuda()
    xs, sc, sh = x.cuda(), scale.cuda(), shift.cuda()
    rs = res.cuda() if res is not None else None
    y = hip.conv2d_h16(xs, packed, sc, sh, 3, stride, 1, act, residual=rs, tile=60, split_k=1)
    _check_bf16(y, ref, f"{cin}->{cout} s{stride} {n}x{h}x{w} res{int(with_res)} tile 60")
    assert torch.equal(y, hip.conv2d_h16(xs, packed, sc, sh, 3, stride, 1, act, residual=rs, tile=60, split_k=1))
    xw = torch.zeros((n, h, w, cin + 24), dtype=half).cuda()
    xw[..., 16:16 + cin] = xs
    wide = torch.zeros((n, ho, wo, cout + 48), dtype=half).cuda()
    rw = None
    if rs is not None:
        rw = torch.zeros((n, ho, wo, cout + 8), dtype=half).cuda()
        rw[..., 8:] = rs
    hip.conv2d_h16(xw[..., 16:16 + cin], packed, sc, sh, 3, stride, 1, act, residual=rw[..., 8:] if rw is not None else None,
                   out=wide[..., 16:16 + cout], tile=60, split_k=1)
    assert torch.equal(wide[..., 16:16 + cout], y), "pitched input / residual / output"
    assert float(wide[..., :16].abs().max()) == 0 and float(wide[..., 16 + cout:].abs().max()) == 0


def test_conv1x1_weight_stationary_refuses(hip_lib):
    from millieye_amd import hip
    x = torch.zeros((1, 8, 8, 256), dtype=torch.bfloat16).cuda()
    w = torch.zeros((128, 1, 1, 256), dtype=torch.bfloat16).cuda()
    one = torch.ones(128).cuda()
    res = torch.zeros((1, 8, 8, 128), dtype=torch.bfloat16).cuda()
    with pytest.raises(hip.MeError):
        hip.conv2d_h16(x, w, one, one, 1, 1, 0, 1, residual=res, tile=50)       # residual
    with pytest.raises(hip.MeError):
        hip.conv2d_h16(x, w, one, one, 1, 1, 0, 1, y_f32=True, tile=50)         # fp32 output
    with pytest.raises(hip.MeError):
        hip.conv2d_h16(x, w[:96], one[:96], one[:96], 1, 1, 0, 1, tile=50)      # no instance for 256 -> 96


def test_conv_p8_refuses_what_it_cannot_do(hip_lib):
    from millieye_amd import hip
    x = torch.zeros((1, 8, 8, 32), dtype=torch.bfloat16).cuda()
    w1 = torch.zeros((128, 1, 1, 32), dtype=torch.bfloat16).cuda()
    w3 = torch.zeros((72, 3, 3, 32), dtype=torch.bfloat16).cuda()
    one = torch.ones(128).cuda()
    with pytest.raises(hip.MeError):
        hip.conv2d_h16(x, w1, one, one, 1, 1, 0, 1, tile=121)          # 1x1
    with pytest.raises(hip.MeError):
        hip.conv2d_h16(x, w3, one[:72], one[:72], 3, 1, 1, 1, tile=121)  # cout % 128 != 0


BNECK_CASES = [
    # name, n, h, w, cin, cmid, cout, tiles, with_res, act2
    ("52x52 block", 3, 52, 52, 256, 128, 256, (1,), True, 1),
    ("52x52 head pair (no residual, 384 in)", 2, 52, 52, 384, 128, 256, (1,), False, 1),
    ("one small image, ragged rows", 1, 7, 11, 64, 128, 256, (1,), True, 0),
    ("many tiny images (tiles cross several)", 9, 5, 6, 32, 128, 256, (1,), True, 1),
    ("104x104 block", 2, 104, 104, 128, 64, 128, (3, 4), True, 1),
    ("rectangular 20x61", 2, 20, 61, 96, 64, 128, (3, 4), True, 1),
]


@pytest.mark.parametrize("half", ["bf16", "f16"])
@pytest.mark.parametrize("case", BNECK_CASES, ids=[c[0] for c in BNECK_CASES])
def test_bottleneck_one_launch_equals_the_two_launches(hip_lib, case, half):
    """``me_bneck_h16`` (csrc/bneck_h16.hip: 1x1 -> 3x3 (+ shortcut) in one launch, the mid tensor rounded to the storage
    type but kept in LDS) against (a) the two ``me_conv2d_h16`` launches it replaces - same rounding points and, on the tiles
    chosen below, the same summation order: the results must be EQUAL bit for bit - and (b) the fp32 CPU convolutions on the same rounded operands with the mid tensor rounded once.  Channel-slice
    input / output (route buffers), determinism, and the pad rows of the padded-linear space (the 3x3 must see ZEROS there,
    not leaky(bn1(0)): shift1 is far from zero in these cases)."""
    from millieye_amd import hip
    half = HALVES[half]
    name, n, h, w, cin, cmid, cout, tiles, with_res, act2 = case
    g = torch.Generator().manual_seed(len(name) * 7 + cin)
    x = _bf(torch.randn((n, h, w, cin), generator=g), half)
    w1 = _bf(torch.randn((cmid, cin, 1, 1), generator=g) / cin ** 0.5, half)
    w2 = _bf(torch.randn((cout, cmid, 3, 3), generator=g) / (9 * cmid) ** 0.5, half)
    s1, t1 = torch.rand(cmid, generator=g) + 0.5, torch.randn(cmid, generator=g) * 0.5 + 0.7
    s2, t2 = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    res = x if (with_res and cin == cout) else (_bf(torch.randn((n, h, w, cout), generator=g), half) if with_res else None)
    mid_ref = _ref(x, w1, s1, t1, 1, 1, 0, 1).to(half)
    ref = _ref(mid_ref, w2, s2, t2, 3, 1, 1, act2, res, 1)
    p1 = w1.permute(0, 2, 3, 1).contiguous().cuda()
    p2 = w2.permute(0, 2, 3, 1).contiguous().cuda()
    xs = x.cuda()
    rs = None if res is None else (xs if res is x else res.cuda())
    c = [t.cuda() for t in (s1, t1, s2, t2)]
    # the pair on a per-tap 1x1 tile and a patch-resident 3x3 tile: both walk K chunk by chunk (3x3: chunk-major, taps inside),
    # the order the one-launch kernel uses - the results must then be EQUAL (tools/bneck_equal.py: 0 of 2.8 M elements differ;
    # against the tap-major per-tap 3x3 tiles or the weight-stationary 1x1 tile 0.007 - 0.3 % flip by one ulp, as those do
    # among themselves)
    mid = hip.conv2d_h16(xs, p1, c[0], c[1], 1, 1, 0, 1, tile=1, split_k=1)
    two = hip.conv2d_h16(mid, p2, c[2], c[3], 3, 1, 1, act2, residual=rs, tile=131, split_k=1)
    def close(got, what):
        # a composite of two stored layers: a rounding flip of a mid element (one ulp of a value of order 1, a few per dot
        # product) moves an output by ~1e-4 whatever the output's own size, so the bar is one output ulp PLUS that absolute term
        got, want = got.float().cpu(), ref.to(half).float()
        tol = torch.maximum(want.abs(), ref.abs()) * ULP[half] + 2e-3
        bad = (got - want).abs() > tol
        assert not bool(bad.any()), f"{what}: {int(bad.sum())} elements off, max {float(((got - want).abs() / tol).max()):.2f} x the bar"
    close(two, f"{name}: two launches")
    for tile in tiles:
        y = hip.bneck_h16(xs, p1, c[0], c[1], p2, c[2], c[3], residual=rs, tile=tile, act2=act2)
        close(y, f"{name}: one launch, tile {tile}")
        assert torch.equal(y, two), f"{name} tile {tile}: {int((y != two).sum())} elements differ from the two-launch result"
        y2 = hip.bneck_h16(xs, p1, c[0], c[1], p2, c[2], c[3], residual=rs, tile=tile, act2=act2)
        assert torch.equal(y, y2), f"{name} tile {tile}: not deterministic"
    # channel slices on both sides
    xw = torch.zeros((n, h, w, cin + 32), dtype=half).cuda()
    xw[..., 32:] = xs
    wide = torch.zeros((n, h, w, cout + 48), dtype=half).cuda()
    r2 = None if rs is None else (xw[..., 32:] if res is x else rs)
    hip.bneck_h16(xw[..., 32:], p1, c[0], c[1], p2, c[2], c[3], residual=r2, out=wide[..., 16:16 + cout], tile=tiles[0], act2=act2)
    y0 = hip.bneck_h16(xs, p1, c[0], c[1], p2, c[2], c[3], residual=rs, tile=tiles[0], act2=act2)
    assert torch.equal(wide[..., 16:16 + cout], y0) and float(wide[..., :16].abs().max()) == 0 \
        and float(wide[..., 16 + cout:].abs().max()) == 0, f"{name}: pitched operands"


def test_bottleneck_one_launch_refuses_what_it_cannot_do(hip_lib):
    from millieye_amd import hip
    x = torch.zeros((1, 26, 26, 512), dtype=torch.bfloat16).cuda()
    w1 = torch.zeros((256, 1, 1, 512), dtype=torch.bfloat16).cuda()
    w2 = torch.zeros((512, 3, 3, 256), dtype=torch.bfloat16).cuda()
    f = [torch.zeros(k).cuda() for k in (256, 256, 512, 512)]
    with pytest.raises(hip.MeError, match="no instance"):   # cmid 256 does not fit the LDS with its patch
        hip.bneck_h16(x, w1, f[0], f[1], w2, f[2], f[3], tile=1)
    x = torch.zeros((1, 8, 80, 256), dtype=torch.bfloat16).cuda()   # 80 wide: 192 + 2 * 82 rows > 320
    w1 = torch.zeros((128, 1, 1, 256), dtype=torch.bfloat16).cuda()
    w2 = torch.zeros((256, 3, 3, 128), dtype=torch.bfloat16).cuda()
    f = [torch.zeros(k).cuda() for k in (128, 128, 256, 256)]
    with pytest.raises(hip.MeError, match="no instance"):
        hip.bneck_h16(x, w1, f[0], f[1], w2, f[2], f[3], tile=1)


KW_CASES = CASES + [
    ("detection conv fp32 out, 255 channels", 1, 13, 13, 1024, 255, 1, 1, 0, False, 1),
    ("13x13 deep 3x3", 1, 13, 13, 512, 1024, 3, 1, 1, True, 1),
    ("52x52 1x1", 1, 52, 52, 256, 128, 1, 1, 1, False, 1),
    ("K steps not a multiple of 8 (two idle waves)", 2, 9, 7, 32, 40, 3, 1, 1, False, 1),
]


@pytest.mark.parametrize("half", ["bf16", "f16"])
@pytest.mark.parametrize("tile", [40, 41])
@pytest.mark.parametrize("case", KW_CASES, ids=[c[0] for c in KW_CASES])
def test_conv_small_batch_tiles_k_split_over_the_waves(hip_lib, case, tile, half):
    """Tile ids 40 / 41 (csrc/conv_kw_h16.hip): one workgroup per 32 x 32 / 32 x 64 output tile, its eight waves take an eighth
    of the K steps each, the partial tiles are added in LDS in wave order and go through the fused epilogue - the one-launch
    form of a split-K layer (no slabs, no reduce launch).  Every filter size / stride, residual, 2x upsample, ragged output
    channels, the fp32 detection output; against the fp32 CPU convolution under the bar of the other tiles, deterministic."""
    from millieye_amd import hip
    half = HALVES[half]
    name, n, h, w, cin, cout, k, s, act, with_res, ups = case
    y_f32 = "fp32 out" in name
    g = torch.Generator().manual_seed(len(name) * 13 + cin + tile)
    pad = (k - 1) // 2
    x = _bf(torch.randn((n, h, w, cin), generator=g), half)
    wgt = _bf(torch.randn((cout, cin, k, k), generator=g) / (k * k * cin) ** 0.5, half)
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.1
    ho = (h + 2 * pad - k) // s + 1
    res = _bf(torch.randn((n, ho, ho if h == w else (w + 2 * pad - k) // s + 1, cout), generator=g), half) if with_res else None
    ref = _ref(x, wgt, scale, shift, k, s, pad, act, res, ups)
    packed = wgt.permute(0, 2, 3, 1).contiguous().cuda()
    y = hip.conv2d_h16(x.cuda(), packed, scale.cuda(), shift.cuda(), k, s, pad, act, residual=res.cuda() if res is not None else None,
                       upsample=ups, y_f32=y_f32, tile=tile, split_k=1)
    if y_f32:
        assert y.dtype == torch.float32
        err = float((y.cpu() - ref).abs().max() / ref.abs().max())
        assert err <= 1e-3, f"{name} tile {tile}: fp32 output {err:.2e}"
    else:
        _check_bf16(y, ref, f"{name} tile {tile}")
    y2 = hip.conv2d_h16(x.cuda(), packed, scale.cuda(), shift.cuda(), k, s, pad, act, residual=res.cuda() if res is not None else None,
                        upsample=ups, y_f32=y_f32, tile=tile, split_k=1)
    assert torch.equal(y, y2), "not deterministic"


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_detector_plan_with_one_launch_bottlenecks(hip_lib, monkeypatch, dtype):
    """The engine's plan with every eligible 1x1 -> 3x3 (+ shortcut) pair of Darknet-53 replaced by ``me_bneck_h16``
    (``MILLIEYE_BNECK=force``: the 52 x 52 and 104 x 104 residual blocks and the 52 x 52 head pairs) against the plan that keeps
    the launch pairs (``MILLIEYE_BNECK=0``): the same decoded rows up to the one-ulp flips of another summation order in the
    layers whose tuned pair tiles walk K tap-major (measured: objectness |d| <= 0.01, boxes <= 0.5 px; bit-equal when the tuner
    picked chunk-major tiles), the same feature tap, deterministic, and the default mode (measured choice) equals one of the two."""
    from tests import parity_helpers as ph
    x = ph.frames(f"bneckplan/{dtype}", 2, 416).cuda()
    outs = {}
    for mode in ("0", "force"):
        monkeypatch.setenv("MILLIEYE_BNECK", mode)
        model = ph.make_darknet("yolov3", tag="bneckplan", trained_like=True).cuda()
        model.compute_dtype = dtype
        with torch.no_grad():
            fm, y = model(x)
            fm2, y2 = model(x)
        assert torch.equal(y, y2) and torch.equal(fm, fm2)
        plan = model.engine_for(dtype).plan_for(x)
        outs[mode] = (fm, y, list(plan.fused_blocks), len(plan.launches))
    assert outs["0"][2] == [] and len(outs["force"][2]) >= 10, outs["force"][2]
    assert outs["force"][3] == outs["0"][3] - len(outs["force"][2])   # one launch less per block
    a, b = outs["0"][1].float(), outs["force"][1].float()
    over = (a[..., 4] >= 0.2) | (b[..., 4] >= 0.2)
    d_obj = float((a[..., 4] - b[..., 4]).abs().max())
    d_box = float(((a[over][:, :4] - b[over][:, :4]).abs() / a[over][:, 2:4].abs().clamp_min(16.0).repeat(1, 2)).max()) if bool(over.any()) else 0.0
    # ... and neither form is further from the fp32 run than the other: the storage format, not the launch structure, sets the error
    monkeypatch.setenv("MILLIEYE_BNECK", "0")
    model = ph.make_darknet("yolov3", tag="bneckplan", trained_like=True).cuda()
    with torch.no_grad():
        fm32, y32 = model(x)
    e_pair, _ = _err(outs["0"][1].cpu(), y32.cpu())
    e_one, _ = _err(outs["force"][1].cpu(), y32.cpu())
    print(f"[bneck plan {dtype}] {len(outs['force'][2])} blocks in one launch each; pair vs one launch: objectness |d| {d_obj:.4f}, boxes over the "
          f"threshold {d_box:.2%} of their size; mean error vs fp32: pair {e_pair:.3e}, one launch {e_one:.3e}")
    # (the direct difference moves with the tiles the tuner picks for the pairs on this box - measured 2.7 % / 0.35 % of a box's size; the
    #  bar that does not move is the one against fp32)
    assert d_obj <= (0.03 if dtype == "bf16" else 0.005) and d_box <= (0.06 if dtype == "bf16" else 0.01)
    assert e_one <= 1.25 * e_pair + 1e-4


@pytest.mark.parametrize("half", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("cin,cout,h", [(64, 128, 26), (128, 256, 13), (256, 64, 10)])
def test_conv_h16_tap_masks_skip_the_zero_taps_bit_exactly(hip_lib, cin, cout, h, half):
    """ABI 13, ``me_conv16_desc.tap_mask``: the 2x2 parity convolution of the 16-bit step's stride-2 data gradient
    (detector_train._parity_weights rounded to the storage type: 7 of its 16 (class, tap) pairs are structurally zero) with the zero
    taps SKIPPED by the masked tile instances equals the same tile multiplying them - bit for bit (a skipped tap would have added
    +0.0) - on every per-tap tile whose width divides the class width; the pixel shuffle of the result is the transposed convolution;
    what the masked instances cannot do is refused."""
    from millieye_amd import hip, synth
    from millieye_amd.detector_train import _PARITY_TAP_MASKS, _parity_weights
    n = 3
    dev = torch.device("cuda")
    w = torch.from_numpy(synth.uniform(f"tm16/w{cin}", (cout, 3, 3, cin), -1, 1)).to(dev) / (9 * cin) ** 0.5   # forward OHWI
    dc = torch.from_numpy(synth.uniform(f"tm16/dc{cin}", (n, h, h, cout), -1, 1)).to(dev).to(half)
    pw = _parity_weights(w).to(half)
    ones, zeros = torch.ones(4 * cin, device=dev), torch.zeros(4 * cin, device=dev)
    masks = (cin, _PARITY_TAP_MASKS)
    seen = 0
    for tile in (1, 2, 3, 11, 12, 13):
        width = 128 if tile in (1, 11) else 64
        if (tile > 10 and cout % 64 != 0) or cin % width != 0:
            with pytest.raises(hip.MeError):
                hip.conv2d_h16(dc, pw, ones, zeros, 2, 1, 1, hip.ACT_LINEAR, tile=tile, split_k=1, tap_masks=masks)
            continue
        base = hip.conv2d_h16(dc, pw, ones, zeros, 2, 1, 1, hip.ACT_LINEAR, tile=tile, split_k=1)
        got = hip.conv2d_h16(dc, pw, ones, zeros, 2, 1, 1, hip.ACT_LINEAR, tile=tile, split_k=1, tap_masks=masks)
        assert torch.equal(got, base), f"tile {tile}: skipping the zero taps changed bits"
        seen += 1
    assert seen >= 2
    dx = got[:, 1:, 1:, :].float().reshape(n, h, h, 2, 2, cin).permute(0, 1, 3, 2, 4, 5).reshape(n, 2 * h, 2 * h, cin)
    ref = F.conv_transpose2d(dc.float().permute(0, 3, 1, 2).cpu(), w.to(half).float().permute(0, 3, 1, 2).cpu(), stride=2, padding=1,
                             output_padding=1)
    err = float((dx.permute(0, 3, 1, 2).cpu() - ref).abs().max()) / float(ref.abs().max())
    assert err < (1e-2 if half == torch.bfloat16 else 2e-3), err   # (one rounding of the output to the storage type)
    with pytest.raises(hip.MeError):
        hip.conv2d_h16(dc, pw, ones, zeros, 2, 1, 1, hip.ACT_LINEAR, tile=3, split_k=1, tap_masks=(cin, (1, 3, 0, 15)))   # a class without taps
    with pytest.raises(hip.MeError):
        hip.conv2d_h16(dc, pw, ones, zeros, 2, 1, 1, hip.ACT_LINEAR, tile=3, split_k=2, tap_masks=masks)                  # masks + K split
    with pytest.raises(hip.MeError):
        hip.conv2d_h16(dc, pw, ones, zeros, 2, 1, 1, hip.ACT_LINEAR, tile=4, split_k=1, tap_masks=masks)                  # a tile without a masked instance
    with pytest.raises(hip.MeError):
        hip.conv2d_h16(dc, pw, ones, zeros, 2, 1, 1, hip.ACT_LINEAR, tile=0, tap_masks=masks)                             # needs an explicit tile
