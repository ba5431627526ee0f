"""-m gpu: per-kernel parity of the detector ops, through the C ABI, against stock torch CPU
fp32 ops (the same library calls the reference's CPU path makes).  Tolerance: 1e-3
(allclose(rtol=atol=1e-3), BASELINE.json north_star); observed errors are ~1e-6."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from millieye_amd import synth
from tests.parity_helpers import assert_close

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _t(tag, shape, lo=-1.0, hi=1.0):
    return torch.from_numpy(synth.uniform(tag, shape, lo, hi))


def _conv_case(hip, tag, n, h, w, cin, cout, k, s, act, residual=False, ups=1, tile=0, nchw_in=False, split_k=0,
               x_slice=0):
    x = _t(tag + "x", (n, cin, h, w))
    wgt = torch.from_numpy(synth.normal(tag + "w", (cout, cin, k, k), 0, (2.0 / (cin * k * k)) ** 0.5))
    scale = _t(tag + "s", (cout,), 0.5, 1.5)
    shift = _t(tag + "b", (cout,), -0.5, 0.5)
    pad = (k - 1) // 2
    ref = F.conv2d(x, wgt, None, s, pad) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    if act == 1:
        ref = F.leaky_relu(ref, 0.1)
    elif act == 2:
        ref = torch.sigmoid(ref)
    res = None
    if residual:
        res = _t(tag + "r", tuple(ref.shape))
        ref = ref + res
    if ups == 2:
        ref = F.interpolate(ref, scale_factor=2, mode="nearest")
    dev = "cuda"
    xd = x.to(dev) if nchw_in else x.permute(0, 2, 3, 1).contiguous().to(dev)
    if x_slice:  # the input is channels [x_slice, x_slice+cin) of a wider NHWC buffer (pitch > cin)
        wide = torch.full((n, h, w, cin + 2 * x_slice), 7.0, device=dev)
        wide[..., x_slice:x_slice + cin] = xd
        xd = wide[..., x_slice:x_slice + cin]
    got = hip.conv2d(xd, hip.pack_conv_weight(wgt).to(dev), scale.to(dev), shift.to(dev), k, s, pad, act,
                     residual=res.permute(0, 2, 3, 1).contiguous().to(dev) if res is not None else None,
                     upsample=ups, x_nchw=nchw_in, tile=tile, split_k=split_k)
    torch.cuda.synchronize()
    return assert_close(got.cpu().permute(0, 3, 1, 2), ref, TOL, tag)


@pytest.mark.parametrize("tile", [1, 2, 3, 4, 5, 7])
def test_conv_buffer_addressed_kernel(hip_lib, tile):
    """conv_igemm_buf_f32 (buffer_load ... lds with range-check zero padding): shapes that stress its addressing -
    tiles spanning 2-3 images, borders on every side, channel-slice inputs (pitch > cin), cin = 16 (a new filter tap
    every K stage), stride 2, 1x1, split-K starting mid-tap, residual + upsample epilogues."""
    from millieye_amd import hip
    _conv_case(hip, f"bk{tile}a", 5, 13, 13, 32, 96, 3, 1, 1, tile=tile)                        # 169 px / image
    _conv_case(hip, f"bk{tile}b", 3, 7, 9, 16, 40, 3, 1, 1, tile=tile, x_slice=16)              # tiny maps, cs = 1
    _conv_case(hip, f"bk{tile}c", 2, 26, 26, 64, 128, 3, 2, 1, tile=tile, x_slice=32)           # stride 2 + slice
    _conv_case(hip, f"bk{tile}d", 2, 13, 13, 128, 255, 1, 1, 0, tile=tile)                      # 1x1, ragged cout
    _conv_case(hip, f"bk{tile}e", 2, 13, 13, 48, 64, 3, 1, 1, residual=True, split_k=4, tile=tile)
    _conv_case(hip, f"bk{tile}f", 1, 13, 13, 64, 48, 3, 1, 1, ups=2, split_k=5, tile=tile, x_slice=64)
    _conv_case(hip, f"bk{tile}g", 1, 40, 24, 32, 32, 3, 1, 2, tile=tile)                        # sigmoid, h != w


@pytest.mark.parametrize("tile", [1, 2, 3, 4, 5, 21, 22, 23, 24, 25, 51, 52, 53, 54])
def test_conv3x3_every_tile(hip_lib, tile):
    from millieye_amd import hip
    # ragged M (n*h*w = 2*13*11 = 286) and ragged cout (not a tile multiple), cin not a BK multiple
    _conv_case(hip, f"c3t{tile}", 2, 13, 11, 24, 72, 3, 1, 1, tile=tile)
    _conv_case(hip, f"c3t{tile}b", 1, 26, 26, 64, 160, 3, 1, 1, tile=tile)
    _conv_case(hip, f"c3t{tile}c", 2, 16, 16, 32, 64, 3, 2, 1, tile=tile, residual=False)   # stride 2
    _conv_case(hip, f"c3t{tile}d", 1, 13, 13, 40, 255, 1, 1, 0, tile=tile)                  # 1x1, one K stage + ragged
    _conv_case(hip, f"c3t{tile}e", 2, 13, 13, 64, 96, 3, 1, 1, residual=True, split_k=3, tile=tile)


@pytest.mark.parametrize("tile", [41, 42, 43, 44, 45, 47])
def test_conv_tail_split_tiles(hip_lib, tile):
    """Tile ids 41-45 = tiles 1-5 with the last partial round of tiles cut split_k ways along K (compact slabs +
    conv_tail_reduce_f32): fewer than 256 tiles (everything is tail), more than 256 (whole tiles + tail pieces in one
    launch), ragged M / cout, 1x1, stride 2, channel-slice input, residual and upsample through the second pass,
    run-to-run determinism."""
    from millieye_amd import hip
    _conv_case(hip, f"ts{tile}a", 2, 13, 11, 32, 72, 3, 1, 1, tile=tile, split_k=3)
    _conv_case(hip, f"ts{tile}b", 1, 13, 13, 48, 255, 1, 1, 0, tile=tile, split_k=2)
    _conv_case(hip, f"ts{tile}c", 2, 26, 26, 64, 128, 3, 2, 1, tile=tile, x_slice=32, split_k=4)
    _conv_case(hip, f"ts{tile}d", 2, 13, 13, 48, 64, 3, 1, 1, residual=True, split_k=0, tile=tile)
    _conv_case(hip, f"ts{tile}e", 1, 13, 13, 64, 48, 3, 1, 1, ups=2, split_k=5, tile=tile, x_slice=64)
    bm, bn = {41: (128, 128), 42: (128, 64), 43: (64, 64), 44: (128, 32), 45: (256, 128), 47: (64, 64)}[tile]
    n, cout = {41: (4, 500), 42: (2, 500), 43: (2, 250), 44: (2, 250), 45: (8, 500), 47: (2, 250)}[tile]
    tiles = -(-n * 52 * 52 // bm) * -(-cout // bn)
    assert 256 < tiles < 512 and tiles % 256, tiles
    _conv_case(hip, f"ts{tile}f", n, 52, 52, 32, cout, 3, 1, 1, residual=True, split_k=3, tile=tile)
    _conv_case(hip, f"ts{tile}g", n, 52, 52, 64, cout, 1, 1, 1, split_k=2, tile=tile)
    x = _t("tsx", (2, 52, 52, 32)).cuda()
    w = torch.from_numpy(synth.normal("tsw", (250, 3, 3, 32), 0, 0.05)).cuda()
    s = torch.ones(250).cuda()
    b = torch.zeros(250).cuda()
    a1 = hip.conv2d(x, w, s, b, 3, 1, 1, 1, tile=tile, split_k=4)
    a2 = hip.conv2d(x, w, s, b, 3, 1, 1, 1, tile=tile, split_k=4)
    assert torch.equal(a1, a2), "the tail split must be run-to-run deterministic"
    a0 = hip.conv2d(x, w, s, b, 3, 1, 1, 1, tile=tile - 40, split_k=1)
    assert_close(a1.cpu(), a0.cpu(), 1e-4, "tail split vs whole tiles")


@pytest.mark.parametrize("tile,split", [(3, 4), (1, 3), (2, 9), (5, 2), (7, 5), (43, 3), (41, 4), (45, 2), (47, 3)])
def test_conv_in_launch_split_reduce(hip_lib, tile, split):
    """Split-K / tail-split slabs reduced inside the launch (arrival counter per tile, the last workgroup to arrive sums the
    slabs in the fixed order and runs the fused epilogue) against the two-pass form (slabs + reduce launch): same bits, every
    time (40 repeats on shapes with many tiles per XCD - a stale slab read across XCDs or a lost arrival would show), and
    the counters are back at zero."""
    from millieye_amd import hip
    for tag, shape, cout, k in (("a", (2, 26, 26, 64), 250, 3), ("b", (4, 52, 52, 32), 500, 3), ("c", (8, 13, 13, 256), 255, 1)):
        x = _t(f"ilr{tag}x", shape).cuda()
        w = torch.from_numpy(synth.normal(f"ilr{tag}w", (cout, k, k, shape[3]), 0, 0.05)).cuda()
        s = _t(f"ilr{tag}s", (cout,), 0.5, 1.5).cuda()
        b = _t(f"ilr{tag}b", (cout,), -0.5, 0.5).cuda()
        r = _t(f"ilr{tag}r", (shape[0], shape[1], shape[2], cout)).cuda()
        pad = (k - 1) // 2
        two = hip.conv2d(x, w, s, b, k, 1, pad, 1, residual=r, tile=tile, split_k=split, in_launch_reduce=False)
        for _ in range(40):
            one = hip.conv2d(x, w, s, b, k, 1, pad, 1, residual=r, tile=tile, split_k=split, in_launch_reduce=True)
            assert torch.equal(one, two), f"in-launch reduction differs from the two-pass result ({tag}, tile {tile})"
        _ptr, _n, counters = hip.tile_counters(x.device)
        assert int(counters.abs().sum()) == 0, "arrival counters must be zero between launches"
        ref = hip.conv2d(x, w, s, b, k, 1, pad, 1, residual=r, tile=tile % 40, split_k=1)
        assert_close(one.cpu(), ref.cpu(), 1e-4, f"split vs whole ({tag}, tile {tile})")


@pytest.mark.parametrize("tile", [3, 43, 2, 5])
def test_conv_column_panels_ragged(hip_lib, tile):
    """Deep layers walk their tiles in column panels (weights of a panel L2-resident, chunk-major K): a case whose tile columns
    do not divide into panels (448 output channels = 7 columns of 64 in panels of 2, 2, 2, 1; 4 columns of 128 in panels of
    1), with whole tiles + tail pieces in one launch (tile 43), ragged M, residual."""
    from millieye_amd import hip
    _conv_case(hip, f"pan{tile}a", 8, 26, 26, 256, 448, 3, 1, 1, residual=True, tile=tile, split_k=3 if tile > 40 else 1)
    _conv_case(hip, f"pan{tile}b", 3, 13, 13, 512, 320, 3, 1, 1, tile=tile, split_k=2 if tile > 40 else 1)
    _conv_case(hip, f"pan{tile}c", 2, 13, 13, 1024, 448, 1, 1, 1, tile=tile, split_k=2 if tile > 40 else 1)  # 1x1: panels, tap-major


def test_conv_variants(hip_lib):
    from millieye_amd import hip
    _conv_case(hip, "s2", 2, 32, 32, 32, 64, 3, 2, 1)                 # stride-2 downsample
    _conv_case(hip, "k1", 2, 13, 13, 128, 255, 1, 1, 0)               # 1x1 linear detection conv (cout 255)
    _conv_case(hip, "res", 2, 16, 16, 32, 64, 3, 1, 1, residual=True)  # fused [shortcut]
    _conv_case(hip, "ups", 2, 13, 13, 64, 32, 1, 1, 1, ups=2)          # fused [upsample]
    _conv_case(hip, "sig", 1, 26, 26, 128, 10, 1, 1, 2)                # radar head: 1x1 + sigmoid, cout 10
    _conv_case(hip, "k1big", 1, 26, 26, 256, 490, 1, 1, 1)             # cnn_layers_1 256->490
    _conv_case(hip, "odd", 1, 7, 5, 16, 40, 3, 1, 1)                   # tiny spatial, cin 16


@pytest.mark.parametrize("split", [2, 3, 9])
def test_conv_split_k_is_exact_and_deterministic(hip_lib, split):
    """Small-M layers are split along K (slabs + ordered reduce): same 1e-3 bar, bit-reproducible,
    and the fused epilogue (residual / upsample / activation) must survive the second pass."""
    from millieye_amd import hip
    _conv_case(hip, f"sk{split}", 2, 13, 13, 64, 96, 3, 1, 1, residual=True, split_k=split)
    _conv_case(hip, f"sku{split}", 1, 13, 13, 48, 40, 3, 1, 1, ups=2, split_k=split, tile=3)
    x = _t("skx", (2, 13, 13, 128)).cuda()
    w = torch.from_numpy(synth.normal("skw", (256, 3, 3, 128), 0, 0.03)).cuda()
    s = torch.ones(256).cuda()
    b = torch.zeros(256).cuda()
    a1 = hip.conv2d(x, w, s, b, 3, 1, 1, 1, split_k=split)
    a2 = hip.conv2d(x, w, s, b, 3, 1, 1, 1, split_k=split)
    assert torch.equal(a1, a2), "split-K must be run-to-run deterministic"
    a0 = hip.conv2d(x, w, s, b, 3, 1, 1, 1, split_k=1)
    assert_close(a1.cpu(), a0.cpu(), 1e-4, "split vs unsplit")


def test_conv_auto_plan_matches_reference_on_small_maps(hip_lib):
    """Darknet-53's 13x13 / 26x26 layers at batch 8: the automatic (tile, split) plan."""
    from millieye_amd import hip
    _conv_case(hip, "auto13", 8, 13, 13, 512, 1024, 3, 1, 1)
    _conv_case(hip, "auto13k1", 8, 13, 13, 1024, 512, 1, 1, 1)


def test_conv_stem_smallcin(hip_lib):
    from millieye_amd import hip
    _conv_case(hip, "stem", 2, 32, 32, 3, 32, 3, 1, 1, nchw_in=True)   # NCHW network input
    _conv_case(hip, "stem16", 1, 40, 24, 3, 16, 3, 1, 1, nchw_in=True)
    _conv_case(hip, "radar", 2, 26, 26, 3, 32, 3, 1, 1, nchw_in=False)  # NHWC cin 3
    _conv_case(hip, "cin4", 1, 9, 9, 4, 20, 3, 1, 0, nchw_in=False)     # cout not multiple of 8
    # couts that are multiples of 32 go to the MFMA stem (csrc/stem_mfma_f32.hip) by default; tile 92 / 91 force the two VALU
    # versions: all three against the CPU convolution, ragged M (not a multiple of 32), sigmoid, 64 channels, pitched output
    for tile in (0, 92, 91):
        _conv_case(hip, f"stem32t{tile}", 3, 13, 21, 3, 32, 3, 1, 1, nchw_in=True, tile=tile)
        _conv_case(hip, f"stem64t{tile}", 2, 17, 9, 3, 64, 3, 1, 2, nchw_in=False, tile=tile)
    x = _t("stemx", (2, 3, 20, 20)).cuda()
    w = torch.from_numpy(synth.normal("stemw", (32, 3, 3, 3), 0, 0.2)).cuda()
    s1, b1 = torch.ones(32).cuda(), torch.zeros(32).cuda()
    dense = hip.conv2d(x, w, s1, b1, 3, 1, 1, 1, x_nchw=True)
    wide = torch.zeros((2, 20, 20, 48)).cuda()
    hip.conv2d(x, w, s1, b1, 3, 1, 1, 1, x_nchw=True, out=wide[..., 8:40])
    assert torch.equal(wide[..., 8:40], dense) and float(wide[..., :8].abs().max()) == 0


def test_conv_pitched_concat_slice(hip_lib):
    """conv writing into a channel slice of a wider buffer and reading a slice ([route] for free)."""
    from millieye_amd import hip
    import ctypes as C
    n, h, w, cin, cout = 2, 13, 13, 32, 48
    big_in = _t("pin", (n, h, w, 96)).cuda()
    big_out = torch.zeros((n, h, w, 128), device="cuda")
    wgt = torch.from_numpy(synth.normal("pw", (cout, cin, 3, 3), 0, 0.06))
    scale, shift = torch.ones(cout), torch.zeros(cout)
    d = hip.ConvDesc()
    pk = hip.pack_conv_weight(wgt).cuda()
    sc, sh = scale.cuda(), shift.cuda()
    d.x = big_in.data_ptr() + 4 * 64          # channels 64..95
    d.x_pitch = 96
    d.wgt, d.scale, d.shift, d.res = pk.data_ptr(), sc.data_ptr(), sh.data_ptr(), None
    d.y = big_out.data_ptr() + 4 * 16         # channels 16..63
    d.y_pitch = 128
    d.n, d.h, d.w, d.cin, d.cout, d.ksize, d.stride, d.pad, d.ho, d.wo = n, h, w, cin, cout, 3, 1, 1, h, w
    d.act, d.upsample, d.x_nchw, d.tile = 0, 1, 0, 0
    hip.check(hip.lib().me_conv2d_f32(C.byref(d), hip.stream_ptr()), "conv")
    torch.cuda.synchronize()
    ref = F.conv2d(big_in.cpu()[..., 64:96].permute(0, 3, 1, 2), wgt, None, 1, 1)
    out = big_out.cpu()
    assert_close(out[..., 16:64].permute(0, 3, 1, 2), ref, TOL, "slice")
    assert out[..., :16].abs().max() == 0 and out[..., 64:].abs().max() == 0, "wrote outside its slice"


def test_conv_rejects_bad_descriptors(hip_lib):
    from millieye_amd import hip
    x = torch.zeros((1, 8, 8, 6), device="cuda")  # cin 6: not % 4
    w = torch.zeros((8, 3, 3, 6), device="cuda")
    s = torch.zeros(8, device="cuda")
    with pytest.raises(hip.MeError):
        hip.conv2d(x, w, s, s, 3, 1, 1, 0)


def test_maxpool(hip_lib):
    from millieye_amd import hip
    x = _t("mp", (2, 16, 26, 26))
    got = hip.maxpool(x.permute(0, 2, 3, 1).contiguous().cuda(), 2, 2).cpu().permute(0, 3, 1, 2)
    assert torch.equal(got, F.max_pool2d(x, 2, 2))
    # darknet's size-2 stride-1 pool: ZERO pad right/bottom (quirk q16) - use negative data so 0 wins
    xn = _t("mpn", (1, 8, 13, 13), -2.0, -1.0)
    got = hip.maxpool(xn.permute(0, 2, 3, 1).contiguous().cuda(), 2, 1, zero_ext=True).cpu().permute(0, 3, 1, 2)
    ref = F.max_pool2d(F.pad(xn, (0, 1, 0, 1), value=0.0), 2, 1)
    assert torch.equal(got, ref)
    x3 = _t("mp3", (1, 6, 9, 9))  # c % 4 != 0 -> scalar path; odd size
    got = hip.maxpool(x3.permute(0, 2, 3, 1).contiguous().cuda(), 2, 2).cpu().permute(0, 3, 1, 2)
    assert torch.equal(got, F.max_pool2d(x3, 2, 2))


def test_upsample_and_layout(hip_lib):
    from millieye_amd import hip
    x = _t("up", (2, 12, 5, 7))
    nhwc = x.permute(0, 2, 3, 1).contiguous().cuda()
    got = hip.upsample(nhwc, 2).cpu().permute(0, 3, 1, 2)
    assert torch.equal(got, F.interpolate(x, scale_factor=2, mode="nearest"))
    assert torch.equal(hip.nhwc_to_nchw(nhwc).cpu(), x)


@pytest.mark.parametrize("g,nc,img", [(13, 12, 416), (3, 80, 96), (26, 12, 416)])
def test_yolo_decode(hip_lib, g, nc, img):
    from millieye_amd import hip
    from oracle import darknet_ref
    anchors = [(81, 82), (135, 169), (344, 319)]
    x = _t(f"yd{g}", (2, 3 * (5 + nc), g, g), -3.0, 3.0)
    ref = darknet_ref.yolo_decode(x, anchors, nc, img)
    got = hip.yolo_decode(x.permute(0, 2, 3, 1).contiguous().cuda(), anchors, nc, img).cpu()
    assert_close(got, ref, 1e-5, "yolo decode")


def test_conv_random_shapes(hip_lib):
    """Seeded sweep of odd shapes through the automatic plan (and the buffer-addressed kernel where it applies):
    filters 1/3/5/7 (7x7 has more taps than the DMA kernels' 32-bit padding mask: register-staged fallback), strides 1-3,
    pads 0-3, h != w, h or w = 1, channel counts around the 16 / 64 boundaries, every activation."""
    from millieye_amd import hip
    rng = np.random.RandomState(20260928)
    done = 0
    for case in range(60):
        k = int(rng.choice([1, 3, 3, 5, 7]))
        s = int(rng.choice([1, 1, 2, 3]))
        pad = (k - 1) // 2   # the wrapper mirrors darknet: pad = (k - 1) // 2
        cin = int(rng.choice([8, 12, 16, 24, 32, 48, 64, 80]))  # cin <= 4 is the stem kernel (3x3 only, fails loudly otherwise)
        cout = int(rng.choice([1, 5, 16, 33, 64, 96, 130]))
        h, w = int(rng.randint(1, 23)), int(rng.randint(1, 23))
        if (h + 2 * pad - k) < 0 or (w + 2 * pad - k) < 0:
            continue
        n = int(rng.randint(1, 4))
        _conv_case(hip, f"rnd{case}", n, h, w, cin, cout, k, s, int(rng.randint(0, 3)), residual=bool(rng.randint(0, 2)),
                   split_k=int(rng.choice([0, 0, 2, 3])))
        done += 1
    assert done >= 45


def test_conv_even_filter_with_explicit_pad(hip_lib):
    """2x2 filters with pad 1 (output one larger than the input): the shape the stride-2 data gradient of
    millieye_amd/detector_train.py runs as (four output-parity classes in one 2x2 convolution); through the automatic
    plan and the per-shape tuner of the training path."""
    from millieye_amd import hip
    for tag, n, h, w, cin, cout in (("e0", 2, 13, 13, 64, 128), ("e1", 1, 26, 20, 128, 256), ("e2", 3, 7, 9, 24, 36)):
        x = _t(tag + "x", (n, cin, h, w))
        wgt = torch.from_numpy(synth.normal(tag + "w", (cout, cin, 2, 2), 0, (2.0 / (cin * 4)) ** 0.5))
        ref = F.conv2d(x, wgt, None, 1, 1)
        one, zero = torch.ones(cout, device="cuda"), torch.zeros(cout, device="cuda")
        xd = x.permute(0, 2, 3, 1).contiguous().cuda()
        for fn in (hip.conv2d, hip.conv2d_auto):
            got = fn(xd, hip.pack_conv_weight(wgt).cuda(), one, zero, 2, 1, 1, hip.ACT_LINEAR)
            torch.cuda.synchronize()
            assert tuple(got.shape) == (n, h + 1, w + 1, cout)
            assert_close(got.cpu().permute(0, 3, 1, 2), ref, TOL, tag + fn.__name__)


P8_F32_TILES_128 = (121, 131, 201, 221, 311, 321)
P8_F32_TILES_256 = (100, 110, 200)


@pytest.mark.parametrize("tile", P8_F32_TILES_128 + P8_F32_TILES_256)
def test_conv_p8_patch_resident_f32(hip_lib, tile):
    """The patch-resident big tiles of the fp32 path (csrc/conv_p8_f32.hip = conv_p8_impl.h on v_mfma_f32_32x32x2_f32):
    tiles are BM consecutive positions of the padded-linear index space (they cross rows and images); every tile shape
    against the CPU convolution at the fp32 bar (1e-3), with residual, on rectangular / tiny / 2-images-per-tile maps, channel
    slices on both sides, and bit-reproducible."""
    from millieye_amd import hip
    cout = 256 if tile in P8_F32_TILES_256 else 128
    _conv_case(hip, f"p8f{tile}a", 5, 13, 13, 32, cout, 3, 1, 1, tile=tile, residual=True)       # two images per tile
    _conv_case(hip, f"p8f{tile}b", 2, 9, 31, 16, cout, 3, 1, 0, tile=tile)                       # rectangular, linear, cs = 1
    _conv_case(hip, f"p8f{tile}c", 1, 52, 52, 48, 2 * cout, 3, 1, 1, tile=tile, residual=True, x_slice=16)
    _conv_case(hip, f"p8f{tile}d", 1, 5, 7, 64, cout, 3, 1, 1, tile=tile)                        # one tiny image
    x = _t("p8fx", (3, 26, 26, 64)).cuda()
    w = torch.from_numpy(synth.normal("p8fw", (cout, 3, 3, 64), 0, 0.04)).cuda()
    s, b = torch.ones(cout).cuda(), torch.zeros(cout).cuda()
    a1 = hip.conv2d(x, w, s, b, 3, 1, 1, 1, tile=tile, split_k=1)
    a2 = hip.conv2d(x, w, s, b, 3, 1, 1, 1, tile=tile, split_k=1)
    assert torch.equal(a1, a2)
    a0 = hip.conv2d(x, w, s, b, 3, 1, 1, 1, tile=3, split_k=1)
    assert_close(a1.cpu(), a0.cpu(), 1e-4, "patch-resident vs per-tap kernel")
    wide = torch.zeros((3, 26, 26, cout + 32), device="cuda")
    hip.conv2d(x, w, s, b, 3, 1, 1, 1, tile=tile, split_k=1, out=wide[..., 16:16 + cout])
    assert torch.equal(wide[..., 16:16 + cout], a1) and float(wide[..., :16].abs().max()) == 0


WS1_PAIRS = ((64, 32), (64, 64), (64, 128), (128, 64), (128, 128), (256, 128), (256, 255), (384, 128), (512, 256), (512, 255))


@pytest.mark.parametrize("cin,cout", WS1_PAIRS)
def test_conv_ws1x1_f32(hip_lib, cin, cout):
    """Tile 50 of me_conv2d_f32 (csrc/conv_ws_f32.hip: weights in registers, activations streamed through an LDS ring,
    persistent grid): every built cin against the CPU convolution at the fp32 bar (1e-3) - ragged M (rows behind the last
    pixel are zero-filled by the DMA range check), more tiles than ring slots and more tiles than workgroups' first round,
    ragged cout (255: the detection convolutions), linear / leaky, a channel slice as input, a pitched output - and BIT-identical
    to the per-tap 64x64 tile without split-K (same accumulation order, same epilogue expression)."""
    from millieye_amd import hip
    _conv_case(hip, f"w1a{cin}_{cout}", 3, 13, 11, cin, cout, 1, 1, 1, tile=50)                 # 429 rows: ragged last tile
    _conv_case(hip, f"w1b{cin}_{cout}", 1, 5, 5, cin, cout, 1, 1, 0, tile=50, x_slice=32)      # fewer rows than a tile, linear, slice
    n, h, w = (6, 52, 52) if cin <= 256 else (4, 26, 26)
    x = _t(f"w1x{cin}", (n, h, w, cin)).cuda()
    wg = torch.from_numpy(synth.normal(f"w1w{cin}_{cout}", (cout, 1, 1, cin), 0, (2.0 / cin) ** 0.5)).cuda()
    sc, sh = _t("w1s", (cout,), 0.5, 1.5).cuda(), _t("w1b", (cout,), -0.5, 0.5).cuda()
    a0 = hip.conv2d(x, wg, sc, sh, 1, 1, 0, 1, tile=3, split_k=1)
    a1 = hip.conv2d(x, wg, sc, sh, 1, 1, 0, 1, tile=50)
    a2 = hip.conv2d(x, wg, sc, sh, 1, 1, 0, 1, tile=50)
    torch.cuda.synchronize()
    assert torch.equal(a1, a2)
    assert torch.equal(a1, a0), f"max diff {float((a1 - a0).abs().max())}"
    wide = torch.zeros((n, h, w, cout + 9), device="cuda")
    hip.conv2d(x, wg, sc, sh, 1, 1, 0, 1, tile=50, out=wide[..., 4:4 + cout])
    assert torch.equal(wide[..., 4:4 + cout], a1) and float(wide[..., :4].abs().max()) == 0 and \
        float(wide[..., 4 + cout:].abs().max()) == 0


@pytest.mark.parametrize("cin,cout,stride", [(32, 64, 1), (32, 64, 2), (64, 128, 1), (64, 128, 2), (32, 48, 1), (64, 255, 2)])
def test_conv_ws3x3_f32(hip_lib, cin, cout, stride):
    """Tile 60 of me_conv2d_f32 (weight-stationary 3x3 for 32 / 64 input channels; 2-D output tiles whose input patch sits in
    one LDS ring slot, the nine taps as pixel shifts): odd map sizes (ragged tiles on both axes, borders everywhere), both
    strides, fused shortcut, ragged cout, channel slices - against the CPU convolution (1e-3) and bit-identical to the per-tap
    64x64 tile without split-K."""
    from millieye_amd import hip
    tag = f"w3{cin}_{cout}_{stride}"
    _conv_case(hip, tag + "a", 2, 13, 11, cin, cout, 3, stride, 1, tile=60, residual=(stride == 1))
    _conv_case(hip, tag + "b", 1, 37, 50, cin, cout, 3, stride, 0, tile=60, x_slice=32)
    _conv_case(hip, tag + "c", 3, 5, 3, cin, cout, 3, stride, 1, tile=60)
    n, h, w = 3, 52, 44
    x = _t(tag + "x", (n, h, w, cin)).cuda()
    wg = torch.from_numpy(synth.normal(tag + "w", (cout, 3, 3, cin), 0, (2.0 / (9 * cin)) ** 0.5)).cuda()
    sc, sh = _t("w3s", (cout,), 0.5, 1.5).cuda(), _t("w3b", (cout,), -0.5, 0.5).cuda()
    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
    res = _t(tag + "r", (n, ho, wo, cout)).cuda() if stride == 1 else None
    a0 = hip.conv2d(x, wg, sc, sh, 3, stride, 1, 1, residual=res, tile=3, split_k=1)
    a1 = hip.conv2d(x, wg, sc, sh, 3, stride, 1, 1, residual=res, tile=60)
    a2 = hip.conv2d(x, wg, sc, sh, 3, stride, 1, 1, residual=res, tile=60)
    torch.cuda.synchronize()
    assert torch.equal(a1, a2)
    assert torch.equal(a1, a0), f"max diff {float((a1 - a0).abs().max())}"


def test_conv_ws_f32_refuses_what_it_cannot_do(hip_lib):
    from millieye_amd import hip
    one, zero = torch.ones(64).cuda(), torch.zeros(64).cuda()
    x = torch.zeros((1, 8, 8, 48)).cuda()
    with pytest.raises(hip.MeError):   # cin without an instance
        hip.conv2d(x, torch.zeros((64, 1, 1, 48)).cuda(), one, zero, 1, 1, 0, 1, tile=50)
    x = torch.zeros((1, 8, 8, 64)).cuda()
    with pytest.raises(hip.MeError):   # sigmoid epilogue
        hip.conv2d(x, torch.zeros((64, 1, 1, 64)).cuda(), one, zero, 1, 1, 0, 2, tile=50)
    with pytest.raises(hip.MeError):   # upsampling epilogue
        hip.conv2d(x, torch.zeros((64, 1, 1, 64)).cuda(), one, zero, 1, 1, 0, 1, tile=50, upsample=2)
    with pytest.raises(hip.MeError):   # 3x3 with 128 input channels
        hip.conv2d(torch.zeros((1, 8, 8, 128)).cuda(), torch.zeros((64, 3, 3, 128)).cuda(), one, zero, 3, 1, 1, 1, tile=60)
    with pytest.raises(hip.MeError):   # no split-K form
        hip.conv2d(x, torch.zeros((64, 1, 1, 64)).cuda(), one, zero, 1, 1, 0, 1, tile=50, split_k=2)


def test_conv_p8_f32_refuses_what_it_cannot_do(hip_lib):
    from millieye_amd import hip
    x = torch.zeros((1, 8, 8, 32)).cuda()
    one = torch.ones(128).cuda()
    with pytest.raises(hip.MeError):
        hip.conv2d(x, torch.zeros((128, 1, 1, 32)).cuda(), one, one, 1, 1, 0, 1, tile=221)          # 1x1
    with pytest.raises(hip.MeError):
        hip.conv2d(x, torch.zeros((72, 3, 3, 32)).cuda(), one[:72], one[:72], 3, 1, 1, 1, tile=221)  # cout % 128
    with pytest.raises(hip.MeError):
        hip.conv2d(x, torch.zeros((128, 3, 3, 32)).cuda(), one, one, 3, 2, 1, 1, tile=221)           # stride 2


def test_pack_conv_single_and_batch(hip_lib):
    """me_pack_conv_f32 / me_pack_conv_batch_f32: every packed copy against torch permutes of the OIHW parameter (ragged
    channel counts, 1x1 / 3x3 / 5x5, with and without BatchNorm / bias / rotated copies), and the one-launch table gives the
    same bytes as the per-layer launches."""
    from millieye_amd import hip
    lib = hip.lib()
    shapes = [(32, 3, 3, True, False), (64, 32, 3, True, True), (51, 48, 1, False, True), (40, 80, 3, True, True),
              (128, 256, 1, True, True), (16, 16, 3, True, True), (255, 1024, 1, False, False), (96, 36, 3, True, True)]
    keep, descs, expect = [], (hip.PackDesc * len(shapes))(), []
    for idx, (cout, cin, k, with_bn, with_rot) in enumerate(shapes):
        tag = f"pk{idx}"
        w = _t(tag + "w", (cout, cin, k, k)).cuda()
        bias = None if with_bn else _t(tag + "b", (cout,)).cuda()
        g, b, m = (_t(tag + c, (cout,)).cuda() for c in "gbm") if with_bn else (None, None, None)
        v = _t(tag + "v", (cout,), 0.5, 1.5).cuda() if with_bn else None
        outs = dict(ohwi=torch.zeros((cout, k, k, cin)), tiled=torch.zeros((k * k, cin // 16, cout, 16)) if cin % 16 == 0 else None,
                    rot=torch.zeros((cin, k, k, cout)) if with_rot else None,
                    rott=torch.zeros((k * k, cout // 16, cin, 16)) if with_rot and cout % 16 == 0 else None,
                    scale=torch.zeros(cout), shift=torch.zeros(cout))
        one = {n_: (t.cuda() if t is not None else None) for n_, t in outs.items()}
        two = {n_: (t.cuda() if t is not None else None) for n_, t in outs.items()}
        ptr = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
        hip.check(lib.me_pack_conv_f32(w.data_ptr(), cout, cin, k, ptr(bias), ptr(g), ptr(b), ptr(m), ptr(v), 1e-5,
                                       one["ohwi"].data_ptr(), ptr(one["tiled"]), ptr(one["rot"]), ptr(one["rott"]),
                                       one["scale"].data_ptr(), one["shift"].data_ptr(), hip.stream_ptr()), "me_pack_conv_f32")
        d = descs[idx]
        d.w, d.bias, d.gamma, d.beta, d.mean, d.var = w.data_ptr(), ptr(bias), ptr(g), ptr(b), ptr(m), ptr(v)
        d.ohwi, d.tiled, d.rot, d.rot_tiled = two["ohwi"].data_ptr(), ptr(two["tiled"]), ptr(two["rot"]), ptr(two["rott"])
        d.scale, d.shift, d.cout, d.cin, d.ksize, d.eps = two["scale"].data_ptr(), two["shift"].data_ptr(), cout, cin, k, 1e-5
        keep.append((w, bias, g, b, m, v))
        # torch restatement
        wc = w.cpu()
        assert torch.equal(one["ohwi"].cpu(), wc.permute(0, 2, 3, 1).contiguous()), tag
        if one["tiled"] is not None:
            ref = wc.permute(2, 3, 1, 0).reshape(k * k, cin // 16, 16, cout).permute(0, 1, 3, 2).contiguous()
            assert torch.equal(one["tiled"].cpu(), ref), tag
        if one["rot"] is not None:
            rot = wc.flip(2, 3).permute(1, 2, 3, 0).contiguous()
            assert torch.equal(one["rot"].cpu(), rot), tag
            if one["rott"] is not None:
                ref = rot.permute(1, 2, 3, 0).reshape(k * k, cout // 16, 16, cin).permute(0, 1, 3, 2).contiguous()
                assert torch.equal(one["rott"].cpu(), ref), tag
        if with_bn:
            sc = g.cpu().double() / torch.sqrt(v.cpu().double() + 1e-5)
            sh = b.cpu().double() - m.cpu().double() * sc
        else:
            sc, sh = torch.ones(cout, dtype=torch.float64), bias.cpu().double()
        assert_close(one["scale"].cpu(), sc.float(), 1e-6, tag + "scale")
        assert_close(one["shift"].cpu(), sh.float(), 1e-6, tag + "shift")
        expect.append((one, two))
    # the stride-2 layers' parity weights ride in the whole-network launch (3x3 only)
    from millieye_amd.detector_train import _parity_weights
    par = {}
    for idx, (cout, cin, k, _bn, _rot) in enumerate(shapes):
        if k == 3 and cin > 4:
            par[idx] = torch.zeros((4 * cin, 2, 2, cout), device="cuda")
            descs[idx].parity = par[idx].data_ptr()
    total = int(lib.me_pack_conv_plan(descs, len(shapes)))
    assert total > 0
    table = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).cuda()
    hip.check(lib.me_pack_conv_batch_f32(table.data_ptr(), len(shapes), total, 3, hip.stream_ptr()), "me_pack_conv_batch_f32")
    torch.cuda.synchronize()
    for one, two in expect:
        for n_ in one:
            if one[n_] is not None:
                assert torch.equal(one[n_], two[n_]), n_
    for idx, t in par.items():
        assert torch.equal(t, _parity_weights(expect[idx][1]["ohwi"])), f"parity weights of shape {shapes[idx]}"
    bad = (hip.PackDesc * 1)()
    assert int(lib.me_pack_conv_plan(bad, 1)) < 0


def test_round3_entry_points_refuse_bad_arguments(hip_lib):
    """me_yolo_loss_fwd_f32 / me_pack_conv_batch_f32 / me_image_pad_resize_flip_u8_f32: null pointers and impossible sizes
    come back as negative ME_E_* codes with a message, nothing is launched."""
    import ctypes as C
    from millieye_amd import hip
    lib = hip.lib()
    buf = torch.zeros(4096, device="cuda")
    ws = torch.zeros(int(lib.me_yolo_loss_workspace_bytes()), dtype=torch.uint8, device="cuda")
    anchors = (C.c_float * 64)(*([1.0] * 64))
    p = buf.data_ptr()
    args = lambda na, raw=p, wsp=ws.data_ptr(): (raw, 3 * 17, 1, 2, na, 12, anchors, p, 1, 0.5, 1.0, 100.0, p, p, p, p, p, p, p, p,  # noqa: E731
                                                p, p, wsp, p, hip.stream_ptr())
    assert lib.me_yolo_loss_fwd_f32(*args(3, raw=None)) < 0 and b"null" in lib.me_last_error()
    assert lib.me_yolo_loss_fwd_f32(*args(17)) < 0 and b"16 anchors" in lib.me_last_error()
    assert lib.me_yolo_loss_fwd_f32(*args(3, wsp=ws.data_ptr() + 4)) < 0
    assert lib.me_pack_conv_batch_f32(None, 1, 1, 3, hip.stream_ptr()) < 0
    assert lib.me_pack_conv_batch_f32(p, 1, 0, 3, hip.stream_ptr()) < 0
    assert lib.me_pack_conv_batch_f32(p, 1, 1, 7, hip.stream_ptr()) < 0 and b"too large" in lib.me_last_error()
    assert lib.me_image_pad_resize_flip_u8_f32(None, 4, 4, p, 8, 1, hip.stream_ptr()) < 0
    assert lib.me_image_pad_resize_flip_u8_f32(p, 0, 4, p, 8, 1, hip.stream_ptr()) < 0
    torch.cuda.synchronize()


@pytest.mark.parametrize("cin,cout,h", [(32, 64, 21), (64, 128, 14), (128, 256, 9), (256, 512, 7)])
def test_conv_tap_masks_skip_the_zero_taps_bit_exactly(hip_lib, cin, cout, h):
    """ABI 10, ``me_conv_desc.tap_mask``: the 2x2 convolution that computes the four output-parity classes of a 3x3 / stride-2
    layer's data gradient (detector_train._parity_weights: 7 of its 16 (class, tap) pairs are structurally zero) with the zero
    taps SKIPPED equals the same launch multiplying them - bit for bit (a skipped tap would have added +0.0), on every tile the
    library accepts for the class width - and equals torch's transposed convolution; bad masks / tiles are refused."""
    from millieye_amd import hip
    from millieye_amd.detector_train import _PARITY_TAP_MASKS, _parity_weights
    n = 3
    dev = torch.device("cuda")
    w = torch.from_numpy(synth.uniform(f"tm/w{cin}", (cout, 3, 3, cin), -1, 1)).to(dev) / (9 * cin) ** 0.5   # forward OHWI
    dc = torch.from_numpy(synth.uniform(f"tm/dc{cin}", (n, h, h, cout), -1, 1)).to(dev)                    # dL/d(conv output)
    pw = _parity_weights(w)
    assert pw.shape == (4 * cin, 2, 2, cout)
    flat = pw.reshape(4, cin, 4, cout)
    for cls, mask in enumerate(_PARITY_TAP_MASKS):   # the masks say exactly which taps are zero
        for t in range(4):
            assert bool((flat[cls, :, t] == 0).all()) == (not (mask >> t) & 1)
    ones, zeros = torch.ones(4 * cin, device=dev), torch.zeros(4 * cin, device=dev)
    plain = hip.conv2d(dc, pw, ones, zeros, 2, 1, 1, hip.ACT_LINEAR, tile=3, split_k=1)
    seen = 0
    for tile in (0, 1, 2, 3, 4, 5):
        try:
            got = hip.conv2d(dc, pw, ones, zeros, 2, 1, 1, hip.ACT_LINEAR, tile=tile, tap_masks=(cin, _PARITY_TAP_MASKS))
        except hip.MeError:
            assert tile != 0, "the automatic tile must exist for every class width"
            continue   # (a tile wider than the class)
        base = hip.conv2d(dc, pw, ones, zeros, 2, 1, 1, hip.ACT_LINEAR, tile=tile if tile else 3, split_k=1)
        assert torch.allclose(got, plain, rtol=1e-5, atol=1e-6), tile
        if tile:
            assert torch.equal(got, base), f"tile {tile}: skipping the zero taps changed bits"
        seen += 1
    assert seen >= 2
    # the gradient itself: pixel shuffle of the cropped classes == conv_transpose2d of dc with the forward weights
    dx = got[:, 1:, 1:, :].reshape(n, h, h, 2, 2, cin).permute(0, 1, 3, 2, 4, 5).reshape(n, 2 * h, 2 * h, cin)
    ref = F.conv_transpose2d(dc.permute(0, 3, 1, 2).cpu(), w.permute(0, 3, 1, 2).cpu(), stride=2, padding=1, output_padding=1)
    assert_close(dx.permute(0, 3, 1, 2).cpu(), ref, 1e-4, "stride-2 data gradient through the masked parity convolution")
    with pytest.raises(hip.MeError):
        hip.conv2d(dc, pw, ones, zeros, 2, 1, 1, hip.ACT_LINEAR, tap_masks=(cin, (1, 3, 0, 15)))      # a class without taps
    with pytest.raises(hip.MeError):
        hip.conv2d(dc, pw, ones, zeros, 2, 1, 1, hip.ACT_LINEAR, tap_masks=(cin + 16, _PARITY_TAP_MASKS))  # cout % cols != 0
    with pytest.raises(hip.MeError):
        hip.conv2d(dc, pw, ones, zeros, 2, 1, 1, hip.ACT_LINEAR, tile=43, tap_masks=(cin, _PARITY_TAP_MASKS))  # tail split
