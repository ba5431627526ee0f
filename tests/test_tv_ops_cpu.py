"""CPU: the torchvision-0.6 boundary (rows a8 / a12 / a13; PARITY UNPINNED - the library is absent).
Two independently written restatements - oracle/tv_ops.c (float32, C, scatter backward) and oracle/tv_ops_np.py (float64,
numpy, explicit sampling-weight matrices) - must agree on fuzzed inputs, forward and backward, and both must reproduce
answers that can be derived by hand from the operator definitions."""
import numpy as np
import pytest
import torch
from hypothesis import given, settings, strategies as st

from oracle import tv_ops, tv_ops_np

P = 7


def _rois(rng, k, n, size, degenerate=False):
    r = np.zeros((k, 5), dtype=np.float32)
    r[:, 0] = rng.randint(0, n, size=k)
    a = rng.uniform(-0.15 * size, 1.1 * size, size=(k, 2))
    wh = rng.uniform(0.5, 0.8 * size, size=(k, 2))
    r[:, 1:3] = a
    r[:, 3:5] = a + wh
    if degenerate and k >= 3:
        r[0, 3:5] = r[0, 1:3]                    # zero-size box
        r[1, 3] = r[1, 1] - 5.0                  # inverted in x
        r[2, 1:5] = [-400.0, -400.0, -350.0, -350.0]  # entirely outside
    return r


@settings(max_examples=25, deadline=None)
@given(seed=st.integers(0, 10 ** 6), h=st.integers(3, 14), w=st.integers(3, 14), k=st.integers(1, 6),
       ratio=st.sampled_from([-1, 1, 2, 3]), aligned=st.booleans())
def test_roi_align_c_vs_numpy(seed, h, w, k, ratio, aligned):
    rng = np.random.RandomState(seed)
    n, c, scale = 2, 3, 1.0 / 16
    x = rng.randn(n, c, h, w).astype(np.float32)
    rois = _rois(rng, k, n, 16 * max(h, w), degenerate=True)
    xt = torch.from_numpy(x).requires_grad_(True)
    y = tv_ops.roi_align(xt, torch.from_numpy(rois), (P, P), scale, ratio, aligned)
    ref = tv_ops_np.roi_align(x, rois, P, scale, ratio, aligned)
    assert np.allclose(y.detach().numpy(), ref, rtol=2e-5, atol=2e-5)
    g = rng.randn(*ref.shape).astype(np.float32)
    y.backward(torch.from_numpy(g))
    gref = tv_ops_np.roi_align_backward(g.astype(np.float64), rois, x.shape, P, scale, ratio, aligned)
    assert np.allclose(xt.grad.numpy(), gref, rtol=2e-5, atol=2e-5)


@settings(max_examples=25, deadline=None)
@given(seed=st.integers(0, 10 ** 6), h=st.integers(3, 14), w=st.integers(3, 14), k=st.integers(1, 6),
       ratio=st.sampled_from([-1, 1, 2]))
def test_ps_roi_align_c_vs_numpy(seed, h, w, k, ratio):
    rng = np.random.RandomState(seed)
    n, co, scale = 2, 2, 1.0 / 16
    x = rng.randn(n, co * P * P, h, w).astype(np.float32)
    rois = _rois(rng, k, n, 16 * max(h, w))      # (degenerate boxes divide by zero in the library: NaN/inf, see below)
    xt = torch.from_numpy(x).requires_grad_(True)
    y = tv_ops.ps_roi_align(xt, torch.from_numpy(rois), (P, P), scale, ratio)
    ref = tv_ops_np.ps_roi_align(x, rois, P, scale, ratio)
    assert np.allclose(y.detach().numpy(), ref, rtol=2e-5, atol=2e-5)
    g = rng.randn(*ref.shape).astype(np.float32)
    y.backward(torch.from_numpy(g))
    gref = tv_ops_np.ps_roi_align_backward(g.astype(np.float64), rois, x.shape, P, scale, ratio)
    assert np.allclose(xt.grad.numpy(), gref, rtol=2e-5, atol=2e-5)


@settings(max_examples=40, deadline=None)
@given(seed=st.integers(0, 10 ** 6), m=st.integers(0, 60), thr=st.sampled_from([0.3, 0.5, 0.7]), classes=st.integers(1, 4))
def test_batched_nms_c_vs_numpy(seed, m, thr, classes):
    rng = np.random.RandomState(seed)
    a = rng.uniform(0, 100, size=(m, 2)).astype(np.float32)
    wh = rng.uniform(2, 60, size=(m, 2)).astype(np.float32)
    boxes = np.concatenate([a, a + wh], 1)
    if m > 6:
        boxes[3] = boxes[2]                      # identical boxes: IoU exactly 1
        boxes[5, :] = boxes[4, :] + np.float32(0.25)
    scores = rng.permutation(m).astype(np.float32) / max(m, 1)   # distinct scores: the order is defined
    idxs = rng.randint(0, classes, size=m).astype(np.float32)
    got = tv_ops.batched_nms(torch.from_numpy(boxes), torch.from_numpy(scores), torch.from_numpy(idxs), thr).numpy()
    ref = tv_ops_np.batched_nms(boxes, scores, idxs, thr)
    assert got.tolist() == ref.tolist()
    got1 = tv_ops.nms(torch.from_numpy(boxes), torch.from_numpy(scores), thr).numpy()
    assert got1.tolist() == tv_ops_np.nms(boxes, scores, thr).tolist()


# ---------------------------------------------------------------------------------------------------------------------
# hand-derived known answers (from the definitions in SURVEY.md Appendix C, no code involved)
# ---------------------------------------------------------------------------------------------------------------------
def _both(fn_name, *args):
    a = getattr(tv_ops, fn_name)(*[torch.from_numpy(v) if isinstance(v, np.ndarray) else v for v in args]).numpy()
    b = getattr(tv_ops_np, fn_name)(*[v[0] if isinstance(v, tuple) else v for v in args])
    return a, b


def test_constant_map_pools_to_the_constant():
    """Bilinear weights of a sample inside the map sum to 1 and the bin mean divides by the sample count: a constant map
    gives the constant for every box that lies inside the map (roi_align and ps_roi_align, any sampling ratio)."""
    x = np.full((1, 2 * P * P, 9, 9), 3.25, dtype=np.float32)
    rois = np.array([[0, 16.0, 24.0, 100.0, 120.0], [0, 40.0, 40.0, 56.0, 72.0]], dtype=np.float32)
    for ratio in (-1, 2):
        a, b = _both("roi_align", x, rois, (P, P), 1.0 / 16, ratio, False)
        assert np.allclose(a, 3.25, atol=1e-6) and np.allclose(b, 3.25, atol=1e-12)
        a, b = _both("ps_roi_align", x, rois, (P, P), 1.0 / 16, ratio)
        assert np.allclose(a, 3.25, atol=1e-6) and np.allclose(b, 3.25, atol=1e-12)


def test_single_sample_per_bin_reads_the_pixel_centre_rule():
    """roi_align, box (0,0)-(7,7) at scale 1, 7 x 7 bins, sampling_ratio 1: bin (ph, pw) is sampled once at
    (ph + 0.5, pw + 0.5) -> the mean of the 2 x 2 pixel block at (ph, pw).  On in[y, x] = 10 y + x that is
    10 (ph + .5) + (pw + .5).  With aligned=True everything shifts by half a pixel: exactly in[ph, pw]."""
    yy, xx = np.meshgrid(np.arange(9.0), np.arange(9.0), indexing="ij")
    x = (10 * yy + xx)[None, None].astype(np.float32)
    rois = np.array([[0, 0.0, 0.0, 7.0, 7.0]], dtype=np.float32)
    ph, pw = np.meshgrid(np.arange(7.0), np.arange(7.0), indexing="ij")
    a, b = _both("roi_align", x, rois, (P, P), 1.0, 1, False)
    assert np.allclose(a[0, 0], 10 * (ph + .5) + (pw + .5), atol=1e-5) and np.allclose(b[0, 0], 10 * (ph + .5) + (pw + .5))
    a, b = _both("roi_align", x, rois, (P, P), 1.0, 1, True)
    assert np.allclose(a[0, 0], 10 * ph + pw, atol=1e-5) and np.allclose(b[0, 0], 10 * ph + pw)


def test_roi_align_clamps_small_boxes_to_one_pixel_but_ps_roi_align_does_not():
    """A 0.25-pixel box: roi_align (aligned=False) widens it to 1 x 1 (bins of 1/7, one sample each since ceil(1/7) = 1);
    ps_roi_align keeps 0.25 (bins of 0.25/7) and shifts by -0.5.  On in[y, x] = x the pooled value is the sample's x."""
    x = np.tile(np.arange(12.0, dtype=np.float32), (12, 1))[None, None]
    x = np.repeat(x, P * P, axis=1)
    rois = np.array([[0, 4.0, 4.0, 4.25, 4.25]], dtype=np.float32)
    a, b = _both("roi_align", x[:, :1], rois, (P, P), 1.0, -1, False)
    want = 4.0 + (np.arange(7.0) + 0.5) / 7.0
    assert np.allclose(a[0, 0, 0], want, atol=1e-5) and np.allclose(b[0, 0, 0], want)
    a, b = _both("ps_roi_align", x, rois, (P, P), 1.0, -1)
    want = 3.5 + (np.arange(7.0) + 0.5) * 0.25 / 7.0
    assert np.allclose(a[0, 0, 0], want, atol=1e-5) and np.allclose(b[0, 0, 0], want)


def test_position_sensitive_channel_map():
    """Output (c, ph, pw) reads input channel (c * 7 + ph) * 7 + pw only: an input whose channel ch is the constant ch."""
    co = 3
    x = np.broadcast_to(np.arange(co * P * P, dtype=np.float32)[None, :, None, None], (1, co * P * P, 8, 8)).copy()
    rois = np.array([[0, 16.0, 16.0, 96.0, 96.0]], dtype=np.float32)
    want = np.arange(co * P * P, dtype=np.float64).reshape(co, P, P)
    a, b = _both("ps_roi_align", x, rois, (P, P), 1.0 / 16, -1)
    assert np.allclose(a[0], want, atol=1e-5) and np.allclose(b[0], want)


def test_single_pixel_impulse_weights():
    """An impulse at (3, 4): a sample at (y, x) picks up (1 - |y - 3|)(1 - |x - 4|) when within one pixel, else 0.
    Box (4,3)-(5,4) at scale 1, aligned, one sample per bin: samples at 3 - .5 + (ph + .5)/7, 4 - .5 + (pw + .5)/7."""
    x = np.zeros((1, 1, 8, 8), dtype=np.float32)
    x[0, 0, 3, 4] = 1.0
    rois = np.array([[0, 4.0, 3.0, 5.0, 4.0]], dtype=np.float32)
    sy = 3.0 - 0.5 + (np.arange(7.0) + 0.5) / 7.0
    sx = 4.0 - 0.5 + (np.arange(7.0) + 0.5) / 7.0
    want = np.outer(1 - np.abs(sy - 3), 1 - np.abs(sx - 4))
    a, b = _both("roi_align", x, rois, (P, P), 1.0, 1, True)
    assert np.allclose(a[0, 0], want, atol=1e-6) and np.allclose(b[0, 0], want)


def test_samples_outside_the_map_contribute_zero_but_still_count():
    """A box hanging over the right edge: samples with x > W read 0 and the bin still divides by the full sample count."""
    x = np.ones((1, 1, 4, 4), dtype=np.float32)
    rois = np.array([[0, 0.0, 0.0, 14.0, 7.0]], dtype=np.float32)   # width 14: bins of 2, ceil(2) = 2 samples in x
    a, b = _both("roi_align", x, rois, (P, P), 1.0, -1, False)
    # sample x positions: pw * 2 + 0.5, pw * 2 + 1.5; inside (<= 4 = W) for 0.5..3.5 (value 1) - also exactly x in (3, 4]
    # clamps to the last pixel - outside beyond: bins 0, 1 -> 1.0; bins >= 2 (x >= 4.5) -> 0
    want_row = np.array([1, 1, 0, 0, 0, 0, 0], dtype=np.float64)
    assert np.allclose(a[0, 0, 0], want_row, atol=1e-6) and np.allclose(b[0, 0, 0], want_row)


def test_nms_threshold_is_strict_and_areas_have_no_plus_one():
    """Boxes (0,0,2,2) and (1,0,3,2): intersection 2, union 6, IoU = 1/3 (area without +1; with the +1 pixel convention it
    would be 6 / 12 = 0.5).  Suppression needs IoU > thr: at thr = 1/3 exactly (in float32: 0.333333343..., and the IoU
    computes to the same float) the second box survives; just below it does not."""
    boxes = np.array([[0, 0, 2, 2], [1, 0, 3, 2]], dtype=np.float32)
    scores = np.array([0.9, 0.8], dtype=np.float32)
    third = float(np.float32(2.0) / np.float32(6.0))
    for fn in (lambda b, s, t: tv_ops.nms(torch.from_numpy(b), torch.from_numpy(s), t).numpy(), tv_ops_np.nms):
        assert fn(boxes, scores, third).tolist() == [0, 1]
        assert fn(boxes, scores, third - 1e-4).tolist() == [0]
        assert fn(boxes, scores, 0.45).tolist() == [0, 1]     # a +1-pixel IoU (0.5) would have suppressed it


def test_batched_nms_separates_classes_and_orders_by_score():
    boxes = np.array([[0, 0, 10, 10], [0, 0, 10, 10], [1, 1, 11, 11], [50, 50, 60, 60]], dtype=np.float32)
    scores = np.array([0.5, 0.9, 0.7, 0.6], dtype=np.float32)
    idxs = np.array([0, 1, 0, 0], dtype=np.float32)
    # class 0: box 2 (0.7) suppresses box 0 (IoU 81/119 = 0.68 > 0.5); box 3 is far away; class 1: box 1 alone
    for got in (tv_ops.batched_nms(torch.from_numpy(boxes), torch.from_numpy(scores), torch.from_numpy(idxs), 0.5).numpy(),
                tv_ops_np.batched_nms(boxes, scores, idxs, 0.5)):
        assert got.tolist() == [1, 2, 3]
    assert tv_ops.batched_nms(torch.zeros((0, 4)), torch.zeros(0), torch.zeros(0), 0.5).dtype == torch.int64
