"""ORACLE (test infrastructure): a SECOND, independently written restatement of the three torchvision 0.6 operators the
reference calls (``roi_align``, ``ps_roi_align``, ``batched_nms`` / ``nms``; call sites module3_our_dataset/my_models.py:495-496,
utils/utils.py:372) - float64, one output sample at a time, written from the operator definitions in SURVEY.md Appendix C
rather than from oracle/tv_ops.c.

PARITY UNPINNED, like tv_ops.c: torchvision is an un-vendored dependency that is absent from this image, so neither
restatement can be checked against the library itself.  What this file buys: two implementations with different structure
(C, float32, forward + scatter backward there; numpy, float64, closed-form sampling weights here - the backward is the
transpose of the forward's weight matrix) that must agree on fuzzed inputs (tests/test_tv_ops_cpu.py), plus hand-derived
known answers.  A slip in either reading of the definition shows up as a disagreement.
"""
import math

import numpy as np


def _bilinear_weights(h, w, y, x):
    """The four (index, weight) pairs torchvision's ``bilinear_interpolate`` uses for sample point (y, x) of an h x w map;
    empty outside [-1, h] x [-1, w]."""
    if y < -1.0 or y > h or x < -1.0 or x > w:
        return []
    y = max(y, 0.0)
    x = max(x, 0.0)
    y0, x0 = int(y), int(x)
    if y0 >= h - 1:
        y0 = y1 = h - 1
        y = float(y0)
    else:
        y1 = y0 + 1
    if x0 >= w - 1:
        x0 = x1 = w - 1
        x = float(x0)
    else:
        x1 = x0 + 1
    ly, lx = y - y0, x - x0
    hy, hx = 1.0 - ly, 1.0 - lx
    return [((y0, x0), hy * hx), ((y0, x1), hy * lx), ((y1, x0), ly * hx), ((y1, x1), ly * lx)]


def _bin_samples(start_h, start_w, bin_h, bin_w, ph, pw, gh, gw):
    for iy in range(gh):
        yy = start_h + ph * bin_h + (iy + 0.5) * bin_h / gh
        for ix in range(gw):
            xx = start_w + pw * bin_w + (ix + 0.5) * bin_w / gw
            yield yy, xx


def roi_align_weights(h, w, box, pooled, spatial_scale, sampling_ratio=-1, aligned=False):
    """Sparse sampling matrix of one RoI: {(ph, pw): {(y, x): weight}} such that out[c, ph, pw] = sum w * in[c, y, x]."""
    P = pooled
    off = 0.5 if aligned else 0.0
    x1, y1, x2, y2 = (float(v) * spatial_scale - off for v in box)
    rw, rh = x2 - x1, y2 - y1
    if not aligned:
        rw, rh = max(rw, 1.0), max(rh, 1.0)
    bh, bw = rh / P, rw / P
    gh = sampling_ratio if sampling_ratio > 0 else int(math.ceil(rh / P))
    gw = sampling_ratio if sampling_ratio > 0 else int(math.ceil(rw / P))
    count = max(gh * gw, 1)
    out = {}
    for ph in range(P):
        for pw in range(P):
            acc = {}
            for yy, xx in _bin_samples(y1, x1, bh, bw, ph, pw, gh, gw):
                for idx, wt in _bilinear_weights(h, w, yy, xx):
                    acc[idx] = acc.get(idx, 0.0) + wt / count
            out[(ph, pw)] = acc
    return out


def ps_roi_align_weights(h, w, box, pooled, spatial_scale, sampling_ratio=-1):
    P = pooled
    x1, y1, x2, y2 = (float(v) * spatial_scale - 0.5 for v in box)
    rw, rh = x2 - x1, y2 - y1  # no clamp
    bh, bw = rh / P, rw / P
    gh = sampling_ratio if sampling_ratio > 0 else int(math.ceil(rh / P))
    gw = sampling_ratio if sampling_ratio > 0 else int(math.ceil(rw / P))
    count = gh * gw  # unguarded, like the library: a degenerate box divides by zero
    out = {}
    for ph in range(P):
        for pw in range(P):
            acc = {}
            for yy, xx in _bin_samples(y1, x1, bh, bw, ph, pw, gh, gw):
                for idx, wt in _bilinear_weights(h, w, yy, xx):
                    acc[idx] = acc.get(idx, 0.0) + (wt / count if count else float("nan"))
            out[(ph, pw)] = acc if count else None
    return out


def roi_align(inp, rois, pooled, spatial_scale=1.0, sampling_ratio=-1, aligned=False):
    """inp [N,C,H,W], rois [K,5] (batch, x1, y1, x2, y2) -> [K,C,P,P] float64."""
    inp = np.asarray(inp, dtype=np.float64)
    _, C, H, W = inp.shape
    out = np.zeros((len(rois), C, pooled, pooled))
    for k, r in enumerate(np.asarray(rois, dtype=np.float64)):
        wts = roi_align_weights(H, W, r[1:], pooled, spatial_scale, sampling_ratio, aligned)
        for (ph, pw), acc in wts.items():
            for (y, x), wt in acc.items():
                out[k, :, ph, pw] += wt * inp[int(r[0]), :, y, x]
    return out


def roi_align_backward(grad, rois, shape, pooled, spatial_scale=1.0, sampling_ratio=-1, aligned=False):
    """Transpose of :func:`roi_align`: grad [K,C,P,P] -> d inp (``shape`` = [N,C,H,W])."""
    g = np.zeros(shape)
    _, C, H, W = shape
    for k, r in enumerate(np.asarray(rois, dtype=np.float64)):
        wts = roi_align_weights(H, W, r[1:], pooled, spatial_scale, sampling_ratio, aligned)
        for (ph, pw), acc in wts.items():
            for (y, x), wt in acc.items():
                g[int(r[0]), :, y, x] += wt * grad[k, :, ph, pw]
    return g


def ps_roi_align(inp, rois, pooled, spatial_scale=1.0, sampling_ratio=-1):
    """inp [N, C_out * P * P, H, W] -> [K, C_out, P, P]; output (c, ph, pw) reads input channel (c * P + ph) * P + pw."""
    inp = np.asarray(inp, dtype=np.float64)
    _, C, H, W = inp.shape
    assert C % (pooled * pooled) == 0
    co = C // (pooled * pooled)
    out = np.zeros((len(rois), co, pooled, pooled))
    for k, r in enumerate(np.asarray(rois, dtype=np.float64)):
        wts = ps_roi_align_weights(H, W, r[1:], pooled, spatial_scale, sampling_ratio)
        for (ph, pw), acc in wts.items():
            if acc is None:
                out[k, :, ph, pw] = np.nan
                continue
            for c in range(co):
                cin = (c * pooled + ph) * pooled + pw
                out[k, c, ph, pw] = sum(wt * inp[int(r[0]), cin, y, x] for (y, x), wt in acc.items())
    return out


def ps_roi_align_backward(grad, rois, shape, pooled, spatial_scale=1.0, sampling_ratio=-1):
    g = np.zeros(shape)
    _, C, H, W = shape
    co = C // (pooled * pooled)
    for k, r in enumerate(np.asarray(rois, dtype=np.float64)):
        wts = ps_roi_align_weights(H, W, r[1:], pooled, spatial_scale, sampling_ratio)
        for (ph, pw), acc in wts.items():
            if acc is None:
                continue
            for c in range(co):
                cin = (c * pooled + ph) * pooled + pw
                for (y, x), wt in acc.items():
                    g[int(r[0]), cin, y, x] += wt * grad[k, c, ph, pw]
    return g


def nms(boxes, scores, thr, arith=np.float32):
    """Greedy NMS: visit in descending score (stable for ties, the order torch's CPU sort gives), suppress j when
    IoU(i, j) > thr (strict), area without +1; IoU arithmetic in ``arith`` (the library's kernel is templated on the box
    type: float32 here, as in the reference).  Returns kept indices in visiting order."""
    b = np.asarray(boxes, dtype=arith)
    order = np.argsort(-np.asarray(scores, dtype=np.float64), kind="stable")
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    dead = np.zeros(len(b), dtype=bool)
    keep = []
    for a in order:
        if dead[a]:
            continue
        keep.append(int(a))
        for c in order:
            if dead[c] or c == a:
                continue
            iw = max(arith(0), min(b[a, 2], b[c, 2]) - max(b[a, 0], b[c, 0]))
            ih = max(arith(0), min(b[a, 3], b[c, 3]) - max(b[a, 1], b[c, 1]))
            inter = arith(iw * ih)
            if inter / arith(area[a] + area[c] - inter) > arith(thr):
                dead[c] = True
        dead[a] = True
    return np.asarray(keep, dtype=np.int64)


def batched_nms(boxes, scores, idxs, thr):
    """torchvision.ops.boxes.batched_nms: classes are separated by shifting every box by ``idx * (max coordinate + 1)``."""
    b = np.asarray(boxes, dtype=np.float32)
    if len(b) == 0:
        return np.zeros((0,), dtype=np.int64)
    off = np.asarray(idxs, dtype=np.float32) * (b.max() + np.float32(1))
    return nms(b + off[:, None], scores, thr)
