"""Detector forward + backward for ``Darknet.forward(x, targets)`` under autograd.

Reference: ``module3_our_dataset/yolov3/models.py:181-267`` - with ``targets`` the module loop returns the summed YOLO
loss of every scale and torch autograd differentiates the whole backbone.  No reference script trains the detector
(``train.py:170`` keeps ``base_detector.eval()`` and Network detaches its outputs).  Both BatchNorm modes are covered, per
layer: **eval** = a per-channel affine ``scale * conv + shift`` whose ``weight`` / ``bias`` still receive gradients, exactly
like ``F.batch_norm(training=False)``; **train** = batch statistics (``me_bn_train_fwd_f32`` / ``me_bn_train_bwd_f32``) with
the running statistics updated by the module's momentum (0.9 in this model, models.py:38).

The graph is static, so forward and backward are explicit launch sequences over ``libmillieye_hip`` and autograd sees
one :class:`torch.autograd.Function` whose inputs are the detector parameters:

forward   every module output is kept (NHWC): ``me_conv2d_f32`` (folded BN + LeakyReLU epilogue), ``me_maxpool_f32``,
          ``me_upsample_f32``, ``me_add_f32`` (shortcut), ``me_copy_f32`` (route concat); every layer's packed weights come
          from ONE ``me_pack_conv_batch_f32`` launch (``DarknetEngine.refresh_train_weights``); the YOLO loss value, the
          metrics and the dense ``build_targets`` tensors come from ``YOLOLayer.loss_from_raw`` = ``me_yolo_loss_fwd_f32``.
backward  ``me_yolo_loss_bwd_f32`` seeds the raw detection maps; modules are walked in reverse:
          ``me_affine_act_bwd_f32`` (activation + BN-affine backward, d gamma / d beta / d bias),
          ``me_conv_wgrad_mfma_oihw_f32`` (weight gradient on the matrix pipe, on a second HIP stream beside the data
          gradient), the data gradient as ``me_conv2d_f32`` on the 180-degree rotated, transposed weights (3x3 / stride 2: one
          2x2 convolution for the four output-parity classes + a pixel shuffle; the 255 / 51-channel detection convs:
          ``me_gemm_f32``), ``me_upsample2_bwd_f32``, ``me_maxpool_bwd_f32``; gradients accumulate in slots with ownership.

"""
import os

import torch

from . import hip


def _ptr(t):
    return t.data_ptr() if t is not None else None


def _resolve(v, i):
    v = int(v)
    return i + v if v < 0 else v


def _add_into(dst, src_ptr, src_pitch, pixels, c, dst_off=0):
    """dst[..., dst_off:dst_off+c] += src (pitched)."""
    lib = hip.lib()
    dp = dst.data_ptr() + 4 * dst_off
    hip.check(lib.me_add_f32(dp, dst.shape[-1], src_ptr, src_pitch, dp, dst.shape[-1], pixels, c, hip.stream_ptr()),
              "me_add_f32")


class _State:
    pass


# Per-pass timing of one step (bench.py: the training line's own roofline).  While a dict sits in ``_TIMING[0]`` every
# forward / data-gradient / weight-gradient convolution is bracketed by HIP events on the launch stream and the weight
# gradients stay on that stream (a kernel's duration is then its own); outside it the brackets cost nothing.
_TIMING = [None]
_SUMS_SIDE = os.environ.get("MILLIEYE_AFFINE_SUMS_SIDE", "1") != "0"   # (A/B: 0 = the per-channel sums' second launch on the main stream)


class _timed:
    def __init__(self, kind, flops):
        self.kind, self.flops = kind, flops

    def __enter__(self):
        if _TIMING[0] is not None:
            self.a = torch.cuda.Event(enable_timing=True)
            self.a.record()
        return self

    def __exit__(self, *exc):
        if _TIMING[0] is not None:
            b = torch.cuda.Event(enable_timing=True)
            b.record()
            _TIMING[0].setdefault(self.kind, []).append((self.a, b, self.flops))
        return False


def profile_step_passes(step_fn):
    """Run ``step_fn()`` once with the brackets on: ``{"fwd" | "dgrad" | "wgrad": (milliseconds, flops, launches)}`` -
    the convolution kernels of each pass of the detector training step, sequential on one stream."""
    _TIMING[0] = {}
    try:
        step_fn()
        torch.cuda.synchronize()
        return {k: (sum(a.elapsed_time(b) for a, b, _f in v), sum(f for _a, _b, f in v), len(v)) for k, v in _TIMING[0].items()}
    finally:
        _TIMING[0] = None


class DetectorTrainer:
    def __init__(self, model):
        self.m = model

    # ------------------------------------------------------------------------------------------ forward
    def forward(self, x):
        if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4):
            raise hip.MeError("Darknet training forward needs a 4-D CUDA float32 tensor; there is no CPU fallback")
        m, lib = self.m, hip.lib()
        from .train_path import _bn_fwd
        eng = m.engine
        x = x.contiguous()
        defs = m.module_defs
        outs, raws, bn_state = [], {}, {}
        ws_t = torch.empty(int(lib.me_bn_workspace_bytes(2048)) + 256, dtype=torch.uint8, device=x.device)
        ws = ws_t.data_ptr() + (-ws_t.data_ptr()) % 256
        eng.refresh_train_weights(x.device)
        for i, d in enumerate(defs):
            t = d["type"]
            if t == "convolutional":
                cw = eng._conv_weights(i)  # (packed by refresh_train_weights above; a no-op stamp check here)
                cw.refresh(x.device)
                k, s = int(d["size"]), int(d["stride"])
                act = hip.ACT_LEAKY if d["activation"] == "leaky" else hip.ACT_LINEAR
                src = x if i == 0 else outs[i - 1]
                seq = m.module_list[i]
                bn = seq[1] if len(seq) > 1 and isinstance(seq[1], torch.nn.BatchNorm2d) else None
                if bn is not None and bn.training:  # batch statistics: plain convolution, then BN(train) + activation
                    cout = cw.wgt.shape[0]
                    if cout > 2048:
                        raise hip.MeError("train-mode BatchNorm: more than 2048 channels")
                    ones, zeros = _const_vectors(cout, x.device)
                    with _timed("fwd", _conv_flops(src, cw.wgt, s, i == 0)):
                        c_raw = hip.conv2d_auto(src, cw.wgt, ones, zeros, k, s, (k - 1) // 2, hip.ACT_LINEAR, x_nchw=(i == 0),
                                                wgt_tiled=cw.wgt_tiled)
                    y = torch.empty_like(c_raw)
                    rows = c_raw.numel() // cout
                    st_bn = _bn_fwd(c_raw, cout, rows, cout, bn, act, y, cout, ws)
                    bn_state[i] = (c_raw, st_bn)
                else:
                    with _timed("fwd", _conv_flops(src, cw.wgt, s, i == 0)):
                        y = hip.conv2d_auto(src, cw.wgt, cw.scale, cw.shift, k, s, (k - 1) // 2, act, x_nchw=(i == 0),
                                            wgt_tiled=cw.wgt_tiled)
            elif t == "maxpool":
                k, s = int(d["size"]), int(d["stride"])
                y = hip.maxpool(outs[i - 1], k, s, zero_ext=(k == 2 and s == 1))
            elif t == "upsample":
                y = hip.upsample(outs[i - 1], int(d["stride"]))
            elif t == "shortcut":
                a, b = outs[i - 1], outs[_resolve(d["from"], i)]
                y = torch.empty_like(a)
                hip.check(lib.me_add_f32(a.data_ptr(), a.shape[-1], b.data_ptr(), b.shape[-1], y.data_ptr(), y.shape[-1],
                                         a.numel() // a.shape[-1], a.shape[-1], hip.stream_ptr()), "me_add_f32")
            elif t == "route":
                parts = [outs[_resolve(v, i)] for v in d["layers"].split(",")]
                if len(parts) == 1:
                    y = parts[0]
                else:
                    ct = sum(p.shape[-1] for p in parts)
                    y = torch.empty(parts[0].shape[:3] + (ct,), device=x.device, dtype=torch.float32)
                    off = 0
                    for p in parts:
                        hip.check(lib.me_copy_f32(p.data_ptr(), p.shape[-1], y.data_ptr() + 4 * off, ct,
                                                  p.numel() // p.shape[-1], p.shape[-1], hip.stream_ptr()), "me_copy_f32")
                        off += p.shape[-1]
            elif t == "yolo":
                raws[i] = outs[i - 1]
                y = None
            else:
                raise ValueError(f"unsupported cfg block [{t}] at module {i}")
            outs.append(y)
        st = _State()
        st.x, st.outs, st.raws, st.bn_state = x, outs, raws, bn_state
        return st

    # ------------------------------------------------------------------------------------------ backward
    def backward(self, st, draws, reducer=None):
        """``draws``: {yolo module index: d loss / d raw map [N,G,G,A*(5+C)]}.  Returns {parameter name: gradient}.
        ``reducer`` (``parallel.GradChunkReducer``): every gradient is handed over as soon as its kernels are enqueued, so
        the data-parallel exchange of the deep layers runs beside the backward of the shallow ones; the caller collects
        the reduced tensors with ``reducer.finish()``."""
        m, lib = self.m, hip.lib()
        eng = m.engine
        defs, outs, x = m.module_defs, st.outs, st.x
        dev = x.device
        L = len(defs)
        # Gradient slots: dout[i] = (tensor, owned).  The first contribution to a slot is taken as it is - a fresh tensor is
        # owned, an alias of somebody else's gradient (a [shortcut] hands its dy to both inputs) is not - and only a second
        # contribution adds: in place when the slot owns its tensor, into a fresh sum otherwise.  Rounds 1-2 zero-filled a
        # buffer per module and added every contribution into it: ~150 fill + add launches per step that moved ~5 GB.
        dout = [None] * L
        grads = {}
        stream = hip.stream_ptr

        def add_out(a_t, b_t, out_t):
            c = a_t.shape[-1]
            hip.check(lib.me_add_f32(a_t.data_ptr(), a_t.stride(-2), b_t.data_ptr(), b_t.stride(-2), out_t.data_ptr(),
                                     out_t.stride(-2), a_t.numel() // c, c, stream()), "me_add_f32")

        def contribute(i, g, fresh):
            """``g``: a gradient w.r.t. module i's output (NHWC, dense or a channel slice); ``fresh``: nobody else holds it."""
            cur = dout[i]
            if cur is None:
                if fresh and g.is_contiguous():
                    dout[i] = (g, True)
                elif g.is_contiguous():
                    dout[i] = (g, False)
                else:  # a channel slice of a wider gradient ([route]): densify once
                    dense = torch.empty(g.shape, device=dev, dtype=torch.float32)
                    c = g.shape[-1]
                    hip.check(lib.me_copy_f32(g.data_ptr(), g.stride(-2), dense.data_ptr(), c, g.numel() // c, c, stream()),
                              "me_copy_f32")
                    dout[i] = (dense, True)
                return
            t_cur, owned = cur
            if owned:
                add_out(t_cur, g, t_cur)
            else:
                total = torch.empty_like(t_cur)
                add_out(t_cur, g, total)
                dout[i] = (total, True)

        x_nhwc = None
        main = torch.cuda.current_stream(dev)
        side = _side_stream(dev) if os.environ.get("MILLIEYE_WGRAD_STREAM", "1") != "0" and _TIMING[0] is None else None
        aff_layer_ws = self.m.__dict__.setdefault("_affine_layer_ws", {})
        for i in reversed(range(L)):
            d = defs[i]
            t = d["type"]
            if t == "yolo":
                contribute(i - 1, draws[i], True)
                continue
            if dout[i] is None:
                continue
            dy, dy_owned = dout[i]
            if t == "convolutional":
                seq = m.module_list[i]
                conv = seq[0]
                bn = seq[1] if len(seq) > 1 and isinstance(seq[1], torch.nn.BatchNorm2d) else None
                cw = eng._conv_weights(i)
                k, s = int(d["size"]), int(d["stride"])
                pad = (k - 1) // 2
                act = hip.ACT_LEAKY if d["activation"] == "leaky" else hip.ACT_LINEAR
                y = outs[i]
                n, ho, wo, cout = y.shape
                rows = n * ho * wo
                # eval-mode BatchNorm: dc = dy * act'(y) * scale is element-wise - in place when this slot owns its gradient
                dc = dy if (dy_owned and i not in st.bn_state) else torch.empty_like(y)
                sums_job = None
                if i in st.bn_state:  # train-mode BatchNorm: gradient through the batch statistics
                    from .train_path import _bn_bwd
                    c_raw, st_bn = st.bn_state[i]
                    ws = torch.empty(int(lib.me_bn_workspace_bytes(cout)) + 256, dtype=torch.uint8, device=dev)
                    dgamma, dshift = _bn_bwd(c_raw, cout, dy, cout, rows, cout, bn, st_bn, act, dc, cout,
                                             ws.data_ptr() + (-ws.data_ptr()) % 256)
                else:
                    dshift = torch.empty(cout, device=dev)
                    dgamma = torch.empty(cout, device=dev) if bn is not None else None
                    need_a = int(lib.me_affine_bwd_workspace_bytes(rows, cout))
                    # d gamma / d beta are not on the data path: with the weight gradient on the side stream the second level of
                    # their sums goes there too (me_affine_bwd_sums_f32) and the main stream is one launch per layer shorter; the
                    # partial rows then live in a scratch of the layer's own (kept on the model) until the side stream has read them
                    defer_sums = side is not None and _SUMS_SIDE
                    if defer_sums:
                        ws = aff_layer_ws.get(i)
                        if ws is None or ws.numel() < need_a:
                            ws = aff_layer_ws[i] = torch.empty(max(need_a, 16), dtype=torch.uint8, device=dev)
                        # (the flag must be the predicate me_affine_act_bwd_f32 derives for ITS plan - channels, pitches and
                        #  16-byte alignment of y / dy / dc - or the two calls could disagree on the chunk count)
                        v4 = cout % 4 == 0 and y.data_ptr() % 16 == 0 and dy.data_ptr() % 16 == 0 and dc.data_ptr() % 16 == 0
                        sums_job = (ws, rows, cout, int(v4), dshift, dgamma)
                    else:
                        ws = torch.empty(need_a, dtype=torch.uint8, device=dev)
                    gam = bn.weight.detach() if bn is not None else None
                    bet = bn.bias.detach() if bn is not None else None
                    hip.check(lib.me_affine_act_bwd_f32(y.data_ptr(), cout, dy.data_ptr(), cout, rows, cout,
                                                        cw.scale.data_ptr() if bn is not None else None, _ptr(gam),
                                                        _ptr(bet), act, dc.data_ptr(), cout,
                                                        None if defer_sums else dshift.data_ptr(),
                                                        None if defer_sums else _ptr(dgamma), ws.data_ptr(), stream()),
                              "me_affine_act_bwd_f32")
                if bn is not None:
                    grads[f"module_list.{i}.batch_norm_{i}.weight"] = dgamma
                    grads[f"module_list.{i}.batch_norm_{i}.bias"] = dshift
                else:
                    grads[f"module_list.{i}.conv_{i}.bias"] = dshift
                # weight gradient, written in the parameter's own OIHW layout by the slab reduction
                if i == 0:
                    if x_nhwc is None:
                        x_nhwc = x.permute(0, 2, 3, 1).contiguous()
                    xin = x_nhwc
                else:
                    xin = outs[i - 1]
                _, h, w, cin = xin.shape
                if side is not None:
                    # the weight gradient only feeds the optimizer: it runs on a second stream beside the data gradient of the
                    # same layer (both read dc), so the tails / slab sums of one fill the other's idle CUs
                    side.wait_stream(main)
                    with torch.cuda.stream(side):
                        if sums_job is not None:
                            ws_j, rows_j, c_j, v4_j, ds_j, dg_j = sums_job
                            hip.check(lib.me_affine_bwd_sums_f32(ws_j.data_ptr(), rows_j, c_j, v4_j, ds_j.data_ptr(), _ptr(dg_j),
                                                                 hip.stream_ptr()), "me_affine_bwd_sums_f32")
                        grads[f"module_list.{i}.conv_{i}.weight"] = hip.conv_wgrad(xin, dc, k, s, pad, oihw=True)
                    dc.record_stream(side)
                    xin.record_stream(side)
                else:
                    with _timed("wgrad", 2.0 * rows * cout * k * k * cin):
                        grads[f"module_list.{i}.conv_{i}.weight"] = hip.conv_wgrad(xin, dc, k, s, pad, oihw=True)
                if reducer is not None:
                    vec_stream = side if sums_job is not None else main   # (where d gamma / d beta were finished)
                    if bn is not None:
                        reducer.push(f"module_list.{i}.batch_norm_{i}.weight", dgamma, vec_stream)
                        reducer.push(f"module_list.{i}.batch_norm_{i}.bias", dshift, vec_stream)
                    else:
                        reducer.push(f"module_list.{i}.conv_{i}.bias", dshift, vec_stream)
                    reducer.push(f"module_list.{i}.conv_{i}.weight", grads[f"module_list.{i}.conv_{i}.weight"],
                                 side if side is not None else main)
                dout[i] = None
                if i == 0:
                    continue
                # data gradient
                tm = _timed("dgrad", 2.0 * rows * cout * k * k * cin).__enter__()
                if cout % 4 != 0:
                    if k != 1 or s != 1:
                        raise NotImplementedError(f"conv {i}: data gradient for cout={cout} (not a multiple of 4) needs k=1")
                    dx = torch.empty((n, h, w, cin), device=dev, dtype=torch.float32)
                    hip.check(lib.me_gemm_f32(0, 0, rows, cin, cout, 1.0, dc.data_ptr(), cout, cw.wgt.data_ptr(), cin, 0.0,
                                              dx.data_ptr(), cin, stream()), "me_gemm_f32")
                    contribute(i - 1, dx, True)
                else:
                    parity = s == 2 and k == 3 and pad == 1 and h == 2 * ho and w == 2 * wo and cin % 4 == 0
                    if parity:
                        wt = wt_tiled = None
                    elif cw.rot is not None:  # [cin][k][k][cout] rotated 180 degrees, from the step's pack launch
                        wt, wt_tiled = cw.rot, cw.rot_tiled
                    else:
                        wt, wt_tiled = cw.wgt.flip(1, 2).permute(3, 1, 2, 0).contiguous(), None
                    ones, zeros = _const_vectors(cin, dev)
                    # a gradient already waiting in the consumer's slot rides in as the conv's residual: dx + existing in one
                    # launch (in place when the slot owns its tensor: every element is read, then written, by one lane)
                    prev = dout[i - 1]
                    res = prev[0] if prev is not None and prev[0].is_contiguous() else None
                    out_t = res if (prev is not None and prev[1] and res is not None) else None
                    if s == 1:
                        dx = hip.conv2d_auto(dc, wt, ones, zeros, k, 1, k - 1 - pad, hip.ACT_LINEAR, residual=res, out=out_t,
                                             wgt_tiled=wt_tiled)
                    elif parity:
                        # Stride-2 transposed convolution by output parity (round 3; it ran as a 3x3 correlation over the
                        # zero-interleaved gradient: 9 taps on 4x the pixels = 4x the forward FLOPs).  dx[2a+py, 2b+px] only
                        # sees the taps with ky = py + 1 (mod 2), kx = px + 1 (mod 2): per axis one tap (W[1], offset 0) for the
                        # even positions, two (W[2] at offset 0, W[0] at offset +1) for the odd ones.  All four parity classes
                        # come out of ONE 2x2 convolution of dc (pad 1) with 4 * cin output channels - 16 tap-units per
                        # gradient pixel instead of 36 - followed by a pixel shuffle of the cropped result.
                        fresh = cw.parity is not None and cw.parity_stamp == cw._stamp
                        pw = cw.parity if fresh else _parity_weights(cw.wgt)  # (from the step's pack launch when it ran)
                        # (round 5: 7 of the 16 (class, tap) pairs of that 2x2 filter are structurally zero - class (py, px)
                        #  has taps only where ky(py, i) and kx(px, j) exist: 1 + 2 + 2 + 4 = 9 of 16 - and the kernel skips
                        #  them: me_conv_desc.tap_mask, tap t = 2 * i + j)
                        masks = _PARITY_TAP_MASKS if cin % 32 == 0 and cout % 16 == 0 and _PARITY_MASKS_ON else None
                        dx4 = hip.conv2d_auto(dc, pw, _const_vectors(4 * cin, dev)[0],
                                              _const_vectors(4 * cin, dev)[1], 2, 1, 1, hip.ACT_LINEAR,
                                              tap_masks=(cin, masks) if masks is not None else None)
                        dx = dx4[:, 1:, 1:, :].reshape(n, ho, wo, 2, 2, cin).permute(0, 1, 3, 2, 4, 5).reshape(n, h, w, cin)
                        dx = dx.contiguous()
                        res = None  # (accumulated below like any fresh contribution)
                    else:
                        # transposed convolution = stride-1 correlation over the zero-interleaved gradient
                        pp = k - 1 - pad
                        z = torch.zeros((n, h + k - 1, w + k - 1, cout), device=dev, dtype=torch.float32)
                        z[:, pp:pp + s * ho:s, pp:pp + s * wo:s, :] = dc
                        dx = hip.conv2d_auto(z, wt, ones, zeros, k, 1, 0, hip.ACT_LINEAR, residual=res, out=out_t,
                                             wgt_tiled=wt_tiled)
                    if res is not None:
                        dout[i - 1] = (dx, True)
                    else:
                        contribute(i - 1, dx, True)
                tm.__exit__()
                continue
            elif t == "shortcut":
                # both inputs receive dy itself; two slots now alias one tensor, so NEITHER may write into it (the conv in
                # front of the shortcut is visited before the block's input and would otherwise run its backward in place)
                contribute(i - 1, dy, False)
                contribute(_resolve(d["from"], i), dy, False)
            elif t == "route":
                srcs = [_resolve(v, i) for v in d["layers"].split(",")]
                if len(srcs) == 1:
                    contribute(srcs[0], dy, dy_owned)
                else:
                    off = 0
                    for sidx in srcs:
                        c = outs[sidx].shape[-1]
                        contribute(sidx, dy[..., off:off + c], False)
                        off += c
            elif t == "upsample":
                if int(d["stride"]) != 2:
                    raise NotImplementedError("upsample backward: stride 2 only")
                n, h, w, c = outs[i - 1].shape
                g = torch.zeros((n, h, w, c), device=dev, dtype=torch.float32)
                hip.check(lib.me_upsample2_bwd_f32(dy.data_ptr(), c, g.data_ptr(), c, n, h, w, c, stream()),
                          "me_upsample2_bwd_f32")
                contribute(i - 1, g, True)
            elif t == "maxpool":
                k, s = int(d["size"]), int(d["stride"])
                zero_ext = 1 if (k == 2 and s == 1) else 0
                xin = outs[i - 1]
                n, h, w, c = xin.shape
                g = torch.zeros((n, h, w, c), device=dev, dtype=torch.float32)
                hip.check(lib.me_maxpool_bwd_f32(xin.data_ptr(), c, dy.data_ptr(), c, g.data_ptr(), c, n, h, w,
                                                 c, k, s, 0 if zero_ext else (k - 1) // 2, zero_ext, stream()),
                          "me_maxpool_bwd_f32")
                contribute(i - 1, g, True)
            dout[i] = None  # free as we go
        if side is not None:
            main.wait_stream(side)
        return grads


def _conv_flops(src, wgt_ohwi, stride, nchw):
    """2 * output pixels * cout * k * k * cin of one convolution (``src``: its input, NHWC - NCHW for the first layer)."""
    n = src.shape[0]
    h, w = (src.shape[2], src.shape[3]) if nchw else (src.shape[1], src.shape[2])
    cout, k, _, cin = wgt_ohwi.shape
    return 2.0 * n * (h // stride) * (w // stride) * cout * k * k * cin


_SIDE = {}


def _side_stream(dev):
    key = str(dev)
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=dev)
    return _SIDE[key]


_PARITY_IDX = {}
# set taps (bit 2 * i + j) of the four parity classes 2 * py + px of _parity_weights: tap (i, j) of class (py, px) exists when
# ky(py, i) and kx(px, j) do - (0, 0): i = 0 only; (1, *): both
_PARITY_TAP_MASKS = tuple(sum(1 << (2 * i + j) for i in (0, 1) for j in (0, 1) if (py == 1 or i == 0) and (px == 1 or j == 0))
                          for py in (0, 1) for px in (0, 1))   # (0b0001, 0b0011, 0b0101, 0b1111)
_PARITY_MASKS_ON = os.environ.get("MILLIEYE_PARITY_MASKS", "1") != "0"  # (A/B: 0 = multiply the zero taps like round 3)


def _parity_weights(wgt_ohwi):
    """OHWI weights [cout, 3, 3, cin] of a stride-2 / pad-1 convolution -> the weights [4 * cin, 2, 2, cout] of the 2x2
    convolution that computes the four output-parity classes of its data gradient at once (class = 2 * py + px, channel
    class * cin + c): tap (i, j) of class (py, px) is W[ky(py, i)][kx(px, j)] with ky(0, 0) = 1, ky(0, 1) = none (zero),
    ky(1, 0) = 2, ky(1, 1) = 0 - see the call site."""
    cout, k, _, cin = wgt_ohwi.shape
    dev = wgt_ohwi.device
    idx = _PARITY_IDX.get(str(dev))
    if idx is None:
        tap_of = {(0, 0): 1, (0, 1): None, (1, 0): 2, (1, 1): 0}
        flat = []
        for py in (0, 1):
            for px in (0, 1):
                for i in (0, 1):
                    for j in (0, 1):
                        ky, kx = tap_of[(py, i)], tap_of[(px, j)]
                        flat.append(9 if ky is None or kx is None else ky * 3 + kx)  # 9 = the appended zero tap
        idx = _PARITY_IDX[str(dev)] = torch.tensor(flat, device=dev)
    wz = torch.cat((wgt_ohwi.reshape(cout, 9, cin), torch.zeros((cout, 1, cin), device=dev)), 1)   # [o, 10, c]
    sel = wz.index_select(1, idx).reshape(cout, 4, 2, 2, cin)                                       # [o, cls, i, j, c]
    return sel.permute(1, 4, 2, 3, 0).reshape(4 * cin, 2, 2, cout).contiguous()


_CONST = {}


def _const_vectors(n, device):
    """(ones[n], zeros[n]) fp32 on ``device``: identity scale / shift of the plain convolutions of the training path, cached
    (they are read-only; re-creating them cost four fill launches per layer and step)."""
    key = (n, str(device))
    hit = _CONST.get(key)
    if hit is None:
        hit = _CONST[key] = (torch.ones(n, device=device), torch.zeros(n, device=device))
    return hit


class _DarknetLoss(torch.autograd.Function):
    """loss = Darknet.forward(x, targets)[0] with the detector parameters as differentiable inputs."""

    @staticmethod
    def forward(ctx, model, x, targets, names, *params):
        if getattr(model, "compute_dtype", "f32") != "f32":   # 16-bit storage mode: the mixed-precision step (detector_train16.py)
            from .detector_train16 import DetectorTrainer16
            trainer = DetectorTrainer16(model)
        else:
            trainer = DetectorTrainer(model)
        st = trainer.forward(x)
        loss = 0
        seeds = {}
        for layer, (idx, raw) in zip(model.yolo_layers, sorted(st.raws.items())):
            layer.img_dim = x.shape[2]
            value, bt = layer.loss_from_raw(raw, targets, return_targets=True)
            loss = loss + value
            seeds[idx] = bt
        ctx.trainer, ctx.st, ctx.seeds, ctx.names = trainer, st, seeds, names
        ctx.model = model
        model._train_state = st  # Darknet._forward_train reads the decoded rows / feature tap from it
        return loss.detach().clone()

    @staticmethod
    def backward(ctx, grad_loss):
        model, st, lib = ctx.model, ctx.st, hip.lib()
        gscale = float(grad_loss)
        draws = {}
        for layer, (idx, raw) in zip(model.yolo_layers, sorted(st.raws.items())):
            bt = ctx.seeds[idx]
            n, g, _, ch = raw.shape
            draw = torch.empty_like(raw)
            hip.check(lib.me_yolo_loss_bwd_f32(raw.data_ptr(), ch, n, g, layer.num_anchors, layer.num_classes,
                                               bt["obj"].data_ptr(), bt["noobj"].data_ptr(), bt["tx"].data_ptr(),
                                               bt["ty"].data_ptr(), bt["tw"].data_ptr(), bt["th"].data_ptr(),
                                               bt["tcls"].data_ptr(), bt["tconf"].data_ptr(), float(bt["n_obj"]),
                                               float(bt["n_noobj"]), float(layer.obj_scale), float(layer.noobj_scale),
                                               gscale, draw.data_ptr(), ch, hip.stream_ptr()), "me_yolo_loss_bwd_f32")
            draws[idx] = draw
        reducer = model.__dict__.get("_grad_reducer")  # parallel.overlap_detector_allreduce: exchange beside the backward
        if reducer is not None:
            reducer.begin(st.x.device)
        grads = ctx.trainer.backward(st, draws, reducer)
        if reducer is not None:
            grads.update(reducer.finish())
        out = []
        for name, needs in zip(ctx.names, ctx.needs_input_grad[4:]):
            g = grads.get(name) if needs else None
            out.append(g)
        return (None, None, None, None) + tuple(out)
