"""Detector forward + backward in a 16-bit storage mode: ``Darknet.forward(x, targets)`` under autograd with
``model.compute_dtype = "bf16"`` (or ``"f16"``) - the mixed-precision twin of :mod:`millieye_amd.detector_train` (round 5).

Reference: ``module3_our_dataset/yolov3/models.py:181-267`` under autograd (the detector is differentiable in the reference;
no reference script trains it, ``train.py:170``).  BASELINE north_star names "backbone forward/backward"; the 16-bit storage
modes were inference only until this round (VERDICT r04 "missing" item 2).

What is stored how:

==========================  ===========================================================================================
activations (every module)  bfloat16 / IEEE half NHWC, one RNE rounding per stored tensor (``me_conv2d_h16`` epilogue)
detection maps, YOLO loss   float32 (``y_f32`` output of the three detection convolutions; ``me_yolo_loss_*_f32``)
activation gradients        16-bit (``me_affine_act_bwd_h16`` and the data-gradient convolutions round once per tensor)
parameters, their grads     float32 (master weights; ``d gamma`` / ``d beta`` / ``d bias`` summed in double from the 16-bit
                            tensors; weight gradients = exact products of 16-bit values, fp32 accumulation, fp32 slabs
                            summed in a fixed order: ``me_conv_wgrad_h16``)
matrix work                 forward, data gradient and weight gradient on ``v_mfma_f32_32x32x16_bf16`` / ``_f16`` (16x the
                            fp32 matrix rate); the stem's and the detection convolutions' weight gradients on the fp32 pipe
==========================  ===========================================================================================

BatchNorm: eval mode is folded into the convolution (its ``weight`` / ``bias`` still receive gradients, like
``F.batch_norm(training=False)``); train() mode (round 6) takes float32 batch statistics over the convolution's raw float32 sums
(``y_f32``), normalises / differentiates in float32 (``me_bn_train_fwd_f32`` / ``_bwd_f32``) and stores the activation and its
gradient in the 16-bit type - slower than the folded form (float32 intermediates), there for the reference's ``model.train()`` semantics.  Parity bar (tests/test_gpu_train16.py): the loss
within 1 % and every parameter gradient at cosine >= 0.99 of the fp32 HIP step's on the same inputs.
"""
import ctypes as C
import os

import torch

from . import hip
from .detector_train import (_PARITY_IDX, _PARITY_MASKS_ON, _PARITY_TAP_MASKS, _State, _const_vectors, _conv_flops, _parity_weights, _resolve, _side_stream, _timed,  # noqa: F401
                             _TIMING)

_HALF = {"bf16": torch.bfloat16, "f16": torch.float16}
_AUTO16 = {}
_MORE_TILES = os.environ.get("MILLIEYE_TRAIN16_MORE_TILES", "1") != "0"   # (A/B: 0 = the round's first candidate list)
_PACK16 = os.environ.get("MILLIEYE_PACK16", "1") != "0"   # (A/B: 0 = fp32 pack + a conversion pass)
_DIRECT = os.environ.get("MILLIEYE_WGRAD16_DIRECT", "1") != "0"   # (A/B: 0 = the weight gradient under torch.cuda.stream(side))
_SUMS_SIDE = os.environ.get("MILLIEYE_AFFINE_SUMS_SIDE", "1") != "0"   # (A/B: 0 = the per-channel sums' second launch on the main stream)
_WGRAD16 = os.environ.get("MILLIEYE_WGRAD16", "1") != "0"   # (A/B: 0 = weight gradients by the fp32 kernels on fp32 copies)


def _ptr(t):
    return t.data_ptr() if t is not None else None


def conv16_auto(x, wgt, scale, shift, ksize, stride, pad, act, residual=None, y_f32=False, x_nchw=False, tap_masks=None):
    """``hip.conv2d_h16`` with the (tile, split_k) pair measured the first time a layer shape is seen (the training path has no engine
    plan whose autotuner would do it).  Candidates: the per-tap tiles with 1 / 2 / 4 K splits and, for 3x3 / stride-1 layers, the
    patch-resident tiles (their tiled weight copy is made per call: a few microseconds against tens)."""
    if x_nchw or wgt.shape[3] <= 4:
        return hip.conv2d_h16(x, wgt, scale, shift, ksize, stride, pad, act, residual=residual, y_f32=y_f32, x_nchw=x_nchw)
    key = (tuple(x.shape), x.stride(2), wgt.shape[0], ksize, stride, pad, residual is not None, y_f32, x.dtype, tap_masks)
    hit = _AUTO16.get(key)
    if hit is None:
        cin = wgt.shape[3]
        # candidates (tile, split, masked): with ``tap_masks`` (the stride-2 data gradient's parity convolution: filters whose zero
        # taps the masked instances skip, me_conv16_desc.tap_mask) every whole per-tap tile is tried both ways - a launch that fits
        # one round of tiles lasts as long as its four-tap class, so the small maps may keep their K splits
        cands = [(0, 0, False)] + [(t, sp, False) for t in ((1, 2, 3, 4, 11, 12, 13, 14) if cin % 64 == 0 else (1, 2, 3, 4)) for sp in (1, 2, 4)]
        if tap_masks is not None:
            cands += [(t, 1, True) for t in ((1, 2, 3, 11, 12, 13) if cin % 64 == 0 else (1, 2, 3)) if tap_masks[0] % (64 if t in (2, 3, 12, 13) else 128) == 0]
        if ksize == 3 and stride == 1 and pad == 1 and not y_f32 and cin % 32 == 0:
            cands += [(t, 1, False) for t in (221, 201, 431, 131, 121, 621)]
        if _MORE_TILES and not y_f32:
            # the streaming kernels of the inference plans (the library refuses the shapes they have no instance for): weights in
            # registers for the pointwise layers without a residual (tile 50), for the 3x3 layers with <= 64 input channels (60)
            if ksize == 1 and stride == 1 and residual is None:
                cands.append((50, 1, False))
            if ksize == 3 and cin <= 64:
                cands.append((60, 1, False))
            cands += [(15, sp, False) for sp in (1, 2)]   # 192 x 128
        best = (float("inf"), 0, 0, False)
        scratch = None
        torch.cuda.synchronize()
        for tile, split, masked in cands:
            tm = tap_masks if masked else None
            try:
                wt = hip.tile_weights_h16(wgt) if tile >= 100 else None
                for _ in range(2):
                    scratch = hip.conv2d_h16(x, wgt, scale, shift, ksize, stride, pad, act, residual=residual, out=scratch,
                                             y_f32=y_f32, tile=tile, split_k=split, wgt_tiled=wt, tap_masks=tm)
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(3):
                    wt = hip.tile_weights_h16(wgt) if tile >= 100 else None
                    hip.conv2d_h16(x, wgt, scale, shift, ksize, stride, pad, act, residual=residual, out=scratch, y_f32=y_f32,
                                   tile=tile, split_k=split, wgt_tiled=wt, tap_masks=tm)
                b.record()
                torch.cuda.synchronize()
                ms = a.elapsed_time(b) / 3
            except hip.MeError:
                continue
            if ms < best[0]:
                best = (ms, tile, split, masked)
        hit = _AUTO16[key] = (best[1], best[2], best[3])
    wt = hip.tile_weights_h16(wgt) if hit[0] >= 100 else None
    return hip.conv2d_h16(x, wgt, scale, shift, ksize, stride, pad, act, residual=residual, y_f32=y_f32, tile=hit[0], split_k=hit[1],
                          wgt_tiled=wt, tap_masks=tap_masks if hit[2] else None)


def _add16(a, b, out):
    c = a.shape[-1]
    hip.check(hip.lib().me_add_h16(a.data_ptr(), a.stride(-2), b.data_ptr(), b.stride(-2), out.data_ptr(), out.stride(-2),
                                   a.numel() // c, c, hip.HALF_TYPES[a.dtype], hip.stream_ptr()), "me_add_h16")


def _copy16(src, dst_ptr, dst_pitch):
    c = src.shape[-1]
    hip.check(hip.lib().me_copy_h16(src.data_ptr(), src.stride(-2), dst_ptr, dst_pitch, src.numel() // c, c, hip.stream_ptr()),
              "me_copy_h16")


class _Weights16:
    """16-bit copies of the packed fp32 weights of every conv block (OHWI, rotated for the data gradient, the parity weights of
    the stride-2 layers), refreshed by ONE multi-tensor copy per step from the buffers ``DarknetEngine.refresh_train_weights``
    fills (stable pointers: the destination tensors are allocated once)."""

    def __init__(self):
        self.dst, self.key = {}, None
        self.direct = False

    def refresh(self, eng, defs, half, dev):
        srcs, dsts = [], []
        for i, d in enumerate(defs):
            if d["type"] != "convolutional":
                continue
            cw = eng._conv_weights(i)
            for name in ("wgt", "rot", "parity"):
                src = getattr(cw, name, None)
                if src is None or src.shape[-1] <= 4 and name == "wgt":
                    continue
                slot = (i, name)
                dst = self.dst.get(slot)
                if dst is None or dst.shape != src.shape or dst.dtype != half or dst.device != dev:
                    dst = self.dst[slot] = torch.empty(src.shape, device=dev, dtype=half)
                srcs.append(src)
                dsts.append(dst)
        if srcs:
            try:
                torch._foreach_copy_(dsts, srcs)   # fp32 -> 16-bit (RNE), a handful of launches for the whole network
            except (RuntimeError, AttributeError):   # (a torch without the mixed-dtype multi-tensor copy)
                for dst, src in zip(dsts, srcs):
                    dst.copy_(src)

    def pack_direct(self, eng, defs, half, dev):
        """The packed fp32 parameters never exist in this form: ONE ``me_pack_conv_batch_f32`` launch reads the OIHW parameters and
        writes the 16-bit OHWI / rotated / parity layouts (``me_pack_desc.ohwi16`` / ``rot16`` / ``parity16``, ABI 11), the folded
        fp32 scale / shift of every block, and fp32 OHWI copies only where this step reads them (the stem's fp32 frames, the
        detection convolutions' fp32 data gradient).  Against the fp32 pack + the conversion pass: 0.5 GB of traffic instead of 2 GB
        at the head of every step, where nothing else runs.  Returns False when a block is not plain fp32 on ``dev`` (the caller
        takes the two-pass path).  The engine's stamps are left alone: its fp32 copies are re-packed when something asks for them."""
        import ctypes as C
        cws, rows = [], []
        for i, d in enumerate(defs):
            if d["type"] != "convolutional":
                continue
            cw = eng._conv_weights(i)
            s2 = int(d["size"]) == 3 and int(d["stride"]) == 2
            cw.want_rot = i > 0 and not s2
            cw.want_parity = i > 0 and s2
            if cw.cin_pad or cw.cout_pad or cw._prepare_packed_f32(dev) is None:
                return False
            to_yolo = i + 1 < len(defs) and defs[i + 1]["type"] == "yolo"
            cws.append((i, cw, to_yolo))
        descs = (hip.PackDesc * len(cws))()
        ht = hip.HALF_TYPES[half]
        for dsc, (i, cw, to_yolo) in zip(descs, cws):
            cw._pack_desc(dsc)
            cout, cin, k, _ = cw.conv.weight.shape
            fp32_too = cin <= 4 or to_yolo          # the stem (fp32 frames) and the detection convolutions (fp32 gradient)
            dsc.tiled = dsc.rot = dsc.rot_tiled = dsc.parity = None
            if not fp32_too:
                dsc.ohwi = None

            def slot(name, shape):
                t = self.dst.get((i, name))
                if t is None or tuple(t.shape) != tuple(shape) or t.dtype != half or t.device != dev:
                    t = self.dst[(i, name)] = torch.empty(shape, device=dev, dtype=half)
                return t.data_ptr()
            dsc.ohwi16 = slot("wgt", (cout, k, k, cin)) if cin > 4 else None
            dsc.rot16 = slot("rot", (cin, k, k, cout)) if cw.want_rot else None
            dsc.parity16 = slot("parity", (4 * cin, 2, 2, cout)) if cw.want_parity and k == 3 else None
            dsc.half_type = ht
        key = bytes(descs)
        cached = self.__dict__.get("_table")
        if cached is None or cached[0] != key:
            total = int(hip.lib().me_pack_conv_plan(descs, len(cws)))
            if total <= 0:
                raise hip.MeError("me_pack_conv_plan: " + hip.lib().me_last_error().decode("utf-8", "replace"))
            table = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(dev)
            cached = self._table = (key, table, total, max(int(d.ksize) for d in descs))
        _, table, total, max_k = cached
        hip.check(hip.lib().me_pack_conv_batch_f32(table.data_ptr(), len(cws), total, max_k, hip.stream_ptr()),
                  "me_pack_conv_batch_f32")
        for _i, cw, _y in cws:   # the fp32 copies of the engine were not (all) written: whoever wants them packs them again
            cw._stamp = None
            cw.parity_stamp = None
        self.direct = True
        return True

    def get(self, i, name):
        return self.dst.get((i, name))


class DetectorTrainer16:
    def __init__(self, model):
        self.m = model
        self.half = _HALF[model.compute_dtype]
        w = model.__dict__.get("_weights16")
        if w is None:
            w = model.__dict__["_weights16"] = _Weights16()
        self.w16 = w

    # ------------------------------------------------------------------------------------------ forward
    def forward(self, x):
        if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4):
            raise hip.MeError("Darknet training forward needs a 4-D CUDA float32 tensor; there is no CPU fallback")
        m, lib, half = self.m, hip.lib(), self.half
        eng = m.engine   # (the fp32 engine owns the packed master weights)
        x = x.contiguous()
        defs = m.module_defs
        outs, raws, bn_state = [], {}, {}
        bn_ws = bn_ws_t = None
        from .train_path import _bn_fwd
        self.w16.direct = False
        if not (_PACK16 and self.w16.pack_direct(eng, defs, half, x.device)):
            eng.refresh_train_weights(x.device)
            self.w16.refresh(eng, defs, half, x.device)
        for i, d in enumerate(defs):
            t = d["type"]
            if t == "convolutional":
                cw = eng._conv_weights(i)
                if not self.w16.direct:
                    cw.refresh(x.device)
                k, s = int(d["size"]), int(d["stride"])
                act = hip.ACT_LEAKY if d["activation"] == "leaky" else hip.ACT_LINEAR
                src = x if i == 0 else outs[i - 1]
                seq = m.module_list[i]
                bn = seq[1] if len(seq) > 1 and isinstance(seq[1], torch.nn.BatchNorm2d) else None
                bn_train = bn is not None and bn.training
                to_yolo = i + 1 < len(defs) and defs[i + 1]["type"] == "yolo"
                co, ci = cw.wgt.shape[0], cw.wgt.shape[3]
                if (ci > 4 and ci % 32) or (not to_yolo and co % 32):
                    raise NotImplementedError(f"16-bit detector training: conv {i} has {ci} -> {co} channels; the 16-bit matrix kernels "
                                              "want multiples of 32 (Darknet-53 has them; the tiny cfgs' 16-channel stem does not - "
                                              "train those in float32)")
                if bn_train:
                    # train()-mode BatchNorm (reference yolov3/models.py:38-40 under model.train()): float32 STATISTICS over the
                    # convolution of the 16-bit operands - the convolution writes its raw float32 sums (y_f32; the stem on the
                    # float32 pipe over the frame rounded to the storage type, the values its MFMA form multiplies),
                    # me_bn_train_fwd_f32 normalises with the batch statistics, updates the running ones and applies the
                    # activation; the stored activation is that result rounded once, like every other stored activation
                    if co > 2048:
                        raise hip.MeError("train-mode BatchNorm: more than 2048 channels")
                    ones, zeros = _const_vectors(co, x.device)
                    with _timed("fwd", _conv_flops(src, cw.wgt, s, i == 0)):
                        if i == 0 or ci <= 4:
                            wq = cw.wgt.to(half).float()
                            c_raw = hip.conv2d_auto(src.to(half).float(), wq, ones, zeros, k, s, (k - 1) // 2, hip.ACT_LINEAR, x_nchw=(i == 0))
                        else:
                            c_raw = conv16_auto(src, self.w16.get(i, "wgt"), ones, zeros, k, s, (k - 1) // 2, hip.ACT_LINEAR, y_f32=True)
                    y32 = torch.empty_like(c_raw)
                    if bn_ws is None:
                        bn_ws_t = torch.empty(int(lib.me_bn_workspace_bytes(2048)) + 256, dtype=torch.uint8, device=x.device)
                        bn_ws = bn_ws_t.data_ptr() + (-bn_ws_t.data_ptr()) % 256
                    st_bn = _bn_fwd(c_raw, co, c_raw.numel() // co, co, bn, act, y32, co, bn_ws)
                    bn_state[i] = (c_raw, st_bn)
                    y = y32 if to_yolo else y32.to(half)
                elif i == 0 or cw.wgt.shape[3] <= 4:   # stem: fp32 frames and fp32 weights holding 16-bit values
                    wq = cw.wgt.to(half).float()
                    with _timed("fwd", _conv_flops(src, cw.wgt, s, i == 0)):
                        y = hip.conv2d_h16(src, wq, cw.scale, cw.shift, k, s, (k - 1) // 2, act, x_nchw=(i == 0), half=half)
                else:
                    with _timed("fwd", _conv_flops(src, cw.wgt, s, False)):
                        y = conv16_auto(src, self.w16.get(i, "wgt"), cw.scale, cw.shift, k, s, (k - 1) // 2, act, y_f32=to_yolo)
            elif t == "maxpool":
                raise NotImplementedError("16-bit detector training: [maxpool] layers are a float32 path (no cfg with max-pooling "
                                          "has 32-multiple channels throughout)")
            elif t == "upsample":
                y = hip.upsample_h16(outs[i - 1], int(d["stride"]))
            elif t == "shortcut":
                a, b = outs[i - 1], outs[_resolve(d["from"], i)]
                y = torch.empty_like(a)
                _add16(a, b, y)
            elif t == "route":
                parts = [outs[_resolve(v, i)] for v in d["layers"].split(",")]
                if len(parts) == 1:
                    y = parts[0]
                else:
                    ct = sum(p.shape[-1] for p in parts)
                    y = torch.empty(parts[0].shape[:3] + (ct,), device=x.device, dtype=half)
                    off = 0
                    for p in parts:
                        _copy16(p, y.data_ptr() + 2 * off, ct)
                        off += p.shape[-1]
            elif t == "yolo":
                raws[i] = outs[i - 1]
                y = None
            else:
                raise ValueError(f"unsupported cfg block [{t}] at module {i}")
            outs.append(y)
        st = _State()
        st.x, st.outs, st.raws, st.bn_state = x, outs, raws, bn_state
        st.bn_ws = bn_ws_t
        return st

    # ------------------------------------------------------------------------------------------ backward
    def backward(self, st, draws, reducer=None):
        """``draws``: {yolo module index: d loss / d raw map, float32}.  Returns {parameter name: float32 gradient}."""
        m, lib, half = self.m, hip.lib(), self.half
        eng = m.engine
        defs, outs, x = m.module_defs, st.outs, st.x
        dev = x.device
        L = len(defs)
        dout = [None] * L
        grads = {}
        ht = hip.HALF_TYPES[half]

        def contribute(i, g, fresh):
            """``g``: a 16-bit gradient w.r.t. module i's output (NHWC, dense or a channel slice); ``fresh``: nobody else holds it."""
            if g.dtype != half:
                g, fresh = g.to(half), True
            cur = dout[i]
            if cur is None:
                if g.is_contiguous():
                    dout[i] = (g, bool(fresh))
                else:  # a channel slice of a wider gradient ([route]): densify once
                    dense = torch.empty(g.shape, device=dev, dtype=half)
                    _copy16(g, dense.data_ptr(), g.shape[-1])
                    dout[i] = (dense, True)
                return
            t_cur, owned = cur
            if owned:
                _add16(t_cur, g, t_cur)
            else:
                total = torch.empty_like(t_cur)
                _add16(t_cur, g, total)
                dout[i] = (total, True)

        x_nhwc = None
        main = torch.cuda.current_stream(dev)
        side = _side_stream(dev) if os.environ.get("MILLIEYE_WGRAD_STREAM", "1") != "0" and _TIMING[0] is None else None
        side_ptr = C.c_void_p(side.cuda_stream) if side is not None else None
        side_ws = self.m.__dict__.setdefault("_wgrad16_ws", [None])   # slab scratch of the side stream, kept from step to step
        aff_ws = self.m.__dict__.setdefault("_affine16_ws", [None])
        aff_layer_ws = self.m.__dict__.setdefault("_affine16_layer_ws", {})
        sp = hip.stream_ptr()   # (the main stream's handle, once: this backward never switches torch's current stream on its hot path)
        stream = lambda: sp  # noqa: E731
        for i in reversed(range(L)):
            d = defs[i]
            t = d["type"]
            if t == "yolo":
                dout[i - 1] = (draws[i], True)   # float32: the detection convolution's output is float32
                continue
            if dout[i] is None:
                continue
            dy, dy_owned = dout[i]
            if t == "convolutional":
                seq = m.module_list[i]
                bn = seq[1] if len(seq) > 1 and isinstance(seq[1], torch.nn.BatchNorm2d) else None
                cw = eng._conv_weights(i)
                k, s = int(d["size"]), int(d["stride"])
                pad = (k - 1) // 2
                act = hip.ACT_LEAKY if d["activation"] == "leaky" else hip.ACT_LINEAR
                y = outs[i]
                n, ho, wo, cout = y.shape
                rows = n * ho * wo
                dshift = torch.empty(cout, device=dev)
                dgamma = torch.empty(cout, device=dev) if bn is not None else None
                gam = bn.weight.detach() if bn is not None else None
                bet = bn.bias.detach() if bn is not None else None
                f32_out = y.dtype == torch.float32   # a detection convolution: float32 map, float32 gradient from the loss
                if i in st.bn_state:
                    # train()-mode BatchNorm: the gradient through the batch statistics in float32 (me_bn_train_bwd_f32 on the raw
                    # float32 sums the forward kept), the activation gradient stored in the 16-bit type like the other layers'
                    from .train_path import _bn_bwd
                    c_raw, st_bn = st.bn_state[i]
                    dy32 = dy.float() if dy.dtype != torch.float32 else dy
                    dc32_bn = torch.empty_like(c_raw)
                    ws_b = st.bn_ws.data_ptr() + (-st.bn_ws.data_ptr()) % 256
                    dgamma, dshift = _bn_bwd(c_raw, cout, dy32.contiguous(), cout, rows, cout, bn, st_bn, act, dc32_bn, cout, ws_b)
                    dc = dc32_bn.to(half)
                    dc32 = None
                    sums_ws = None
                    f32_out = False
                elif f32_out:
                    dc32 = dy if dy_owned else torch.empty_like(y)
                    ws = torch.empty(lib.me_affine_bwd_workspace_bytes(rows, cout), dtype=torch.uint8, device=dev)
                    hip.check(lib.me_affine_act_bwd_f32(y.data_ptr(), cout, dy.data_ptr(), cout, rows, cout,
                                                        cw.scale.data_ptr() if bn is not None else None, _ptr(gam), _ptr(bet), act,
                                                        dc32.data_ptr(), cout, dshift.data_ptr(), _ptr(dgamma), ws.data_ptr(),
                                                        stream()), "me_affine_act_bwd_f32")
                    dc = None
                    sums_ws = None
                else:
                    if dy.dtype != half or not dy.is_contiguous():
                        dy = dy.to(half).contiguous()
                    dc = dy if dy_owned else torch.empty_like(y)
                    need_a = int(lib.me_affine_bwd_h16_workspace_bytes(rows, cout))
                    # the second level of the per-channel sums (d gamma / d beta) is not on the data path: when this layer's weight
                    # gradient goes to the side stream anyway, the sums go with it (me_affine_bwd_h16_sums) and the main stream is
                    # one launch per layer shorter; the partial rows then need a scratch of the layer's own until the side stream
                    # has read them (kept on the model: the same addresses every step)
                    defer_sums = side is not None and _DIRECT and _SUMS_SIDE and _WGRAD16 and i > 0 and outs[i - 1].dtype == half \
                        and outs[i - 1].shape[-1] % 8 == 0 and cout % 8 == 0
                    if defer_sums:
                        ws = aff_layer_ws.get(i)
                        if ws is None or ws.numel() < need_a:
                            ws = aff_layer_ws[i] = torch.empty(need_a, dtype=torch.uint8, device=dev)
                    else:
                        if aff_ws[0] is None or aff_ws[0].numel() < need_a:   # one partial-sum scratch for the other layers (stream-ordered)
                            aff_ws[0] = torch.empty(max(need_a, 8 << 20), dtype=torch.uint8, device=dev)
                        ws = aff_ws[0]
                    hip.check(lib.me_affine_act_bwd_h16(y.data_ptr(), cout, dy.data_ptr(), cout, rows, cout,
                                                        cw.scale.data_ptr() if bn is not None else None, _ptr(gam), _ptr(bet), act,
                                                        dc.data_ptr(), cout, None if defer_sums else dshift.data_ptr(),
                                                        None if defer_sums else _ptr(dgamma), ws.data_ptr(), ht, stream()),
                              "me_affine_act_bwd_h16")
                    sums_ws = ws if defer_sums else None
                    dc32 = None
                if bn is not None:
                    grads[f"module_list.{i}.batch_norm_{i}.weight"] = dgamma
                    grads[f"module_list.{i}.batch_norm_{i}.bias"] = dshift
                else:
                    grads[f"module_list.{i}.conv_{i}.bias"] = dshift
                # weight gradient: the fp32 matrix kernels on fp32 copies (products of 16-bit values are exact in fp32)
                if i == 0:
                    if x_nhwc is None:
                        x_nhwc = x.permute(0, 2, 3, 1).contiguous()
                    xin = x_nhwc
                else:
                    xin = outs[i - 1]
                _, h, w, cin = xin.shape

                def wgrad():
                    if _WGRAD16 and dc is not None and xin.dtype == half and cin % 8 == 0 and cout % 8 == 0:
                        return hip.conv_wgrad_h16(xin, dc, k, s, pad, oihw=True)   # 16-bit operands, fp32 accumulators and slabs
                    # stem (fp32 frames) and the detection convolutions (fp32 gradient, 255 / 51 channels): the fp32 kernels
                    x32 = xin if xin.dtype == torch.float32 else xin.float()
                    d32 = dc32 if dc32 is not None else dc.float()
                    return hip.conv_wgrad(x32, d32, k, s, pad, oihw=True)
                fast16 = _WGRAD16 and dc is not None and xin.dtype == half and cin % 8 == 0 and cout % 8 == 0
                if side is not None and fast16 and _DIRECT:
                    # the 16-bit weight gradient straight onto the side stream's handle: no switch of torch's current stream (a
                    # context manager, two current-stream lookups and a per-stream workspace lookup: ~20 us of host time per layer of
                    # a step whose host is the slower side), the slab scratch is this backward's own buffer on that stream
                    side.wait_stream(main)
                    if sums_ws is not None:
                        hip.check(lib.me_affine_bwd_h16_sums(sums_ws.data_ptr(), rows, cout, dshift.data_ptr(), _ptr(dgamma), side_ptr),
                                  "me_affine_bwd_h16_sums")
                    dwt = torch.empty((cout, cin, k, k), device=dev, dtype=torch.float32)
                    need = max(int(lib.me_conv_wgrad_workspace_bytes(n, ho, wo, cin, cout, k)), 4 * cout * cin * k * k)
                    if side_ws[0] is None or side_ws[0].numel() < need + 256:
                        with torch.cuda.stream(side):
                            side_ws[0] = torch.empty(max(need + 256, 64 << 20), dtype=torch.uint8, device=dev)
                    wsp = side_ws[0].data_ptr() + (-side_ws[0].data_ptr()) % 256
                    hip.check(lib.me_conv_wgrad_h16(xin.data_ptr(), cin, dc.data_ptr(), cout, dwt.data_ptr(), n, h, w, cin, cout, k, s,
                                                    pad, wsp, need, 1, ht, side_ptr), "me_conv_wgrad_h16")
                    grads[f"module_list.{i}.conv_{i}.weight"] = dwt
                    dc.record_stream(side)
                    xin.record_stream(side)
                elif side is not None:
                    side.wait_stream(main)
                    with torch.cuda.stream(side):
                        grads[f"module_list.{i}.conv_{i}.weight"] = wgrad()
                    (dc32 if dc32 is not None else dc).record_stream(side)
                    xin.record_stream(side)
                else:
                    with _timed("wgrad", 2.0 * rows * cout * k * k * cin):
                        grads[f"module_list.{i}.conv_{i}.weight"] = wgrad()
                if reducer is not None:
                    vec_stream = side if sums_ws is not None else main   # (where d gamma / d beta were finished)
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
                if f32_out or cout % 32 != 0:
                    if k != 1 or s != 1:
                        raise NotImplementedError(f"conv {i}: 16-bit data gradient for cout={cout} needs k=1 (detection convolutions)")
                    d32 = dc32 if dc32 is not None else dc.float()
                    dx = torch.empty((n, h, w, cin), device=dev, dtype=torch.float32)
                    hip.check(lib.me_gemm_f32(0, 0, rows, cin, cout, 1.0, d32.data_ptr(), cout, cw.wgt.data_ptr(), cin, 0.0,
                                              dx.data_ptr(), cin, stream()), "me_gemm_f32")
                    contribute(i - 1, dx.to(half), True)
                else:
                    ones, zeros = _const_vectors(cin, dev)
                    parity = s == 2 and k == 3 and pad == 1 and h == 2 * ho and w == 2 * wo and cin % 8 == 0
                    prev = dout[i - 1]
                    res = prev[0] if (prev is not None and prev[0].is_contiguous() and prev[0].dtype == half) else None
                    if s == 1:
                        rot = self.w16.get(i, "rot")
                        if rot is None:
                            rot = cw.wgt.flip(1, 2).permute(3, 1, 2, 0).contiguous().to(half)
                        dx = conv16_auto(dc, rot, ones, zeros, k, 1, k - 1 - pad, hip.ACT_LINEAR, residual=res)
                        if res is not None:
                            dout[i - 1] = (dx, True)
                        else:
                            contribute(i - 1, dx, True)
                    elif parity:
                        pw = self.w16.get(i, "parity")
                        if pw is None or (not self.w16.direct and cw.parity_stamp != cw._stamp):
                            pw = _parity_weights(cw.wgt).to(half)
                        o4, z4 = _const_vectors(4 * cin, dev)
                        # (round 6, as the fp32 step since round 5: 7 of the 16 (class, tap) pairs of the 2x2 parity filter are
                        #  structurally zero and the masked tile instances skip them - a measured candidate of conv16_auto)
                        masks = (cin, _PARITY_TAP_MASKS) if (cin % 64 == 0 and _PARITY_MASKS_ON) else None
                        dx4 = conv16_auto(dc, pw, o4, z4, 2, 1, 1, hip.ACT_LINEAR, tap_masks=masks)
                        dx = dx4[:, 1:, 1:, :].reshape(n, ho, wo, 2, 2, cin).permute(0, 1, 3, 2, 4, 5).reshape(n, h, w, cin)
                        contribute(i - 1, dx.contiguous(), True)
                    else:
                        raise NotImplementedError(f"conv {i}: 16-bit data gradient of a {k}x{k} / stride {s} layer on a {h}x{w} map")
                tm.__exit__()
                continue
            elif t == "shortcut":
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
                g = dy.float().view(n, h, 2, w, 2, c).sum(dim=(2, 4)).to(half)   # two small layers: fp32 sum of the 2x2 block, one rounding
                contribute(i - 1, g, True)
            dout[i] = None
        if side is not None:
            main.wait_stream(side)
        return grads
