"""Stage-3 training step of ``Network.forward(..., targets)`` on the HIP library.

Reference: ``module3_our_dataset/my_models.py:545-639`` (labels, sampling, focal + BCE losses) on top
of the train-mode forward of the heads, differentiated by torch autograd (``train.py:185-191``).
Here the graph is fixed, so forward and backward are explicit launch sequences over
``libmillieye_hip`` (``csrc/train.hip``, ``csrc/heads.hip``, ``csrc/conv.hip``); autograd only sees one
:class:`torch.autograd.Function` whose inputs are the head parameters, so ``loss.backward()`` /
``optimizer.step()`` / ``requires_grad=False`` freezing (``train.py:146-149``) behave as in the reference.

What gets a gradient: ``ensemble_head.fc1/fc2``, ``refinement_head.net0``,
``net2`` (rows 0-1; the other rows are zero), ``radar_net``; ``img_cnn_layers`` through the PS-RoIAlign
backward; ``radar_cnn_layers`` through the RoIAlign backward.  ``net1`` (regression loss is excluded from the
loss, my_models.py:635), ``net3`` and ``fusion_head`` (never used) get ``None`` like in the reference.
The detector is frozen and detached (models.py:255,266).

Host-side parts are the ones that are host-side in the reference too: IoU labelling
(``obtain_iou_labels``, a python loop over boxes) and negative sampling with python's ``random``.
"""
import ctypes as C
import random

import numpy as np
import os

import torch

from . import hip

_EPS = 1e-5
_MOM = 0.1


def _ptr(t):
    return t.data_ptr() if t is not None else None


def _gemm(ta, tb, m, n, k, a, lda, b, ldb, c, ldc, alpha=1.0, beta=0.0):
    if ta and not tb and alpha == 1.0 and beta == 0.0 and ldc == n and k >= 64:
        # C[m,n] = A^T B with A [k,m], B [k,n]: the weight gradient of a 1x1 convolution over k "pixels" -> matrix pipe
        lib = hip.lib()
        need = lib.me_conv_wgrad_workspace_bytes(1, k, 1, n, m, 1)
        ws_ptr, _keep = (None, None)
        if need > 0:
            ws_ptr, _keep = hip._workspace(need, c.device, slot="wgrad")
        hip.check(lib.me_conv_wgrad_mfma_f32(_ptr(b), ldb, _ptr(a), lda, _ptr(c), 1, k, 1, n, m, 1, 1, 0, ws_ptr, need,
                                             hip.stream_ptr()), "me_conv_wgrad_mfma_f32")
        return
    hip.check(hip.lib().me_gemm_f32(int(ta), int(tb), m, n, k, alpha, _ptr(a), lda, _ptr(b), ldb, beta, _ptr(c), ldc,
                                    hip.stream_ptr()), "me_gemm_f32")


def _colsum(x, ld, rows, cols, out):
    hip.check(hip.lib().me_colsum_f32(_ptr(x), ld, rows, cols, _ptr(out), hip.stream_ptr()), "me_colsum_f32")


def _f32(dev, *shape):
    return torch.empty(shape, device=dev, dtype=torch.float32)


_CONSTS = {}


def _const(dev, c, value):
    """A cached read-only vector of ``c`` ones / zeros (per-channel scale / shift of a plain convolution): the step asked torch for a
    fresh one - an allocation and a fill launch - eight times per call."""
    key = (str(dev), int(c), float(value))
    t = _CONSTS.get(key)
    if t is None:
        t = _CONSTS[key] = torch.full((int(c),), float(value), device=dev, dtype=torch.float32)
    return t


class _BnState:
    def __init__(self, c, dev, buf=None):
        if buf is None:
            buf = torch.empty((3, c), device=dev, dtype=torch.float32)   # (one allocation for the three vectors)
        self.mean, self.var, self.rstd = buf[0], buf[1], buf[2]


def _bn_fwd(x, ldx, rows, c, bn, act, y, ldy, ws, buf=None):
    st = _BnState(c, x.device, buf)   # (``buf``: a [3, c] arena buffer - the captured backward reads the statistics from fixed addresses)
    if rows == 1:
        raise ValueError(f"Expected more than 1 value per channel when training, got input size [1, {c}]")
    hip.check(hip.lib().me_bn_train_fwd_f32(_ptr(x), ldx, rows, c, _ptr(bn.weight), _ptr(bn.bias), float(bn.eps),
                                            float(bn.momentum), _ptr(bn.running_mean), _ptr(bn.running_var), act,
                                            _ptr(y), ldy, _ptr(st.mean), _ptr(st.var), _ptr(st.rstd), ws,
                                            hip.stream_ptr()), "me_bn_train_fwd_f32")
    # the kernel wrote the running statistics through raw pointers: bump their version counters so the packed
    # scale/shift caches (engine.ConvWeights, my_models._HeadPack) re-fold them at the next eval-mode forward
    torch.autograd.graph.increment_version((bn.running_mean, bn.running_var))
    with torch.no_grad():
        bn._buffers["num_batches_tracked"].add_(1)   # (in place on the buffer: `bn.num_batches_tracked += 1` goes through Module.__setattr__)
    return st


def _bn_bwd(x, ldx, dy, lddy, rows, c, bn, st, act, dx, lddx, ws, rows_dev=None):
    dg, db = _f32(x.device, c), _f32(x.device, c)
    if rows_dev is not None:   # captured step: ``rows`` is the buffers' capacity, the live row count is the device word
        hip.check(hip.lib().me_bn_train_bwd_dev_f32(_ptr(x), ldx, _ptr(dy), lddy, rows, rows_dev.data_ptr(), c, _ptr(bn.weight),
                                                    _ptr(bn.bias), _ptr(st.mean), _ptr(st.rstd), act, _ptr(dx), lddx, _ptr(dg),
                                                    _ptr(db), ws, hip.stream_ptr()), "me_bn_train_bwd_dev_f32")
        return dg, db
    hip.check(hip.lib().me_bn_train_bwd_f32(_ptr(x), ldx, _ptr(dy), lddy, rows, c, _ptr(bn.weight), _ptr(bn.bias),
                                            _ptr(st.mean), _ptr(st.rstd), act, _ptr(dx), lddx, _ptr(dg), _ptr(db), ws,
                                            hip.stream_ptr()), "me_bn_train_bwd_f32")
    return dg, db


class _BnEval:
    """Eval-mode BatchNorm = the per-channel affine ``scale * x + shift`` from the running statistics (no update)."""

    def __init__(self, bn, dev):
        with torch.no_grad():
            scale = bn.weight.detach().double() / torch.sqrt(bn.running_var.detach().double() + bn.eps)
            shift = bn.bias.detach().double() - bn.running_mean.detach().double() * scale
        self.scale = scale.to(dev, torch.float32).contiguous()
        self.shift = shift.to(dev, torch.float32).contiguous()

    def with_bias(self, bias):
        """shift of ``scale * (conv + bias) + shift`` folded for the conv epilogue."""
        return (self.shift + bias.detach().to(self.shift) * self.scale).contiguous()


def _bn_eval_bwd(y, ldy, dy, lddy, rows, c, bn, st, act, dc, lddc):
    """Backward of ``y = act(scale * x + shift)`` (eval-mode BatchNorm, like ``F.batch_norm(training=False)`` under
    autograd): dx, d gamma = sum dy act' xhat, d beta = sum dy act' (me_affine_act_bwd_f32, fixed-order reductions)."""
    dev = y.device
    dg, db = _f32(dev, c), _f32(dev, c)
    lib = hip.lib()
    ws = torch.empty(max(int(lib.me_affine_bwd_workspace_bytes(rows, c)), 256), dtype=torch.uint8, device=dev)
    gam, bet = bn.weight.detach().contiguous(), bn.bias.detach().contiguous()
    hip.check(lib.me_affine_act_bwd_f32(_ptr(y), ldy, _ptr(dy), lddy, rows, c, _ptr(st.scale), _ptr(gam), _ptr(bet), act,
                                        _ptr(dc), lddc, _ptr(db), _ptr(dg), ws.data_ptr(), hip.stream_ptr()),
              "me_affine_act_bwd_f32")
    return dg, db


def _conv(x, x_pitch, n, h, w, cin, wgt, scale, shift, k, pad, act, out):
    d = hip.ConvDesc()
    d.x, d.x_pitch, d.x_nchw = _ptr(x), x_pitch, 0
    d.wgt, d.scale, d.shift, d.res, d.res_pitch = _ptr(wgt), _ptr(scale), _ptr(shift), None, 0
    d.y, d.y_pitch = _ptr(out), out.shape[-1]
    d.n, d.h, d.w, d.cin, d.cout = n, h, w, cin, wgt.shape[0]
    d.ksize, d.stride, d.pad, d.ho, d.wo = k, 1, pad, h, w
    d.act, d.upsample, d.tile, d.split_k = act, 1, 0, 1
    hip.check(hip.lib().me_conv2d_f32(C.byref(d), hip.stream_ptr()), "me_conv2d_f32")
    return out


def _wgrad(x, x_pitch, dy, dy_pitch, n, h, w, cin, cout, k, pad):
    """Weight gradient on the matrix pipe (me_conv_wgrad_mfma_f32); returns OIHW like the parameter."""
    dw = _f32(x.device, cout, k, k, cin)
    lib = hip.lib()
    ho, wo = h + 2 * pad - k + 1, w + 2 * pad - k + 1
    need = lib.me_conv_wgrad_workspace_bytes(n, ho, wo, cin, cout, k)
    ws_ptr, _keep = (None, None)
    if need > 0:
        ws_ptr, _keep = hip._workspace(need, x.device, slot="wgrad")
    hip.check(lib.me_conv_wgrad_mfma_f32(_ptr(x), x_pitch, _ptr(dy), dy_pitch, _ptr(dw), n, h, w, cin, cout, k, 1, pad,
                                         ws_ptr, need, hip.stream_ptr()), "me_conv_wgrad_mfma_f32")
    return dw.permute(0, 3, 1, 2).contiguous()  # OHWI -> OIHW (the parameter's layout)


def iou_labels_vectorized(boxes, targets):
    """Same result as ``my_models.obtain_iou_labels(boxes, targets, multi_boxes=<truthy>)`` (reference
    my_models.py:317-375, quirk q5: the call site always passes a truthy ``multi_boxes``, so the ``detected``
    bookkeeping never changes the outcome) without the per-box python loop: one ``[P, Q]`` IoU matrix (+1 pixel
    convention, identical fp32 formula), masked to same-image / same-class targets, first-maximum per row."""
    P, Q = boxes.shape[0], targets.shape[0]
    iou_labels = torch.zeros((P, 1))
    target_location = torch.zeros((P, 4))
    if P == 0 or Q == 0:
        return iou_labels, target_location
    # numpy float32, single-threaded: torch's CPU ops fan a [P, Q] reduction of a few thousand elements out over every
    # core of the host and take ~10 ms for it on a 128-core box; the arithmetic (IEEE fp32, same operation order) is identical
    bx = boxes.detach().to(torch.float32).numpy()
    tg = targets.detach().to(torch.float32).numpy()
    b, t = bx[:, 2:6], tg[:, 2:6]
    one = np.float32(1.0)
    ix1 = np.maximum(b[:, None, 0], t[None, :, 0])
    iy1 = np.maximum(b[:, None, 1], t[None, :, 1])
    ix2 = np.minimum(b[:, None, 2], t[None, :, 2])
    iy2 = np.minimum(b[:, None, 3], t[None, :, 3])
    inter = np.maximum(ix2 - ix1 + one, np.float32(0)) * np.maximum(iy2 - iy1 + one, np.float32(0))
    a1 = ((b[:, 2] - b[:, 0] + one) * (b[:, 3] - b[:, 1] + one))[:, None]
    a2 = ((t[:, 2] - t[:, 0] + one) * (t[:, 3] - t[:, 1] + one))[None, :]
    iou = inter / (a1 + a2 - inter + np.float32(1e-16))
    same = (bx[:, None, 0] == tg[None, :, 0]) & (bx[:, None, 1] == tg[None, :, 1])
    masked = np.where(same, iou, np.float32(-1.0))
    arg = masked.argmax(1)  # first maximum, like ious.max(0) over the filtered targets in target order
    best = masked[np.arange(P), arg]
    has = same.any(1)
    iou_labels[:, 0] = torch.from_numpy(np.where(has, best, np.float32(0)).astype(np.float32))
    target_location[:] = torch.from_numpy(np.where(has[:, None], t[arg], np.float32(0)).astype(np.float32))
    return iou_labels, target_location


def _head_named(net):
    """(names, parameters) of everything outside the detector, cached on the module: walking named_parameters() of a
    107-module Darknet costs ~1 ms, and the training step asks several times.  The cache follows the Parameter objects
    (re-assigning a parameter re-walks)."""
    cache = net.__dict__.get("_head_cache")
    if cache is not None and all(a is b for a, b in zip(cache[2], (net._parameters, net._modules))):
        ids = cache[3]
        if len(ids) == len(cache[1]) and all(id(p) == i for p, i in zip(cache[1], ids)):
            return cache[0], cache[1]
    names, params = [], []
    for name, p in net.named_parameters():
        if not name.startswith("base_detector."):
            names.append(name)
            params.append(p)
    object.__setattr__(net, "_head_cache", (names, params, (net._parameters, net._modules), [id(p) for p in params]))
    return names, params


def head_parameters(net):
    """Fixed order of every non-detector parameter (the autograd inputs of the training step)."""
    return list(_head_named(net)[1])


def _head_names(net):
    return list(_head_named(net)[0])


# ---------------------------------------------------------------------------------------------------------------------------------
# The backward of the step as ONE captured hipGraph (round 6).  The step is host-bound (profiles/r06_host_profile_train16_tottime.txt:
# 0.9 ms of the 2.4 ms step issues the ~45 launches of ``_backward``), but its launch arguments change from step to step with the
# number of proposals k.  Captured form: every tensor the backward reads lives in a per-network ARENA of fixed-capacity buffers
# (capacity rows = 200 N + the radar boxes rounded up to 16; same addresses every step - ``forward_train`` writes them eagerly, with
# the true k), the launches run over the CAPACITY and the four kernels whose arithmetic depends on the row count read it from a device
# word (``me_heads_tail_bwd_dev_f32``, ``me_bn_train_bwd_dev_f32``, ``me_[ps_]roi_align_bwd_dev_f32``: zeros behind the live rows, so the
# dense products over the capacity add exact zeros).  The first two backwards of a signature run eagerly, the third is captured,
# later ones are one copy of the upstream gradient + one replay + one clone of the flat gradient buffer.  MILLIEYE_TRAIN_GRAPH=0: off.
# ---------------------------------------------------------------------------------------------------------------------------------
def _train_graph_enabled():
    return os.environ.get("MILLIEYE_TRAIN_GRAPH", "1") != "0"


class _Arena:
    """Named fixed-shape device buffers of one Network's training step, zero-filled when first asked for and kept."""

    def __init__(self):
        self.t = {}
        self.graphs = {}

    def get(self, name, shape, dev, dtype=torch.float32):
        key = (name, tuple(int(v) for v in shape), dtype, str(dev))
        t = self.t.get(key)
        if t is None:
            if len(self.t) > 512:   # (shapes follow the batch geometry: bounded in any real loop; never let it grow without limit)
                self.t.clear()
                self.graphs.clear()
            t = self.t[key] = torch.zeros(key[1], device=dev, dtype=dtype)
        return t


def _arena(net):
    a = net.__dict__.get("_train_arena")
    if a is None:
        a = net.__dict__["_train_arena"] = _Arena()
    return a


class _BackwardGraph:
    def __init__(self):
        self.runs = 0
        self.graph = None
        self.g_in = None
        self.flat = None
        self.layout = None   # [(name, shape, offset, numel)]
        self.scratch = None  # addresses of the library's per-stream workspaces at capture time


def _scratch_ptrs():
    """The library's per-stream workspaces (``hip._workspace``: the weight-gradient slabs of ``_gemm``, the NMS lists): their addresses are
    baked into a capture, and a later call with a larger request - another model on the same stream, a bigger batch - REPLACES them.
    Checked before every replay (like ``detector_graph.GraphedDetectorStep._scratch_ptrs``): a moved buffer means a new capture."""
    return {k: t.data_ptr() for k, t in hip._ws_cache.items() if t is not None}


def _scratch_moved(snapshot):
    cache = hip._ws_cache
    for k, ptr in snapshot.items():   # (workspaces that appeared since the capture cannot be in it)
        t = cache.get(k)
        if t is None or t.data_ptr() != ptr:
            return True
    return False


def _graphed_backward(S, grad_out, needed):
    arena = S["arena"]
    sig = S["graph_sig"] + (tuple(sorted(needed)) if needed is not None else None,)
    rec = arena.graphs.get(sig)
    if rec is None:
        rec = arena.graphs[sig] = _BackwardGraph()
    if rec.graph is False or torch.cuda.is_current_stream_capturing():
        return _backward(S, grad_out, needed)
    rec.runs += 1
    if rec.graph not in (None, False) and _scratch_moved(rec.scratch):
        rec.graph, rec.flat = None, None   # a workspace the capture points into was replaced: capture again (this call, eager warm-up done)
    if rec.graph is None and rec.runs <= 2:
        return _backward(S, grad_out, needed)   # eager warm-up: lazy one-time state (workspaces, kernel attributes) settles
    dev = S["fm"].device
    if rec.graph is None:
        rec.g_in = torch.zeros((), device=dev, dtype=torch.float32)
        rec.g_in.copy_(grad_out.reshape(()))
        torch.cuda.synchronize(dev)
        graph = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                G = _backward(S, rec.g_in, needed, rows=S["cap_rows"], k_dev=S["k_dev"])
                names = sorted(G)
                rec.flat = torch.cat([G[nm].reshape(-1) for nm in names])
            off, layout = 0, []
            for nm in names:
                layout.append((nm, tuple(G[nm].shape), off, G[nm].numel()))
                off += G[nm].numel()
            rec.layout, rec.graph, rec.scratch = layout, graph, _scratch_ptrs()
        except Exception as exc:   # a failed capture is not fatal and is not retried: this signature stays eager
            import warnings
            warnings.warn(f"millieye_amd: hipGraph capture of the stage-3 backward failed ({exc!r}); it runs eagerly")
            rec.graph = False
            torch.cuda.synchronize(dev)
            return _backward(S, grad_out, needed)
    else:
        rec.g_in.copy_(grad_out.reshape(()))
    rec.graph.replay()
    fresh = rec.flat.clone()   # the caller's gradients must not alias the graph's static buffer (accumulation over several batches)
    return {nm: fresh[off:off + cnt].view(shape) for nm, shape, off, cnt in rec.layout}


class _StageThree(torch.autograd.Function):
    @staticmethod
    def forward(ctx, state, *params):
        ctx.state = state
        return state["loss"].clone()

    @staticmethod
    def backward(ctx, grad_out):
        names = ctx.state["names"]
        # frozen tensors (train.py:146-149 freezes the stage-2 part: requires_grad = False) need no gradient: with the
        # score-map conv / BatchNorm frozen, the largest GEMMs of the step (490 x 256 over every pixel, the PS-RoIAlign
        # scatter, the BatchNorm backward over N*h*w x 490) are skipped altogether
        needed = {n for n, need in zip(names, ctx.needs_input_grad[1:]) if need}
        grads = _graphed_backward(ctx.state, grad_out, needed) if ctx.state.get("arena") is not None else _backward(ctx.state, grad_out, needed)
        return (None,) + tuple(grads.get(n) if n in needed else None for n in names)



# ---------------------------------------------------------------------------------------------------------------------------------
# The frozen part of the step, one batch ahead (round 5).  Stage-3 training keeps the detector frozen (reference train.py:170), so the
# detector + NMS + proposal assembly of batch k + 1 depend on nothing batch k's step computes.  The rest of the step is a chain of
# small launches behind one host read (the RoI count) and is bound by the HOST (2.1 ms of issue time at batch 8 for ~1 ms of kernels,
# tools/train_host_profile.py): the GPU idles through most of it.  ``Network.queue_detector_prefetch(images_next)`` in front of the
# call for batch k makes ``forward_train`` issue batch k + 1's frozen part on a second stream right after it has picked up its own -
# the detector of the next batch runs under the current batch's tail.  Results are private copies (the engine's arena belongs to the
# prefetch stream while it runs; ``Darknet._run`` waits for a prefetch in flight before anything else touches the engine).
# ---------------------------------------------------------------------------------------------------------------------------------
_PREFETCH_STREAMS = {}


def _prefetch_stream(dev):
    key = str(dev)
    st = _PREFETCH_STREAMS.get(key)
    if st is None:
        st = _PREFETCH_STREAMS[key] = torch.cuda.Stream(dev)
    return st


def _frozen_detector_block(net, images):
    """Detector forward (whatever storage mode ``Darknet.compute_dtype`` names), NMS, per-class proposal rows (stage 3: the rows of
    ``net.class_idx``; stage 2 - module2/my_models.py has no ``class_idx`` - the rows of every class), an fp32 NHWC copy of the
    feature tap - everything of the step that has no trainable parameter in it.  Runs on the current stream."""
    from .my_models import _DETECTIONS_PER_IMG, _NMS_THRESH
    lib = hip.lib()
    dev = images.device
    n = images.shape[0]
    f32 = dict(device=dev, dtype=torch.float32)
    plan, yolo_out = net.base_detector._run(images, nms_conf=float(net.conf_thresh))
    det, cnt = hip.nms_batched(yolo_out, float(net.conf_thresh), _NMS_THRESH, _DETECTIONS_PER_IMG,
                               writeback_xyxy=False, prepped=plan.nms_prepped == float(net.conf_thresh))
    num_classes = yolo_out.shape[2] - 5
    cols = 8 + net.class_num
    cap_img = n * _DETECTIONS_PER_IMG
    img_boxes = torch.empty((cap_img, cols), **f32)
    n_img_dev = torch.empty((1,), device=dev, dtype=torch.int32)
    hip.check(lib.me_gather_class_boxes_f32(det.data_ptr(), cnt.data_ptr(), n, _DETECTIONS_PER_IMG, num_classes,
                                            int(getattr(net, "class_idx", -1)), int(net.class_num), img_boxes.data_ptr(),
                                            n_img_dev.data_ptr(), hip.stream_ptr()), "me_gather_class_boxes_f32")
    if plan.tap is None:
        raise AttributeError("'Darknet' object has no attribute 'featuremap'")
    fm = plan.tap.permute(0, 2, 3, 1).float().contiguous()  # fp32 NHWC copy: the arena is reused by the next forward
    return dict(num_classes=num_classes, cols=cols, cap_img=cap_img, img_boxes=img_boxes, n_img_dev=n_img_dev, fm=fm,
                tap_shape=tuple(plan.tap_shape))


def _prefetch_key(net, images, wstamp=None):
    """What a look-ahead result was computed from: the frames, the thresholds, the storage mode and the detector's weights - the
    engine's own weight stamp (live ``(data_ptr, _version)`` of every source tensor + the invalidation epoch) and the identity of
    the detector module.  ``wstamp``: a stamp the caller computed a moment ago (one pass over the ~370 tensors per step, not two)."""
    det = net.base_detector
    if wstamp is None:
        wstamp = det.engine.weight_stamp(images.device)
    return (images.data_ptr(), tuple(images.shape), images._version, float(net.conf_thresh), int(getattr(net, "class_idx", -1)),
            int(net.class_num), det.compute_dtype, id(det), wstamp)


def _issue_prefetch(net, images, wstamp=None):
    if not (torch.is_tensor(images) and images.is_cuda and images.dtype == torch.float32 and images.dim() == 4):
        return
    key = _prefetch_key(net, images, wstamp)
    if key is None or net.base_detector._any_bn_training():
        return
    dev = images.device
    main, side = torch.cuda.current_stream(dev), _prefetch_stream(dev)
    side.wait_stream(main)   # the frames are there, and whoever ran the engine before is done with it
    from . import engine as _engine
    import contextlib
    # (the detector as one graph replay: what matters here is the host time of issuing it - the tail this runs under is host-bound)
    replay = _engine.graph_replay() if os.environ.get("MILLIEYE_PREFETCH_GRAPH", "1") != "0" else contextlib.nullcontext()
    with torch.cuda.stream(side), torch.no_grad(), replay:
        res = _frozen_detector_block(net, images)
        done = torch.cuda.Event()
        done.record(side)
    images.record_stream(side)
    net.__dict__["_det_prefetch"] = (key, res, done, images)
    net.base_detector.__dict__["_prefetch_event"] = done   # Darknet._run: nobody else touches the engine before this


def _take_prefetch(net, images, wstamp=None):
    rec = net.__dict__.pop("_det_prefetch", None)
    if rec is None:
        return None
    key, res, done, _held = rec
    main = torch.cuda.current_stream(images.device)
    main.wait_event(done)
    if key != _prefetch_key(net, images, wstamp):
        return None   # other frames, another threshold, new detector weights: computed again (behind the prefetch, see _run)
    for t in (res["img_boxes"], res["n_img_dev"], res["fm"]):
        t.record_stream(main)
    return res


def forward_train(net, images, maps, radar_boxes_location, targets, model_mode=0):
    """``targets`` given: returns ``(loss, output, metric, radar_attention)`` like the reference's training call
    (my_models.py:545-641).  ``targets is None``: the inference return of a model left in ``train()`` mode (reference
    :433-539 with batch-statistics BatchNorm: the score-map / radar-CNN BatchNorms normalise over the batch's pixels, the
    ``radar_net`` BatchNorm over the RoIs, and every one of them updates its running statistics) -> ``output [m, 8]``."""
    from .my_models import _DETECTIONS_PER_IMG, _NMS_THRESH
    from .utils.utils import xywh2xyxy

    if not images.is_cuda:
        raise hip.MeError("Network.forward needs CUDA tensors (MI355X); there is no CPU fallback")
    dev = images.device
    lib = hip.lib()
    n = images.shape[0]
    size = images.shape[-1]
    f32 = dict(device=dev, dtype=torch.float32)
    rh, eh = net.refinement_head, net.ensemble_head
    head_bns = [net.img_cnn_layers.net[1], net.radar_cnn_layers.conv1[1], net.radar_cnn_layers.conv2[1],
                net.radar_cnn_layers.conv3[1], rh.radar_net[1]]
    bn_train = all(b.training for b in head_bns)
    if not bn_train and any(b.training for b in head_bns):
        raise NotImplementedError("Network.forward(targets=...): the head BatchNorms are partly in train() and partly in "
                                  "eval() mode; call model.train() or model.eval() on the whole Network")
    # bn_train False = an eval()-mode model called with targets: the reference computes the same loss tuple with
    # running-statistics BatchNorm (my_models.py:545-641 has no mode check) and its autograd still reaches every head
    # parameter; nothing updates the running statistics
    if targets is None and model_mode == 2:  # radar only: permanent, like the reference (quirk q3)
        net.refine_threshold_img = 1

    # ---- frozen detector, NMS, proposal assembly (no grad) -------------------------------------------
    grad_on = torch.is_grad_enabled()
    with torch.no_grad():
        nxt = net.__dict__.pop("_next_images", None)
        look_ahead = nxt is not None or "_det_prefetch" in net.__dict__
        wstamp = net.base_detector.engine.weight_stamp(images.device) if look_ahead else None   # once per step, for both uses
        pre = _take_prefetch(net, images, wstamp)
        if pre is None:
            pre = _frozen_detector_block(net, images)
        num_classes, cols, cap_img = pre["num_classes"], pre["cols"], pre["cap_img"]
        img_boxes, n_img_dev, fm = pre["img_boxes"], pre["n_img_dev"], pre["fm"]
        fh, fw, fc = pre["tap_shape"]
        if nxt is not None:   # (Network.queue_detector_prefetch: the NEXT batch's frozen part, beside this batch's host-bound tail)
            _issue_prefetch(net, nxt, wstamp)
        if len(radar_boxes_location) > 0:
            radar_boxes_location[:, 1:] *= size
        n_radar = int(radar_boxes_location.shape[0])
        rb = radar_boxes_location.to(**f32).contiguous() if n_radar else torch.zeros((0, 5), **f32)
        pix = n * fh * fw
        # captured backward (_graphed_backward): everything the backward reads is allocated from the network's arena - fixed capacity,
        # the same addresses every step.  Only for the plain training configuration; everything else keeps per-step tensors.
        arena = _arena(net) if (_train_graph_enabled() and targets is not None and bn_train and model_mode == 0 and grad_on) else None
        r_cap = -(-max(n_radar, 1) // 16) * 16
        cap_rows = cap_img + r_cap

        def A(name, *shape, dtype=torch.float32):
            return arena.get(name, shape, dev, dtype) if arena is not None else torch.empty(shape, device=dev, dtype=dtype)

        def stbuf(name, c):
            return arena.get(name, (3, c), dev) if arena is not None else None

        if arena is not None:   # the prefetched block's results are per-step tensors: into the arena (three small copies)
            ib_a, nd_a, fm_a = A("img_boxes", cap_img, cols), A("n_img", 1, dtype=torch.int32), A("fm", n, fh, fw, fc)
            ib_a.copy_(img_boxes)
            nd_a.copy_(n_img_dev)
            fm_a.copy_(fm)
            img_boxes, n_img_dev, fm = ib_a, nd_a, fm_a
            rb_a = A("rb", r_cap, 5)
            if n_radar:
                rb_a[:n_radar].copy_(rb)
            rb = rb_a[:n_radar]
        ws_t = A("bn_ws", int(lib.me_bn_workspace_bytes(512)) + 256, dtype=torch.uint8)
        ws = ws_t.data_ptr() + (-ws_t.data_ptr()) % 256

        # ---- image score map: conv1x1 (+bias) -> BN(train) -> leaky ---------------------------------
        icl = net.img_cnn_layers.net
        w_img = icl[0].weight.detach().reshape(490, fc).contiguous()
        ones490, b_img = _const(dev, 490, 1.0), icl[0].bias.detach().contiguous()
        a1 = A("a1", pix, 490)
        if bn_train:
            z1 = A("z1", pix, 490)
            _conv(fm, fc, n, fh, fw, fc, w_img.view(490, 1, 1, fc), ones490, b_img, 1, 0, hip.ACT_LINEAR,
                  z1.view(n, fh, fw, 490))
            st_img = _bn_fwd(z1, 490, pix, 490, icl[1], hip.ACT_LEAKY, a1, 490, ws, stbuf("st_img", 490))
        else:  # folded affine + LeakyReLU in the conv epilogue; the backward recovers the pre-activation from a1
            z1, st_img = None, _BnEval(icl[1], dev)
            _conv(fm, fc, n, fh, fw, fc, w_img.view(490, 1, 1, fc), st_img.scale, st_img.with_bias(b_img), 1, 0,
                  hip.ACT_LEAKY, a1.view(n, fh, fw, 490))

        # ---- radar CNN (train-mode BN) ---------------------------------------------------------------
        rc = net.radar_cnn_layers
        maps = maps.to(**f32)
        mh, mw = maps.shape[2], maps.shape[3]  # may differ from (fh, fw): the demos feed the raw 32 x 32 map (quirk q15)
        pix_r = n * mh * mw
        if arena is not None:
            x0 = A("x0", n, mh, mw, 3)
            x0.copy_(maps.permute(0, 2, 3, 1))
        else:
            x0 = maps.permute(0, 2, 3, 1).contiguous()  # NHWC, 3 channels
        radar = {"x0": x0}
        prev, prev_c = x0, 3
        for li, seq in enumerate((rc.conv1, rc.conv2, rc.conv3), start=1):
            conv, bn = seq[0], seq[1]
            cout = conv.weight.shape[0]
            wp = conv.weight.detach().permute(0, 2, 3, 1).contiguous()
            r_act = A(f"r{li}", n, mh, mw, cout)
            if bn_train:
                c_raw = A(f"c{li}", n, mh, mw, cout)
                _conv(prev, prev_c, n, mh, mw, prev_c, wp, _const(dev, cout, 1.0), conv.bias.detach().contiguous(), 3, 1,
                      hip.ACT_LINEAR, c_raw)
                st = _bn_fwd(c_raw, cout, pix_r, cout, bn, hip.ACT_LEAKY, r_act, cout, ws, stbuf(f"st{li}", cout))
            else:
                c_raw, st = None, _BnEval(bn, dev)
                _conv(prev, prev_c, n, mh, mw, prev_c, wp, st.scale, st.with_bias(conv.bias), 3, 1, hip.ACT_LEAKY, r_act)
            radar[f"c{li}"], radar[f"r{li}"], radar[f"st{li}"] = c_raw, r_act, st
            prev, prev_c = r_act, cout
        conv4 = rc.conv3[3]
        w4 = conv4.weight.detach().reshape(10, 128).contiguous()
        r4 = A("r4", n, mh, mw, 10)
        _conv(prev, 128, n, mh, mw, 128, w4.view(10, 1, 1, 128), _const(dev, 10, 1.0), conv4.bias.detach().contiguous(),
              1, 0, hip.ACT_SIGMOID, r4)
        radar["r4"] = r4

        # ---- heads, part A: pooling + net0 + small dot products (saved for backward) -----------------
        wts = dict(
            w0t=(A("w0t", 490, 256).copy_(rh.net0[0].weight.detach().t()) if arena is not None
                 else rh.net0[0].weight.detach().t().contiguous()), b0=rh.net0[0].bias.detach().contiguous(),
            w1=rh.net1[0].weight.detach().contiguous(), b1=rh.net1[0].bias.detach().contiguous(),
            w2=rh.net2[0].weight.detach().contiguous(), b2=rh.net2[0].bias.detach().contiguous(),
            rw=rh.radar_net[0].weight.detach().reshape(10, 490).contiguous(),
            rb=rh.radar_net[0].bias.detach().contiguous(),
            rscale=(A("rscale", 10) if arena is not None else torch.ones(10, **f32)),
            rshift=(A("rshift", 10) if arena is not None else torch.zeros(10, **f32)),
            rw2=rh.radar_net[3].weight.detach().reshape(10).contiguous(),
            rb2=rh.radar_net[3].bias.detach().reshape(1).contiguous(),
            e1w=eh.fc1[0].weight.detach().contiguous(), e1b=eh.fc1[0].bias.detach().contiguous(),
            e2w=eh.fc2[0].weight.detach().contiguous(), e2b=eh.fc2[0].bias.detach().contiguous())
        # The one host sync of the forward: the number of image proposals (the labelling below is host work anyway, reference
        # :556).  Everything above it - both score maps with their BatchNorms, the weight layouts - needs nothing from the
        # detector's boxes and is ISSUED before the host waits, so it queues up behind the detector instead of starting when the
        # GPU has already gone idle (the step's second half is bound by the host).
        n_img = int(n_img_dev.item())
        k = n_img + n_radar
        cap = max(k, 1) if arena is None else cap_rows
        # pooled features [cap, 980] = image half | radar half (me_heads_desc.pool_scratch): the RoI pooling as its own launch and net0
        # on the matrix pipe, as in inference (same operations in the same order: the same bits as the fused launch); the halves are
        # the backward's feature operands, pitch 980
        pooled = A("pooled", cap, 980)
        feat_img, feat_rad, feat_ld = pooled[:, :490], pooled[:, 490:], 980
        if os.environ.get("MILLIEYE_TRAIN_HEADS_MFMA", "1") == "0":   # (A/B: the fused VALU launch with separate feature saves)
            pooled = None
            feat_img, feat_rad, feat_ld = A("feat_img", cap, 490), A("feat_rad", cap, 490), 490
        hidden, small = A("hidden", cap, 256), A("small", cap, 16)
        regress, refine, mask1 = A("regress", cap, 4), A("refine", cap, 2), A("mask1", cap)
        rows, key = A("rows", cap, 8), A("key", cap)
        keep = A("keep", cap, dtype=torch.uint8).zero_() if arena is not None else torch.zeros((cap,), device=dev, dtype=torch.uint8)
        if arena is not None:
            # rows [k, previous k) of the feature operands still hold the last step's values.  The captured backward multiplies them by
            # exact zeros - harmless for finite values, but a degenerate proposal pools NaN (0 / 0, like the library) and NaN x 0 would
            # poison every later weight gradient: clear what this step does not overwrite
            prev_k = arena.t.get(("prev_k", cap), 0)
            if prev_k > k:
                hidden[k:prev_k].zero_()
                if pooled is not None:
                    pooled[k:prev_k].zero_()
                else:
                    feat_img[k:prev_k].zero_()
                    feat_rad[k:prev_k].zero_()
            arena.t[("prev_k", cap)] = k
        d = hip.HeadsDesc()
        d.img_map, d.radar_map, d.img_pitch, d.radar_pitch = a1.data_ptr(), r4.data_ptr(), 490, 10
        d.n, d.fh, d.fw, d.spatial_scale = n, fh, fw, 1.0 / 16
        d.rh, d.rw = mh, mw
        d.img_boxes, d.n_img, d.n_img_cap, d.box_cols = img_boxes.data_ptr(), n_img_dev.data_ptr(), n_img, cols
        d.radar_boxes, d.n_radar = (rb.data_ptr() if n_radar else None), n_radar
        d.thr_img, d.thr_radar = float(net.refine_threshold_img), float(net.refine_threshold_radar)
        d.regress = 0 if (targets is None and model_mode == 2) else 1
        for name, t in wts.items():
            setattr(d.wts, name, t.data_ptr())
        d.regress_out, d.refine_out, d.mask1_out = regress.data_ptr(), refine.data_ptr(), mask1.data_ptr()
        d.out_rows, d.keep, d.sort_key = rows.data_ptr(), keep.data_ptr(), key.data_ptr()
        if pooled is not None:
            d.pool_scratch = pooled.data_ptr()
        else:
            d.save_feat_img, d.save_feat_rad = feat_img.data_ptr(), feat_rad.data_ptr()
        d.save_hidden, d.save_small = hidden.data_ptr(), small.data_ptr()
        if k > 0:
            hip.check(lib.me_roi_heads_f32(C.byref(d), hip.stream_ptr()), "me_roi_heads_f32")
        rh.count += 1

        # ---- radar_net BatchNorm over the RoIs, then the scalar tail ----------------------------------
        bn_r = rh.radar_net[1]
        st_r = None
        if k > 0:
            if bn_train:
                dummy = _f32(dev, k, 10)
                st_r = _bn_fwd(small[:, 6:], 16, k, 10, bn_r, hip.ACT_LINEAR, dummy, 10, ws, stbuf("st_r", 10))
                rscale = bn_r.weight.detach() * st_r.rstd
                rshift = bn_r.bias.detach() - st_r.mean * rscale
                if arena is not None:   # (fixed addresses: the descriptor the captured backward holds points at them)
                    rscale, rshift = wts["rscale"].copy_(rscale), wts["rshift"].copy_(rshift)
            else:
                st_r = _BnEval(bn_r, dev)
                rscale, rshift = st_r.scale, st_r.shift
            wts["rscale"], wts["rshift"] = rscale.contiguous(), rshift.contiguous()
            d.wts.rscale, d.wts.rshift = wts["rscale"].data_ptr(), wts["rshift"].data_ptr()
            hip.check(lib.me_heads_tail_f32(C.byref(d), small.data_ptr(), k, hip.stream_ptr()), "me_heads_tail_f32")

        # ---- output rows (same ordering rule as inference, reference :517-539) -------------------------
        ordered = _f32(dev, cap, 8)
        n_out = torch.empty((1,), device=dev, dtype=torch.int32)
        hip.check(lib.me_compact_sort_rows_f32(rows.data_ptr(), keep.data_ptr(), key.data_ptr(), k, 8, ordered.data_ptr(),
                                               n_out.data_ptr(), hip.stream_ptr()), "me_compact_sort_rows_f32")
        if targets is None:
            return ordered[:int(n_out.item())]

        # ---- labels (reference :545-604): the IoU labelling runs on the device (me_iou_labels_f32, bit-identical with
        # iou_labels_vectorized); what the host-side metric and the python-`random` negative sampling (q7) need - labels,
        # kept flags, the two confidences and the output row count - comes back in ONE copy: the step's second and last
        # host read (the first is the proposal count that sizes everything) ------------------------------------------------
        targets[:, 2:] = xywh2xyxy(targets[:, 2:])
        targets[:, 2:] *= images.shape[3]
        ib = img_boxes[:n_img]
        tg_d = targets.detach().to(**f32).contiguous()
        packed = _f32(dev, 4 * cap + 1)
        hip.check(lib.me_iou_labels_f32(img_boxes.data_ptr(), n_img, cols, rb.data_ptr() if n_radar else None, n_radar,
                                        tg_d.data_ptr() if len(tg_d) else None, int(tg_d.shape[0]), refine.data_ptr(),
                                        mask1.data_ptr(), keep.data_ptr(), packed.data_ptr(), hip.stream_ptr()),
                  "me_iou_labels_f32")
        packed[4 * cap:] = n_out  # (device-side cast + copy)
        host = packed.cpu()
        output = ordered[:int(host[4 * cap])]
        host = host[:4 * k].view(k, 4)
        iou_labels = host[:, 0:1].contiguous()
        pos_filter = (iou_labels > net.iou_thresh[1]).flatten()
        neg_filter = (iou_labels < net.iou_thresh[0]).flatten()
        positive_masks = host[:, 1] > 0
        conf_1, conf_2 = host[:, 2].contiguous(), host[:, 3].contiguous()
        flat = iou_labels.flatten()
        confs = dict(conf_1_pos=conf_1[flat > 0.5], conf_1_neg=conf_1[flat < 0.5], conf_2_pos=conf_2[flat > 0.5],
                     conf_2_neg=conf_2[flat < 0.5])
        metric = dict(total=len(iou_labels), true=pos_filter.sum(), positive=positive_masks.sum(),
                      tp=(positive_masks * pos_filter).sum().float(), conf=confs)
        pos_idx = np.where(pos_filter)[0]
        neg_idx = np.where(neg_filter)[0]
        top_k = min(len(pos_idx) * net.balance_factor, len(neg_idx))
        sample_filter = pos_filter.clone()
        selected = neg_idx[random.sample(range(len(neg_idx)), k=top_k)]  # python RNG, like the reference (q7)
        sample_filter[selected] = True
        in_conf = sample_filter.clone()
        in_focal = sample_filter.clone()
        in_focal[n_img:] = False
        masks_d = torch.stack((pos_filter, in_focal, in_conf)).to(torch.uint8).to(dev)  # one upload
        lab_d, foc_d, cnf_d = masks_d[0], masks_d[1], masks_d[2]

        # ---- loss terms + gradient seeds ----------------------------------------------------------------
        terms, seed_p, seed_c = A("terms", cap, 2), A("seed_p", cap), A("seed_c", cap)
        sums = torch.zeros(2, **f32)
        if k > 0:
            hip.check(lib.me_heads_loss_f32(mask1.data_ptr(), refine.data_ptr(), lab_d.data_ptr(), foc_d.data_ptr(),
                                            cnf_d.data_ptr(), k, float(net.alpha), float(net.loss_lambda[0]),
                                            terms.data_ptr(), seed_p.data_ptr(), seed_c.data_ptr(), hip.stream_ptr()),
                      "me_heads_loss_f32")
            _colsum(terms, 2, k, 2, sums)
        loss_value = sums[0] + sums[1]  # masks_loss + conf_loss / lambda (reference :635)
        radar_attention = r4[..., :1].permute(0, 3, 1, 2).contiguous()

    state = dict(net=net, names=_head_names(net), loss=loss_value, bn_train=bn_train, n=n, fh=fh, fw=fw, fc=fc, pix=pix, k=k, n_img=n_img,
                 mh=mh, mw=mw,
                 n_radar=n_radar, fm=fm, z1=z1, a1=a1, st_img=st_img, radar=radar, feat_img=feat_img,
                 feat_rad=feat_rad, feat_ld=feat_ld, hidden=hidden, small=small, refine=refine, mask1=mask1, seed_p=seed_p,
                 seed_c=seed_c, desc=d, keepalive=(wts, img_boxes, n_img_dev, rb, regress, rows, keep, key), st_r=st_r,
                 ws=(ws, ws_t), w_img=w_img, w4=w4, rois=None,
                 losses=dict(masks_loss=sums[0], conf_loss=sums[1]))
    with torch.no_grad():
        if arena is None:
            state["rois"] = torch.cat((ib[:, :5], rb), 0).contiguous()
        else:
            rois = A("rois", cap_rows, 5)
            if n_img:
                rois[:n_img].copy_(ib[:, :5])
            if n_radar:
                rois[n_img:k].copy_(rb)
            k_dev = A("k_dev", 1, dtype=torch.int32)
            torch.add(n_img_dev, n_radar, out=k_dev)
            state.update(rois=rois, arena=arena if k > 0 else None, cap_rows=cap_rows, k_dev=k_dev,
                         graph_sig=(n, fh, fw, fc, mh, mw, cap_rows, cols, str(dev), tuple(p.data_ptr() for p in head_parameters(net))))
    loss = _StageThree.apply(state, *head_parameters(net))
    net._last_train = state
    return loss, output, metric, radar_attention


def _backward(S, grad_out, needed=None, rows=None, k_dev=None):
    """Manual backward of the stage-3 graph; returns {parameter name: gradient}.  ``needed``: names whose gradient the
    caller wants (None = all); whole branches whose every consumer is frozen are skipped.  ``rows`` / ``k_dev`` (the captured
    form, ``_graphed_backward``): every launch runs over ``rows`` = the capacity of the arena buffers and the kernels whose
    arithmetic depends on the proposal count read it from the device word ``k_dev``."""
    net, dev = S["net"], S["fm"].device
    lib = hip.lib()
    f32 = dict(device=dev, dtype=torch.float32)
    k, n_img, n, fh, fw, fc, pix = S["k"], S["n_img"], S["n"], S["fh"], S["fw"], S["fc"], S["pix"]
    mh, mw = S["mh"], S["mw"]
    pix_r = n * mh * mw
    rh, eh, rc = net.refinement_head, net.ensemble_head, net.radar_cnn_layers
    ws = S["ws"][0]
    G = {}
    with torch.no_grad():
        if k == 0 and rows is None:
            return G
        if rows is not None:
            k = rows
        g = grad_out.to(**f32).reshape(())
        seed_p, seed_c = S["seed_p"][:k] * g, S["seed_c"][:k] * g
        d = S["desc"]
        g_o, g_hpre, h_act, xin = _f32(dev, k, 2), _f32(dev, k, 64), _f32(dev, k, 64), _f32(dev, k, 4)
        g_z2, g_rl, rl, g_rlogit = _f32(dev, k, 2), _f32(dev, k, 10), _f32(dev, k, 10), _f32(dev, k, 1)
        if k_dev is not None:
            hip.check(lib.me_heads_tail_bwd_dev_f32(C.byref(d), S["small"].data_ptr(), S["refine"].data_ptr(),
                                                    S["mask1"].data_ptr(), seed_p.data_ptr(), seed_c.data_ptr(), k, k_dev.data_ptr(),
                                                    g_o.data_ptr(), g_hpre.data_ptr(), h_act.data_ptr(), xin.data_ptr(),
                                                    g_z2.data_ptr(), g_rl.data_ptr(), rl.data_ptr(), g_rlogit.data_ptr(),
                                                    hip.stream_ptr()), "me_heads_tail_bwd_dev_f32")
        else:
            hip.check(lib.me_heads_tail_bwd_f32(C.byref(d), S["small"].data_ptr(), S["refine"].data_ptr(),
                                                S["mask1"].data_ptr(), seed_p.data_ptr(), seed_c.data_ptr(), k,
                                                g_o.data_ptr(), g_hpre.data_ptr(), h_act.data_ptr(), xin.data_ptr(),
                                                g_z2.data_ptr(), g_rl.data_ptr(), rl.data_ptr(), g_rlogit.data_ptr(),
                                                hip.stream_ptr()), "me_heads_tail_bwd_f32")
        # ---- ensemble head ------------------------------------------------------------------------------
        # (me_gemm_f32 with beta = 0 never reads C: the full-size weight gradients need no zero fill)
        dw = torch.empty((2, 64), **f32); _gemm(1, 0, 2, 64, k, g_o, 2, h_act, 64, dw, 64)
        db = _f32(dev, 2); _colsum(g_o, 2, k, 2, db)
        G["ensemble_head.fc2.0.weight"], G["ensemble_head.fc2.0.bias"] = dw, db
        dw = torch.empty((32, 2), **f32); _gemm(1, 0, 32, 2, 2 * k, g_hpre, 32, xin, 2, dw, 2)
        db = _f32(dev, 32); _colsum(g_hpre, 32, 2 * k, 32, db)
        G["ensemble_head.fc1.0.weight"], G["ensemble_head.fc1.0.bias"] = dw, db
        # ---- radar_net: 1x1, BN over RoIs (+leaky), 7x7 conv ---------------------------------------------
        dw = torch.empty((1, 10), **f32); _gemm(1, 0, 1, 10, k, g_rlogit, 1, rl, 10, dw, 10)
        db = _f32(dev, 1); _colsum(g_rlogit, 1, k, 1, db)
        G["refinement_head.radar_net.3.weight"], G["refinement_head.radar_net.3.bias"] = dw.view(1, 10, 1, 1), db
        bn_r = rh.radar_net[1]
        g_rconv = _f32(dev, k, 10)
        if S["bn_train"]:
            dg, dbt = _bn_bwd(S["small"][:, 6:], 16, g_rl, 10, k, 10, bn_r, S["st_r"], hip.ACT_LEAKY, g_rconv, 10, ws, rows_dev=k_dev)
        else:  # rl = leaky(rscale * small + rshift), the stored activated value
            dg, dbt = _bn_eval_bwd(rl, 10, g_rl, 10, k, 10, bn_r, S["st_r"], hip.ACT_LEAKY, g_rconv, 10)
        G["refinement_head.radar_net.1.weight"], G["refinement_head.radar_net.1.bias"] = dg, dbt
        dw = torch.empty((10, 490), **f32); _gemm(1, 0, 10, 490, k, g_rconv, 10, S["feat_rad"], S["feat_ld"], dw, 490)
        db = _f32(dev, 10); _colsum(g_rconv, 10, k, 10, db)
        G["refinement_head.radar_net.0.weight"], G["refinement_head.radar_net.0.bias"] = dw.view(10, 10, 7, 7), db
        wr = rh.radar_net[0].weight.detach().reshape(10, 490).contiguous()
        d_prad = _f32(dev, k, 490); _gemm(0, 0, k, 490, 10, g_rconv, 10, wr, 490, d_prad, 490)
        # ---- net2 (rows 0, 1) and net0 -----------------------------------------------------------------------
        w2 = rh.net2[0].weight.detach().contiguous()
        dw2 = torch.zeros((13, 256), **f32); _gemm(1, 0, 2, 256, k, g_z2, 2, S["hidden"], 256, dw2, 256)
        db2 = torch.zeros(13, **f32); _colsum(g_z2, 2, k, 2, db2)
        G["refinement_head.net2.0.weight"], G["refinement_head.net2.0.bias"] = dw2, db2
        dt = _f32(dev, k, 256); _gemm(0, 0, k, 256, 2, g_z2, 2, w2, 256, dt, 256)
        g_pre = _f32(dev, k, 256)
        hip.check(lib.me_act_bwd_f32(S["hidden"].data_ptr(), 256, dt.data_ptr(), 256, g_pre.data_ptr(), 256, k, 256,
                                     hip.ACT_LEAKY, hip.stream_ptr()), "me_act_bwd_f32")
        dw0 = torch.empty((256, 490), **f32); _gemm(1, 0, 256, 490, k, g_pre, 256, S["feat_img"], S["feat_ld"], dw0, 490)
        db0 = _f32(dev, 256); _colsum(g_pre, 256, k, 256, db0)
        G["refinement_head.net0.0.weight"], G["refinement_head.net0.0.bias"] = dw0, db0
        img_names = ("img_cnn_layers.net.batch_norm_0.weight", "img_cnn_layers.net.batch_norm_0.bias",
                     "img_cnn_layers.net.conv_0.weight", "img_cnn_layers.net.conv_0.bias")
        want_img = needed is None or any(nm in needed for nm in img_names)
        w0 = rh.net0[0].weight.detach().contiguous()
        d_pimg = _f32(dev, k, 490)
        if want_img:
            _gemm(0, 0, k, 490, 256, g_pre, 256, w0, 490, d_pimg, 490)
        # ---- RoI pooling backward (atomic scatter) ----------------------------------------------------------
        rois = S["rois"]
        d_r4 = torch.zeros((pix_r, 10), **f32)
        if want_img:
            d_a1 = torch.zeros((pix, 490), **f32)
            if k_dev is not None:
                hip.check(lib.me_ps_roi_align_bwd_dev_f32(d_pimg.data_ptr(), rois.data_ptr(), k, k_dev.data_ptr(), n, fh, fw, 490, 7,
                                                          1.0 / 16, d_a1.data_ptr(), 490, hip.stream_ptr()), "me_ps_roi_align_bwd_dev_f32")
            else:
                hip.check(lib.me_ps_roi_align_bwd_f32(d_pimg.data_ptr(), rois.data_ptr(), k, n, fh, fw, 490, 7, 1.0 / 16,
                                                      d_a1.data_ptr(), 490, hip.stream_ptr()), "me_ps_roi_align_bwd_f32")
        if k_dev is not None:
            hip.check(lib.me_roi_align_bwd_dev_f32(d_prad.data_ptr(), rois.data_ptr(), k, k_dev.data_ptr(), n, mh, mw, 10, 7, 1.0 / 16,
                                                   d_r4.data_ptr(), 10, hip.stream_ptr()), "me_roi_align_bwd_dev_f32")
        else:
            hip.check(lib.me_roi_align_bwd_f32(d_prad.data_ptr(), rois.data_ptr(), k, n, mh, mw, 10, 7, 1.0 / 16,
                                               d_r4.data_ptr(), 10, hip.stream_ptr()), "me_roi_align_bwd_f32")
        # ---- image score map: BN(+leaky) backward, 1x1 conv weight / bias gradient -------------------------
        if want_img:
            icl = net.img_cnn_layers.net
            dz1 = _f32(dev, pix, 490)
            if S["bn_train"]:
                dg, dbt = _bn_bwd(S["z1"], 490, d_a1, 490, pix, 490, icl[1], S["st_img"], hip.ACT_LEAKY, dz1, 490, ws)
            else:
                dg, dbt = _bn_eval_bwd(S["a1"], 490, d_a1, 490, pix, 490, icl[1], S["st_img"], hip.ACT_LEAKY, dz1, 490)
            G["img_cnn_layers.net.batch_norm_0.weight"], G["img_cnn_layers.net.batch_norm_0.bias"] = dg, dbt
            dw = torch.empty((490, fc), **f32); _gemm(1, 0, 490, fc, pix, dz1, 490, S["fm"], fc, dw, fc)
            db = _f32(dev, 490); _colsum(dz1, 490, pix, 490, db)
            G["img_cnn_layers.net.conv_0.weight"], G["img_cnn_layers.net.conv_0.bias"] = dw.view(490, fc, 1, 1), db
        # ---- radar CNN ---------------------------------------------------------------------------------------
        R = S["radar"]
        dc4 = _f32(dev, pix_r, 10)
        hip.check(lib.me_act_bwd_f32(R["r4"].data_ptr(), 10, d_r4.data_ptr(), 10, dc4.data_ptr(), 10, pix_r, 10,
                                     hip.ACT_SIGMOID, hip.stream_ptr()), "me_act_bwd_f32")
        dw = torch.empty((10, 128), **f32); _gemm(1, 0, 10, 128, pix_r, dc4, 10, R["r3"], 128, dw, 128)
        db = _f32(dev, 10); _colsum(dc4, 10, pix_r, 10, db)
        G["radar_cnn_layers.conv3.3.weight"], G["radar_cnn_layers.conv3.3.bias"] = dw.view(10, 128, 1, 1), db
        d_act = _f32(dev, pix_r, 128); _gemm(0, 0, pix_r, 128, 10, dc4, 10, S["w4"], 128, d_act, 128)
        for li, seq, cin in ((3, rc.conv3, 64), (2, rc.conv2, 32), (1, rc.conv1, 3)):
            conv, bn = seq[0], seq[1]
            cout = conv.weight.shape[0]
            dc = _f32(dev, pix_r, cout)
            if S["bn_train"]:
                dg, dbt = _bn_bwd(R[f"c{li}"], cout, d_act, cout, pix_r, cout, bn, R[f"st{li}"], hip.ACT_LEAKY, dc, cout, ws)
            else:
                dg, dbt = _bn_eval_bwd(R[f"r{li}"], cout, d_act, cout, pix_r, cout, bn, R[f"st{li}"], hip.ACT_LEAKY, dc, cout)
            G[f"radar_cnn_layers.conv{li}.1.weight"], G[f"radar_cnn_layers.conv{li}.1.bias"] = dg, dbt
            x_in = R["x0"] if li == 1 else R[f"r{li - 1}"]
            G[f"radar_cnn_layers.conv{li}.0.weight"] = _wgrad(x_in, cin, dc, cout, n, mh, mw, cin, cout, 3, 1)
            db = _f32(dev, cout); _colsum(dc, cout, pix_r, cout, db)
            G[f"radar_cnn_layers.conv{li}.0.bias"] = db
            if li > 1:  # data gradient = the forward conv kernel on the rotated, transposed weights
                wd = conv.weight.detach().flip(2, 3).permute(1, 2, 3, 0).contiguous()  # [cin][ky][kx][cout]
                d_prev = _f32(dev, n, mh, mw, cin)
                _conv(dc, cout, n, mh, mw, cout, wd, _const(dev, cin, 1.0), _const(dev, cin, 0.0), 3, 1,
                      hip.ACT_LINEAR, d_prev)
                d_act = d_prev.view(pix_r, cin)
    return G
