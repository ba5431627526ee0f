"""The detector's training step as ONE captured hipGraph: forward + the three YOLO losses + backward of every layer (round 5).

Reference: ``module3_our_dataset/yolov3/models.py:181-267`` under autograd - ``loss = model(x, targets)[0]; loss.backward()``.
The eager form of that step (:mod:`millieye_amd.detector_train` / :mod:`millieye_amd.detector_train16`) issues ~520 launches from
Python; in the 16-bit storage modes the GPU finishes them faster than the host can issue them (8.0 ms of host time per 8.1 ms
step at batch 8, ``profiles/r05_kernel_evolution.md`` section 4b).  The launch sequence of a step depends on the batch SHAPE only,
so it is captured once and replayed:

==========================  =============================================================================================
frames, targets             copied into static buffers in front of the replay; the target table has a fixed capacity and a device
                            row count (``me_yolo_loss_fwd_counted_f32``) - the step's target count is not a launch argument
``n_obj`` / ``n_noobj``     stay in device memory (``me_yolo_loss_bwd_dev_f32`` reads them from the forward's ``result[16]``):
                            the eager step reads them back between the forward and the backward, three host syncs per step
weights                     the packing launches (fp32 OHWI / rotated / parity layouts, the 16-bit copies) are part of the graph:
                            every replay packs the parameters as they are then (an optimizer step between replays is seen)
gradients                   static fp32 tensors of the graph's pool, attached as ``p.grad`` after the replay (added to a ``.grad``
                            that is already there, like autograd's accumulation)
metrics, bad targets        ``metrics()`` reads the three ``result[16]`` rows when asked (one host read); a target outside the
                            batch / grid / class range raises IndexError THERE (the eager step raises it in the forward)
==========================  =============================================================================================

The kernels, their order and their arguments are those of the eager step, so loss and gradients equal the eager step's bit for
bit (``tests/test_gpu_graph_step.py``).  Scope: BatchNorm in eval() mode (the reference keeps the detector in eval(),
``train.py:170``; train()-mode statistics are an eager fp32 path), one device, CUDA frames or host frames of a fixed shape.
With a process group the gradients are exchanged after the replay (``parallel.allreduce_gradients``), not inside the graph.
"""
import ctypes as C

import numpy as np
import torch

from . import hip
from .yolov3.models import _yolo_loss_workspace

_KEYS = ("loss", "x", "y", "w", "h", "conf", "cls", "cls_acc", "recall50", "recall75", "precision", "conf_obj", "conf_noobj")


class GraphedDetectorStep:
    """``step = GraphedDetectorStep(model); loss = step(x, targets); optimizer.step(); optimizer.zero_grad()``.

    ``model``: a :class:`millieye_amd.yolov3.models.Darknet` on a CUDA device, any ``compute_dtype``.  ``max_targets``: capacity
    of the static target table (rows of ``targets`` [m, 6] = image, class, cx, cy, w, h in [0, 1]).  ``group`` / ``average``:
    all-reduce of the gradients after the replay when a process group is up (``False``: never).  The graph is captured at the
    first call and again whenever the frame shape, a parameter's storage or ``compute_dtype`` changes.
    """

    def __init__(self, model, max_targets=256, group=None, average=False, exchange=True):
        self.m = model
        self.cap = int(max_targets)
        self.group, self.average, self.exchange = group, bool(average), bool(exchange)
        self.graph = None
        self._key = None
        self.params = []
        self._ring, self._ring_at = [], 0

    # ------------------------------------------------------------------------------------------ the step, eager (this is what is captured)
    def _losses_and_backward(self):
        m, lib = self.m, hip.lib()
        x = self.x
        if m.compute_dtype != "f32":
            from .detector_train16 import DetectorTrainer16
            trainer = DetectorTrainer16(m)
        else:
            from .detector_train import DetectorTrainer
            trainer = DetectorTrainer(m)
        st = trainer.forward(x)
        dev = x.device
        f32 = dict(device=dev, dtype=torch.float32)
        results = torch.zeros((len(m.yolo_layers), 16), **f32)
        draws, keep = {}, []
        ws = _yolo_loss_workspace(dev)
        sp = hip.stream_ptr()
        for row, (layer, (idx, raw)) in enumerate(zip(m.yolo_layers, sorted(st.raws.items()))):
            layer.img_dim = x.shape[2]
            n, g = raw.shape[0], raw.shape[1]
            na, nc = layer.num_anchors, layer.num_classes
            if raw.dtype != torch.float32 or raw.stride(3) != 1 or raw.stride(1) != g * raw.stride(2) or \
                    raw.stride(0) != g * g * raw.stride(2):
                raise hip.MeError("captured detector step: a raw detection map is not a pitched float32 NHWC tensor")
            stride = layer.img_dim / g
            scaled = [float(np.float32(v)) for aw, ah in layer.anchors for v in (aw / stride, ah / stride)]
            anchors_c = (C.c_float * len(scaled))(*scaled)
            cells = (n, na, g, g)
            obj, noobj = torch.empty(cells, device=dev, dtype=torch.uint8), torch.empty(cells, device=dev, dtype=torch.uint8)
            tx, ty, tw, th, tconf, cmask, ious = (torch.empty(cells, **f32) for _ in range(7))
            tcls = torch.empty(cells + (nc,), **f32)
            res = results[row]
            hip.check(lib.me_yolo_loss_fwd_counted_f32(
                raw.data_ptr(), raw.stride(2), n, g, na, nc, anchors_c, self.table.data_ptr(), self.cap, self.count.data_ptr(),
                float(layer.ignore_thres), float(layer.obj_scale), float(layer.noobj_scale), obj.data_ptr(), noobj.data_ptr(),
                tx.data_ptr(), ty.data_ptr(), tw.data_ptr(), th.data_ptr(), tcls.data_ptr(), tconf.data_ptr(), cmask.data_ptr(),
                ious.data_ptr(), ws.data_ptr(), res.data_ptr(), sp), "me_yolo_loss_fwd_counted_f32")
            draw = torch.empty_like(raw)
            hip.check(lib.me_yolo_loss_bwd_dev_f32(
                raw.data_ptr(), raw.stride(2), n, g, na, nc, obj.data_ptr(), noobj.data_ptr(), tx.data_ptr(), ty.data_ptr(),
                tw.data_ptr(), th.data_ptr(), tcls.data_ptr(), tconf.data_ptr(), res.data_ptr(), float(layer.obj_scale),
                float(layer.noobj_scale), None, draw.data_ptr(), draw.stride(2), sp), "me_yolo_loss_bwd_dev_f32")
            draws[idx] = draw
            keep.append((obj, noobj, tx, ty, tw, th, tconf, cmask, ious, tcls))
            layer.grid_size, layer.stride = g, stride
        loss = results[0, 0].clone()
        for row in range(1, results.shape[0]):   # the eager step's order: ((l0 + l1) + l2)
            loss = loss + results[row, 0]
        self.bad.add_(results[:, 15].sum())
        grads = trainer.backward(st, draws, None)
        return loss, results, grads, (st, keep, draws)

    # ------------------------------------------------------------------------------------------ capture
    def _signature(self, x):
        # (the Parameter objects are looked up once per capture: walking the module tree costs more than the rest of this call;
        #  a parameter whose STORAGE moved - .to(), .half(), a re-assigned .data - is seen here and the step is captured again)
        return (tuple(x.shape), self.m.compute_dtype, tuple([p.data_ptr() for p in self.params]), self._scratch_ptrs())

    def _scratch_ptrs(self):
        """Addresses of every scratch buffer the captured launches were handed that an EAGER step on the same model (or on the
        shared side stream) may later outgrow and replace: the model-level slab / partial-sum scratches of the two trainers and
        the library's per-stream workspaces.  A replaced buffer changes this tuple and the step is captured again instead of
        replaying launches that point into freed memory."""
        d = self.m.__dict__
        out = []
        for name in ("_wgrad16_ws", "_affine16_ws"):
            for t in d.get(name) or ():
                out.append(t.data_ptr() if t is not None else 0)
        for name in ("_affine16_layer_ws", "_affine_layer_ws"):
            tab = d.get(name) or {}
            out.extend((k, tab[k].data_ptr()) for k in sorted(tab))
        out.extend((k[0], k[2], t.data_ptr()) for k, t in sorted(hip._ws_cache.items(), key=lambda kv: kv[0]) if t is not None)
        return tuple(out)

    def _device(self):
        return next(self.m.parameters()).device

    def _capture(self, x):
        m = self.m
        dev = self._device()
        if dev.type != "cuda":
            raise hip.MeError("GraphedDetectorStep needs the model on a CUDA device (MI355X); there is no CPU fallback")
        if m._any_bn_training():
            raise NotImplementedError("captured detector step: BatchNorm in train() mode (batch statistics, running-stat updates on "
                                      "the host side) is an eager float32 path; model.eval() keeps the statistics fixed, as the "
                                      "reference's loops do (train.py:170)")
        self.graph = None
        self.names = [k for k, _ in m.named_parameters()]
        self.params = [p for _, p in m.named_parameters()]
        self.x = torch.empty(tuple(x.shape), device=dev, dtype=torch.float32)
        self.x.copy_(x)
        words = self.cap * 6 + 2   # the table, then the row count (int32 bits) in the word behind it
        self.dev_words = torch.zeros(words, device=dev, dtype=torch.float32)
        self.table = self.dev_words[: self.cap * 6]
        self.count = self.dev_words.view(torch.int32)[self.cap * 6:]
        self.bad = torch.zeros((), device=dev, dtype=torch.float32)
        self._ring = [(torch.zeros(words, dtype=torch.float32).pin_memory(), torch.cuda.Event()) for _ in range(4)]
        self._ring_at = 0
        stream = self.stream = torch.cuda.Stream(dev)
        stream.wait_stream(torch.cuda.current_stream(dev))
        eng = m.engine
        with torch.cuda.stream(stream), torch.no_grad():
            for _ in range(2):   # eager: autotuner picks, kernel attributes, workspaces and descriptor tables of THIS stream settle
                self._losses_and_backward()
            torch.cuda.synchronize(dev)
            for i, d in enumerate(m.module_defs):   # every replay packs the weights: the packing launches must be in the graph
                if d["type"] == "convolutional":
                    eng._conv_weights(i)._stamp = None
            graph = torch.cuda.CUDAGraph()
            # thread_local: only THIS thread's calls are checked against the capture - a process group's watchdog thread polls
            # events of its own streams while a rank captures (the default "global" mode would fail the capture on that)
            with torch.cuda.graph(graph, stream=stream, capture_error_mode="thread_local"):
                self.loss, self.results, grads, self._held = self._losses_and_backward()
        torch.cuda.current_stream(dev).wait_stream(stream)
        self.static_grads = [grads.get(k) for k in self.names]   # (None: a parameter the backward produces no gradient for)
        self.graph = graph
        self.bad.zero_()
        self._key = self._signature(x)

    # ------------------------------------------------------------------------------------------ replay
    def __call__(self, x, targets):
        """One training step's forward + backward: returns the loss (a device scalar) and leaves the gradients in ``p.grad``."""
        if self.graph is None or self._key != self._signature(x):
            self._capture(x)
        m_rows = 0 if targets is None else int(targets.shape[0])
        if m_rows > self.cap:
            raise ValueError(f"GraphedDetectorStep: {m_rows} target rows, the static table holds {self.cap} (max_targets)")
        if m_rows and targets.is_cuda:   # device targets: straight into the table, the count by a fill (no host read of the rows)
            self.table[: m_rows * 6].copy_(targets.detach().reshape(-1))
            self.count.fill_(m_rows)
        else:                            # host targets: table + count through a pinned slot, one asynchronous copy
            host, done = self._ring[self._ring_at]
            self._ring_at = (self._ring_at + 1) % len(self._ring)
            done.synchronize()   # (the copy that last read this pinned slot: four steps ago)
            if m_rows:
                host[: m_rows * 6].copy_(targets.detach().reshape(-1))
            host.view(torch.int32)[self.cap * 6] = m_rows
            self.dev_words.copy_(host, non_blocking=True)
            done.record()
        if x.data_ptr() != self.x.data_ptr():
            self.x.copy_(x, non_blocking=True)
        for p, g in zip(self.params, self.static_grads):   # a .grad that still IS the static tensor (nobody zeroed it): its values
            if g is not None and p.grad is g:              # are about to be overwritten - keep them, the new ones are added below
                p.grad = g.clone()
        self.graph.replay()
        add_to, add_from = [], []
        for p, g in zip(self.params, self.static_grads):
            if g is None or not p.requires_grad:
                continue
            if p.grad is None:
                p.grad = g
            else:
                add_to.append(p.grad)
                add_from.append(g)
        if add_to:
            torch._foreach_add_(add_to, add_from)
        if self.exchange and torch.distributed.is_available() and torch.distributed.is_initialized() and \
                torch.distributed.get_world_size(self.group) > 1:
            from . import parallel
            parallel.allreduce_gradients([p for p in self.params if p.grad is not None], group=self.group, static_pattern=True)
            if self.average:
                torch._foreach_div_([p.grad for p in self.params if p.grad is not None], torch.distributed.get_world_size(self.group))
        return self.loss.clone()

    def metrics(self):
        """The reference's ``YOLOLayer.metrics`` dicts of the LAST step, one per scale (one host read); raises IndexError when a
        target of any step since the last call lay outside the batch, the grid or the class range."""
        if self.graph is None:
            raise RuntimeError("GraphedDetectorStep.metrics(): no step has run yet")
        rows = torch.cat((self.results.reshape(-1), self.bad.reshape(1))).tolist()
        if rows[-1] != 0.0:
            self.bad.zero_()
            raise IndexError("YOLO loss: a target lies outside the batch, the grid (cx / cy must be < 1) or the class range")
        out = []
        for i, layer in enumerate(self.m.yolo_layers):
            d = dict(zip(_KEYS, rows[16 * i: 16 * i + 13]))
            d["grid_size"] = layer.grid_size
            layer.metrics = d
            out.append(d)
        return out
