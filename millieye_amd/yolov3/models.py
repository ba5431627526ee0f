"""``Darknet(cfg)`` - the YOLOv3 / yolov3-tiny detector, executed by HIP kernels.

Host-side mirror of ``module3_our_dataset/yolov3/models.py`` (identical copies live in
``module2_mixed/yolov3/models.py``): same constructor, same attributes (``module_defs``,
``hyperparams``, ``module_list``, ``yolo_layers``, ``img_size``, ``seen``, ``header_info``,
``featuremap``), same ``forward(x, targets=None)`` return tuples and the same darknet
``.weights`` reader / writer.

What differs is *execution*: the module tree below only **holds parameters** under the
reference's exact names (``module_list.{i}.conv_{i}.weight``,
``module_list.{i}.batch_norm_{i}.running_mean`` ... so ``state_dict`` / ``load_state_dict`` /
``weights_init_normal`` / freeze-by-name keep working).  ``forward`` never
calls those torch modules - it hands the graph to :class:`millieye_amd.engine.DarknetEngine`,
which launches fused gfx950 kernels from ``libmillieye_hip.so``.
"""
from __future__ import division

import os

import numpy as np
import torch
import torch.nn as nn

from .. import hip
from ..engine import DarknetEngine
from ..utils.parse_config import parse_model_config, parse_data_config  # noqa: F401 (re-export)
from ..utils.utils import build_targets, to_cpu  # noqa: F401 (re-export, reference :9-10)

__all__ = ["create_modules", "Upsample", "EmptyLayer", "YOLOLayer", "Darknet",
           "parse_model_config", "parse_data_config"]


class Upsample(nn.Module):
    """Parameter-less marker for ``[upsample]`` (reference :82-92).  Executed as a fused conv
    epilogue or ``me_upsample_f32``; ``forward`` exists for API completeness only."""

    def __init__(self, scale_factor, mode="nearest"):
        super().__init__()
        self.scale_factor = scale_factor
        self.mode = mode

    def forward(self, x):
        from .. import hip
        if self.mode != "nearest":
            raise NotImplementedError("only nearest-neighbour upsampling exists in darknet cfgs")
        nhwc = x.permute(0, 2, 3, 1).contiguous()
        return hip.upsample(nhwc, int(self.scale_factor)).permute(0, 3, 1, 2)


class EmptyLayer(nn.Module):
    """Placeholder for ``[route]`` / ``[shortcut]`` (reference :95-99)."""


class YOLOLayer(nn.Module):
    """Detection layer bookkeeping (reference :102-232).

    Holds anchors / class count / loss hyper-parameters; the inference decode
    (``me_yolo_decode_f32``) is issued by the engine as part of the whole-network launch list.
    ``metrics`` must exist: ``Darknet`` finds its yolo layers by that attribute (reference :241)."""

    def __init__(self, anchors, num_classes, img_dim=416):
        super().__init__()
        self.anchors = anchors
        self.num_anchors = len(anchors)
        self.num_classes = num_classes
        self.ignore_thres = 0.5
        self.mse_loss = nn.MSELoss()
        self.bce_loss = nn.BCELoss()
        self.obj_scale = 1
        self.noobj_scale = 100
        self.metrics = {}
        self.img_dim = img_dim
        self.grid_size = 0

    def forward(self, x, targets=None, img_dim=None):
        """Stand-alone use on one raw detection map ``x`` [N, A*(5+C), G, G] (NCHW, as the reference
        passes it): returns ``(output [N, A*G*G, 5+C], 0)`` or, with ``targets``, ``(output, loss)``."""
        from .. import hip
        self.img_dim = img_dim if img_dim is not None else self.img_dim
        self.grid_size = x.size(2)
        self.stride = self.img_dim / self.grid_size
        nhwc = x.permute(0, 2, 3, 1).contiguous()
        output = hip.yolo_decode(nhwc, self.anchors, self.num_classes, self.img_dim)
        if targets is None:
            return output, 0
        return output, self.loss_from_raw(nhwc, targets)

    def loss_from_raw(self, raw_nhwc, targets, return_targets=False):
        """YOLO loss + metrics of this scale (reference :181-232 with ``utils.build_targets`` :381-440) from the raw detection
        map ``raw_nhwc`` [N, G, G, A*(5+C)] (CUDA) and ``targets`` [m,6] = (image_i, class, cx, cy, w, h) normalised to [0,1]:
        ``me_yolo_loss_fwd_f32`` (csrc/yolo_loss.hip) - target assignment, the six terms and the metric sums in three launches
        and one read-back (the reference runs ~200 torch ops and ~15 host reads per scale).  Sets ``self.metrics`` like the
        reference; ``return_targets`` adds the dense build_targets tensors ``me_yolo_loss_bwd_f32`` takes.  There is no CPU
        path (the torch-op restatement the tests compare with is ``oracle/darknet_ref.py:yolo_loss_terms``)."""
        if not raw_nhwc.is_cuda:
            raise hip.MeError("YOLOLayer.loss_from_raw needs CUDA tensors (MI355X); there is no CPU fallback")
        import ctypes as C
        n, g = raw_nhwc.shape[0], raw_nhwc.shape[1]
        dev, na, nc = raw_nhwc.device, self.num_anchors, self.num_classes
        raw = raw_nhwc
        if raw.dtype != torch.float32 or raw.stride(3) != 1 or raw.stride(1) != g * raw.stride(2) or \
                raw.stride(0) != g * g * raw.stride(2):
            raw = raw.float().contiguous()
        stride = self.img_dim / g
        scaled = [float(np.float32(v)) for aw, ah in self.anchors for v in (aw / stride, ah / stride)]  # float32 like the reference's tensor
        self.scaled_anchors = torch.tensor(scaled, dtype=torch.float32).view(na, 2)  # (host copy; the kernel takes the floats)
        tg = targets.to(device=dev, dtype=torch.float32).contiguous()
        cells = (n, na, g, g)
        f32 = dict(device=dev, dtype=torch.float32)
        obj, noobj = torch.empty(cells, device=dev, dtype=torch.uint8), torch.empty(cells, device=dev, dtype=torch.uint8)
        tx, ty, tw, th, tconf, cmask, ious = (torch.empty(cells, **f32) for _ in range(7))
        tcls = torch.empty(cells + (nc,), **f32)
        result = torch.zeros(16, **f32)  # (never uninitialised memory as a loss, whatever happens to the launch)
        anchors_c = (C.c_float * len(scaled))(*scaled)
        ws = _yolo_loss_workspace(dev)
        hip.check(hip.lib().me_yolo_loss_fwd_f32(
            raw.data_ptr(), raw.stride(2), n, g, na, nc, anchors_c, tg.data_ptr(), tg.shape[0], float(self.ignore_thres),
            float(self.obj_scale), float(self.noobj_scale), obj.data_ptr(), noobj.data_ptr(), tx.data_ptr(), ty.data_ptr(),
            tw.data_ptr(), th.data_ptr(), tcls.data_ptr(), tconf.data_ptr(), cmask.data_ptr(), ious.data_ptr(), ws.data_ptr(),
            result.data_ptr(), hip.stream_ptr()), "me_yolo_loss_fwd_f32")
        scalars = result.tolist()  # the one host read of this scale
        if scalars[15] != 0.0:
            raise IndexError("YOLO loss: a target lies outside the batch, the grid (cx / cy must be < 1) or the class range")
        keys = ("loss", "x", "y", "w", "h", "conf", "cls", "cls_acc", "recall50", "recall75", "precision", "conf_obj",
                "conf_noobj")
        self.metrics = dict(zip(keys, scalars[:13]))
        self.metrics["grid_size"] = g
        total_loss = result[0].clone()
        if return_targets:
            bt = dict(obj=obj, noobj=noobj, tx=tx, ty=ty, tw=tw, th=th, tcls=tcls, tconf=tconf, n_obj=int(scalars[13]),
                      n_noobj=int(scalars[14]))
            return total_loss, bt
        return total_loss




_LOSS_WS = {}


def _yolo_loss_workspace(dev):
    """One per (device, stream): two YOLO layers / models computing their loss on different streams never share the block
    partials and the ticket.  Zeroed at creation; ``me_yolo_loss_fwd_f32`` resets its ticket words in front of every call."""
    key = (str(dev), torch.cuda.current_stream(dev).cuda_stream)
    ws = _LOSS_WS.get(key)
    if ws is None:
        ws = _LOSS_WS[key] = torch.zeros(int(hip.lib().me_yolo_loss_workspace_bytes()), dtype=torch.uint8, device=dev)
    return ws


def _conv_block(index, spec, in_channels):
    seq = nn.Sequential()
    bn = int(spec["batch_normalize"])
    filters = int(spec["filters"])
    k = int(spec["size"])
    seq.add_module(f"conv_{index}", nn.Conv2d(in_channels, filters, kernel_size=k, stride=int(spec["stride"]),
                                              padding=(k - 1) // 2, bias=not bn))
    if bn:
        seq.add_module(f"batch_norm_{index}", nn.BatchNorm2d(filters, momentum=0.9, eps=1e-5))
    if spec["activation"] == "leaky":
        seq.add_module(f"leaky_{index}", nn.LeakyReLU(0.1))
    return seq, filters


def _maxpool_block(index, spec, in_channels):
    seq = nn.Sequential()
    k, s = int(spec["size"]), int(spec["stride"])
    if k == 2 and s == 1:
        seq.add_module(f"_debug_padding_{index}", nn.ZeroPad2d((0, 1, 0, 1)))
    seq.add_module(f"maxpool_{index}", nn.MaxPool2d(kernel_size=k, stride=s, padding=int((k - 1) // 2)))
    return seq, in_channels


def create_modules(module_defs):
    """cfg blocks -> ``(hyperparams, nn.ModuleList)`` with the reference's child names
    (reference :12-79).  Pops the ``[net]`` block off ``module_defs`` like the reference."""
    hyperparams = module_defs.pop(0)
    widths = [int(hyperparams["channels"])]  # widths[1 + i] = output channels of module i
    module_list = nn.ModuleList()
    for index, spec in enumerate(module_defs):
        kind = spec["type"]
        if kind == "convolutional":
            seq, filters = _conv_block(index, spec, widths[-1])
        elif kind == "maxpool":
            seq, filters = _maxpool_block(index, spec, widths[-1])
        elif kind == "upsample":
            seq = nn.Sequential()
            seq.add_module(f"upsample_{index}", Upsample(scale_factor=int(spec["stride"]), mode="nearest"))
            filters = widths[-1]
        elif kind == "route":
            seq = nn.Sequential()
            seq.add_module(f"route_{index}", EmptyLayer())
            filters = sum(widths[1:][int(l)] for l in spec["layers"].split(","))
        elif kind == "shortcut":
            seq = nn.Sequential()
            seq.add_module(f"shortcut_{index}", EmptyLayer())
            filters = widths[1:][int(spec["from"])]
        elif kind == "yolo":
            seq = nn.Sequential()
            flat = [int(v) for v in spec["anchors"].split(",")]
            pairs = [(flat[j], flat[j + 1]) for j in range(0, len(flat), 2)]
            chosen = [pairs[int(m)] for m in spec["mask"].split(",")]
            seq.add_module(f"yolo_{index}", YOLOLayer(chosen, int(spec["classes"]), int(hyperparams["height"])))
            filters = widths[-1]
        else:
            raise ValueError(f"unsupported cfg block [{kind}] (module {index})")
        module_list.append(seq)
        widths.append(filters)
    return hyperparams, module_list


class Darknet(nn.Module):
    """YOLOv3 object detector (reference :235-352), forward on MI355X."""

    def __init__(self, config_path, img_size=416):
        super().__init__()
        self.module_defs = parse_model_config(config_path)
        self.hyperparams, self.module_list = create_modules(self.module_defs)
        self.yolo_layers = [layer[0] for layer in self.module_list if hasattr(layer[0], "metrics")]
        self.img_size = img_size
        self.seen = 0
        self.header_info = np.array([0, 0, 0, self.seen, 0], dtype=np.int32)
        # not a submodule / not in state_dict: plain attribute via object.__setattr__
        object.__setattr__(self, "_engines", {})
        # storage type of the accelerated inference path: "f32" (default, the mode the 1e-3 parity bar is quoted on) or
        # "bf16" / "f16" (BASELINE configs[2]/[4]; 16-bit activations + weights, fp32 accumulation).  Training always runs fp32.
        object.__setattr__(self, "compute_dtype", os.environ.get("MILLIEYE_DTYPE", "f32"))

    # -- execution -----------------------------------------------------------------------------
    def engine_for(self, dtype):
        eng = self._engines.get(dtype)
        if eng is None:
            eng = DarknetEngine(self, dtype)
            if self._engines:
                eng.tap_module = next(iter(self._engines.values())).tap_module
            self._engines[dtype] = eng
        return eng

    @property
    def engine(self):
        """The fp32 engine (training paths, tools); inference goes through ``engine_for(self.compute_dtype)``."""
        return self.engine_for("f32")

    @property
    def featuremap_module(self):
        """Module index tapped as ``featuremap`` (8 = the reference's ``conv_8``; see
        :func:`millieye_amd.engine.pick_tap_module` for the Darknet-53 extension)."""
        return self.engine.tap_module

    @featuremap_module.setter
    def featuremap_module(self, index):
        self.engine  # make sure at least the fp32 engine exists
        for eng in self._engines.values():
            eng.tap_module = index
            eng._plans.clear()

    def _run(self, x, keep_raw=False, nms_conf=None):
        """Internal: (plan, yolo_outputs) without cloning the feature tap (used by Network).  ``nms_conf``: see
        ``DarknetEngine.run`` (the decode fills the NMS candidate lists)."""
        ev = self.__dict__.pop("_prefetch_event", None)
        if ev is not None:   # train_path._issue_prefetch ran the engine on its own stream: its arena is busy until then
            torch.cuda.current_stream(x.device).wait_event(ev)
        eng = self.engine if keep_raw else self.engine_for(self.compute_dtype)
        return eng.run(x, keep_raw, nms_conf=nms_conf)

    def forward(self, x, targets=None):
        """``(featuremap, yolo_outputs)``, or with ``targets`` ``(loss, featuremap, yolo_outputs)`` where
        ``loss`` is the summed YOLO loss of every scale (reference :261-267).  Under autograd with parameters that require
        gradients the loss is differentiable (``_forward_train``: millieye_amd/detector_train.py, or - ``compute_dtype`` "bf16" /
        "f16" - the mixed-precision step of millieye_amd/detector_train16.py); otherwise it is a value."""
        if targets is not None and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return self._forward_train(x, targets)
        if self._any_bn_training():
            # model.train() without autograd (reference :35,247-267: legal, batch statistics, running statistics updated):
            # the inference engine folds BatchNorm into the convolution and cannot do that - the training path's forward can
            return self._forward_batch_stats(x, targets)
        plan, yolo_outputs = self._run(x, keep_raw=targets is not None)
        if getattr(plan, "graph", None) is not None:
            yolo_outputs = yolo_outputs.clone()  # graph replays write a static buffer: the caller gets its own tensor
        if plan.tap is not None:
            # fresh tensor per call like the reference's ``x.detach()`` of a fresh activation;
            # memory stays channels-last (NHWC), shape is the reference's [N,256,S/16,S/16]
            self.featuremap = plan.tap.clone() if plan.tap.dtype == torch.float32 else plan.tap.float()
        if not hasattr(self, "featuremap"):
            # same failure the reference has for cfgs without a ``conv_8`` child
            raise AttributeError("'Darknet' object has no attribute 'featuremap'")
        if targets is not None:
            loss = 0
            for layer, raw in zip(self.yolo_layers, plan.yolo_raw):
                layer.img_dim = x.shape[2]
                loss = loss + layer.loss_from_raw(raw, targets)
            return loss, self.featuremap, yolo_outputs
        return self.featuremap, yolo_outputs

    def _any_bn_training(self):
        """True when a BatchNorm of the detector is in train() mode.  The (container, key, BatchNorm) triples are collected once
        (a walk over ``self.modules()`` - ~250 modules - was 50-100 us of host time in front of the first launch of every
        inference call) and re-validated per call by one dict lookup each, so a replaced block or BatchNorm
        triggers a fresh walk."""
        cached = self.__dict__.get("_bn_triples")
        blocks = self._modules["module_list"]._modules
        if cached is not None and cached[0] == len(blocks):
            ok = True
            for cont, key, bn in cached[1]:
                if cont.get(key) is not bn:
                    ok = False
                    break
                if bn.training:
                    return True
            if ok and all(blocks.get(k) is b for k, b in cached[2]):
                return False
        triples = [(m._modules, k, c) for m in self.modules() for k, c in m._modules.items() if isinstance(c, nn.BatchNorm2d)]
        self.__dict__["_bn_triples"] = (len(blocks), triples, list(blocks.items()))
        return any(bn.training for _c, _k, bn in triples)

    def _forward_batch_stats(self, x, targets):
        """``Darknet.forward`` under ``model.train()`` outside autograd: BatchNorm layers in train() mode normalise with the
        batch statistics and update their running statistics (``me_bn_train_fwd_f32``), the others stay folded; outputs (and,
        with ``targets``, the loss VALUE) come from that forward (either storage mode: detector_train.py / detector_train16.py)."""
        from ..detector_train import DetectorTrainer
        with torch.no_grad():
            if self.compute_dtype != "f32":   # 16-bit storage modes: float32 batch statistics over the 16-bit operands' products
                from ..detector_train16 import DetectorTrainer16
                st = DetectorTrainer16(self).forward(x)
            else:
                st = DetectorTrainer(self).forward(x)
            yolo_outputs = self._decode_state(st, x)
            if targets is None:
                return self.featuremap, yolo_outputs
            loss = 0
            for layer, (_idx, raw) in zip(self.yolo_layers, sorted(st.raws.items())):
                layer.img_dim = x.shape[2]
                loss = loss + layer.loss_from_raw(raw, targets)
            return loss, self.featuremap, yolo_outputs

    def _decode_state(self, st, x):
        """Decoded rows ``[N, R, 5 + C]`` + the feature tap from a training-path forward state (both BatchNorm modes)."""
        from .. import hip
        rows_total = sum(l.num_anchors * r.shape[1] * r.shape[2] for l, (_i, r) in zip(self.yolo_layers, sorted(st.raws.items())))
        yolo_outputs, off = None, 0
        for layer, (_idx, raw) in zip(self.yolo_layers, sorted(st.raws.items())):
            yolo_outputs = hip.yolo_decode(raw, layer.anchors, layer.num_classes, x.shape[2], out=yolo_outputs,
                                           rows_total=rows_total, row_offset=off)
            off += layer.num_anchors * raw.shape[1] * raw.shape[2]
            layer.grid_size, layer.stride = raw.shape[1], x.shape[2] / raw.shape[1]
        tap = self.engine.tap_module
        if tap is not None and tap < len(st.outs) and st.outs[tap] is not None:
            tap_t = st.outs[tap]
            self.featuremap = hip.nhwc_to_nchw(tap_t if tap_t.dtype == torch.float32 else tap_t.float())
        if not hasattr(self, "featuremap"):
            raise AttributeError("'Darknet' object has no attribute 'featuremap'")
        return yolo_outputs

    def _forward_train(self, x, targets):
        """``(loss, featuremap, yolo_outputs)`` with a differentiable loss (reference :181-267 under autograd): every
        module output is kept and ``loss.backward()`` runs the HIP backward of millieye_amd/detector_train.py.  The
        decoded ``yolo_outputs`` / ``featuremap`` are taken from that same forward."""
        from .. import hip
        from ..detector_train import _DarknetLoss
        named = [(k, p) for k, p in self.named_parameters()]
        loss = _DarknetLoss.apply(self, x, targets, [k for k, _ in named], *[p for _, p in named])
        st = self.__dict__.pop("_train_state")
        with torch.no_grad():  # decoded rows + feature tap straight from the training forward (works for both BN modes)
            rows_total = sum(l.num_anchors * r.shape[1] * r.shape[2] for l, (_i, r) in zip(self.yolo_layers, sorted(st.raws.items())))
            yolo_outputs, off = None, 0
            for layer, (_idx, raw) in zip(self.yolo_layers, sorted(st.raws.items())):
                yolo_outputs = hip.yolo_decode(raw, layer.anchors, layer.num_classes, x.shape[2], out=yolo_outputs,
                                               rows_total=rows_total, row_offset=off)
                off += layer.num_anchors * raw.shape[1] * raw.shape[2]
                layer.grid_size, layer.stride = raw.shape[1], x.shape[2] / raw.shape[1]
            tap = self.engine.tap_module
            if tap is not None and tap < len(st.outs) and st.outs[tap] is not None:
                tap_t = st.outs[tap]
                self.featuremap = hip.nhwc_to_nchw(tap_t if tap_t.dtype == torch.float32 else tap_t.float())
        if not hasattr(self, "featuremap"):
            raise AttributeError("'Darknet' object has no attribute 'featuremap'")
        return loss, self.featuremap, yolo_outputs

    # -- darknet .weights I/O (reference :269-352) ---------------------------------------------
    def _conv_blocks(self, stop=None):
        for i, (spec, module) in enumerate(zip(self.module_defs, self.module_list)):
            if stop is not None and i == stop:
                return
            if spec["type"] == "convolutional":
                yield spec, module

    def load_darknet_weights(self, weights_path):
        """5 x int32 header, then per conv: [bn.bias, bn.weight, bn.running_mean, bn.running_var]
        or [conv.bias], then conv.weight (OIHW), all float32."""
        with open(weights_path, "rb") as fh:
            header = np.fromfile(fh, dtype=np.int32, count=5)
            self.header_info = header
            self.seen = header[3]
            stream = np.fromfile(fh, dtype=np.float32)
        cutoff = None
        if "darknet53.conv.74" in weights_path:
            cutoff = 75
        if "yolov3-tiny.conv.15" in weights_path:
            cutoff = 15
        cursor = 0

        def take(dst):
            nonlocal cursor
            count = dst.numel()
            with torch.no_grad():  # not ``.data.copy_`` (reference style): that leaves ``dst._version`` where it was
                dst.copy_(torch.from_numpy(stream[cursor: cursor + count]).view_as(dst))
            cursor += count

        for spec, module in self._conv_blocks(stop=cutoff):
            conv = module[0]
            if spec["batch_normalize"]:
                bn = module[1]
                for dst in (bn.bias, bn.weight, bn.running_mean, bn.running_var):
                    take(dst)
            else:
                take(conv.bias)
            take(conv.weight)
        self.invalidate_weights()

    @staticmethod
    def invalidate_weights():
        """Rebuild every packed device copy of the parameters before the next forward.  Only needed after writes that
        bypass the autograd version counter (``param.data.copy_()`` / ``.data.normal_()`` in reference-style user code);
        ``load_state_dict``, optimizers, ``nn.init`` and this class's own loaders are tracked automatically."""
        from ..engine import invalidate_weights
        invalidate_weights()

    def save_darknet_weights(self, path, cutoff=-1):
        """Inverse of :meth:`load_darknet_weights`; ``cutoff`` slices ``module_defs[:cutoff]``
        exactly like the reference (so the default -1 drops the last module)."""
        with open(path, "wb") as fp:
            self.header_info[3] = self.seen
            self.header_info.tofile(fp)
            for spec, module in zip(self.module_defs[:cutoff], self.module_list[:cutoff]):
                if spec["type"] != "convolutional":
                    continue
                conv = module[0]
                if spec["batch_normalize"]:
                    bn = module[1]
                    for src in (bn.bias, bn.weight, bn.running_mean, bn.running_var):
                        src.data.cpu().numpy().tofile(fp)
                else:
                    conv.bias.data.cpu().numpy().tofile(fp)
                conv.weight.data.cpu().numpy().tofile(fp)
