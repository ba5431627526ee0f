"""Host-side planner / executor of the Darknet graph on the HIP library.

``Darknet.forward`` of the reference (``module3_our_dataset/yolov3/models.py:247-267``) is a
Python loop that dispatches one library op per cfg module and keeps every activation
alive.  Here the cfg graph is compiled once per input shape into a short list of
``libmillieye_hip`` launches:

* conv + BN(eval) + LeakyReLU + bias are one kernel (``me_conv2d_f32``), with the
  ``[shortcut]`` add and a following ``[upsample]`` fused into its epilogue when nobody
  else reads the intermediate;
* ``[route]`` concatenation is free: the producers write straight into channel slices of
  one wider NHWC buffer (pitched stores), single-source routes are aliases;
* activations live in one arena with liveness-based reuse (only tensors that a later
  ``[route]``/``[shortcut]`` reads stay alive), so the working set of Darknet-53 at batch 8
  stays inside the 256 MiB Infinity Cache instead of the reference's ~2.5 GB;
* each ``[yolo]`` decode writes its rows directly at its offset of the final
  ``[N, R, 5+C]`` tensor (no ``torch.cat``).

Nothing in this file computes: it owns shapes, pointers and launch order.
"""
import ctypes as C
import os

import torch

from . import hip

__all__ = ["DarknetEngine", "ConvWeights", "pick_tap_module", "invalidate_weights", "bump_versions"]

_ALIGN = 256  # bytes

# Packed-weight caches compare ``(data_ptr, _version)`` stamps.  Writers that bypass the version counter - reference-style
# ``param.data.copy_()`` / ``.data.normal_()`` in user code, or a kernel writing through a raw pointer - call
# :func:`invalidate_weights` (or ``Darknet.invalidate_weights()``): the epoch is part of every stamp.
_EPOCH = [0]


def invalidate_weights():
    """Force every packed device copy (conv weights, folded BN scale/shift, head packs) to be rebuilt before its next use."""
    _EPOCH[0] += 1


def bump_versions(*tensors):
    """Make a raw-pointer write (HIP kernel) to ``tensors`` visible to the ``(data_ptr, _version)`` stamps - no launch."""
    torch.autograd.graph.increment_version(tensors)

_DTYPES = ("f32", "bf16", "f16")
_TORCH_HALF = {"bf16": torch.bfloat16, "f16": torch.float16}


def _resolve(idx, current):
    """darknet layer reference -> absolute module index (negative = relative to ``current``)."""
    idx = int(idx)
    return current + idx if idx < 0 else idx


def pick_tap_module(module_defs):
    """Index of the module whose output is ``Darknet.featuremap``.

    Reference rule (models.py:254-255): the ``nn.Sequential`` whose first child is named
    ``conv_8`` - i.e. module 8 when it is convolutional (true for the tiny cfgs).  For cfgs where
    module 8 is not a convolution (yolov3.cfg: a shortcut) the reference raises AttributeError;
    documented extension: the last 256-filter convolution before the second
    ``[yolo]`` block (yolov3.cfg module 91: 512->256 @ stride 16) - the only tensor compatible
    with ``cnn_layers_1((256, 490))`` and ``spatial_scale = 1/16`` (my_models.py:427,495).
    Returns ``None`` when no such module exists."""
    if len(module_defs) > 8 and module_defs[8]["type"] == "convolutional":
        return 8
    yolos = [i for i, d in enumerate(module_defs) if d["type"] == "yolo"]
    if len(yolos) < 2:
        return None
    tap = None
    for i in range(yolos[0] + 1, yolos[1]):
        d = module_defs[i]
        if d["type"] == "convolutional" and int(d["filters"]) == 256:
            tap = i
    return tap


class ConvWeights:
    """Packed device copies of one conv block's parameters (weights OHWI, folded scale/shift).

    The ``nn.Parameter``s of the module tree stay the source of truth (state_dict / optimizer /
    checkpoint compatibility); this object re-packs them in place
    whenever their version counters or storage change, so descriptor pointers stay valid."""

    def __init__(self, conv, bn=None, dtype="f32", cin_pad=0, cout_pad=0):
        """``dtype="bf16"`` / ``"f16"``: weights stored as bfloat16 / IEEE half (one RNE rounding; the cin <= 4 stem keeps
        fp32 - it runs on the VALU).  ``cin_pad`` / ``cout_pad``: zero-extend the channel dimensions (bf16 MFMA kernel: cin % 32 == 0)."""
        self.conv, self.bn = conv, bn
        self.dtype, self.cin_pad, self.cout_pad = dtype, cin_pad, cout_pad
        self.wgt = self.scale = self.shift = None
        self.wgt_tiled = None  # second packing [taps][cin/32][cout][32] (16-bit 3x3 layers) / [taps][cin/16][cout][16] (fp32)
        # data-gradient weights (fp32 only, built on demand by ``want_rot``): [cin][k][k][cout] rotated by 180 degrees with the
        # channels transposed, and their tiled copy [k*k][cout/16][cin][16] - the detector's training step runs the data
        # gradient as me_conv2d_f32 on these (millieye_amd/detector_train.py)
        self.want_rot = False
        self.rot = self.rot_tiled = None
        # weights of the stride-2 layers' output-parity data gradient ([4*cin][2][2][cout]); written by the whole-network pack
        self.want_parity, self.parity, self.parity_stamp = False, None, None
        self._stamp = None

    def _sources(self):
        # (straight from the modules' dicts: nn.Module.__getattr__ is ~5x slower and this runs in every forward)
        cp = self.conv._parameters
        ts = [cp["weight"]]
        if cp.get("bias") is not None:
            ts.append(cp["bias"])
        if self.bn is not None:
            bp, bb = self.bn._parameters, self.bn._buffers
            ts += [bp["weight"], bp["bias"], bb["running_mean"], bb["running_var"]]
        return ts

    def stamp(self):
        return tuple((t.data_ptr(), t._version) for t in self._sources()) + (_EPOCH[0],)

    def _prepare_packed_f32(self, device):
        """Eligibility + the stable destination buffers of the one-launch pack: None (not fp32 parameters on ``device``),
        "realloc" (new buffers: descriptors holding the old pointers are stale) or True."""
        conv, bn = self.conv, self.bn
        w = conv.weight
        cout, cin, k, k2 = w.shape
        if k != k2:
            return None
        srcs = [w] + [t for t in ((conv.bias,) if conv.bias is not None else ())]
        if bn is not None:
            srcs += [bn.weight, bn.bias, bn.running_mean, bn.running_var]
        if not all(t.device == device and t.dtype == torch.float32 and t.is_contiguous() for t in srcs):
            return None
        f32 = dict(device=device, dtype=torch.float32)
        realloc = self.wgt is None or self.wgt.device != device or tuple(self.wgt.shape) != (cout, k, k, cin) \
            or self.wgt.dtype != torch.float32
        if realloc:
            self.wgt = torch.empty((cout, k, k, cin), **f32)
            self.scale, self.shift = torch.empty(cout, **f32), torch.empty(cout, **f32)
            self.wgt_tiled = torch.empty((k * k, cin // 16, cout, 16), **f32) if cin % 16 == 0 else None
            self.rot = self.rot_tiled = None
        if self.want_rot and self.rot is None:
            self.rot = torch.empty((cin, k, k, cout), **f32)
            self.rot_tiled = torch.empty((k * k, cout // 16, cin, 16), **f32) if cout % 16 == 0 else None
        if realloc:
            self.parity = None
        if self.want_parity and k == 3 and self.parity is None:
            self.parity = torch.empty((4 * cin, 2, 2, cout), **f32)
        return "realloc" if realloc else True

    def _pack_desc(self, d):
        """Fill one ``hip.PackDesc`` (me_pack_desc) from the module's parameters and the buffers of _prepare_packed_f32."""
        conv, bn = self.conv, self.bn
        ptr = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
        cout, cin, k, _ = conv.weight.shape
        d.w, d.bias = conv.weight.data_ptr(), ptr(conv.bias)
        d.gamma, d.beta = ptr(bn.weight if bn is not None else None), ptr(bn.bias if bn is not None else None)
        d.mean, d.var = ptr(bn.running_mean if bn is not None else None), ptr(bn.running_var if bn is not None else None)
        d.ohwi, d.tiled = self.wgt.data_ptr(), ptr(self.wgt_tiled)
        d.rot, d.rot_tiled = ptr(self.rot if self.want_rot else None), ptr(self.rot_tiled if self.want_rot else None)
        d.scale, d.shift = self.scale.data_ptr(), self.shift.data_ptr()
        d.parity = ptr(self.parity if self.want_parity else None)
        d.cout, d.cin, d.ksize, d.eps = cout, cin, k, float(bn.eps) if bn is not None else 0.0
        return d

    def _refresh_packed_f32(self, device):
        """fp32, parameters already on ``device``: ONE launch of ``me_pack_conv_f32`` writes every packed copy into the
        stable buffers (a training step re-packs every layer: through torch that was ~25 launches per layer)."""
        done = self._prepare_packed_f32(device)
        if done is None:
            return None
        self.parity_stamp = None  # (the per-layer entry point has no parity output)
        d = self._pack_desc(hip.PackDesc())
        hip.check(hip.lib().me_pack_conv_f32(d.w, d.cout, d.cin, d.ksize, d.bias, d.gamma, d.beta, d.mean, d.var, d.eps, d.ohwi,
                                             d.tiled, d.rot, d.rot_tiled, d.scale, d.shift, hip.stream_ptr()),
                  "me_pack_conv_f32")
        return done

    def refresh(self, device):
        stamp = self.stamp() + (self.want_rot,)
        if stamp == self._stamp and self.wgt is not None and self.wgt.device == device:
            return False
        dev = torch.device(device)
        if self.dtype == "f32" and not self.cin_pad and not self.cout_pad and dev.type == "cuda":
            if dev.index is None:
                dev = torch.device("cuda", torch.cuda.current_device())
            done = self._refresh_packed_f32(dev)
            if done is not None:
                self._stamp = stamp
                return done
        with torch.no_grad():
            w = self.conv.weight.detach().to(device=device, dtype=torch.float32)
            packed = w.permute(0, 2, 3, 1).contiguous()
            cout = w.shape[0]
            if self.bn is not None:
                g = self.bn.weight.detach().to(device, torch.float64)
                b = self.bn.bias.detach().to(device, torch.float64)
                mean = self.bn.running_mean.detach().to(device, torch.float64)
                var = self.bn.running_var.detach().to(device, torch.float64)
                scale = g / torch.sqrt(var + self.bn.eps)
                shift = b - mean * scale
                if self.conv.bias is not None:
                    shift = shift + self.conv.bias.detach().to(device, torch.float64) * scale
            else:
                scale = torch.ones(cout, dtype=torch.float64, device=device)
                if self.conv.bias is not None:
                    shift = self.conv.bias.detach().to(device, torch.float64)
                else:
                    shift = torch.zeros(cout, dtype=torch.float64, device=device)
            scale, shift = scale.to(torch.float32), shift.to(torch.float32)
            cin = packed.shape[3]
            if self.cin_pad > cin:
                packed = torch.nn.functional.pad(packed, (0, self.cin_pad - cin))
            if self.cout_pad > cout:  # padded output channels: weight 0, scale 0, shift 0 -> exactly 0 after any activation
                extra = self.cout_pad - cout
                packed = torch.nn.functional.pad(packed, (0, 0, 0, 0, 0, 0, 0, extra))
                scale = torch.nn.functional.pad(scale, (0, extra))
                shift = torch.nn.functional.pad(shift, (0, extra))
            if self.dtype in _TORCH_HALF and cin > 4:
                packed = packed.to(_TORCH_HALF[self.dtype])
            elif self.dtype in _TORCH_HALF:  # stem: fp32 container (the VALU fallback reads floats), 16-bit values
                packed = packed.to(_TORCH_HALF[self.dtype]).to(torch.float32)
            packed = packed.contiguous()
            realloc = (self.wgt is None or self.wgt.device != device or self.wgt.shape != packed.shape
                       or self.wgt.dtype != packed.dtype)
            tiled = None
            if packed.dtype in hip.HALF_TYPES and packed.shape[1] in (1, 3) and packed.shape[3] % 32 == 0:
                # (3x3: the patch-resident kernels; 1x1: the first half of a one-launch bottleneck, csrc/bneck_h16.hip)
                tiled = hip.tile_weights_h16(packed)
            elif packed.dtype == torch.float32 and self.dtype == "f32" and packed.shape[3] % 16 == 0:
                # the fp32 MFMA kernels stream their weight tiles from this copy (a K stage of a tile is contiguous - csrc/conv.hip,
                # conv_igemm_buf_f32); the patch-resident 3x3 kernels (csrc/conv_p8_f32.hip) need it
                tiled = hip.tile_weights_f32(packed)
            elif self.dtype in _TORCH_HALF and cin == 3 and packed.shape[1] == 3 and packed.shape[0] == 32:
                # MFMA stem (csrc/stem_mfma_h16.hip): [32 cout][32 taps] in the storage type, taps 27..31 zero
                tiled = torch.nn.functional.pad(packed.reshape(32, 27), (0, 5)).to(_TORCH_HALF[self.dtype]).contiguous()
            if realloc:
                self.wgt, self.scale, self.shift = packed, scale.contiguous(), shift.contiguous()
                self.wgt_tiled = tiled
            else:  # keep the pointers the plans hold
                self.wgt.copy_(packed)
                self.scale.copy_(scale)
                self.shift.copy_(shift)
                if tiled is not None:
                    self.wgt_tiled.copy_(tiled)
        self._stamp = stamp
        return "realloc" if realloc else True


class _Tensor:
    __slots__ = ("h", "w", "c", "parent", "chan_off", "producers", "readers", "offset", "pinned", "external", "esize",
                 "padded")

    def __init__(self, h, w, c, esize=4):
        self.h, self.w, self.c = h, w, c
        self.esize = esize    # bytes per element (4 = float32, 2 = bfloat16)
        self.padded = 0       # zero channels appended behind the logical ones (bf16 mode, tiny cfgs' 16-channel stem)
        self.parent = None
        self.chan_off = 0
        self.producers = []
        self.readers = []
        self.offset = None
        self.pinned = False
        self.external = False  # the network input (NCHW, caller owned)

    def root(self):
        t, off = self, 0
        while t.parent is not None:
            off += t.chan_off
            t = t.parent
        return t, off


class _Plan:
    pass


class DarknetEngine:
    """Compiled execution of a :class:`millieye_amd.yolov3.models.Darknet` module tree."""

    def __init__(self, model, dtype="f32"):
        """``dtype``: storage of the activations and MFMA-conv weights between layers - ``"f32"`` (default; the mode the
        1e-3 parity bar is quoted on) or ``"bf16"`` / ``"f16"`` (BASELINE configs[2]/[4]: 16-bit operands, fp32 accumulation and
        epilogue, fp32 detection maps into the YOLO decode; the engine plans are inference - the 16-bit training step is
        millieye_amd/detector_train16.py)."""
        if dtype not in _DTYPES:
            raise ValueError(f"unknown engine dtype {dtype!r} (expected one of {_DTYPES})")
        self.model = model
        self.dtype = dtype
        self._plans = {}
        self._weights = {}
        self.tap_module = pick_tap_module(model.module_defs)

    # ---------------------------------------------------------------------------------- weights
    def _conv_weights(self, i, cin_pad=0, cout_pad=0):
        cw = self._weights.get(i)
        if cw is None:
            seq = self.model.module_list[i]
            conv = seq[0]
            bn = seq[1] if isinstance(seq[1] if len(seq) > 1 else None, torch.nn.BatchNorm2d) else None
            cw = ConvWeights(conv, bn, self.dtype, cin_pad, cout_pad)
            self._weights[i] = cw
        elif (cin_pad, cout_pad) != (0, 0) and (cw.cin_pad, cw.cout_pad) != (cin_pad, cout_pad):
            cw.cin_pad, cw.cout_pad = cin_pad, cout_pad
            cw._stamp = None
        return cw

    def refresh_train_weights(self, device):
        """Training step (millieye_amd/detector_train.py): every conv block's parameters changed, so all of them are packed
        by ONE ``me_pack_conv_batch_f32`` launch from a descriptor table kept in device memory (parameters and packed buffers
        of a module tree are stable, so the table is rebuilt only when a pointer moves).  The rotated copies for the data
        gradient ride along, except for layer 0 (no data gradient) and the 3x3 / stride-2 layers (their data gradient is the
        output-parity convolution built from the forward layout).  Falls back to the per-layer launches when a block is not
        plain fp32 on ``device``."""
        import ctypes as C
        dev = torch.device(device)
        if dev.type == "cuda" and dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        cws = []
        for i, d in enumerate(self.model.module_defs):
            if d["type"] != "convolutional":
                continue
            cw = self._conv_weights(i)
            s2 = int(d["size"]) == 3 and int(d["stride"]) == 2
            cw.want_rot = i > 0 and not s2
            cw.want_parity = i > 0 and s2  # (only the batch launch writes it; the data gradient builds it itself otherwise)
            cws.append(cw)
        stale = []
        for cw in cws:
            stamp = cw.stamp() + (cw.want_rot,)
            if not (stamp == cw._stamp and cw.wgt is not None and cw.wgt.device == dev):
                stale.append((cw, stamp))
        if not stale:
            return
        batch = self.dtype == "f32" and dev.type == "cuda" and len(stale) >= 2 and \
            not any(cw.cin_pad or cw.cout_pad for cw, _ in stale)
        if batch:
            states = [cw._prepare_packed_f32(dev) for cw, _ in stale]
            batch = all(st is not None for st in states)
        if not batch:
            for cw, _ in stale:
                if cw.refresh(dev) == "realloc" and self._plans:
                    self._plans.clear()
            return
        if "realloc" in states and self._plans:
            self._plans.clear()
        descs = (hip.PackDesc * len(stale))()
        for d, (cw, _) in zip(descs, stale):
            cw._pack_desc(d)
        key = bytes(descs)
        cached = self.__dict__.get("_pack_table")
        if cached is None or cached[0] != key:
            total = int(hip.lib().me_pack_conv_plan(descs, len(stale)))
            if total <= 0:
                raise hip.MeError("me_pack_conv_plan: " + hip.lib().me_last_error().decode("utf-8", "replace"))
            table = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(dev)
            cached = self._pack_table = (key, table, total, max(int(d.ksize) for d in descs))
        _, table, total, max_k = cached
        hip.check(hip.lib().me_pack_conv_batch_f32(table.data_ptr(), len(stale), total, max_k, hip.stream_ptr()),
                  "me_pack_conv_batch_f32")
        for cw, stamp in stale:
            cw._stamp = stamp
            cw.parity_stamp = stamp if cw.parity is not None else None

    def refresh_weights(self, device):
        """Re-pack whatever parameter changed since the last run.  The fast path is one flat tuple of
        ``(data_ptr, _version)`` over the ~370 source tensors, read straight from the modules' ``_parameters`` /
        ``_buffers`` dicts (``nn.Module.__getattr__`` is 5x slower and this runs in front of every forward)."""
        fast = self.__dict__.get("_fast_sources")
        if fast is None:
            slots, bns = [], []
            for i, d in enumerate(self.model.module_defs):
                if d["type"] != "convolutional":
                    continue
                seq = self.model.module_list[i]
                conv = seq[0]
                slots.append((conv._parameters, "weight"))
                if conv._parameters.get("bias") is not None:
                    slots.append((conv._parameters, "bias"))
                if len(seq) > 1 and isinstance(seq[1], torch.nn.BatchNorm2d):
                    bn = seq[1]
                    bns.append(bn)
                    slots += [(bn._parameters, "weight"), (bn._parameters, "bias"), (bn._buffers, "running_mean"),
                              (bn._buffers, "running_var")]
            fast = self._fast_sources = (slots, bns)
            self._fast_stamp = None
        slots, bns = fast
        for bn in bns:
            if bn.training:
                raise NotImplementedError(
                    "Darknet BatchNorm in training mode (batch statistics) is not on the accelerated "
                    "inference path; call model.eval() (the reference keeps base_detector.eval(), "
                    "module3_our_dataset/train.py:170)")
        stamp = tuple([(t.data_ptr(), t._version) for t in [dct[key] for dct, key in slots]]) + (str(device), _EPOCH[0])
        if stamp == self._fast_stamp:
            return False
        for i, d in enumerate(self.model.module_defs):
            if d["type"] == "convolutional":
                if self._conv_weights(i).refresh(device) == "realloc" and self._plans:
                    self._plans.clear()  # descriptors hold the old pointers
        self._fast_stamp = stamp
        return True

    def weight_stamp(self, device):
        """``(data_ptr, _version)`` of every tensor the packed weights are built from, read through the modules' LIVE parameter /
        buffer dicts (a re-assigned ``nn.Parameter`` is seen), + the device and the invalidation epoch: the identity of the detector's
        weights, one pass over ~370 dict entries (what ``refresh_weights`` compares in front of every forward)."""
        fast = self.__dict__.get("_fast_sources")
        if fast is None:
            self.refresh_weights(device)
            fast = self._fast_sources
        return tuple([(t.data_ptr(), t._version) for t in [dct[key] for dct, key in fast[0]]]) + (str(device), _EPOCH[0])

    # ---------------------------------------------------------------------------------- planning
    def _build(self, n, h, w, device, keep_raw=False):
        defs = self.model.module_defs
        L = len(defs)
        hyper_c = int(self.model.hyperparams["channels"])

        # who reads layer k?
        readers = [[] for _ in range(L)]
        srcs = [None] * L
        for i, d in enumerate(defs):
            t = d["type"]
            if t in ("convolutional", "upsample", "maxpool", "yolo"):
                srcs[i] = [i - 1]
            elif t == "route":
                srcs[i] = [_resolve(x, i) for x in d["layers"].split(",")]
            elif t == "shortcut":
                srcs[i] = [i - 1, _resolve(d["from"], i)]
            else:
                raise ValueError(f"unsupported cfg block [{t}] at module {i}")
            for s in srcs[i]:
                if s >= 0:
                    readers[s].append(i)
        tap = self.tap_module

        tensors = []
        bf16 = self.dtype in _TORCH_HALF  # any 16-bit storage mode
        half_type = hip.HALF_TYPES[_TORCH_HALF[self.dtype]] if bf16 else 0
        if bf16 and keep_raw:
            raise NotImplementedError("a 16-bit ENGINE PLAN keeps no raw maps (the loss value of an evaluation call comes from the fp32 engine; "
                                      "training in a 16-bit storage mode is millieye_amd/detector_train16.py, not an engine plan)")
        act_esize = 2 if bf16 else 4

        def new_tensor(hh, ww, cc, esize=None):
            t = _Tensor(hh, ww, cc, act_esize if esize is None else esize)
            tensors.append(t)
            return t

        def feeds_yolo_only(idx):
            return bool(readers[idx]) and all(defs[r]["type"] == "yolo" for r in readers[idx])

        t_in = new_tensor(h, w, hyper_c, 4)
        t_in.external = True
        out = [None] * L  # layer index -> _Tensor
        ops = []  # dicts
        fused_away = set()
        yolo_rows = []
        i = 0
        while i < L:
            d = defs[i]
            t = d["type"]
            if t == "convolutional":
                x = t_in if i == 0 else out[i - 1]
                if x is None:
                    raise RuntimeError(f"module {i} reads a fused-away tensor")
                k, s = int(d["size"]), int(d["stride"])
                pad = (k - 1) // 2
                cout = int(d["filters"])
                ho = (x.h + 2 * pad - k) // s + 1
                wo = (x.w + 2 * pad - k) // s + 1
                act = hip.ACT_LEAKY if d["activation"] == "leaky" else hip.ACT_LINEAR
                op = dict(kind="conv", module=i, x=x, res=None, k=k, s=s, pad=pad, act=act, ups=1, ho=ho, wo=wo,
                          cout=cout)
                y_esize, y_c = act_esize, cout
                if bf16:
                    if x.esize != (4 if x.external else 2):
                        raise RuntimeError(f"module {i}: a convolution reads an fp32 detection map in bf16 mode")
                    if feeds_yolo_only(i):
                        y_esize = 4   # raw detection maps stay fp32 for the YOLO decode
                    elif cout % 32:
                        y_c = -(-cout // 32) * 32  # the next MFMA conv needs cin % 32 == 0: zero channels behind the real ones
                nxt = defs[i + 1] if i + 1 < L else None
                only_next = readers[i] == [i + 1] and i != tap
                if (nxt is not None and nxt["type"] == "shortcut" and only_next and x.c > 4
                        and srcs[i + 1][0] == i and out[srcs[i + 1][1]] is not None
                        and srcs[i + 1][1] != i):
                    res = out[srcs[i + 1][1]]
                    if (res.h, res.w, res.c) == (ho, wo, cout) and y_c == cout and y_esize == res.esize:
                        y = new_tensor(ho, wo, cout, y_esize)
                        op["res"] = res
                        op["y"] = y
                        op["covers"] = (i, i + 1)
                        out[i + 1] = y
                        fused_away.add(i)
                        ops.append(op)
                        i += 2
                        continue
                if (nxt is not None and nxt["type"] == "upsample" and int(nxt["stride"]) == 2 and only_next
                        and x.c > 4 and y_c == cout):
                    y = new_tensor(ho * 2, wo * 2, cout, y_esize)
                    op["ups"] = 2
                    op["y"] = y
                    op["covers"] = (i, i + 1)
                    out[i + 1] = y
                    fused_away.add(i)
                    ops.append(op)
                    i += 2
                    continue
                y = new_tensor(ho, wo, y_c, y_esize)
                y.padded = y_c - cout
                op["y"] = y
                op["covers"] = (i,)
                out[i] = y
                ops.append(op)
            elif t == "maxpool":
                x = out[i - 1]
                k, s = int(d["size"]), int(d["stride"])
                zero_ext = (k == 2 and s == 1)
                pad = (k - 1) // 2
                ext = 1 if zero_ext else 0
                ho = (x.h + ext + 2 * pad - k) // s + 1
                wo = (x.w + ext + 2 * pad - k) // s + 1
                y = new_tensor(ho, wo, x.c, x.esize)
                y.padded = x.padded
                ops.append(dict(kind="pool", module=i, x=x, y=y, k=k, s=s, pad=pad, zero_ext=ext, ho=ho, wo=wo))
                out[i] = y
            elif t == "upsample":
                x = out[i - 1]
                f = int(d["stride"])
                y = new_tensor(x.h * f, x.w * f, x.c, x.esize)
                y.padded = x.padded
                ops.append(dict(kind="upsample", module=i, x=x, y=y, f=f))
                out[i] = y
            elif t == "shortcut":
                a, b = out[srcs[i][0]], out[srcs[i][1]]
                if a.esize != b.esize or a.padded or b.padded:
                    raise NotImplementedError(f"shortcut {i}: mixed storage types / padded channels")
                y = new_tensor(a.h, a.w, a.c, a.esize)
                ops.append(dict(kind="add", module=i, a=a, b=b, y=y))
                out[i] = y
            elif t == "route":
                parts = [out[s] for s in srcs[i]]
                if any(p is None for p in parts):
                    raise RuntimeError(f"route {i} reads a fused-away tensor")
                if len(parts) == 1:
                    out[i] = parts[0]
                else:
                    if any(p.padded or p.esize != parts[0].esize for p in parts):
                        raise NotImplementedError(f"route {i}: mixed storage types / padded channels")
                    cat = new_tensor(parts[0].h, parts[0].w, sum(p.c for p in parts), parts[0].esize)
                    off = 0
                    for p in parts:
                        if (p.h, p.w) != (cat.h, cat.w):
                            raise ValueError(f"route {i}: spatial size mismatch")
                        if p.parent is None and not p.external and p is not cat and not _in_family(cat, p):
                            p.parent, p.chan_off = cat, off
                        else:  # already part of another concat: materialise a copy
                            piece = new_tensor(p.h, p.w, p.c, p.esize)
                            piece.parent, piece.chan_off = cat, off
                            ops.append(dict(kind="copy", module=i, x=p, y=piece))
                        off += p.c
                    out[i] = cat
            elif t == "yolo":
                x = out[i - 1]
                if x.h != x.w or h != w:
                    raise ValueError("YOLO decode needs square inputs (the reference uses one grid_size)")
                yl = self.model.module_list[i][0]
                na, nc = yl.num_anchors, yl.num_classes
                if x.c != na * (nc + 5):
                    raise ValueError(f"yolo {i}: {x.c} channels != {na}*({nc}+5)")
                if x.esize != 4:
                    raise RuntimeError(f"yolo {i}: the detection map is shared with another reader (bf16 mode)")
                ops.append(dict(kind="yolo", module=i, x=x, layer=yl, g=x.h, row_offset=sum(yolo_rows)))
                yolo_rows.append(na * x.h * x.h)
                out[i] = None  # decoded rows are never routed
            i += 1

        # the [yolo] decodes go behind the last convolution: one launch for all scales (me_yolo_decode_cand_multi_f32) instead of
        # three small ones in the middle of the dependent chain; the detection maps stay live until then (liveness below)
        if os.environ.get("MILLIEYE_DECODE_LAST", "1") != "0":
            ops = [op for op in ops if op["kind"] != "yolo"] + [op for op in ops if op["kind"] == "yolo"]

        # liveness (op index granularity)
        for oi, op in enumerate(ops):
            for key in ("x", "res", "a", "b"):
                tt = op.get(key)
                if tt is not None:
                    tt.readers.append(oi)
            if op.get("y") is not None:
                op["y"].producers.append(oi)
        tap_tensor = out[tap] if tap is not None and tap < L else None
        if tap_tensor is not None:
            tap_tensor.pinned = True
        if keep_raw:  # the YOLO loss reads the raw detection maps after the run: exempt them from reuse
            for op in ops:
                if op["kind"] == "yolo":
                    op["x"].pinned = True

        fam = {}
        for tt in tensors:
            if tt.external:
                continue
            root, _ = tt.root()
            first, last, pin = fam.get(id(root), (10 ** 9, -1, False))
            if tt.producers:
                first = min(first, min(tt.producers))
            if tt.readers:
                last = max(last, max(tt.readers))
            pin = pin or tt.pinned
            fam[id(root)] = (first, last, pin)
        roots = []
        for tt in tensors:
            if tt.external or tt.parent is not None or id(tt) not in fam:
                continue
            first, last, pin = fam[id(tt)]
            if first == 10 ** 9:
                continue  # never produced (should not happen)
            if pin:
                last = len(ops)
            last = max(last, first)
            size = -(-(n * tt.h * tt.w * tt.c * tt.esize) // _ALIGN) * _ALIGN  # bytes
            roots.append((first, last, size, tt))
        roots.sort(key=lambda r: (r[0], -r[2]))
        placed = []  # (offset, size, first, last)
        total = 0
        for first, last, size, tt in roots:
            busy = sorted((o, s) for (o, s, f, l) in placed if not (l < first or f > last))
            off = 0
            for o, s in busy:
                if off + size <= o:
                    break
                off = max(off, o + s)
            tt.offset = off
            placed.append((off, size, first, last))
            total = max(total, off + size)

        plan = _Plan()
        plan.n, plan.h, plan.w = n, h, w
        plan.arena = torch.empty(max(total, _ALIGN), dtype=torch.uint8, device=device)
        if plan.arena.data_ptr() % _ALIGN:
            raise RuntimeError("the caching allocator returned an arena that is not 256-byte aligned")
        plan.arena_bytes = total
        plan.dtype = self.dtype
        plan.rows = sum(yolo_rows)
        plan.num_classes = None
        base = plan.arena.data_ptr()

        def view(tt):
            root, coff = tt.root()
            return base + root.offset + tt.esize * coff, root.c

        def typed_view(tt, nchw):
            """torch view of an arena tensor ([n,h,w,c], or the NCHW permutation of it) - API boundary only"""
            root, coff = tt.root()
            flat = plan.arena.view(_TORCH_HALF[self.dtype] if tt.esize == 2 else torch.float32)
            pitch, off = root.c, root.offset // tt.esize + coff
            if nchw:
                return torch.as_strided(flat, (n, tt.c, tt.h, tt.w), (tt.h * tt.w * pitch, 1, tt.w * pitch, pitch), off)
            return torch.as_strided(flat, (n, tt.h, tt.w, tt.c), (tt.h * tt.w * pitch, tt.w * pitch, pitch, 1), off)

        lib = hip.lib()
        launches = []
        plan.input_descs = []
        plan.yolo_raw = []  # raw detection maps [n,g,g,A*(5+C)] (arena views) for the YOLO loss
        plan.yolo_descs = []
        plan.conv_descs = []
        flops = 0
        for op in ops:
            kind = op["kind"]
            if kind == "conv":
                x, y = op["x"], op["y"]
                cw = self._conv_weights(op["module"], x.c if x.padded else 0, y.c if y.padded else 0)
                if cw.refresh(device) == "realloc" and self._plans:
                    self._plans.clear()
                dsc = hip.Conv16Desc() if bf16 else hip.ConvDesc()
                if x.external:
                    dsc.x, dsc.x_pitch, dsc.x_nchw = None, x.c, 1
                    plan.input_descs.append(dsc)
                else:
                    dsc.x, dsc.x_pitch = view(x)
                    dsc.x_nchw = 0
                dsc.wgt, dsc.scale, dsc.shift = cw.wgt.data_ptr(), cw.scale.data_ptr(), cw.shift.data_ptr()
                if op["res"] is not None:
                    dsc.res, dsc.res_pitch = view(op["res"])
                else:
                    dsc.res, dsc.res_pitch = None, 0
                dsc.y, dsc.y_pitch = view(y)
                dsc.n, dsc.h, dsc.w, dsc.cin = n, x.h, x.w, x.c
                dsc.cout, dsc.ksize, dsc.stride, dsc.pad = cw.wgt.shape[0], op["k"], op["s"], op["pad"]
                dsc.ho, dsc.wo, dsc.act, dsc.upsample, dsc.tile = op["ho"], op["wo"], op["act"], op["ups"], 0
                dsc.split_k, dsc.workspace, dsc.workspace_bytes = 0, None, 0
                dsc.wgt_tiled = cw.wgt_tiled.data_ptr() if cw.wgt_tiled is not None else None
                if bf16:
                    dsc.y_f32 = 1 if y.esize == 4 else 0
                    dsc.half_type = half_type
                op["launch"] = len(launches)
                op["desc"] = dsc
                launches.append((lib.me_conv2d_h16 if bf16 else lib.me_conv2d_f32, (C.byref(dsc),), dsc,
                                 f"conv{op['module']}"))
                plan.conv_descs.append((op["module"], dsc))
                flops += 2 * n * op["ho"] * op["wo"] * op["cout"] * op["k"] * op["k"] * (x.c - x.padded)
            elif kind == "pool" and bf16:
                x, y = op["x"], op["y"]
                (xp, xpitch), (yp, ypitch) = view(x), view(y)
                launches.append((lib.me_maxpool_h16,
                                 (xp, xpitch, yp, ypitch, n, x.h, x.w, x.c, op["k"], op["s"],
                                  0 if op["zero_ext"] else op["pad"], op["zero_ext"], op["ho"], op["wo"], half_type), None,
                                 f"pool{op['module']}"))
            elif kind == "pool":
                x, y = op["x"], op["y"]
                dsc = hip.PoolDesc()
                dsc.x, dsc.x_pitch = view(x)
                dsc.y, dsc.y_pitch = view(y)
                dsc.n, dsc.h, dsc.w, dsc.c = n, x.h, x.w, x.c
                dsc.size, dsc.stride, dsc.pad, dsc.zero_ext = op["k"], op["s"], (0 if op["zero_ext"] else op["pad"]), \
                    op["zero_ext"]
                dsc.ho, dsc.wo = op["ho"], op["wo"]
                launches.append((lib.me_maxpool_f32, (C.byref(dsc),), dsc, f"pool{op['module']}"))
            elif kind == "upsample":
                x, y = op["x"], op["y"]
                (xp, xpitch), (yp, ypitch) = view(x), view(y)
                launches.append((lib.me_upsample_h16 if x.esize == 2 else lib.me_upsample_f32,
                                 (xp, xpitch, yp, ypitch, n, x.h, x.w, x.c, op["f"]), None,
                                 f"upsample{op['module']}"))
            elif kind == "add":
                a, b, y = op["a"], op["b"], op["y"]
                (ap, apitch), (bp, bpitch), (yp, ypitch) = view(a), view(b), view(y)
                launches.append((lib.me_add_h16 if a.esize == 2 else lib.me_add_f32,
                                 (ap, apitch, bp, bpitch, yp, ypitch, n * a.h * a.w, a.c)
                                 + ((half_type,) if a.esize == 2 else ()), None,
                                 f"add{op['module']}"))
            elif kind == "copy":
                x, y = op["x"], op["y"]
                (xp, xpitch), (yp, ypitch) = view(x), view(y)
                launches.append((lib.me_copy_h16 if x.esize == 2 else lib.me_copy_f32,
                                 (xp, xpitch, yp, ypitch, n * x.h * x.w, x.c), None,
                                 f"copy{op['module']}"))
            elif kind == "yolo":
                x, yl = op["x"], op["layer"]
                dsc = hip.YoloDesc()
                dsc.x, dsc.x_pitch = view(x)
                dsc.out = None
                dsc.n, dsc.g, dsc.num_anchors, dsc.num_classes = n, op["g"], yl.num_anchors, yl.num_classes
                dsc.rows_total, dsc.row_offset = plan.rows, op["row_offset"]
                stride = h / op["g"]
                dsc.stride = stride
                for k, (aw, ah) in enumerate(yl.anchors):
                    dsc.anchors[2 * k] = aw / stride
                    dsc.anchors[2 * k + 1] = ah / stride
                plan.num_classes = yl.num_classes
                plan.yolo_descs.append(dsc)
                plan.yolo_raw.append(typed_view(x, nchw=False))
                launches.append((lib.me_yolo_decode_f32, (C.byref(dsc),), dsc, f"yolo{op['module']}"))
                # side effects the reference's YOLOLayer.forward has (models.py:135-156)
                yl.img_dim = h
                yl.grid_size = op["g"]
                yl.stride = stride
        k = len(plan.yolo_descs)
        plan.yolo_tail = None  # the decodes are the last launches: Network.forward's candidate-list decode takes them in one launch
        if 1 <= k <= 3 and all(name.startswith("yolo") for _f, _a, _k, name in launches[-k:]):
            plan.yolo_tail = (C.c_void_p * k)(*[C.addressof(dsc) for dsc in plan.yolo_descs])
        # shared scratch for the deterministic split-K slabs (launches are serial on one stream)
        ws_fn = lib.me_conv2d_h16_workspace_bytes if bf16 else lib.me_conv2d_workspace_bytes
        need = max([ws_fn(C.byref(d)) for _m, d in plan.conv_descs] + [0])
        # arrival counters of the in-launch split-K reduction (zero between launches; one array per plan = per stream).  Opt-in:
        # on MI355X the serial tail of the last workgroup (drain, ticket, slab reads past the L2) costs more than the launch it
        # saves - batch-1 detector 1.49 -> 1.89 ms, batch 8 4.93 -> 5.15 ms (profiles/r02_kernel_evolution.md)
        plan.tile_counters = None
        if os.environ.get("MILLIEYE_INLAUNCH_REDUCE", "0") == "1":
            plan.tile_counters = torch.zeros(hip.TILE_COUNTERS, dtype=torch.int32, device=device)
            for _m, d in plan.conv_descs:
                if d.cin > 4:
                    d.tile_counters, d.tile_counters_len = plan.tile_counters.data_ptr(), hip.TILE_COUNTERS
        plan.conv_ws = None
        plan.graph = None
        if need > 0:
            plan.conv_ws = torch.empty(need + 256, dtype=torch.uint8, device=device)
            ws_ptr = plan.conv_ws.data_ptr() + (-plan.conv_ws.data_ptr()) % 256
            for _m, d in plan.conv_descs:
                if ws_fn(C.byref(d)) > 0:
                    d.workspace, d.workspace_bytes = ws_ptr, need
        plan.launches = launches
        plan.conv_flops = flops
        if _autotune_enabled():
            _autotune(plan, lib)
        plan.fused_blocks = []
        if bf16 and _bneck_mode() != "0" and (_autotune_enabled() or _bneck_mode() == "force"):
            _fuse_bottlenecks(plan, ops, lib)   # (a measured choice, like the tiles: not without the autotuner - pinned plans stay pinned)
        if tap_tensor is not None:
            # NCHW view of the feature tap in its storage type (bf16 mode: callers that need fp32 convert at the API
            # boundary - Darknet.forward; Network.forward hands the bf16 tap straight to the score-map conv)
            plan.tap = typed_view(tap_tensor, nchw=True)
            plan.tap_ptr, plan.tap_pitch = view(tap_tensor)
            plan.tap_shape = (tap_tensor.h, tap_tensor.w, tap_tensor.c)
        else:
            plan.tap = None
        return plan

    def plan_for(self, x, keep_raw=False):
        n, _, h, w = x.shape
        key = (n, h, w, x.device.index, bool(keep_raw))
        plan = self._plans.get(key)
        if plan is None:
            plan = self._build(n, h, w, x.device, keep_raw)
            self._plans[key] = plan
            if len(self._plans) > 8:  # multiscale callers: keep the arena count bounded
                self._plans.pop(next(iter(self._plans)))
        return plan

    # ---------------------------------------------------------------------------------- execution
    def run(self, x, keep_raw=False, nms_conf=None):
        """x: CUDA fp32 NCHW [N,C,H,W].  Returns (plan, yolo_outputs [N,R,5+C]); the feature tap is
        ``plan.tap`` (a view into the plan's arena, valid until the next ``run`` of that plan).
        ``nms_conf``: the caller will run NMS on the rows at this confidence threshold next (Network.forward): every [yolo]
        decode then also appends its passing rows to the candidate lists of the shared NMS workspace
        (``me_yolo_decode_cand_f32``) and ``plan.nms_prepped`` is set to the threshold - ``hip.nms_batched(..., prepped=True)``
        skips its own pass over the 5 + C columns of every row."""
        if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4):
            raise hip.MeError("Darknet input must be a 4-D CUDA float32 tensor [N,C,H,W]; this path has no CPU "
                              "fallback (the CPU restatement is oracle/, test infrastructure only)")
        x = x.contiguous()
        # The parameter check (one pass over ~370 tensors, ~50 us) normally finds nothing, and whatever the host does in front of
        # the first launch is idle time of the GPU: with a plan in hand the launches go out FIRST on the packed weights of the
        # last run, then the check runs behind them; if a parameter did change, the weights are re-packed and the forward is
        # issued again on the same stream (same buffers: the stale results are overwritten before anyone can read them).
        n_, _, h_, w_ = x.shape
        plan = self._plans.get((n_, h_, w_, x.device.index, bool(keep_raw))) \
            if self.__dict__.get("_fast_stamp") is not None else None
        # ... except where a second issue would be wrong or wasteful: under a caller's stream capture (the stale
        # pass and the re-pack would both be captured) and with autograd recording (a forward that follows an optimizer step -
        # the evaluation inside a training loop - would pay the launches twice); MILLIEYE_CHECK_FIRST=1 turns it off for good.
        early = plan is not None and _SPECULATIVE and not _graphs_enabled() and not torch.is_grad_enabled() \
            and not torch.cuda.is_current_stream_capturing()
        if not early:
            self.refresh_weights(x.device)
            plan = self.plan_for(x, keep_raw)
            if _graphs_enabled() and not torch.cuda.is_current_stream_capturing():
                return self._run_graph(plan, x, self._nms_candidates(plan, x, nms_conf))
        while True:
            yolo_out = torch.empty((plan.n, plan.rows, 5 + (plan.num_classes or 0)), dtype=torch.float32,
                                   device=x.device)
            cand = self._nms_candidates(plan, x, nms_conf)
            plan.nms_prepped = cand[0] if cand is not None else None
            self._launch_all(plan, x, yolo_out, cand)
            plan.last_input = x  # keep the caller's tensor alive until the stream has consumed it
            if not early or not self.refresh_weights(x.device):
                return plan, yolo_out
            early = False  # a parameter changed since the last run: again, on the fresh weights (and a fresh plan if they moved)
            plan = self.plan_for(x, keep_raw)

    @staticmethod
    def _nms_candidates(plan, x, nms_conf):
        """``(threshold, workspace pointer)`` when the [yolo] decodes of this run are to fill the NMS candidate lists (the caller's
        stream's workspace: ``hip.nms_batched`` on the same stream finds them there), else None."""
        if nms_conf is None or not plan.yolo_descs or 5 + (plan.num_classes or 0) > 128 or plan.rows > 32768 or plan.n > 65535:
            return None
        ws_ptr, _keep = hip.nms_workspace(plan.n, plan.rows, x.device)
        return (float(nms_conf), ws_ptr)

    @staticmethod
    def _launch_all(plan, x, yolo_out, cand=None):
        xp = x.data_ptr()
        for dsc in plan.input_descs:
            dsc.x = xp
        yp = yolo_out.data_ptr()
        for dsc in plan.yolo_descs:
            dsc.out = yp
        stream = hip.stream_ptr()
        first = 1
        decode_cand = hip.lib().me_yolo_decode_cand_f32 if cand is not None else None
        for fn, args, _keep, name in plan.launches:
            if decode_cand is not None and name.startswith("yolo"):
                if plan.yolo_tail is not None:
                    if first:
                        rc = hip.lib().me_yolo_decode_cand_multi_f32(plan.yolo_tail, len(plan.yolo_tail), cand[0], cand[1], 1,
                                                                     stream)
                    else:
                        continue
                else:
                    rc = decode_cand(args[0], cand[0], cand[1], first, stream)
                first = 0
            else:
                rc = fn(*args, stream)
            if rc != 0:
                hip.check(rc, name)

    def _run_graph(self, plan, x, cand=None):
        """hipGraph replay of the plan's launch sequence (80-110 kernels): the first two runs of a plan are eager (lazy
        one-time state such as kernel attributes settles), the third is captured with ``torch.cuda.graph`` into static
        input / output buffers, later runs are one device-to-device copy of the frames + one graph launch.  Descriptors
        are passed to the kernels by value, so the capture holds everything; a plan rebuild (weight reallocation, new
        shape) drops the graph with the plan.  ``yolo_outputs`` is then a static buffer that the next run of the same
        plan overwrites (``Darknet.forward`` hands out a copy; ``Network.forward`` consumes it immediately).
        ``cand`` (``_nms_candidates``): the decodes also fill the NMS candidate lists, like the eager run with ``nms_conf``.  The
        threshold and the workspace pointer are baked into the capture, so a plan holds one graph per ``cand`` value (round 6: the
        stage-3 loop's look-ahead detector is a replay - without this it paid the separate decode + candidate pass, 96 us per
        step at batch 8); a workspace that was replaced by a bigger one gives a new key and a new capture."""
        shape = (plan.n, plan.rows, 5 + (plan.num_classes or 0))
        plan.nms_prepped = cand[0] if cand is not None else None
        rec = None
        if plan.graph is not False:   # (False: a capture failed; None: no graph run yet; else the dict {cand: record})
            if plan.graph is None:
                plan.graph = {}
            rec = plan.graph.get(cand)
            if rec is None:
                if len(plan.graph) >= 4:   # thresholds that keep changing: drop the oldest capture
                    plan.graph.pop(next(iter(plan.graph)))
                rec = plan.graph[cand] = {"eager": 0, "graph": None}
        if rec is not None and rec["graph"] is None:
            rec["eager"] += 1
            if rec["eager"] <= 2:
                yolo_out = torch.empty(shape, dtype=torch.float32, device=x.device)
                self._launch_all(plan, x, yolo_out, cand)
                plan.last_input = x
                return plan, yolo_out
            if getattr(plan, "x_static", None) is None:
                plan.x_static = torch.empty_like(x)
                plan.y_static = torch.empty(shape, dtype=torch.float32, device=x.device)
            torch.cuda.synchronize(x.device)
            graph = torch.cuda.CUDAGraph()
            try:
                # thread_local: a DataLoader pin-memory thread or the NCCL watchdog calling into the runtime during the capture
                # window must not invalidate it (the global mode would) - same choice as detector_graph.GraphedDetectorStep
                with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                    self._launch_all(plan, plan.x_static, plan.y_static, cand)
            except Exception as exc:  # a failed capture is not fatal and is not retried: this plan stays on eager launches
                import warnings
                warnings.warn(f"millieye_amd: hipGraph capture of the detector plan failed ({exc!r}); the plan runs eagerly")
                plan.graph = False
                plan.x_static = plan.y_static = None
                torch.cuda.synchronize(x.device)
            else:
                rec["graph"] = graph
        if plan.graph is False:
            yolo_out = torch.empty(shape, dtype=torch.float32, device=x.device)
            self._launch_all(plan, x, yolo_out, cand)
            plan.last_input = x
            return plan, yolo_out
        plan.x_static.copy_(x)
        rec["graph"].replay()
        return plan, plan.y_static


import threading

_GRAPH_SCOPE = threading.local()   # (per thread: a `with graph_replay()` block of one thread must not switch another thread's engine runs)


class graph_replay:
    """``with engine.graph_replay():`` - engine runs inside the block replay their plan's captured hipGraph (``_run_graph``).  For
    callers whose HOST time is what counts: the stage-3 training loop issues the next batch's frozen detector while its own tail is
    host-bound (train_path._issue_prefetch) - one copy + one graph launch instead of ~130 launches (0.4 ms of issue time)."""

    def __enter__(self):
        _GRAPH_SCOPE.depth = getattr(_GRAPH_SCOPE, "depth", 0) + 1
        return self

    def __exit__(self, *exc):
        _GRAPH_SCOPE.depth = getattr(_GRAPH_SCOPE, "depth", 1) - 1
        return False


def _graphs_enabled():
    """Opt-in (``MILLIEYE_HIPGRAPH=1``, or inside ``graph_replay()``).  Measured on MI355X (Darknet-53 @416): 655 vs 661 frames/s at
    batch 1, 1508 vs 1514 at batch 8 with / without the graph - the GPU side of the path is bound by the kernels' own latency, not by
    launches, so the eager sequence stays the default."""
    import os
    return getattr(_GRAPH_SCOPE, "depth", 0) > 0 or os.environ.get("MILLIEYE_HIPGRAPH", "0") in ("1", "true", "on")


# ------------------------------------------------------------------------------------------ autotuner
_TUNE_CACHE = {}
_TUNE_FILE_LOADED = [False]
_TUNE_STATS = {"measured": 0, "synced": 0}  # layer shapes whose candidates were timed in this process (0 = every choice came from the file)
_TUNE_TILES = (1, 2, 3, 4, 5)  # (7 / 47 = 64x64 with a 6 / 9-stage LDS ring: no gain at batch 1 / 8 - a lone wave per SIMD is bound by its
# own MFMA chain, not by DMA latency - so they stay forced-only ids)
_TUNE_SPLITS = (1, 2, 3, 4, 6, 8)
_KW_MAX_POSITIONS = 6144   # tiles 40 / 41 are offered to layers with at most this many output positions (batch 1 up to 52 x 52)
_TUNE_TILES_BF16 = (1, 2, 3, 4, 11, 12, 13, 14, 15)  # 1x = one 32-channel sub-stage per pipeline stage (more workgroups / CU)
_TILE_SHAPES_BF16 = {40: (32, 32), 41: (32, 64),   # small-batch tiles: the K split over the waves of one workgroup (csrc/conv_kw_h16.hip)
                     1: (128, 128), 2: (128, 64), 3: (64, 64), 4: (256, 128),
                     11: (128, 128), 12: (128, 64), 13: (64, 64), 14: (256, 128), 15: (192, 128),
                     # patch-resident big tiles (csrc/conv_p8_h16.hip; 3x3 stride 1 only, no split-K): 1xx one workgroup per
                     # CU with register-pipelined fragments, 2xx two workgroups per CU
                     100: (128, 256), 110: (192, 256), 120: (256, 256), 101: (128, 128), 121: (256, 128), 131: (384, 128),
                     141: (512, 128), 200: (128, 256), 201: (128, 128), 221: (256, 128), 301: (128, 128), 311: (192, 128),
                     321: (256, 128), 331: (256, 128), 421: (256, 128), 431: (384, 128), 441: (512, 128),
                     # round 4: 6xx / 7xx = 221 / 421 / 431 with the DMA duty split (weights by waves 0-3, patch by waves 4-7);
                     # 8xx = ping-pong halves (one workgroup per CU, its halves half a stage apart)
                     621: (256, 128), 721: (256, 128), 731: (384, 128),
                     810: (192, 256), 820: (256, 256), 821: (256, 128), 831: (384, 128), 841: (512, 128),
                     1210: (192, 256), 1221: (256, 128), 1231: (384, 128),  # ping-pong + four loader waves
                     # weight-stationary streaming 1x1 (csrc/conv1x1_ws_h16.hip): persistent grid, 32-row tiles, no split-K
                     50: (32, 256),
                     # weight-stationary 3x3 for cin 32 / 64 (csrc/conv3x3_ws_h16.hip): 2-D tiles, persistent grid, no split-K
                     60: (256, 128)}
_TUNE_TILES_TAIL = (41, 42, 43, 44, 45)  # fp32: tiles 1-5 with the last partial round of tiles cut split_k ways along K
_TUNE_TILES_P8_F32 = (201, 221)  # fp32 is matrix-pipe bound: the big tiles' pad positions / quantisation cost more than
# their traffic saves (tools/p8_bench_f32.py: only the 2-workgroup tiles come close to the 64x64 per-tap tile)
_TUNE_TILES_P8 = (100, 110, 120, 101, 121, 131, 141, 200, 201, 221, 311, 321, 421, 431, 441,   # 4xx: a barrier per three taps
                  621, 721, 731, 810, 831, 1210, 1231)  # round 4: duty split, ping-pong, ping-pong + loader waves


_SPECULATIVE = os.environ.get("MILLIEYE_CHECK_FIRST", "0") != "1"  # (=1: parameter check in front of the launches, as before)


def _autotune_enabled():
    import os
    return os.environ.get("MILLIEYE_AUTOTUNE", "1") not in ("0", "false", "off")


def _tune_file():
    import os
    path = os.environ.get("MILLIEYE_TUNE_CACHE")
    if path:
        return path
    base = os.environ.get("XDG_CACHE_HOME", os.path.join(os.path.expanduser("~"), ".cache"))
    return os.path.join(base, "millieye_amd", "conv_tune_v15.json")  # bump with every kernel generation


def _tune_load():
    import json
    import os
    if _TUNE_FILE_LOADED[0]:
        return
    _TUNE_FILE_LOADED[0] = True
    try:
        with open(_tune_file()) as fh:
            for k, v in json.load(fh).items():
                _TUNE_CACHE[tuple(int(x) for x in k.split(","))] = (int(v[0]), int(v[1]))
    except (OSError, ValueError):
        pass


def _tune_save():
    import json
    import os
    path = _tune_file()
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        tmp = f"{path}.{os.getpid()}.tmp"
        with open(tmp, "w") as fh:
            json.dump({",".join(str(int(x)) for x in k): list(v) for k, v in _TUNE_CACHE.items()}, fh)
        os.replace(tmp, path)
    except OSError:
        pass  # the cache is an optimisation only


def _autotune(plan, lib):
    """Pick (tile, split_k) per MFMA conv layer by measuring the candidates on this GPU (HIP events on the
    launch stream, clocks pre-warmed).  The analytic planner inside ``me_conv2d_f32`` is within ~6 % of the
    best on average but misjudges the wave-quantisation of individual layers; measuring is cheap (about a
    second per plan, outside any timed region) and cached per layer shape for the life of the process.
    Every candidate computes the same convolution (parity tests cover all tiles and split-K)."""
    import os
    import time

    stream = hip.stream_ptr()
    _tune_load()
    bf16 = plan.dtype in _TORCH_HALF
    conv_fn = lib.me_conv2d_h16 if bf16 else lib.me_conv2d_f32
    todo, keyed = [], []
    for _m, d in plan.conv_descs:
        if d.cin <= 4:
            continue
        key = (d.n, d.h, d.w, d.cin, d.cout, d.ksize, d.stride, d.upsample, int(bool(d.res)))
        if bf16:
            key += (16 + d.y_f32 + 2 * d.half_type,)
        keyed.append((key, d))
        hit = _TUNE_CACHE.get(key)
        if hit is not None:
            d.tile, d.split_k = hit
        else:
            todo.append((key, d))

    def agree_on_rank0():
        """Data-parallel replicas run the SAME (tile, split_k) per layer: every rank measures (or finds) its own choices,
        then rank 0's table for this plan replaces them - per-rank measurement noise would otherwise give the replicas
        different accumulation orders (harmless for fp32 parity bars, visible in the 16-bit modes; VERDICT r03 robustness).
        One small object broadcast per plan build; every rank reaches it whichever way it got its choices."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            return
        # opt-in (MILLIEYE_TUNE_SYNC=1): every rank must build the same plans at the same points of the program - true for the
        # lock-step loops of bench.py, NOT for a loop in which rank 0 evaluates alone while the others wait in a barrier
        if os.environ.get("MILLIEYE_TUNE_SYNC", "0") not in ("1", "true", "on"):
            return
        box = [{k: (int(dd.tile), int(dd.split_k)) for k, dd in keyed} if dist.get_rank() == 0 else None]
        dist.broadcast_object_list(box, src=0)
        _TUNE_STATS["synced"] += 1
        for k, dd in keyed:
            hit0 = box[0].get(k)
            if hit0 is not None:
                dd.tile, dd.split_k = hit0
                _TUNE_CACHE[k] = hit0

    def ensure_workspace(nbytes):
        have = plan.conv_ws.numel() - 256 if plan.conv_ws is not None else 0
        if nbytes > have:
            plan.conv_ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=plan.arena.device)
            have = nbytes
        ptr = plan.conv_ws.data_ptr() + (-plan.conv_ws.data_ptr()) % 256 if plan.conv_ws is not None else None
        for _m2, d2 in plan.conv_descs:
            if d2.cin > 4:
                d2.workspace, d2.workspace_bytes = ptr, have
        return have

    def scratch_of(d2):
        if not bf16 and d2.tile in _TUNE_TILES_TAIL + (47,):  # tail split: compact slabs of the last partial round (the library knows)
            return int(lib.me_conv2d_workspace_bytes(C.byref(d2)))
        if bf16 and d2.tile >= 100:  # patch tiles cut along K: compact slabs of every workgroup
            return int(lib.me_conv2d_h16_workspace_bytes(C.byref(d2)))
        return d2.split_k * d2.n * d2.ho * d2.wo * d2.cout * 4 if d2.split_k > 1 else 0

    def required():
        return max([scratch_of(d2) for _m2, d2 in plan.conv_descs if d2.cin > 4] + [0])

    _TUNE_STATS["measured"] += len(todo)
    if not todo:
        agree_on_rank0()
        ensure_workspace(required())  # cached choices may need more scratch than the analytic plan asked for
        return
    ws_bytes = plan.conv_ws.numel() - 256 if plan.conv_ws is not None else 0
    # one shared scratch big enough for 8 slabs of the smaller layers (bounded: 512 MiB)
    need = 0
    for _k, d in todo:
        slab = d.n * d.ho * d.wo * d.cout * 4
        if slab * 2 <= 512 * 2 ** 20:
            need = max(need, min(8 * slab, 512 * 2 ** 20))
        need = max(need, (128 if bf16 else 64) * 2 ** 20)  # tail-split / K-split patch-tile candidates: compact slabs
    ws_bytes = ensure_workspace(max(need, required()))

    def run(d, reps):
        for _ in range(reps):
            rc = conv_fn(C.byref(d), stream)
            if rc != 0:
                return False
        return True

    def timed(d, reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        ok = run(d, reps)
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps if ok else None

    # clock ramp-up on the first layer
    t_end = time.perf_counter() + 0.3
    d0 = todo[0][1]
    d0.tile, d0.split_k = 0, 1
    while time.perf_counter() < t_end:
        run(d0, 5)
        torch.cuda.synchronize()
    for key, d in todo:
        hit = _TUNE_CACHE.get(key)
        if hit is not None:  # the same layer shape earlier in this list (Darknet-53 repeats its blocks up to eight times)
            d.tile, d.split_k = hit
            continue
        slab = d.n * d.ho * d.wo * d.cout * 4
        stages = d.ksize * d.ksize * ((d.cin + 15) // 16)
        best = (float("inf"), 0, 1)
        cands = []
        tiles = _TUNE_TILES
        if bf16:
            stages = d.ksize * d.ksize * (d.cin // 32)
            tiles = _TUNE_TILES_BF16 if d.cin % 64 == 0 else _TUNE_TILES_BF16[:4] + (15,)
            if d.ksize == 3 and d.stride == 1 and d.upsample == 1 and d.wgt_tiled and not d.y_f32:
                tiles = tiles + _TUNE_TILES_P8  # the library refuses the ones that do not apply (cout % width, LDS)
            no_ws = os.environ.get("MILLIEYE_NO_WS", "")  # A/B: "50", "60" or "50,60" keep the weight-stationary kernels out
            if d.ksize == 1 and d.stride == 1 and d.upsample == 1 and not d.y_f32 and not d.res and "50" not in no_ws:
                tiles = tiles + (50,)  # refused by the library unless one of its (cin, cout) instances fits
            if d.ksize == 3 and d.cin <= 64 and d.upsample == 1 and not d.y_f32 and "60" not in no_ws:
                tiles = tiles + (60,)
            if d.n * d.ho * d.wo <= _KW_MAX_POSITIONS and os.environ.get("MILLIEYE_NO_KW", "0") != "1":
                tiles = tiles + (40, 41)  # round 6: one launch instead of split-K slabs + a reduce launch (batch-1 layers)
        else:
            if d.cin % 16 == 0 and os.environ.get("MILLIEYE_TUNE_TAIL", "1") != "0":
                tiles = tiles + _TUNE_TILES_TAIL
            if d.ksize == 3 and d.stride == 1 and d.upsample == 1 and d.wgt_tiled:
                tiles = tiles + _TUNE_TILES_P8_F32
            # round 5: weight-stationary streaming kernels for the short-K layers (csrc/conv_ws_f32.hip; the library refuses
            # shapes without an instance).  MILLIEYE_NO_WS32 = "50", "60" or "50,60" keeps them out (A/B)
            no_ws32 = os.environ.get("MILLIEYE_NO_WS32", "")
            if d.ksize == 1 and d.stride == 1 and d.upsample == 1 and not d.res and d.act != hip.ACT_SIGMOID \
                    and d.cin in (64, 128, 256, 384, 512) and "50" not in no_ws32:
                tiles = tiles + (50,)
            if d.ksize == 3 and d.pad == 1 and d.cin in (32, 64) and d.upsample == 1 and d.act != hip.ACT_SIGMOID \
                    and "60" not in no_ws32:
                tiles = tiles + (60,)
        for tile in tiles:
            bm, bn = _TILE_SHAPES_BF16[tile] if (bf16 or tile >= 100 or tile in (50, 60)) else \
                {1: (128, 128), 2: (128, 64), 3: (64, 64), 4: (128, 32), 5: (256, 128), 7: (64, 64)}[tile % 40]
            ntiles = -(-d.n * d.ho * d.wo // bm) * -(-d.cout // bn)
            tail = not bf16 and tile in _TUNE_TILES_TAIL
            if tail and ntiles % 256 == 0:
                continue
            # (patch tiles can be cut along K - split_k > 1, compact fp32 slabs + a reduce launch - but the slabs are twice the
            #  fp32 size of the layer's output: 60 -> 76 us at 13x13, tools/p8_bench.py 32 121,121/2; not offered to the tuner)
            p8_split = False
            for split in (_TUNE_SPLITS[1:] if tail else (1,) if tile in (50, 60) or (bf16 and tile in (40, 41)) else _TUNE_SPLITS if tile < 100
                          else (1, 2, 3, 4) if p8_split else (1,)):
                d.tile, d.split_k = tile, split
                if tail or (tile >= 100 and split > 1):
                    if split > stages or scratch_of(d) > ws_bytes:
                        continue
                elif split > 1 and (split * slab > ws_bytes or ntiles * split > 4096 or split > stages):
                    continue
                if not run(d, 2):
                    continue
                ms = timed(d, 4)
                if ms is not None:
                    cands.append((ms, tile, split, split > 1 or tail))
        # second look at the front runners (the best whole-tile candidate always among them), interleaved so that clock /
        # cache state drift hits all of them alike; a candidate with a second pass (slab reduce) has to beat the best
        # whole-tile one by more than the measurement noise
        cands.sort()
        finalists = cands[:4]
        whole = [c for c in cands if not c[3]]
        if whole and whole[0] not in finalists:
            finalists.append(whole[0])
        score = {}
        for _round in range(3):
            for c in finalists:
                d.tile, d.split_k = c[1], c[2]
                ms = timed(d, 6)
                if ms is not None:
                    score.setdefault(c, []).append(ms)
        ranked = sorted((sorted(v)[len(v) // 2], c) for c, v in score.items())
        if os.environ.get("MILLIEYE_TUNE_VERBOSE"):
            import sys
            print("[tune]", key, " ".join(f"{c[1]}/{c[2]}:{1e3 * c[0]:.0f}>{1e3 * t:.0f}us" for t, c in ranked), file=sys.stderr, flush=True)
        if ranked:
            best = ranked[0]
            whole = [r for r in ranked if not r[1][3]]
            if best[1][3] and whole and best[0] > 0.98 * whole[0][0]:
                best = whole[0]
            best = (best[0], best[1][1], best[1][2])
        d.tile, d.split_k = best[1], best[2]
        _TUNE_CACHE[key] = (best[1], best[2])
    # (round 4 also tried a second opinion "in situ" - the finalists of a shape swapped into the WHOLE forward and the forward
    #  timed: it confirmed the isolated choice for all but two of 37 shapes and moved the step by < 0.1 %,
    #  profiles/r04_micro_tune_in_situ_ab.txt; not kept)
    agree_on_rank0()
    ensure_workspace(required())
    _tune_save()


def _bneck_mode():
    """MILLIEYE_BNECK: "0" never, "1" (default) where measured faster than the launch pair, "force" wherever an instance exists."""
    return os.environ.get("MILLIEYE_BNECK", "1")


_BNECK_CACHE = {}   # block shape -> bneck tile id, or 0 (the launch pair stays)
_BNECK_STATS = {"fused": 0, "kept": 0}


def _fuse_bottlenecks(plan, ops, lib):
    """16-bit plans: a ``[convolutional] 1x1`` block whose output is read by nothing but the ``3x3 / stride 1`` block right
    behind it (+ that block's fused ``[shortcut]``) can run as ONE launch that keeps the mid tensor in LDS (``me_bneck_h16``,
    csrc/bneck_h16.hip; reference yolov3/models.py:22-41, 258-260 - the residual blocks of yolov3.cfg).  Same rounding points
    as the pair; which of the two forms runs is decided by MEASUREMENT per block shape (the pair with its tuned tiles against
    every instance of the one-launch kernel, interleaved, HIP events on the launch stream), cached for the life of the process."""
    import time
    stream = hip.stream_ptr()
    launches = plan.launches
    conv_ops = [op for op in ops if op["kind"] == "conv"]
    index = {id(op): k for k, op in enumerate(ops)}
    pairs = []
    for a, b in zip(conv_ops, conv_ops[1:]):
        da, db = a["desc"], b["desc"]
        if not (index[id(b)] == index[id(a)] + 1 and b["x"] is a["y"]):
            continue
        mid = a["y"]
        if mid.readers != [index[id(b)]] or mid.pinned or mid.parent is not None or mid.padded or mid.esize != 2:
            continue
        if not (da.ksize == 1 and da.stride == 1 and da.upsample == 1 and not da.res and not da.y_f32 and not da.x_nchw
                and db.ksize == 3 and db.stride == 1 and db.pad == 1 and db.upsample == 1 and not db.y_f32
                and da.wgt_tiled and db.wgt_tiled and da.cout == db.cin and da.y_pitch == db.x_pitch):
            continue
        bd = hip.Bneck16Desc()
        bd.x, bd.x_pitch = da.x, da.x_pitch
        bd.w1_tiled, bd.scale1, bd.shift1 = da.wgt_tiled, da.scale, da.shift
        bd.w2_tiled, bd.scale2, bd.shift2 = db.wgt_tiled, db.scale, db.shift
        bd.res, bd.res_pitch, bd.y, bd.y_pitch = db.res, db.res_pitch, db.y, db.y_pitch
        bd.n, bd.h, bd.w, bd.cin, bd.cmid, bd.cout = da.n, da.h, da.w, da.cin, da.cout, db.cout
        bd.act1, bd.act2, bd.half_type = da.act, db.act, da.half_type
        tiles = []
        for tile in hip.BNECK_TILES:
            bd.tile = tile
            if lib.me_bneck_h16_supported(C.byref(bd)):
                tiles.append(tile)
        if tiles:
            pairs.append((a, b, bd, tiles))
    if not pairs:
        return

    def timed(fn, reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            if fn() != 0:
                return None
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    warmed = False
    for a, b, bd, tiles in pairs:
        da, db = a["desc"], b["desc"]
        key = (bd.n, bd.h, bd.w, bd.cin, bd.cmid, bd.cout, int(bool(bd.res)), bd.half_type, bd.act1, bd.act2)
        choice = tiles[0] if _bneck_mode() == "force" else _BNECK_CACHE.get(key)
        if choice is None:
            def pair():
                rc = lib.me_conv2d_h16(C.byref(da), stream)
                return rc if rc else lib.me_conv2d_h16(C.byref(db), stream)
            if not warmed:   # clocks up
                t_end = time.perf_counter() + 0.2
                while time.perf_counter() < t_end:
                    timed(pair, 5)
                warmed = True
            cands = {0: []}
            for tile in tiles:
                cands[tile] = []
            for _round in range(3):
                for tile in cands:
                    if tile == 0:
                        ms = timed(pair, 6)
                    else:
                        bd.tile = tile
                        ms = timed(lambda: lib.me_bneck_h16(C.byref(bd), stream), 6)
                    if ms is not None:
                        cands[tile].append(ms)
            med = {t: sorted(v)[len(v) // 2] for t, v in cands.items() if v}
            best = min(med, key=med.get)
            # the one-launch form has to beat the pair by more than the measurement noise
            choice = best if (best != 0 and med[best] < 0.98 * med.get(0, float("inf"))) else 0
            _BNECK_CACHE[key] = choice
            if os.environ.get("MILLIEYE_TUNE_VERBOSE"):
                import sys
                print("[bneck]", key, " ".join(f"{t}:{1e3 * v:.0f}us" for t, v in sorted(med.items())), "->", choice, file=sys.stderr, flush=True)
        if not choice:
            _BNECK_STATS["kept"] += 1
            continue
        _BNECK_STATS["fused"] += 1
        bd.tile = choice
        launches[a["launch"]] = (lib.me_bneck_h16, (C.byref(bd),), bd, f"bneck{a['module']}+{b['module']}")
        launches[b["launch"]] = None
        plan.fused_blocks.append((a["module"], b["module"], int(choice)))
    plan.launches = [entry for entry in launches if entry is not None]


def _in_family(cat, p):
    """True if ``cat`` is (transitively) a slice of ``p`` - guards against cyclic concat parents."""
    t = cat
    while t is not None:
        if t is p:
            return True
        t = t.parent
    return False
