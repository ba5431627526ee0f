"""milliEye fusion network (stage 3): YOLO proposals + radar proposals -> refined detections.

Host-side mirror of ``module3_our_dataset/my_models.py``: ``define_yolo``, ``init_yolo``,
``Network(base_detector, conf_thresh)`` with the reference's attributes / children
(``base_detector``, ``img_cnn_layers``, ``radar_cnn_layers``, ``refinement_head``,
``ensemble_head``) and parameter names, and
``forward(images, maps, radar_boxes_location, model_mode=0, targets=None)``.

Execution: the sub-modules are parameter containers; ``forward`` stays on the GPU from the
frames to the final ``[m, 8]`` rows (the reference bounces through the host three times,
my_models.py:457,470,520):

    Darknet engine -> me_nms_batched_f32 -> me_gather_class_boxes_f32
      -> me_conv2d_f32 (1x1 256->490 score map; radar CNN 3->32->64->128->10)
      -> me_roi_heads_f32 (PS-RoIAlign + RoIAlign + refinement head + ensemble head + box regress)
      -> one compaction / ordering step and a single host sync for the data-dependent row count.
"""
import ctypes as C
import random  # noqa: F401  (the reference's negative sampling uses python's RNG; training tail)

import numpy as np  # noqa: F401
import itertools
import os

import torch
from torch import nn

from . import engine as _engine
from . import hip
from .engine import ConvWeights
from .utils.utils import *  # noqa: F401,F403  (reference re-exports its utils through this module)
from .utils.utils import xywh2xyxy, xyxy2xywh, bbox_iou
from .yolov3.models import Darknet

__all__ = ["define_yolo", "init_yolo", "cnn_layers_1", "cnn_layers_3", "ensemble_head", "refinement_head",
           "FocalLoss", "obtain_iou_labels", "box_regress", "regression_loss", "Network"]

_DETECTIONS_PER_IMG = 200  # non_max_suppression_cpp default (utils/utils.py:337)
_NMS_THRESH = 0.5


def define_yolo(model_def):
    """cfg path -> :class:`Darknet` (reference my_models.py:13-24)."""
    return Darknet(model_def)


def init_yolo(model, weights_path):
    """Load detector weights: darknet ``.weights``, ultralytics ``.pt`` (positional copy of
    ``["model"]``), or a plain ``state_dict`` checkpoint (reference my_models.py:27-44)."""
    if weights_path.endswith(".weights"):
        model.load_darknet_weights(weights_path)
    elif weights_path.endswith(".pt"):
        param = torch.load(weights_path)["model"]
        own = model.state_dict()
        names = list(param)
        for i, name in enumerate(own):
            own[name] = param[names[i]]
        model.load_state_dict(own)
    else:
        model.load_state_dict(torch.load(weights_path))


# --------------------------------------------------------------------------------------------------
# parameter containers (names / shapes / construction order = reference, so default-init RNG
# consumption, weights_init_normal and checkpoints line up)
# --------------------------------------------------------------------------------------------------
class cnn_layers_1(nn.Module):
    """1x1 conv + BN + LeakyReLU stack producing the RoI score maps (reference :47-77)."""

    def __init__(self, channels):
        super().__init__()
        self.net = nn.Sequential()
        for i in range(len(channels) - 1):
            self.net.add_module(f"conv_{i}", nn.Conv2d(channels[i], channels[i + 1], kernel_size=(1, 1), stride=(1, 1)))
            self.net.add_module(f"batch_norm_{i}", nn.BatchNorm2d(channels[i + 1], momentum=0.1))
            self.net.add_module(f"leaky_{i}", nn.LeakyReLU(0.1))

    def forward(self, x):
        raise RuntimeError("cnn_layers_1 is executed by Network.forward through me_conv2d_f32")


class cnn_layers_3(nn.Module):
    """Radar heat-map CNN 3->32->64->128->10 + sigmoid (reference :130-157)."""

    def __init__(self):
        super().__init__()

        def block(cin, cout):
            return [nn.Conv2d(cin, cout, kernel_size=3, stride=1, padding=1), nn.BatchNorm2d(cout, momentum=0.1),
                    nn.LeakyReLU(0.1)]

        self.conv1 = nn.Sequential(*block(3, 32))
        self.conv2 = nn.Sequential(*block(32, 64))
        self.conv3 = nn.Sequential(*block(64, 128), nn.Conv2d(128, 10, kernel_size=1, stride=1))

    def forward(self, x):
        raise RuntimeError("cnn_layers_3 is executed by Network.forward through me_conv2d_f32")


class ensemble_head(nn.Module):
    """(refinement vector, yolo vector) -> 2-way softmax (reference :176-210)."""

    def __init__(self, channels, activation_softmax=True):
        super().__init__()
        self.activation_softmax = activation_softmax
        self.fc1 = nn.Sequential(nn.Linear(channels[0], channels[1]), nn.LeakyReLU(0.1))
        self.fc2 = nn.Sequential(nn.Linear(channels[2], channels[3]))
        self.softmax = nn.Softmax(dim=1)

    def forward(self, refinement_vector, yolo_vector):
        raise RuntimeError("ensemble_head is fused into me_roi_heads_f32")


class refinement_head(nn.Module):
    """RoI features -> box regression + (confidence, class) vector (reference :213-284).
    ``net3`` / ``fusion_head`` exist in the reference's state_dict but are never used."""

    def __init__(self, channels):
        super().__init__()
        self.count = 0
        tmp = 49
        self.net0 = nn.Sequential(nn.Linear(channels[0], channels[1]), nn.LeakyReLU(0.1))
        self.net1 = nn.Sequential(nn.Linear(channels[1], 4))
        self.net2 = nn.Sequential(nn.Linear(channels[1], 13), nn.Sigmoid())
        self.net3 = nn.Sequential(nn.Linear(channels[1], tmp), nn.Sigmoid())
        self.radar_net = nn.Sequential(
            nn.Conv2d(10, 10, kernel_size=7, stride=1, padding=0),
            nn.BatchNorm2d(10, momentum=0.1),
            nn.LeakyReLU(0.1),
            nn.Conv2d(10, 1, kernel_size=1, stride=1, padding=0),
            nn.Sigmoid(),
        )
        self.fusion_head = nn.Sequential(nn.Linear(2 * tmp, 1), nn.Sigmoid())

    def forward(self, radar_maps, img_maps):
        raise RuntimeError("refinement_head is fused into me_roi_heads_f32")


# --------------------------------------------------------------------------------------------------
# losses / helpers of the training tail (reference :287-408) - host-side float code
# --------------------------------------------------------------------------------------------------
class FocalLoss(nn.Module):
    """alpha-balanced focal loss on one-hot labels (reference :287-314)."""

    def __init__(self, device, alpha, gamma=2, reduction="sum"):
        super().__init__()
        self.alpha, self.gamma, self.reduction, self.device = alpha, gamma, reduction, device

    def forward(self, inputs, labels):
        alpha_pos = torch.full((labels.shape[0], 1), self.alpha).to(self.device)
        alpha_neg = torch.full((labels.shape[0], 1), 1 - self.alpha).to(self.device)
        alpha = torch.where(labels[:, 1:2] == 1, alpha_pos, alpha_neg)
        probs = (inputs * labels).sum(1).view(-1, 1)
        batch_loss = -alpha * (torch.pow((1 - probs), self.gamma)) * probs.log()
        if self.reduction == "mean":
            return batch_loss.mean()
        return batch_loss.sum()


def obtain_iou_labels(boxes, targets, multi_boxes=True):
    """Max IoU (+1 convention) of every box with the same-image, same-class targets
    (reference :317-375, incl. quirk q5; the reference's ``b.txt`` debug file is not written)."""
    image_index, pred_classes, pred_boxes = boxes[:, :1], boxes[:, 1:2], boxes[:, 2:]
    detected = []
    iou_labels = torch.zeros((len(image_index), 1))
    target_location = torch.zeros((len(image_index), 4))
    for box_i in range(len(boxes)):
        sel = (targets[:, 0] == image_index[box_i]) & (targets[:, 1] == pred_classes[box_i])
        if not bool(sel.any()):
            continue
        target_boxes = targets[sel][:, 2:]
        ious = bbox_iou(pred_boxes[box_i].unsqueeze(0), target_boxes)
        if len(ious) > 0:
            iou, target_index = ious.max(0)
            if (target_index not in detected) or multi_boxes:
                iou_labels[box_i] = iou
                target_location[box_i] = target_boxes[target_index]
                if iou > 0.7:
                    detected += [target_index]
    return iou_labels, target_location


def box_regress(regress_param, roi_location):
    """Apply (dx, dy, dw, dh) to xyxy RoIs (reference :378-391)."""
    x, y, w, h = xyxy2xywh(roi_location).t()
    xr = regress_param[:, 0] * w + x
    yr = regress_param[:, 1] * h + y
    wr = torch.exp(regress_param[:, 2]) * w
    hr = torch.exp(regress_param[:, 3]) * h
    return xywh2xyxy(torch.stack((xr, yr, wr, hr), 1))


def regression_loss(regress_param, target_location, roi_location):
    """SmoothL1 (sum) on the encoded regression targets (reference :394-408)."""
    x, y, w, h = xyxy2xywh(roi_location).t()
    xt, yt, wt, ht = xyxy2xywh(target_location).t()
    p01 = torch.stack(((xt - x) / (w + 1e-16), (yt - y) / (h + 1e-16)), -1)
    p23 = torch.stack((torch.log(wt / w + 1e-16), torch.log(ht / h + 1e-16)), -1)
    loss_xy = torch.nn.SmoothL1Loss(reduction="sum")(p01, regress_param[:, :2])
    loss_wh = torch.nn.SmoothL1Loss(reduction="sum")(p23, regress_param[:, 2:])
    return loss_xy, loss_wh


# --------------------------------------------------------------------------------------------------
def _leaf(mod, *path):
    """``mod.a[0]`` as ``_leaf(mod, "a", 0)``: plain ``_modules`` dict lookups (no ``nn.Module.__getattr__`` /
    ``Sequential.__getitem__``); an int is a position inside a container."""
    for name in path:
        mods = mod._modules
        mod = mods[name] if isinstance(name, str) else next(itertools.islice(mods.values(), name, None))
    return mod


class _Leaves:
    """The leaf modules under a fixed set of paths, re-read on every call by three dict lookups each (the positions of a
    path are resolved to the containers' key names once and re-resolved when a key disappears), so a replaced inner module
    is seen at once at ~0.3 us per leaf."""

    def __init__(self, paths):
        self.paths, self.named = paths, None

    def _resolve(self, root):
        named = []
        for path in self.paths:
            mod, keys = root, []
            for name in path:
                mods = mod._modules
                key = name if isinstance(name, str) else next(itertools.islice(mods.keys(), name, None))
                keys.append(key)
                mod = mods[key]
            named.append(tuple(keys))
        self.named = named

    def __call__(self, root):
        if self.named is None:
            self._resolve(root)
        try:
            return [root._modules[a]._modules[b]._modules[c] for a, b, c in self.named]
        except KeyError:
            self._resolve(root)
            return [root._modules[a]._modules[b]._modules[c] for a, b, c in self.named]


_HEAD_LEAVES = (("refinement_head", "net0", 0), ("refinement_head", "net1", 0), ("refinement_head", "net2", 0),
                ("refinement_head", "radar_net", 0), ("refinement_head", "radar_net", 1),
                ("refinement_head", "radar_net", 3), ("ensemble_head", "fc1", 0), ("ensemble_head", "fc2", 0))


_COUNT_SPIN = os.environ.get("MILLIEYE_COUNT_SPIN", "0") == "1"  # row count of Network.forward: .item() / polled pinned word (A/B)

_HEAD_BN_PATHS = (("img_cnn_layers", "net", 1), ("radar_cnn_layers", "conv1", 1), ("radar_cnn_layers", "conv2", 1),
                  ("radar_cnn_layers", "conv3", 1), ("refinement_head", "radar_net", 1))


class _HeadPack:
    """Device copies of the small head weights in the layouts ``me_roi_heads_f32`` reads."""

    def __init__(self, net):
        self.net = net
        self._stamp = None
        self.t = {}
        self._leaves = _Leaves(_HEAD_LEAVES)

    def _sources(self):
        rh, eh = self.net.refinement_head, self.net.ensemble_head
        bn = rh.radar_net[1]
        return [rh.net0[0].weight, rh.net0[0].bias, rh.net1[0].weight, rh.net1[0].bias, rh.net2[0].weight,
                rh.net2[0].bias, rh.radar_net[0].weight, rh.radar_net[0].bias, bn.weight, bn.bias, bn.running_mean,
                bn.running_var, rh.radar_net[3].weight, rh.radar_net[3].bias, eh.fc1[0].weight, eh.fc1[0].bias,
                eh.fc2[0].weight, eh.fc2[0].bias]

    def _slots(self):
        """(dict, key) of every source tensor, read straight from the modules' ``_parameters`` / ``_buffers`` (the attribute chains
        of ``_sources`` cost ~40 us per forward through ``nn.Module.__getattr__``); rebuilt when a child module is replaced."""
        # keyed by the LEAF modules (``refinement_head.radar_net[1] = new BatchNorm`` or ``net0 = Sequential(...)``
        # replaces an inner module under an unchanged top-level child)
        leaves = self._leaves(self.net)
        key = tuple(map(id, leaves))
        cached = self.__dict__.get("_slot_cache")
        if cached is None or cached[0] != key:
            n0, n1, n2, r0, bn, r3, f1, f2 = leaves
            pairs = []
            for mod in (n0, n1, n2, r0):
                pairs += [(mod._parameters, "weight"), (mod._parameters, "bias")]
            pairs += [(bn._parameters, "weight"), (bn._parameters, "bias"), (bn._buffers, "running_mean"),
                      (bn._buffers, "running_var")]
            for mod in (r3, f1, f2):
                pairs += [(mod._parameters, "weight"), (mod._parameters, "bias")]
            cached = self._slot_cache = (key, pairs, leaves)  # (the leaves stay referenced: their ids cannot be recycled)
        return cached[1]

    def refresh(self, device):
        stamp = tuple([(t.data_ptr(), t._version) for t in [dct[key] for dct, key in self._slots()]]) \
            + (str(device), _engine._EPOCH[0])
        if stamp == self._stamp:
            return self.t
        rh, eh = self.net.refinement_head, self.net.ensemble_head
        f = dict(device=device, dtype=torch.float32)
        with torch.no_grad():
            bn = rh.radar_net[1]
            rscale = bn.weight.double() / torch.sqrt(bn.running_var.double() + bn.eps)
            rshift = (rh.radar_net[0].bias.double() - bn.running_mean.double()) * rscale + bn.bias.double()
            self.t = dict(
                w0t=rh.net0[0].weight.detach().t().contiguous().to(**f), b0=rh.net0[0].bias.detach().to(**f),
                w1=rh.net1[0].weight.detach().contiguous().to(**f), b1=rh.net1[0].bias.detach().to(**f),
                w2=rh.net2[0].weight.detach().contiguous().to(**f), b2=rh.net2[0].bias.detach().to(**f),
                rw=rh.radar_net[0].weight.detach().reshape(10, 490).contiguous().to(**f),
                rscale=rscale.to(**f), rshift=rshift.to(**f),
                rw2=rh.radar_net[3].weight.detach().reshape(10).contiguous().to(**f),
                rb2=rh.radar_net[3].bias.detach().reshape(1).to(**f),
                e1w=eh.fc1[0].weight.detach().contiguous().to(**f), e1b=eh.fc1[0].bias.detach().to(**f),
                e2w=eh.fc2[0].weight.detach().contiguous().to(**f), e2b=eh.fc2[0].bias.detach().to(**f),
            )
        self._stamp = stamp
        return self.t


class _BinMajorRows:
    """``cnn_layers_1``'s packed 1x1 weights with the 490 output channels reordered bin-major: channel ``(ph*7+pw)*10 + c_out``
    holds the reference's channel ``(c_out*7+ph)*7+pw`` (my_models.py:47-52 -> ps_roi_align, :495).  The score map is only read
    by the RoI pooling launch, which then finds the ten values of a sample point in 40 contiguous bytes
    (``me_heads_desc.img_bin_major``).  A row permutation of weight / scale / shift: every output value is computed exactly as
    before, it only lands in another channel slot.  Re-derived when the underlying pack changes."""

    def __init__(self, cw):
        self.cw = cw
        self.wgt = self.scale = self.shift = None
        self._key = None
        self._perm = None

    def refresh(self, device):
        self.cw.refresh(device)
        key = (self.cw._stamp, self.cw.wgt.data_ptr())
        if key != self._key:
            if self._perm is None or self._perm.device != self.cw.wgt.device:
                q = torch.arange(490, device=self.cw.wgt.device)
                self._perm = (q % 10) * 49 + q // 10
            with torch.no_grad():
                self.wgt = self.cw.wgt.index_select(0, self._perm).contiguous()
                self.scale = self.cw.scale.index_select(0, self._perm).contiguous()
                self.shift = self.cw.shift.index_select(0, self._perm).contiguous()
            self._key = key
        return self


class Network(nn.Module):
    """milliEye stage-3 network (reference my_models.py:411-641)."""

    def __init__(self, base_detector, conf_thresh):
        super().__init__()
        self.device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.conf_thresh = conf_thresh
        self.seen = 0
        self.iou_thresh = (0.3, 0.7)
        self.alpha = 0.75
        self.balance_factor = 5
        self.loss_lambda = (6, 1)
        self.refine_threshold_img, self.refine_threshold_radar = 0, 0
        self.class_num = 1
        self.class_idx = 0

        self.base_detector = base_detector.eval()
        self.img_cnn_layers = cnn_layers_1((256, 490))
        self.radar_cnn_layers = cnn_layers_3()
        self.refinement_head = refinement_head((490, 256, 128, self.class_num + 1))
        self.ensemble_head = ensemble_head((2, 32, 32 * (1 + self.class_num), 2))

        object.__setattr__(self, "_packs", None)

    # ---------------------------------------------------------------------------------- helpers
    def _get_packs(self):
        if self._packs is None:
            rc = self.radar_cnn_layers
            packs = dict(
                img=_BinMajorRows(ConvWeights(self.img_cnn_layers.net[0], self.img_cnn_layers.net[1])),
                bf16=_BinMajorRows(ConvWeights(self.img_cnn_layers.net[0], self.img_cnn_layers.net[1], "bf16")),
                f16=_BinMajorRows(ConvWeights(self.img_cnn_layers.net[0], self.img_cnn_layers.net[1], "f16")),
                r1=ConvWeights(rc.conv1[0], rc.conv1[1]),
                r2=ConvWeights(rc.conv2[0], rc.conv2[1]),
                r3=ConvWeights(rc.conv3[0], rc.conv3[1]),
                r4=ConvWeights(rc.conv3[3], None),
                heads=_HeadPack(self),
            )
            object.__setattr__(self, "_packs", packs)
        return self._packs

    def _head_bns(self):
        # (looked up once per set of child modules: five Sequential.__getitem__ chains per call were ~15 us, three calls in
        #  front of the detector's first launch)
        # the five BatchNorm leaves themselves, by ``_modules`` lookups: nothing cached, so a replaced inner module
        #  is seen at once
        get = self.__dict__.get("_bn_leaves")
        if get is None:
            get = self.__dict__["_bn_leaves"] = _Leaves(_HEAD_BN_PATHS)
        return get(self)

    def _check_eval(self):
        if any(b.training for b in self._head_bns()):  # (forward() sends an all-train()-mode model to train_path)
            raise NotImplementedError("Network.forward: the head BatchNorms are partly in train() and partly in eval() "
                                      "mode; call model.train() or model.eval() on the whole Network")

    @staticmethod
    def _conv16(x_ptr, x_pitch, n, h, w, cin, cw, ksize, pad, act, out):
        """16-bit feature tap (detector in a 16-bit storage mode) -> float32 score map: ``me_conv2d_h16`` with ``y_f32``."""
        d = hip.Conv16Desc()
        d.x, d.x_pitch, d.x_nchw, d.y_f32 = x_ptr, x_pitch, 0, 1
        d.half_type = hip.HALF_TYPES[cw.wgt.dtype]
        d.wgt, d.scale, d.shift, d.res, d.res_pitch = cw.wgt.data_ptr(), cw.scale.data_ptr(), cw.shift.data_ptr(), None, 0
        d.y, d.y_pitch = out.data_ptr(), out.shape[-1]
        d.n, d.h, d.w, d.cin, d.cout = n, h, w, cin, cw.wgt.shape[0]
        d.ksize, d.stride, d.pad, d.ho, d.wo = ksize, 1, pad, h, w
        d.act, d.upsample, d.tile, d.split_k = act, 1, 0, 1
        hip.check(hip.lib().me_conv2d_h16(C.byref(d), hip.stream_ptr()), "me_conv2d_h16")
        return out

    @staticmethod
    def _conv(x_ptr, x_pitch, x_nchw, n, h, w, cin, cw, ksize, pad, act, out):
        d = hip.ConvDesc()
        d.x, d.x_pitch, d.x_nchw = x_ptr, x_pitch, 1 if x_nchw else 0
        d.wgt, d.scale, d.shift, d.res, d.res_pitch = cw.wgt.data_ptr(), cw.scale.data_ptr(), cw.shift.data_ptr(), None, 0
        d.y, d.y_pitch = out.data_ptr(), out.shape[-1]
        d.n, d.h, d.w, d.cin, d.cout = n, h, w, cin, cw.wgt.shape[0]
        d.ksize, d.stride, d.pad, d.ho, d.wo = ksize, 1, pad, h, w
        d.act, d.upsample, d.tile, d.split_k = act, 1, 0, 0
        need = hip.lib().me_conv2d_workspace_bytes(C.byref(d))
        if need > 0:
            d.workspace, _keep = hip._workspace(need, out.device, slot="conv")
            d.workspace_bytes = need
        hip.check(hip.lib().me_conv2d_f32(C.byref(d), hip.stream_ptr()), "me_conv2d_f32")
        return out

    def _buf(self, name, shape, dev, dtype=torch.float32):
        """Scratch tensor of the post-detector tail, kept from call to call (round 5: the tail issued ~18 ``torch.empty`` per
        forward, 3 - 4 us of host time each, in the one part of a small-batch step where the host is the slower side).  Every
        launch that reads or writes these buffers is ordered on the forward's main stream or on the side stream behind an event
        of the main stream, so the next forward's launches queue behind this forward's.  Contents are valid until the next
        ``forward`` (``self._last`` hands some of them out, like ``plan.tap``).  The returned rows are NOT from this pool.  Keyed by the
        issuing stream; forwards of one network still run one at a time (the detector's plan arena is per network, not per stream)."""
        pool = self.__dict__.get("_tail_bufs")
        if pool is None:
            pool = self.__dict__["_tail_bufs"] = {}
        key = (name, shape, dtype, dev, hip.stream_ptr().value)   # (per stream: two forwards of one network on two streams never share scratch)
        t = pool.get(key)
        if t is None:
            if len(pool) > 96:  # (the RoI capacity follows the number of radar boxes of a call: keep the pool bounded)
                pool.clear()
            t = pool[key] = torch.empty(shape, device=dev, dtype=dtype)
        return t

    def _side_stream(self, dev):
        st = self.__dict__.get("_side")
        if st is None or st.device != dev:
            st = torch.cuda.Stream(device=dev)
            object.__setattr__(self, "_side", st)
        return st

    def _radar_score_map(self, maps, n, dev):
        """radar_score_map [n,h,w,10 (pitch 12)] = cnn_layers_3 on the radar maps, on the current stream.  It depends on
        nothing the detector computes, so ``forward`` starts it on the side stream BEFORE the detector (four small launches
        that used to sit behind the detector on the critical path of the score-map branch)."""
        f32 = dict(device=dev, dtype=torch.float32)
        packs = self._get_packs()
        for key in ("r1", "r2", "r3", "r4"):
            packs[key].refresh(dev)
        maps = maps.contiguous()
        if not (maps.is_cuda and maps.dtype == torch.float32):
            raise hip.MeError("radar maps must be CUDA float32 [N,3,h,w]")
        mh, mw = maps.shape[2], maps.shape[3]
        t1 = self._buf("r1", (n, mh, mw, 32), dev)
        t2 = self._buf("r2", (n, mh, mw, 64), dev)
        t3 = self._buf("r3", (n, mh, mw, 128), dev)
        radar_score_map = self._buf("rmap", (n, mh, mw, 12), dev)  # 10 channels, pitch 12
        self._conv(maps.data_ptr(), 3, True, n, mh, mw, 3, packs["r1"], 3, 1, hip.ACT_LEAKY, t1)
        self._conv(t1.data_ptr(), 32, False, n, mh, mw, 32, packs["r2"], 3, 1, hip.ACT_LEAKY, t2)
        self._conv(t2.data_ptr(), 64, False, n, mh, mw, 64, packs["r3"], 3, 1, hip.ACT_LEAKY, t3)
        self._conv(t3.data_ptr(), 128, False, n, mh, mw, 128, packs["r4"], 1, 0, hip.ACT_SIGMOID, radar_score_map)
        # the reference hands both maps to the RoI ops with the same spatial_scale; a radar map of another size (the
        # demos feed the raw 32 x 32 map, quirk q15) is legal there: the pooling kernel takes per-map sizes
        self._keep_side = (t1, t2, t3, maps)  # alive until the next forward: the side stream may still read them
        return radar_score_map, mh, mw

    def _roi_score_map(self, plan, n, dev):
        """roi_score_map [n,fh,fw,490] = cnn_layers_1 on the detector's feature tap, on the current stream; channels bin-major
        (:class:`_BinMajorRows`)."""
        packs = self._get_packs()
        tap16 = getattr(plan, "dtype", "f32") != "f32"
        packs[plan.dtype if tap16 else "img"].refresh(dev)
        fh, fw, fc = plan.tap_shape
        roi_score_map = self._buf("imap", (n, fh, fw, 490), dev)
        if tap16:
            self._conv16(plan.tap_ptr, plan.tap_pitch, n, fh, fw, fc, packs[plan.dtype], 1, 0, hip.ACT_LEAKY, roi_score_map)
        else:
            self._conv(plan.tap_ptr, plan.tap_pitch, False, n, fh, fw, fc, packs["img"], 1, 0, hip.ACT_LEAKY, roi_score_map)
        return roi_score_map, fh, fw

    def _score_maps(self, plan, maps, n, dev, radar_job=None):
        """Both score maps on the current stream (``radar_job``: the radar half was already started there); returns
        (roi_score_map, radar_score_map, fh, fw, mh, mw)."""
        roi_score_map, fh, fw = self._roi_score_map(plan, n, dev)
        radar_score_map, mh, mw = radar_job if radar_job is not None else self._radar_score_map(maps, n, dev)
        return roi_score_map, radar_score_map, fh, fw, mh, mw

    def queue_detector_prefetch(self, images_next):
        """Training loops with a FROZEN detector (the reference's stage 3, train.py:170): name the next batch's frames in front of the
        call for the current batch, and that call issues the next batch's detector + NMS + proposal assembly on a second stream as
        soon as it has its own - the detector of batch k + 1 runs under the host-bound tail of batch k (millieye_amd/train_path.py).
        Purely an overlap: the next call takes the prefetched result only when frames, thresholds and detector weights are the
        ones it was computed from, and computes it itself otherwise.  No reference counterpart (a scheduling hint, not an op)."""
        self.__dict__["_next_images"] = images_next

    # ---------------------------------------------------------------------------------- forward
    def forward(self, images, maps, radar_boxes_location, model_mode=0, targets=None):
        """See the reference docstring (my_models.py:434-450).  Returns ``output [m, 8]`` rows
        ``(image_i, x1, y1, x2, y2, object_conf, class_score, class_pred)`` sorted by confidence.

        Compat rule: ``train.py:185`` passes ``targets`` in the ``model_mode`` slot; a
        tensor there is therefore taken as ``targets`` with mode 0."""
        if isinstance(model_mode, torch.Tensor):
            targets, model_mode = model_mode, 0
        if targets is not None:  # training call: (loss, output, metric, radar_attention), reference :545-641
            from .train_path import forward_train
            return forward_train(self, images, maps, radar_boxes_location, targets)
        if not images.is_cuda:
            raise hip.MeError("Network.forward needs CUDA tensors (MI355X); there is no CPU fallback")
        if model_mode != 1 and all(b.training for b in self._head_bns()):
            # a model left in train() mode, called without targets (reference :433-539 under model.train()): same rows,
            # BatchNorm on batch statistics - the training path's forward half
            from .train_path import forward_train
            return forward_train(self, images, maps, radar_boxes_location, None, model_mode)
        dev = images.device
        n = images.shape[0]
        f32 = dict(device=dev, dtype=torch.float32)

        # ---- candidate boxes from the base detector (reference :454-473), all on device
        cb = getattr(self, "_stage_cb", None)
        mark = cb or (lambda _name: None)  # bench.py: per-stage HIP events
        mark("start")
        tr = self.__dict__.get("_trace_cb") or (lambda _name: None)  # tools/b1_tail_events.py: (host time, HIP event on the
        # current stream) per point, WITHOUT changing the stream structure (the stage marks above switch the overlap off)
        tr("entry")
        # The radar CNN needs nothing from the detector: it runs on the side stream beside it (mode 0 / 2 / 3).  Its four launches are
        # ISSUED behind the detector's (the host is the slower side while the detector's ~125 launches go out, and whatever the
        # host does in front of the first one is idle time of the GPU - ~0.1 ms of a 1.8 ms step at batch 1); the side stream only
        # waits for the work that was queued when forward() was entered.
        radar_job = None
        entered = None
        if cb is None and model_mode != 1:
            self._check_eval()
            entered = torch.cuda.Event()
            entered.record()  # (on the current stream)
        if entered is not None and os.environ.get("MILLIEYE_RADAR_FIRST") == "1":  # (A/B: the previous order)
            side = self._side_stream(dev)
            side.wait_event(entered)
            with torch.cuda.stream(side):
                radar_job = self._radar_score_map(maps, n, dev)
            entered = None
        plan, yolo_out = self.base_detector._run(images, nms_conf=float(self.conf_thresh))  # the decode fills the NMS lists
        tr("detector issued")
        if entered is not None:
            side = self._side_stream(dev)
            side.wait_event(entered)
            with torch.cuda.stream(side):
                radar_job = self._radar_score_map(maps, n, dev)
                tr("side: radar CNN issued")
        mark("detector")
        # The score maps (reference :486-487) only need the feature tap, NMS only the decoded rows: NMS keeps 32
        # workgroups busy for ~0.3 ms, so the score-map convolutions run beside it on a second stream (mode 0 / 2 / 3).
        overlap = cb is None and model_mode != 1 and plan.tap is not None
        maps_job = None
        if overlap:
            self._check_eval()
            side = self._side_stream(dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                maps_job = self._score_maps(plan, maps, n, dev, radar_job)
                tr("side: score map issued")
        det, cnt = hip.nms_batched(yolo_out, float(self.conf_thresh), _NMS_THRESH, _DETECTIONS_PER_IMG,
                                   writeback_xyxy=False, prepped=plan.nms_prepped == float(self.conf_thresh))
        mark("nms")
        tr("nms issued")
        num_classes = yolo_out.shape[2] - 5
        cols = 8 + self.class_num
        cap_img = n * _DETECTIONS_PER_IMG
        img_boxes = self._buf("img_boxes", (cap_img, cols), dev)
        n_img_dev = self._buf("n_img", (1,), dev, torch.int32)
        lib = hip.lib()
        hip.check(lib.me_gather_class_boxes_f32(det.data_ptr(), cnt.data_ptr(), n, _DETECTIONS_PER_IMG, num_classes,
                                                int(self.class_idx), int(self.class_num), img_boxes.data_ptr(),
                                                n_img_dev.data_ptr(), hip.stream_ptr()), "me_gather_class_boxes_f32")
        mark("proposals")
        tr("proposals issued")
        if model_mode == 1:  # yolo only
            return img_boxes[: int(n_img_dev.item()), :8].clone()  # (img_boxes is a pooled scratch buffer: the caller gets its own rows)
        if model_mode == 2:  # radar only: permanent, like the reference (quirk q3)
            self.refine_threshold_img = 1

        # ---- score maps (reference :486-487)
        if maps_job is None:
            self._check_eval()
            if plan.tap is None:
                raise AttributeError("'Darknet' object has no attribute 'featuremap'")
            maps_job = self._score_maps(plan, maps, n, dev)
        else:
            torch.cuda.current_stream(dev).wait_stream(self._side_stream(dev))
        roi_score_map, radar_score_map, fh, fw, rh_, rw_ = maps_job
        packs = self._get_packs()
        mark("score_maps")
        tr("joined side stream")

        # ---- RoIs: image proposals then radar proposals (reference :490-492)
        if len(radar_boxes_location) > 0:
            radar_boxes_location[:, 1:] *= images.shape[-1]  # in place on the caller's tensor, like the reference
        n_radar = int(radar_boxes_location.shape[0])
        rb = radar_boxes_location.to(**f32).contiguous() if n_radar else None
        cap = cap_img + n_radar
        # the pool is keyed by a capacity BUCKET (multiples of 64 rows): the radar box count changes from frame to frame on real
        # data, and an exact-shape key would miss the pool on most calls; the launches get leading-row views of ``cap`` rows
        cap_b = -(-cap // 64) * 64
        regress = self._buf("regress", (cap_b, 4), dev)[:cap]
        refine = self._buf("refine", (cap_b, 2), dev)[:cap]
        mask1 = self._buf("mask1", (cap_b,), dev)[:cap]
        rows = self._buf("rows", (cap_b, 8), dev)[:cap]
        keep = self._buf("keep", (cap_b,), dev, torch.uint8)[:cap]  # (the heads launch clears the slots behind the last RoI)
        key = self._buf("key", (cap_b,), dev)[:cap]

        hw = packs["heads"].refresh(dev)
        d = hip.HeadsDesc()
        d.img_map, d.radar_map = roi_score_map.data_ptr(), radar_score_map.data_ptr()
        d.img_pitch, d.radar_pitch, d.img_bin_major = 490, 12, 1
        d.n, d.fh, d.fw, d.spatial_scale = n, fh, fw, 1.0 / 16
        d.rh, d.rw = rh_, rw_
        d.img_boxes, d.n_img, d.n_img_cap, d.box_cols = img_boxes.data_ptr(), n_img_dev.data_ptr(), cap_img, cols
        d.radar_boxes, d.n_radar = (rb.data_ptr() if n_radar else None), n_radar
        d.thr_img, d.thr_radar = float(self.refine_threshold_img), float(self.refine_threshold_radar)
        d.regress = 0 if model_mode == 2 else 1
        for name in ("w0t", "b0", "w1", "b1", "w2", "b2", "rw", "rscale", "rshift", "rw2", "rb2", "e1w", "e1b", "e2w",
                     "e2b"):
            setattr(d.wts, name, hw[name].data_ptr())
        d.regress_out, d.refine_out, d.mask1_out = regress.data_ptr(), refine.data_ptr(), mask1.data_ptr()
        d.out_rows, d.keep, d.sort_key = rows.data_ptr(), keep.data_ptr(), key.data_ptr()
        if not getattr(self, "_fused_heads", False):
            pooled = self._buf("pooled", (cap_b, 980), dev)[:cap]  # the RoI pooling as its own launch (me_heads_desc.pool_scratch)
            d.pool_scratch = pooled.data_ptr()
        # (``_fused_heads``: the single-launch VALU kernel with the pooling inside - kept as the cross-check of the two-launch
        # path, tests/test_gpu_network.py)
        hip.check(lib.me_roi_heads_f32(C.byref(d), hip.stream_ptr()), "me_roi_heads_f32")
        mark("roi_heads")
        tr("heads issued")
        self.refinement_head.count += 1

        # ---- keep positives, order by confidence (reference :517-539): one launch, then the one host sync (the row count)
        ordered = torch.empty((cap, 8), **f32)
        if _COUNT_SPIN:
            # A/B form of the one host sync of the call (round 5): the kernel stores the count straight into a pinned host word
            # (system-scope store) that the host preset to -1 and polls, instead of ``n_out.item()`` (a 4-byte blocking copy).
            # tools/b1_tail_events.py suggested ~70 us between the end of the compaction kernel and the count on the host; the
            # poll showed that gap to be the pick-up latency of the trace's FIRST event on an idle GPU, not a read-back cost:
            # 1.637 vs 1.628 ms (fp32) and 0.851 vs 0.845 ms (bf16) per batch-1 step with / without the poll - no gain, off.
            word = self.__dict__.get("_count_word")
            if word is None:
                host = torch.zeros((16,), dtype=torch.int32).pin_memory()
                word = self.__dict__["_count_word"] = (host, C.cast(host.data_ptr(), C.POINTER(C.c_int32)))
            host, flag = word
            flag[0] = -1
            hip.check(lib.me_compact_sort_rows_f32(rows.data_ptr(), keep.data_ptr(), key.data_ptr(), cap, 8, ordered.data_ptr(),
                                                   host.data_ptr(), hip.stream_ptr()), "me_compact_sort_rows_f32")
            tr("compaction issued")
            spins = 0
            while flag[0] == -1:
                spins += 1
                if spins > 20_000_000:  # (seconds: a wedged stream - let the runtime report it)
                    torch.cuda.current_stream(dev).synchronize()
                    if flag[0] == -1:
                        raise hip.MeError("Network.forward: the compaction launch never delivered its row count")
            n_rows = int(flag[0])
        else:
            n_out = self._buf("n_out", (1,), dev, torch.int32)
            hip.check(lib.me_compact_sort_rows_f32(rows.data_ptr(), keep.data_ptr(), key.data_ptr(), cap, 8, ordered.data_ptr(),
                                                   n_out.data_ptr(), hip.stream_ptr()), "me_compact_sort_rows_f32")
            tr("compaction issued")
            n_rows = int(n_out.item())
        tr("row count on the host")
        output = ordered[:n_rows]
        mark("output")
        # (views of pooled scratch: valid only until the next forward of this network on this stream)
        self._last = dict(regress=regress, refine=refine, mask1=mask1, n_img=n_img_dev, img_boxes=img_boxes)
        return output
