"""Stage-2 network of milliEye: ``module2_mixed/my_models.py``, inference path.

``Network(base_detector, conf_thresh).forward(images)`` -> ``output [m,8]`` **on the CPU** (reference :361) with rows
``(image_i, x1, y1, x2, y2, new_confidence, class_conf, class_pred)`` sorted by the new confidence.  Same device
pipeline as stage 3 (``millieye_amd/my_models.py``) minus the radar branch, plus every class instead of class 0:

    DarknetEngine -> me_nms_batched_f32 -> me_gather_class_boxes_f32(class_idx=-1: all classes, 8+12 columns)
                  -> me_conv2d_f32 (fcn_layers: 1x1 256->490 + BN + LeakyReLU on the feature-map tap)
                  -> me_m2_heads_f32 (PS-RoIAlign + refinement_head + ensemble_head + box_regress, csrc/heads.hip)

The module tree carries the reference's parameter names (``fcn_layers.net.conv_0.weight``, ``refinement_head.net0.0.weight``,
``ensemble_head.fc2.0.weight`` ...) so stage-2 checkpoints load unchanged and ``train.load_pretrained_module2`` can hand them
to stage 3.  With ``targets`` the call returns ``(output, loss, metric)`` and ``loss.backward()`` runs the HIP backward of
``millieye_amd/module2/train_path.py`` (focal + confidence + category + SmoothL1 box losses, reference :366-459); in
``train()`` mode without targets it returns the rows of the train-mode forward (batch-statistics BatchNorm, Dropout).
"""
import ctypes as C

import torch
from torch import nn

from .. import engine as _engine
from .. import hip
from ..engine import ConvWeights
from ..my_models import FocalLoss, _DETECTIONS_PER_IMG, _NMS_THRESH, box_regress, define_yolo, init_yolo  # noqa: F401
from ..my_models import cnn_layers_1 as _cnn_layers_1

__all__ = ["Network", "define_yolo", "init_yolo", "fcn_layers", "refinement_head", "ensemble_head", "FocalLoss",
           "box_regress"]


class fcn_layers(_cnn_layers_1):
    """1x1 conv + BatchNorm + LeakyReLU stack (reference :47-77); same children names as stage 3's ``cnn_layers_1``."""


class refinement_head(nn.Module):
    """Parameter container of reference :96-127, channels e.g. (490, 256, c+1)."""

    def __init__(self, channels):
        super().__init__()
        self.net0 = nn.Sequential(nn.Linear(channels[0], channels[1]), nn.LeakyReLU(0.1), nn.Dropout(0.5))
        self.net1 = nn.Sequential(nn.Linear(channels[1], 4))
        self.net2 = nn.Sequential(nn.Linear(channels[1], channels[2]), nn.Sigmoid())

    def forward(self, img_maps):
        raise hip.MeError("refinement_head is a parameter container here; Network.forward runs me_m2_heads_f32")


class ensemble_head(nn.Module):
    """Parameter container of reference :130-164, channels e.g. (2, 32, 32*(c+1), 2); note the LeakyReLU after fc2."""

    def __init__(self, channels, use_activation=True):
        super().__init__()
        self.use_activation = use_activation
        self.fc1 = nn.Sequential(nn.Linear(channels[0], channels[1]), nn.LeakyReLU(0.1))
        self.fc2 = nn.Sequential(nn.Linear(channels[2], channels[3]), nn.LeakyReLU(0.1))
        self.softmax = nn.Softmax(dim=1)

    def forward(self, refinement_vector, yolo_vector):
        raise hip.MeError("ensemble_head is a parameter container here; Network.forward runs me_m2_heads_f32")


class _HeadPack:
    def __init__(self, net):
        self.net, self._stamp, self.t = net, None, {}

    def refresh(self, device):
        rh, eh = self.net.refinement_head, self.net.ensemble_head
        src = [rh.net0[0].weight, rh.net0[0].bias, rh.net1[0].weight, rh.net1[0].bias, rh.net2[0].weight, rh.net2[0].bias,
               eh.fc1[0].weight, eh.fc1[0].bias, eh.fc2[0].weight, eh.fc2[0].bias]
        stamp = tuple((t.data_ptr(), t._version) for t in src) + (str(device), _engine._EPOCH[0])
        if stamp != self._stamp:
            f = dict(device=device, dtype=torch.float32)
            with torch.no_grad():
                self.t = dict(w0t=src[0].detach().t().contiguous().to(**f), b0=src[1].detach().to(**f),
                              w1=src[2].detach().contiguous().to(**f), b1=src[3].detach().to(**f),
                              w2=src[4].detach().contiguous().to(**f), b2=src[5].detach().to(**f),
                              e1w=src[6].detach().contiguous().to(**f), e1b=src[7].detach().to(**f),
                              e2w=src[8].detach().contiguous().to(**f), e2b=src[9].detach().to(**f))
            self._stamp = stamp
        return self.t


class Network(nn.Module):
    """Reference module2_mixed/my_models.py:282-461."""

    def __init__(self, base_detector, conf_thresh):
        super().__init__()
        self.device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.conf_thresh = conf_thresh
        self.seen = 0
        self.iou_thresh = (0.3, 0.7)
        self.alpha = 0.75
        self.balance_fac = 5
        self.loss_lambda = (15, 5)
        self.refine_threshold = 0
        self.class_num = 12
        self.base_detector = base_detector.eval()
        self.fcn_layers = fcn_layers((256, 490))
        self.refinement_head = refinement_head((490, 256, self.class_num + 1))
        self.ensemble_head = ensemble_head((2, 32, 32 * (self.class_num + 1), 2))
        object.__setattr__(self, "_packs", None)
        # where nn.Dropout's keep mask of the training step comes from (module2/train_path.py): "philox" - drawn on the device by
        # me_dropout_mask_u8 from one CPU-generator seed per step (default; reproducible under torch.manual_seed); "device" - torch's
        # GPU generator (what the reference does on a CUDA machine); "cpu" - torch's CPU generator element by element (the reference's
        # CPU run bit for bit: the goldens and the oracle comparisons; 3.6 ms of host time per step)
        self.dropout_generator = "philox"

    def queue_detector_prefetch(self, images_next):
        """The stage-3 Network's look-ahead hint (millieye_amd/my_models.py) for the stage-2 loop: the detector is frozen here too
        (module2_mixed/train.py:129 ``model.base_detector.eval()``, its parameters are not handed to the optimizer), so the next
        batch's detector + NMS + proposal rows can run under this batch's host-bound tail.  Purely an overlap."""
        self.__dict__["_next_images"] = images_next

    def forward(self, images, targets=None):
        if targets is not None or self.fcn_layers.net[1].training or self.refinement_head.training:
            # training call: (output, loss, metric), reference :366-461; without targets in train() mode: the output rows of
            # the train-mode forward (batch-statistics BatchNorm, active Dropout)
            from .train_path import forward_train
            return forward_train(self, images, targets)
        if not images.is_cuda:
            raise hip.MeError("Network.forward needs CUDA tensors (MI355X); there is no CPU fallback")
        dev, n = images.device, images.shape[0]
        f32 = dict(device=dev, dtype=torch.float32)
        lib = hip.lib()
        plan, yolo_out = self.base_detector._run(images, nms_conf=float(self.conf_thresh))
        det, cnt = hip.nms_batched(yolo_out, float(self.conf_thresh), _NMS_THRESH, _DETECTIONS_PER_IMG,
                                   writeback_xyxy=False, prepped=plan.nms_prepped == float(self.conf_thresh))
        num_classes = yolo_out.shape[2] - 5
        if num_classes < self.class_num:
            raise hip.MeError(f"the detector has {num_classes} classes, module 2 needs {self.class_num}")
        cols, cap = 8 + self.class_num, n * _DETECTIONS_PER_IMG
        boxes = torch.empty((cap, cols), **f32)
        n_dev = torch.empty((1,), device=dev, dtype=torch.int32)
        hip.check(lib.me_gather_class_boxes_f32(det.data_ptr(), cnt.data_ptr(), n, _DETECTIONS_PER_IMG, num_classes, -1,
                                                self.class_num, boxes.data_ptr(), n_dev.data_ptr(), hip.stream_ptr()),
                  "me_gather_class_boxes_f32")
        if plan.tap is None:
            raise AttributeError("'Darknet' object has no attribute 'featuremap'")
        if self._packs is None:
            object.__setattr__(self, "_packs", dict(img=ConvWeights(self.fcn_layers.net[0], self.fcn_layers.net[1]),
                                                    bf16=ConvWeights(self.fcn_layers.net[0], self.fcn_layers.net[1], "bf16"),
                                                    f16=ConvWeights(self.fcn_layers.net[0], self.fcn_layers.net[1], "f16"),
                                                    heads=_HeadPack(self)))
        tap16 = getattr(plan, "dtype", "f32") != "f32"  # detector in a 16-bit storage mode (Darknet.compute_dtype)
        self._packs[plan.dtype if tap16 else "img"].refresh(dev)
        hw = self._packs["heads"].refresh(dev)
        fh, fw, fc = plan.tap_shape
        score_map = torch.empty((n, fh, fw, 490), **f32)
        from ..my_models import Network as _N3
        if tap16:
            _N3._conv16(plan.tap_ptr, plan.tap_pitch, n, fh, fw, fc, self._packs[plan.dtype], 1, 0, hip.ACT_LEAKY, score_map)
        else:
            _N3._conv(plan.tap_ptr, plan.tap_pitch, False, n, fh, fw, fc, self._packs["img"], 1, 0, hip.ACT_LEAKY,
                      score_map)
        w = hip.HeadsWeights()
        for name, t in hw.items():
            setattr(w, name, t.data_ptr())
        regress = torch.empty((cap, 4), **f32)
        refine = torch.empty((cap, self.class_num + 1), **f32)
        mask = torch.empty((cap,), **f32)
        rows = torch.empty((cap, 8), **f32)
        keep = torch.zeros((cap,), device=dev, dtype=torch.uint8)
        key = torch.empty((cap,), **f32)
        pooled = torch.empty((cap, 490), **f32)  # the PS-RoIAlign as its own launch (one workgroup per box)
        hip.check(lib.me_m2_heads_f32(score_map.data_ptr(), 490, n, fh, fw, 1.0 / 16, boxes.data_ptr(), n_dev.data_ptr(),
                                      cap, cols, self.class_num, C.byref(w), float(self.refine_threshold),
                                      regress.data_ptr(), refine.data_ptr(), mask.data_ptr(), rows.data_ptr(),
                                      keep.data_ptr(), key.data_ptr(), pooled.data_ptr(), hip.stream_ptr()), "me_m2_heads_f32")
        ordered = torch.empty((cap, 8), **f32)
        n_out = torch.empty((1,), device=dev, dtype=torch.int32)
        hip.check(lib.me_compact_sort_rows_f32(rows.data_ptr(), keep.data_ptr(), key.data_ptr(), cap, 8, ordered.data_ptr(),
                                               n_out.data_ptr(), hip.stream_ptr()), "me_compact_sort_rows_f32")
        self._last = dict(regress=regress, refine=refine, mask=mask, n_boxes=n_dev, boxes=boxes)
        return ordered[:int(n_out.item())].cpu()
