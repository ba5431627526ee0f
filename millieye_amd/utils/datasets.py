"""Input producer: ``MyDataset`` with the batch assembled on the GPU.

Mirror of ``module3_our_dataset/utils/datasets.py:109-326``: same constructor, file layout
(``dataset.txt``, ``image/*.jpg``, ``label/*.txt``, ``radar_box/*.pkl``, ``radar_point/*.pkl``), scene split, label /
radar-box arithmetic and ``collate_fn`` return tuple ``(paths, imgs, targets, radar_boxes, radar_maps)``.

What moved to the device (``csrc/input.hip``):

* ``ToTensor`` + ``pad_to_square`` + nearest ``resize`` of every frame  -> ``me_image_pad_resize_u8_f32``
* ``plot_radar_heatmap`` (3 x ``np.histogram2d``) + ``ToTensor().float()`` + ``pad_to_square`` + bilinear
  ``F.interpolate(align_corners=True)`` of every radar map             -> ``me_radar_heatmap_f32`` (one launch per batch)

``__getitem__`` therefore only decodes (PIL, like the reference) and parses the small text / pickle files - it stays
picklable CPU data, so ``DataLoader(num_workers>0)`` still works - and ``collate_fn`` returns *staged* batches for
``imgs`` / ``radar_maps``: objects whose ``.to(device)`` (the call ``train.py:181-182`` / ``test_fusion.py:66-67`` make on
them anyway) uploads the raw bytes / points and launches the kernels in the consumer process.  ``targets`` and
``radar_boxes`` are the same CPU tensors as in the reference.  There is no CPU implementation here:
``.to("cpu")`` raises (the CPU restatement used by the tests lives in ``oracle/datasets_ref.py``).
"""
import os
import pickle
import random

import numpy as np
import torch
from torch.utils.data import Dataset

from .. import hip

__all__ = ["MyDataset", "StagedImages", "StagedRadarMaps", "obtain_bboxs"]

_SCENES = ["0", "1", "2", "3", "4"]


def obtain_bboxs(path):
    """Annotation lines ``label x y w h`` of a text file (datasets.py:41-56); ``%`` starts a comment line."""
    out = []
    with open(path, "r") as fh:
        for line in fh.read().split("\n"):
            if not line or line.startswith("%"):
                continue
            items = line.strip().split(" ")
            out.append([items[0]] + [float(v) for v in items[1:5]])
    return out


def _pad_amounts(h, w):
    """(left, right, top, bottom) of ``pad_to_square`` (datasets.py:16-27)."""
    diff = abs(h - w)
    pad1, pad2 = diff // 2, diff - diff // 2
    return (0, 0, pad1, pad2) if h <= w else (pad1, pad2, 0, 0)


def read_label_rows(label_path):
    """A YOLO label file as a float64 ``[k,5]`` tensor ``(class, cx, cy, w, h)`` (``np.loadtxt``, like the reference)."""
    return torch.from_numpy(np.loadtxt(label_path).reshape(-1, 5))


def letterbox_labels(lab, scale_hw, pad, padded_hw):
    """Label geometry shared by both producers (stage 3 ``MyDataset``, stage 2 ``module2.ListDataset``).

    ``lab``: float64 ``[k,5]`` rows ``(class, cx, cy, w, h)``; ``scale_hw = (sy, sx)`` turns its coordinates into pixels of
    the *unpadded* frame (the frame's height / width for normalised labels, ``(1, 1)`` for pixel labels); ``pad`` is
    ``_pad_amounts`` of that frame, ``padded_hw`` the padded size.  Returns float32 ``[k,6]`` rows ``(0, class, cx, cy, w, h)``
    relative to the padded square.  float64 arithmetic in the reference's order (module3 ``datasets.py:224-245``, module2
    ``utils/datasets.py:111-135``): corners in pixels, each shifted by the padding on its side, centre back to [0,1];
    extents scaled by one python-float ratio."""
    sy, sx = scale_hw
    padded_h, padded_w = padded_hw
    half_w, half_h = lab[:, 3] / 2, lab[:, 4] / 2
    left = (lab[:, 1] - half_w) * sx + pad[0]
    right = (lab[:, 1] + half_w) * sx + pad[1]
    top = (lab[:, 2] - half_h) * sy + pad[2]
    bottom = (lab[:, 2] + half_h) * sy + pad[3]
    out = torch.zeros((len(lab), 6))
    out[:, 1] = lab[:, 0]
    out[:, 2] = ((left + right) / 2) / padded_w
    out[:, 3] = ((top + bottom) / 2) / padded_h
    out[:, 4] = lab[:, 3] * (sx / padded_w)
    out[:, 5] = lab[:, 4] * (sy / padded_h)
    return out


def decode_rgb_u8(img_path):
    """PIL decode -> uint8 ``[h,w,3]`` tensor (the only image work left on the host)."""
    from PIL import Image
    return torch.from_numpy(np.array(Image.open(img_path).convert("RGB"), dtype=np.uint8))


class StagedImages:
    """A batch of decoded frames waiting for ``.to(device)``: uint8 HWC tensors + the target side."""

    def __init__(self, frames, size, flips=None):
        self.frames, self.size = list(frames), int(size)
        # stage-2 augmentation (module2_mixed/utils/datasets.py:143-146): mirror the padded square before the resize
        self.flips = [bool(f) for f in flips] if flips is not None else [False] * len(self.frames)
        self.shape = torch.Size((len(self.frames), 3, self.size, self.size))
        self.dtype = torch.float32

    def __len__(self):
        return len(self.frames)

    def size_(self, dim=None):
        return self.shape if dim is None else self.shape[dim]

    def to(self, device, *_, **__):
        device = torch.device(device)
        if device.type != "cuda":
            raise hip.MeError("StagedImages.to(): the batch is assembled by the HIP library - CUDA device required")
        lib = hip.lib()
        out = torch.empty(tuple(self.shape), device=device, dtype=torch.float32)
        stream = hip.stream_ptr()
        for i, frame in enumerate(self.frames):
            h, w, c = frame.shape
            if c != 3 or frame.dtype != torch.uint8:
                raise hip.MeError(f"frame {i}: expected uint8 [h,w,3], got {frame.dtype} {tuple(frame.shape)}")
            d = frame.contiguous().to(device, non_blocking=True)
            hip.check(lib.me_image_pad_resize_flip_u8_f32(d.data_ptr(), h, w, out[i].data_ptr(), self.size,
                                                          int(self.flips[i]), stream), "me_image_pad_resize_flip_u8_f32")
        return out


class StagedRadarMaps:
    """Radar points of a batch waiting for ``.to(device)``: per frame ``[n_i,4]`` float64 (u, v, depth, velocity) and
    the original image size ``(w, h)``."""

    def __init__(self, points, sizes, map_size, radar_maps_size=32):
        self.points, self.sizes = list(points), list(sizes)
        self.map_size, self.radar_maps_size = int(map_size), int(radar_maps_size)
        self.shape = torch.Size((len(self.points), 3, self.map_size, self.map_size))
        self.dtype = torch.float32

    def __len__(self):
        return len(self.points)

    def to(self, device, *_, **__):
        device = torch.device(device)
        if device.type != "cuda":
            raise hip.MeError("StagedRadarMaps.to(): the maps are computed by the HIP library - CUDA device required")
        n = len(self.points)
        out = torch.empty(tuple(self.shape), device=device, dtype=torch.float32)
        if n == 0:
            return out
        rows = [np.asarray(p, dtype=np.float64).reshape(-1, 4) for p in self.points]
        offsets = np.zeros(n + 1, dtype=np.int32)
        offsets[1:] = np.cumsum([len(r) for r in rows])
        flat = np.concatenate(rows, 0) if offsets[-1] else np.zeros((1, 4), np.float64)
        d_pts = torch.from_numpy(np.ascontiguousarray(flat)).to(device)
        d_off = torch.from_numpy(offsets).to(device)
        d_sz = torch.tensor(self.sizes, dtype=torch.int32).reshape(n, 2).to(device)
        hip.check(hip.lib().me_radar_heatmap_f32(d_pts.data_ptr(), d_off.data_ptr(), d_sz.data_ptr(), n,
                                                 self.radar_maps_size, out.data_ptr(), self.map_size, hip.stream_ptr()),
                  "me_radar_heatmap_f32")
        return out


class MyDataset(Dataset):
    """``__getitem__`` -> ``(img_path, frame_u8 [h,w,3], targets [k,6] | None, radar_box [r,5] | None, (points [n,4]
    float64, (w, h)))``; ``collate_fn`` -> ``(paths, StagedImages, targets [q,6], radar_boxes [r,5], StagedRadarMaps)``."""

    def __init__(self, mode, illumination, img_size=416, augment=False, multiscale=True, test_list=0,
                 dataset_folder="../data/our_dataset"):
        self.mode = mode
        self.illumination = illumination
        self.img_size = img_size
        self.map_size = int(self.img_size / 16)
        self.augment = augment
        self.multiscale = multiscale
        self.test_list = _SCENES[test_list:test_list + 1]
        self.train_list = _SCENES[:test_list] + _SCENES[test_list + 1:]
        self.max_objects = 100
        self.min_size = self.img_size - 3 * 32
        self.max_size = self.img_size + 3 * 32
        self.batch_count = 0
        self.chosen_classes = list(range(12))
        self.get_paths(dataset_folder)

    def get_paths(self, dataset_folder):
        """``dataset.txt`` lines look like ``<light><scene>-<...>-<appendix>``; the scene digit decides train / test
        (5-fold by ``test_list``), the light letter must be in ``illumination`` (datasets.py:156-194)."""
        split = {"train": dict(img=[], label=[], box=[], point=[]), "test": dict(img=[], label=[], box=[], point=[])}
        with open(f"{dataset_folder}/dataset.txt", "r") as fh:
            lines = [x.strip() for x in fh.read().split("\n") if x and not x.startswith("#")]
        for line in lines:
            head = line.split("-")[0]
            light, scene = head[0], head[1]
            _appendix = line.split("-")[2]  # the reference indexes it too: malformed lines must fail the same way
            if light not in self.illumination:
                continue
            for which, scenes in (("train", self.train_list), ("test", self.test_list)):
                if scene in scenes:
                    split[which]["img"].append(os.path.join(f"{dataset_folder}/image", line + ".jpg"))
                    split[which]["label"].append(os.path.join(f"{dataset_folder}/label", line + ".txt"))
                    split[which]["box"].append(os.path.join(f"{dataset_folder}/radar_box", line + ".pkl"))
                    split[which]["point"].append(os.path.join(f"{dataset_folder}/radar_point", line + ".pkl"))
        self.paths = split

    def __len__(self):
        return len(self.paths[self.mode]["img"])

    def __getitem__(self, idx):
        from PIL import Image

        sel = self.paths[self.mode]
        img_path, label_path, box_path, point_path = (sel[k][idx] for k in ("img", "label", "box", "point"))
        frame = torch.from_numpy(np.array(Image.open(img_path).convert("RGB"), dtype=np.uint8))  # [h,w,3]
        h, w = frame.shape[0], frame.shape[1]
        pad = _pad_amounts(h, w)
        padded_h, padded_w = h + pad[2] + pad[3], w + pad[0] + pad[1]

        targets = self._load_targets(label_path, (h, w), pad, (padded_h, padded_w))
        radar_box_output = self._load_radar_boxes(box_path, pad, padded_h)

        with open(point_path, "rb") as handle:
            points = np.asarray(pickle.load(handle), dtype=np.float64).reshape(-1, 4)  # (u, v, depth, velocity) rows
        return img_path, frame, targets, radar_box_output, (points, (w, h))

    @staticmethod
    def _load_targets(label_path, hw, pad, padded_hw):
        """``[k,6]`` target rows of one frame (``letterbox_labels``), or ``None`` without a label file."""
        if not os.path.exists(label_path):
            return None
        return letterbox_labels(read_label_rows(label_path), hw, pad, padded_hw)

    @staticmethod
    def _load_radar_boxes(box_path, pad, padded_side):
        """Radar proposals: pickled ``[r,4]`` xyxy pixels of the unpadded frame -> ``[r',5]`` rows ``(0, x1,y1,x2,y2)`` in
        [0,1] of the padded square (clamped, empty boxes dropped), or ``None`` (datasets.py:250-264)."""
        with open(box_path, "rb") as handle:
            rb = torch.from_numpy(pickle.load(handle))
        if len(rb) == 0:
            return None
        shift = torch.tensor([pad[0], pad[2], pad[1], pad[3]], dtype=rb.dtype)
        rb += shift  # in place on the unpickled array, like the reference's column-wise +=
        rb = torch.clamp(rb / padded_side, 0, 1)
        rb = rb[(rb[:, 0] < rb[:, 2]) & (rb[:, 1] < rb[:, 3])]
        if len(rb) == 0:
            return None
        out = torch.zeros((len(rb), 5))
        out[:, 1:] = rb
        return out

    def collate_fn(self, batch):
        paths, frames, targets, radar_boxes, radar_points = list(zip(*batch))
        for i, boxes in enumerate(targets):
            if boxes is not None:
                boxes[:, 0] = i
        for i, boxes in enumerate(radar_boxes):
            if boxes is not None:
                boxes[:, 0] = i
        targets = [b for b in targets if b is not None]
        targets = torch.cat(targets, 0) if len(targets) > 0 else torch.empty(0, 6)
        radar_boxes = [b for b in radar_boxes if b is not None]
        radar_boxes = torch.cat(radar_boxes, 0) if len(radar_boxes) > 0 else torch.empty(0, 5)
        if self.multiscale and self.batch_count % 10 == 0:  # a new input size every tenth batch (datasets.py:312-314)
            self.img_size = random.choice(range(self.min_size, self.max_size + 1, 32))
            self.map_size = int(self.img_size / 16)
        imgs = StagedImages(frames, self.img_size)
        radar_maps = StagedRadarMaps([p for p, _ in radar_points], [s for _, s in radar_points], self.map_size)
        self.batch_count += 1
        return paths, imgs, targets, radar_boxes, radar_maps
