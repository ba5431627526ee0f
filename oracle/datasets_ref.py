"""ORACLE (test infrastructure only - never imported by the product): CPU restatement of the input producer,
``module3_our_dataset/utils/datasets.py``, with the stock numpy / torch CPU calls the reference itself makes.
Pinned by ``tests/golden/dataset_small.npz`` (outputs of the real reference ``MyDataset`` on the committed
mini-dataset ``tests/golden/dataset_small/``; ``transforms.ToTensor`` is torchvision's and absent here - its two
branches (uint8 HWC -> float CHW / 255; float ndarray -> CHW as is) are restated in ``to_tensor``)."""
import os
import pickle

import numpy as np
import torch
import torch.nn.functional as F


def to_tensor(pic):
    """torchvision.transforms.ToTensor for the two inputs datasets.py feeds it (:203 PIL RGB, :267 float ndarray HWC)."""
    arr = np.asarray(pic)
    t = torch.from_numpy(np.ascontiguousarray(arr.transpose(2, 0, 1)))
    return t.float().div(255) if arr.dtype == np.uint8 else t


def pad_to_square(img, pad_value):
    """datasets.py:16-27."""
    _, h, w = img.shape
    diff = abs(h - w)
    pad1, pad2 = diff // 2, diff - diff // 2
    pad = (0, 0, pad1, pad2) if h <= w else (pad1, pad2, 0, 0)
    return F.pad(img, pad, "constant", value=pad_value), pad


def resize(image, size, mode="nearest"):
    """datasets.py:30-32."""
    return F.interpolate(image.unsqueeze(0), size=size, mode=mode).squeeze(0)


def plot_radar_heatmap(points, img_size, radar_maps_size=32):
    """datasets.py:59-106; ``points`` [4,n] rows (u, v, depth, velocity); returns float64 [bin_h, bin_w, 3]."""
    scale = max(img_size) / radar_maps_size
    bin_w, bin_h = round(img_size[0] / scale), round(img_size[1] / scale)
    kw = dict(x=points[0, :], y=points[1, :], bins=[bin_w, bin_h], range=[[0, img_size[0]], [0, img_size[1]]])
    h0 = np.histogram2d(**kw)[0].T
    h1 = np.histogram2d(weights=points[2, :], **kw)[0].T / (h0 + 1e-6)
    h1 = np.where(h1 < 1, 100, h1)
    h2 = np.absolute(np.histogram2d(weights=points[3, :], **kw)[0].T / (h0 + 1e-6))
    maps = np.stack((h0, h1, h2), axis=-1)
    for i, (lo, hi) in enumerate(((0, 5), (12, 0), (0, 4))):
        maps[..., i] = np.clip((maps[..., i] - lo) / (hi - lo), 0, 1)
    return maps


def load_item(img_path, label_path, box_path, point_path):
    """``MyDataset.__getitem__`` (datasets.py:196-279) -> (img [3,P,P], targets | None, radar_box | None, radar_map)."""
    from PIL import Image

    img = to_tensor(Image.open(img_path).convert("RGB"))
    _, h, w = img.shape
    img, pad = pad_to_square(img, 0)
    _, padded_h, padded_w = img.shape
    targets = None
    if os.path.exists(label_path):
        boxes = torch.from_numpy(np.loadtxt(label_path).reshape(-1, 5))
        x1 = (boxes[:, 1] - boxes[:, 3] / 2) * w + pad[0]
        y1 = (boxes[:, 2] - boxes[:, 4] / 2) * h + pad[2]
        x2 = (boxes[:, 1] + boxes[:, 3] / 2) * w + pad[1]
        y2 = (boxes[:, 2] + boxes[:, 4] / 2) * h + pad[3]
        boxes[:, 1] = ((x1 + x2) / 2) / padded_w
        boxes[:, 2] = ((y1 + y2) / 2) / padded_h
        boxes[:, 3] *= w / padded_w
        boxes[:, 4] *= h / padded_h
        targets = torch.zeros((len(boxes), 6))
        targets[:, 1:] = boxes
    with open(box_path, "rb") as fh:
        rb = torch.from_numpy(pickle.load(fh))
    rb_out = None
    if len(rb) > 0:
        rb[:, 0] += pad[0]
        rb[:, 2] += pad[1]
        rb[:, 1] += pad[2]
        rb[:, 3] += pad[3]
        rb = torch.clamp(rb / padded_h, 0, 1)
        rb = rb[torch.logical_and(rb[:, 0] < rb[:, 2], rb[:, 1] < rb[:, 3])]
        if len(rb) > 0:
            rb_out = torch.zeros((len(rb), 5))
            rb_out[:, 1:] = rb
    with open(point_path, "rb") as fh:
        points = pickle.load(fh)
    radar_map = to_tensor(plot_radar_heatmap(points.transpose(), (w, h))).float()
    radar_map, _ = pad_to_square(radar_map, 0)
    return img, targets, rb_out, radar_map


def collate(items, img_size, map_size):
    """``collate_fn`` (datasets.py:282-322) without the multiscale draw: items = list of load_item() tuples."""
    imgs, targets, radar_boxes, radar_maps = list(zip(*items))
    for i, b in enumerate(targets):
        if b is not None:
            b[:, 0] = i
    for i, b in enumerate(radar_boxes):
        if b is not None:
            b[:, 0] = i
    targets = [b for b in targets if b is not None]
    targets = torch.cat(targets, 0) if targets else torch.empty(0, 6)
    radar_boxes = [b for b in radar_boxes if b is not None]
    radar_boxes = torch.cat(radar_boxes, 0) if radar_boxes else torch.empty(0, 5)
    imgs = torch.stack([resize(img, img_size) for img in imgs])
    radar_maps = torch.stack([F.interpolate(m.unsqueeze(0), map_size, mode="bilinear", align_corners=True).squeeze(0)
                              for m in radar_maps])
    return imgs, targets, radar_boxes, radar_maps


# ---------------------------------------------------------------------------------------------------------------------
# stage 2: ListDataset of module2_mixed/utils/datasets.py:75-166 (pinned by tests/golden/m2_listdataset.npz: outputs of the
# real class on generated PNGs, tests/golden/make_golden.py --module2-loops)
# ---------------------------------------------------------------------------------------------------------------------
def list_item(img_path, label_path, flip):
    """``ListDataset.__getitem__`` (:93-148) with the augmentation draw made by the caller: ToTensor(convert('RGB')),
    pad_to_square, label arithmetic in float64 then float32 targets, horisontal_flip (utils/augmentations.py:6-9) on the
    PADDED image.  Returns (img [3,P,P], targets [k,6] | None)."""
    from PIL import Image
    img = to_tensor(np.asarray(Image.open(img_path).convert("RGB")))
    _, h, w = img.shape
    img, pad = pad_to_square(img, 0)
    _, padded_h, padded_w = img.shape
    targets = None
    if os.path.exists(label_path):
        boxes = torch.from_numpy(np.loadtxt(label_path).reshape(-1, 5))
        x1 = w * (boxes[:, 1] - boxes[:, 3] / 2) + pad[0]
        y1 = h * (boxes[:, 2] - boxes[:, 4] / 2) + pad[2]
        x2 = w * (boxes[:, 1] + boxes[:, 3] / 2) + pad[1]
        y2 = h * (boxes[:, 2] + boxes[:, 4] / 2) + pad[3]
        boxes[:, 1] = ((x1 + x2) / 2) / padded_w
        boxes[:, 2] = ((y1 + y2) / 2) / padded_h
        boxes[:, 3] *= w / padded_w
        boxes[:, 4] *= h / padded_h
        targets = torch.zeros((len(boxes), 6))
        targets[:, 1:] = boxes
    if flip:
        img = torch.flip(img, [-1])
        targets[:, 2] = 1 - targets[:, 2]
    return img, targets


def list_collate(items, img_size):
    """``ListDataset.collate_fn`` (:150-163) without the multiscale draw: the sample index counts the items that HAVE
    targets (an unlabelled frame shifts the following indices - kept as in the reference)."""
    imgs, targets = list(zip(*items))
    targets = [b for b in targets if b is not None]
    for i, b in enumerate(targets):
        b[:, 0] = i
    targets = torch.cat(targets, 0)
    return torch.stack([resize(img, img_size) for img in imgs]), targets
