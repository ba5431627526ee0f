"""Stage-2 input producer: ``ListDataset`` of ``module2_mixed/utils/datasets.py:75-166`` with the batch assembled on the GPU.

Same constructor, file layout (a list file of image paths; labels next to them with ``images`` -> ``labels`` and the
extension replaced by ``.txt``), label arithmetic, augmentation draw (``np.random.random() < 0.5`` per item: horizontal flip)
and ``collate_fn`` (sample index into column 0, ``random.choice`` of a new size every tenth batch when ``multiscale``).  As in
``millieye_amd/utils/datasets.py`` the host only decodes (PIL): ``__getitem__`` returns the decoded uint8 frame and the flip
flag instead of the padded float tensor, and ``collate_fn`` returns a ``StagedImages`` whose ``.to(device)`` uploads the bytes
and runs ``me_image_pad_resize_flip_u8_f32`` (ToTensor + pad_to_square + flip + nearest resize in one pass).  The python /
numpy / torch random streams are consumed exactly like the reference's, so a seeded run sees the same flips, sizes and
shuffles.  There is no CPU implementation (the restatement the tests use is ``oracle/datasets_ref.py``).
"""
import os
import random

import numpy as np
import torch
from torch.utils.data import Dataset

from ..utils.datasets import StagedImages, _pad_amounts

__all__ = ["ListDataset"]


class ListDataset(Dataset):
    def __init__(self, list_path, img_size=416, augment=True, multiscale=True, normalized_labels=True):
        with open(list_path, "r") as file:
            self.img_files = file.readlines()
        self.label_files = [path.replace("images", "labels").replace(".png", ".txt").replace(".jpg", ".txt")
                            for path in self.img_files]
        self.img_size = img_size
        self.max_objects = 100
        self.augment = augment
        self.multiscale = multiscale
        self.normalized_labels = normalized_labels
        self.min_size = self.img_size - 3 * 32
        self.max_size = self.img_size + 3 * 32
        self.batch_count = 0

    def __getitem__(self, index):
        """-> ``(img_path, (frame_u8 [h,w,3], flip), targets [k,6] | None)``"""
        from PIL import Image
        img_path = self.img_files[index % len(self.img_files)].rstrip()
        frame = torch.from_numpy(np.array(Image.open(img_path).convert("RGB"), dtype=np.uint8))
        h, w = frame.shape[0], frame.shape[1]
        h_factor, w_factor = (h, w) if self.normalized_labels else (1, 1)
        pad = _pad_amounts(h, w)
        padded_h = padded_w = max(h, w)

        label_path = self.label_files[index % len(self.img_files)].rstrip()
        targets = None
        if os.path.exists(label_path):
            boxes = torch.from_numpy(np.loadtxt(label_path).reshape(-1, 5))  # float64, like the reference
            x1 = w_factor * (boxes[:, 1] - boxes[:, 3] / 2)
            y1 = h_factor * (boxes[:, 2] - boxes[:, 4] / 2)
            x2 = w_factor * (boxes[:, 1] + boxes[:, 3] / 2)
            y2 = h_factor * (boxes[:, 2] + boxes[:, 4] / 2)
            x1 += pad[0]
            y1 += pad[2]
            x2 += pad[1]
            y2 += pad[3]
            boxes[:, 1] = ((x1 + x2) / 2) / padded_w
            boxes[:, 2] = ((y1 + y2) / 2) / padded_h
            boxes[:, 3] *= w_factor / padded_w
            boxes[:, 4] *= h_factor / padded_h
            targets = torch.zeros((len(boxes), 6))
            targets[:, 1:] = boxes

        flip = False
        if self.augment:
            if np.random.random() < 0.5:
                flip = True
                targets[:, 2] = 1 - targets[:, 2]  # (an image without a label file fails here in the reference too)
        return img_path, (frame, flip), targets

    def collate_fn(self, batch):
        paths, imgs, targets = list(zip(*batch))
        targets = [boxes for boxes in targets if boxes is not None]
        for i, boxes in enumerate(targets):
            boxes[:, 0] = i
        targets = torch.cat(targets, 0)
        if self.multiscale and self.batch_count % 10 == 0:
            self.img_size = random.choice(range(self.min_size, self.max_size + 1, 32))
        staged = StagedImages([f for f, _ in imgs], self.img_size, flips=[fl for _, fl in imgs])
        self.batch_count += 1
        return paths, staged, targets

    def __len__(self):
        return len(self.img_files)
