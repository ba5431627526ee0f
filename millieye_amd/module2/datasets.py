"""Stage-2 input producer: the behaviour of ``ListDataset`` (``module2_mixed/utils/datasets.py:75-166``) with the batch
assembled on the GPU.

What a caller of the reference class can rely on is kept: the constructor arguments, the file convention (a list file of
image paths; the label of ``.../images/x.png`` is ``.../labels/x.txt``), ``(path, image, targets)`` items, a ``collate_fn``
that numbers the targets by sample and draws a new input size every tenth batch when ``multiscale``, and the consumption
of the python / numpy random streams (one ``np.random.random()`` per augmented item, one ``random.choice`` per resize), so
a seeded run sees the same flips, sizes and shuffles (``tests/golden/m2_listdataset.npz``).

The work is split differently.  The host only decodes (``decode_rgb_u8``) and does the label geometry, which is the one
helper shared with the stage-3 producer (``utils.datasets.letterbox_labels``); an item carries the raw uint8 frame and the
flip decision, and ``collate_fn`` hands back a ``StagedImages`` whose ``.to(device)`` uploads the bytes and runs
``me_image_pad_resize_flip_u8_f32`` (ToTensor + pad_to_square + flip + nearest resize in one pass).  There is no CPU
implementation; the restatement the tests compare with is ``oracle/datasets_ref.py``.
"""
import os
import random

import numpy as np
import torch
from torch.utils.data import Dataset

from ..utils.datasets import StagedImages, _pad_amounts, decode_rgb_u8, letterbox_labels, read_label_rows

__all__ = ["ListDataset"]

_IMAGE_SUFFIXES = (".png", ".jpg")
_SIZE_STEP, _SIZE_SPAN = 32, 3   # multiscale: img_size +- 3 strides of 32, redrawn every _RESIZE_EVERY batches
_RESIZE_EVERY = 10


def _label_file_of(image_line):
    """``images`` -> ``labels`` and the image suffix -> ``.txt``, on the raw line of the list file (trailing newline
    included, as the reference keeps it: callers strip at use)."""
    out = image_line.replace("images", "labels")
    for suffix in _IMAGE_SUFFIXES:
        out = out.replace(suffix, ".txt")
    return out


class ListDataset(Dataset):
    def __init__(self, list_path, img_size=416, augment=True, multiscale=True, normalized_labels=True):
        with open(list_path, "r") as fh:
            self.img_files = fh.readlines()
        self.label_files = [_label_file_of(line) for line in self.img_files]
        self.img_size = img_size
        self.augment, self.multiscale, self.normalized_labels = augment, multiscale, normalized_labels
        self.min_size = img_size - _SIZE_SPAN * _SIZE_STEP
        self.max_size = img_size + _SIZE_SPAN * _SIZE_STEP
        self.max_objects = 100
        self.batch_count = 0

    def __len__(self):
        return len(self.img_files)

    def _targets_of(self, slot, h, w):
        """``[k,6]`` rows of item ``slot`` relative to the padded square, or ``None`` when it has no label file."""
        label_path = self.label_files[slot].rstrip()
        if not os.path.exists(label_path):
            return None
        side = max(h, w)
        return letterbox_labels(read_label_rows(label_path), (h, w) if self.normalized_labels else (1, 1),
                                _pad_amounts(h, w), (side, side))

    def __getitem__(self, index):
        """-> ``(img_path, (frame_u8 [h,w,3], flip), targets [k,6] | None)``"""
        slot = index % len(self.img_files)
        img_path = self.img_files[slot].rstrip()
        frame = decode_rgb_u8(img_path)
        targets = self._targets_of(slot, frame.shape[0], frame.shape[1])
        flip = bool(self.augment) and np.random.random() < 0.5
        if flip:
            targets[:, 2] = 1 - targets[:, 2]  # (an unlabelled image fails here in the reference too)
        return img_path, (frame, flip), targets

    def collate_fn(self, batch):
        paths = tuple(item[0] for item in batch)
        frames = [item[1][0] for item in batch]
        flips = [item[1][1] for item in batch]
        labelled = [item[2] for item in batch if item[2] is not None]
        for sample, rows in enumerate(labelled):   # numbered among the labelled items, as the reference does
            rows[:, 0] = sample
        targets = torch.cat(labelled, 0)
        if self.multiscale and self.batch_count % _RESIZE_EVERY == 0:
            self.img_size = random.choice(range(self.min_size, self.max_size + 1, _SIZE_STEP))
        self.batch_count += 1
        return paths, StagedImages(frames, self.img_size, flips=flips), targets
