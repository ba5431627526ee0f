"""Programmatic darknet cfg emitters for the three networks the path is quoted on.

The reference ships cfg *files* (``module3_our_dataset/config/yolov3-tiny-12.cfg``,
``yolov3-tiny-coco.cfg``, ``yolov3.cfg``); this repo ships the *architecture* as
code and writes an equivalent cfg on demand, so ``Darknet(cfg_path)`` keeps the
reference's file-based constructor while nothing is copied.  The emitted text
parses (``utils.parse_config.parse_model_config``) to the same module list:
yolov3-tiny -> 24 modules, yolov3 (Darknet-53) -> 107 modules
(75 convolutional / 23 shortcut / 4 route / 2 upsample / 3 yolo).
"""
import os
import tempfile

__all__ = ["yolov3_cfg_text", "yolov3_tiny_cfg_text", "write_cfg", "KNOWN"]

_TINY_ANCHORS = "10,14,  23,27,  37,58,  81,82,  135,169,  344,319"
_V3_ANCHORS = "10,13,  16,30,  33,23,  30,61,  62,45,  59,119,  116,90,  156,198,  373,326"


class _Emitter:
    def __init__(self, size):
        self.out = [
            "[net]",
            "batch=1",
            "subdivisions=1",
            f"width={size}",
            f"height={size}",
            "channels=3",
            "momentum=0.9",
            "decay=0.0005",
            "",
        ]
        self.count = 0

    def _block(self, kind, pairs):
        self.out.append(f"# module {self.count}")
        self.out.append(f"[{kind}]")
        for k, v in pairs:
            self.out.append(f"{k}={v}")
        self.out.append("")
        self.count += 1

    def conv(self, filters, size, stride=1, bn=True, act="leaky"):
        pairs = []
        if bn:
            pairs.append(("batch_normalize", 1))
        pairs += [("filters", filters), ("size", size), ("stride", stride), ("pad", 1), ("activation", act)]
        self._block("convolutional", pairs)

    def maxpool(self, size, stride):
        self._block("maxpool", [("size", size), ("stride", stride)])

    def shortcut(self, frm=-3):
        self._block("shortcut", [("from", frm), ("activation", "linear")])

    def route(self, *layers):
        self._block("route", [("layers", ", ".join(str(l) for l in layers))])

    def upsample(self, stride=2):
        self._block("upsample", [("stride", stride)])

    def yolo(self, mask, anchors, classes, num):
        self._block(
            "yolo",
            [
                ("mask", ",".join(str(m) for m in mask)),
                ("anchors", anchors),
                ("classes", classes),
                ("num", num),
                ("jitter", ".3"),
                ("ignore_thresh", ".7"),
                ("truth_thresh", 1),
                ("random", 1),
            ],
        )

    def text(self):
        return "\n".join(self.out) + "\n"


def yolov3_tiny_cfg_text(classes=80, size=416):
    """yolov3-tiny: 24 modules; ``classes=12`` is the reference default (tiny-12)."""
    e = _Emitter(size)
    det = 3 * (classes + 5)
    for filters in (16, 32, 64, 128, 256):  # modules 0..9
        e.conv(filters, 3)
        e.maxpool(2, 2)
    e.conv(512, 3)  # 10
    e.maxpool(2, 1)  # 11 (zero-padded 2x2 stride-1 pool)
    e.conv(1024, 3)  # 12
    e.conv(256, 1)  # 13
    e.conv(512, 3)  # 14
    e.conv(det, 1, bn=False, act="linear")  # 15
    e.yolo((3, 4, 5), _TINY_ANCHORS, classes, 6)  # 16
    e.route(-4)  # 17
    e.conv(128, 1)  # 18
    e.upsample(2)  # 19
    e.route(-1, 8)  # 20
    e.conv(256, 3)  # 21
    e.conv(det, 1, bn=False, act="linear")  # 22
    e.yolo((1, 2, 3), _TINY_ANCHORS, classes, 6)  # 23
    assert e.count == 24
    return e.text()


def yolov3_cfg_text(classes=80, size=416):
    """YOLOv3 / Darknet-53: 107 modules."""
    e = _Emitter(size)
    det = 3 * (classes + 5)

    def residual_stage(width, repeats):
        for _ in range(repeats):
            e.conv(width // 2, 1)
            e.conv(width, 3)
            e.shortcut(-3)

    e.conv(32, 3)  # 0
    for width, repeats in ((64, 1), (128, 2), (256, 8), (512, 8), (1024, 4)):
        e.conv(width, 3, stride=2)
        residual_stage(width, repeats)
    assert e.count == 75

    def head(width):
        for _ in range(3):
            e.conv(width, 1)
            e.conv(width * 2, 3)

    head(512)  # 75..80
    e.conv(det, 1, bn=False, act="linear")  # 81
    e.yolo((6, 7, 8), _V3_ANCHORS, classes, 9)  # 82
    e.route(-4)  # 83
    e.conv(256, 1)  # 84
    e.upsample(2)  # 85
    e.route(-1, 61)  # 86
    head(256)  # 87..92
    e.conv(det, 1, bn=False, act="linear")  # 93
    e.yolo((3, 4, 5), _V3_ANCHORS, classes, 9)  # 94
    e.route(-4)  # 95
    e.conv(128, 1)  # 96
    e.upsample(2)  # 97
    e.route(-1, 36)  # 98
    head(128)  # 99..104
    e.conv(det, 1, bn=False, act="linear")  # 105
    e.yolo((0, 1, 2), _V3_ANCHORS, classes, 9)  # 106
    assert e.count == 107
    return e.text()


KNOWN = {
    "yolov3": lambda: yolov3_cfg_text(80),
    "yolov3-tiny": lambda: yolov3_tiny_cfg_text(80),
    "yolov3-tiny-coco": lambda: yolov3_tiny_cfg_text(80),
    "yolov3-tiny-12": lambda: yolov3_tiny_cfg_text(12),
}


def write_cfg(name, directory=None):
    """Write the named cfg (see ``KNOWN``) into ``directory`` (default: a per-user cache
    dir) and return its path, ready for ``Darknet(path)`` / ``define_yolo(path)``."""
    if name not in KNOWN:
        raise KeyError(f"unknown cfg {name!r}; known: {sorted(KNOWN)}")
    if directory is None:
        directory = os.path.join(tempfile.gettempdir(), f"millieye_amd_cfg_{os.getuid()}")
    os.makedirs(directory, exist_ok=True)
    path = os.path.join(directory, f"{name}.cfg")
    text = KNOWN[name]()
    tmp = f"{path}.{os.getpid()}.tmp"
    with open(tmp, "w") as fh:
        fh.write(text)
    os.replace(tmp, path)
    return path
