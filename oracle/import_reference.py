"""ORACLE helper, BUILD CONTAINER ONLY: import the real reference from /root/reference.

Recipe of SURVEY.md Appendix E.  The reference is Python and importable here once the missing
third-party modules are stubbed: ``torchvision`` (absent from this image) gets stand-ins whose
three operators point at oracle/tv_ops.py (the C restatement - that boundary is therefore
parity-unpinned, everything else the reference computes itself), ``terminaltables`` /
``tensorboard`` get inert stubs.  /root/reference does not exist on the GPU box: nothing under
``tests -m gpu``, ``smoke()`` or ``bench.py`` may call this; only tests/golden/make_golden.py and
CPU tests that skip when the tree is absent do.

Never both module2 and module3 in one process (same top-level module names).
"""
import importlib
import os
import sys
import tempfile
import types

REFERENCE_ROOT = "/root/reference"


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "module3_our_dataset"))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_stubs():
    from oracle import tv_ops

    sys.dont_write_bytecode = True  # the reference tree is read-only for us: no __pycache__ droppings next to its sources

    if "torchvision" not in sys.modules:
        boxes = _stub("torchvision.ops.boxes", batched_nms=tv_ops.batched_nms, nms=tv_ops.nms)
        ops = _stub("torchvision.ops", ps_roi_align=tv_ops.ps_roi_align, roi_align=tv_ops.roi_align, boxes=boxes,
                    nms=tv_ops.nms, batched_nms=tv_ops.batched_nms)

        class _ToTensor:  # torchvision.transforms.ToTensor for the two inputs utils/datasets.py feeds it
            def __call__(self, pic):
                import numpy as np
                import torch

                arr = np.asarray(pic)
                if arr.ndim == 2:
                    arr = arr[:, :, None]
                t = torch.from_numpy(np.ascontiguousarray(arr.transpose(2, 0, 1)))
                return t.float().div(255) if arr.dtype == np.uint8 else t  # only byte images are scaled

        transforms = _stub("torchvision.transforms", ToTensor=_ToTensor)
        _stub("torchvision.datasets")
        _stub("torchvision.utils", make_grid=lambda t, **k: t)
        _stub("torchvision", ops=ops, transforms=transforms, datasets=sys.modules["torchvision.datasets"],
              utils=sys.modules["torchvision.utils"])
    if "terminaltables" not in sys.modules:
        class _AsciiTable:
            def __init__(self, rows):
                self.table = "\n".join(" | ".join(str(c) for c in r) for r in rows)

        _stub("terminaltables", AsciiTable=_AsciiTable)


def import_module3(chdir=True):
    """Returns a namespace with the reference's m3 modules: ``models`` (yolov3.models),
    ``utils`` (utils.utils), ``parse_config``, ``my_models``.  Changes CWD to a temp dir
    (the training tail appends to ./b.txt)."""
    if not reference_available():
        raise RuntimeError("/root/reference is not present (GPU box?) - fixtures are committed under tests/golden")
    install_stubs()
    root = os.path.join(REFERENCE_ROOT, "module3_our_dataset")
    for name in ("utils", "utils.utils", "utils.parse_config", "yolov3", "yolov3.models", "my_models"):
        mod = sys.modules.get(name)
        if mod is not None and not getattr(mod, "__file__", "").startswith(root):
            raise RuntimeError(f"top-level module name {name!r} already imported from {mod.__file__}")
    if root not in sys.path:
        sys.path.insert(0, root)
    if chdir:
        os.chdir(tempfile.mkdtemp(prefix="millieye_ref_"))
    ns = types.SimpleNamespace()
    ns.parse_config = importlib.import_module("utils.parse_config")
    ns.utils = importlib.import_module("utils.utils")
    ns.models = importlib.import_module("yolov3.models")
    ns.my_models = importlib.import_module("my_models")
    ns.root = root
    ns.load_datasets = lambda: importlib.import_module("utils.datasets")  # needs PIL + matplotlib (both present)
    return ns


def import_module2(chdir=True):
    """The stage-2 tree (``module2_mixed``): ``models``, ``utils``, ``parse_config``, ``my_models``.  Same top-level
    module names as module 3 - use a fresh interpreter (tests/golden/make_golden.py --module2)."""
    if not reference_available():
        raise RuntimeError("/root/reference is not present (GPU box?) - fixtures are committed under tests/golden")
    install_stubs()
    root = os.path.join(REFERENCE_ROOT, "module2_mixed")
    for name in ("utils", "utils.utils", "utils.parse_config", "yolov3", "yolov3.models", "my_models"):
        mod = sys.modules.get(name)
        if mod is not None and not getattr(mod, "__file__", "").startswith(root):
            raise RuntimeError(f"top-level module name {name!r} already imported from {mod.__file__}")
    if root not in sys.path:
        sys.path.insert(0, root)
    if chdir:
        os.chdir(tempfile.mkdtemp(prefix="millieye_ref_m2_"))
    ns = types.SimpleNamespace()
    ns.parse_config = importlib.import_module("utils.parse_config")
    ns.utils = importlib.import_module("utils.utils")
    ns.models = importlib.import_module("yolov3.models")
    ns.my_models = importlib.import_module("my_models")
    ns.root = root
    return ns
