"""millieye_amd - MI355X (gfx950 / CDNA4) native implementation of milliEye's dense
detection + fusion hot path: ``Darknet.forward`` -> NMS -> ``Network.forward``.

Layout:
  csrc/        hand-written HIP kernels + the C-ABI (``include/millieye_hip.h``)
  hip.py       ctypes binding of that C-ABI (the "FFI stub" a maintainer would add)
  engine.py    host-side planner: darknet cfg graph -> fused kernel launch list
  yolov3/, utils/, my_models.py, test_fusion.py
               host-side mirror of the reference's Python interface for this path
  dropin/      top-level module names the reference scripts import (yolov3.models, ...)

PyTorch is used for device memory, streams and torch.distributed only.
"""
__version__ = "0.1.0"
