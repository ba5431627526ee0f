import os, sys  # noqa: E401
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _bootstrap  # noqa: F401,E402
from millieye_amd.utils.utils import *  # noqa: F401,F403,E402
from millieye_amd.utils.utils import load_classes, weights_init_normal, xywh2xyxy, get_batch_statistics  # noqa: F401,E402
from millieye_amd.utils.utils import ap_per_class as _ap  # noqa: E402


def ap_per_class(tp, conf, pred_cls, target_cls):
    """The stage-2 tree's variant: the curve tuple carries the sorted confidences (module2_mixed/utils/utils.py:219-295)."""
    return _ap(tp, conf, pred_cls, target_cls, with_conf=True)
