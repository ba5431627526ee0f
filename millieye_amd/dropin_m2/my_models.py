import os, sys  # noqa: E401
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _bootstrap  # noqa: F401,E402
from millieye_amd.module2.my_models import *  # noqa: F401,F403,E402
from millieye_amd.module2.my_models import Network, define_yolo, init_yolo  # noqa: F401,E402
