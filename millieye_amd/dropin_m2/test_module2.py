import os, sys  # noqa: E401
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _bootstrap  # noqa: F401,E402
from millieye_amd.module2.test_module2 import *  # noqa: F401,F403,E402
from millieye_amd.module2.test_module2 import evaluate  # noqa: F401,E402
