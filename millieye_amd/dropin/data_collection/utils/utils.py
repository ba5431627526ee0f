import os, sys  # noqa: E401
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import _bootstrap  # noqa: F401,E402
from millieye_amd.radar_proposals import projection_xyr_to_uv, from_3d_to_2d  # noqa: F401,E402
from .tracking import *  # noqa: F401,F403,E402  (the reference's utils.py re-exports tracking.py the same way)
