"""Make ``millieye_amd`` importable when a script is launched from this directory."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(1, _ROOT)
