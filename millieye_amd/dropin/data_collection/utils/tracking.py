import os, sys  # noqa: E401
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import _bootstrap  # noqa: F401,E402
from millieye_amd.radar_proposals import (radar_dbscan, associate_clusters, KalmanClusterTracker, Tracker,  # noqa: F401,E402
                                          LinearKalmanFilter as KalmanFilter)
