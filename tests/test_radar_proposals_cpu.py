"""the demos' radar proposal generator (millieye_amd/radar_proposals.py).

Projection, FOV / velocity filter, DBSCAN clustering and frame-to-frame association are compared with the REAL
reference functions (tests/golden/radar_proposals_synth.npz, tests/golden/make_golden.py --radar) - bit for bit.
The Kalman tracker is parity-unpinned (filterpy is absent from the image): it is checked against the textbook recursion
written out here and for the track life-cycle rules of tracking.py:178-238."""
import os

import numpy as np
import pytest

from millieye_amd import radar_proposals as rp
from tests.golden.make_golden import RADAR_CALIB, RADAR_DEMO, RADAR_FRAMES, radar_points

GOLD = os.path.join(os.path.dirname(__file__), "golden", "radar_proposals_synth.npz")


def _clusters_of(g, k):
    c = np.zeros(len(g[k + "num_points"]), dtype=rp.CLUSTER_DTYPE)
    c["num_points"], c["center"], c["size"], c["avgV"] = g[k + "num_points"], g[k + "center"], g[k + "size"], g[k + "avgV"]
    return c


def test_projection_filter_clustering_association_match_reference():
    g = np.load(GOLD)
    calib = np.array(RADAR_CALIB)
    assert np.array_equal(calib, g["calib"])
    prev = None
    for f in range(RADAR_FRAMES):
        k = f"f{f}/"
        pts = radar_points(f)
        assert np.array_equal(pts, g[k + "points"]), "the seeded generator drifted from the fixture"
        uv, xyzv = rp.from_3d_to_2d(pts, calib)
        assert uv.dtype == np.int64 and np.array_equal(uv, g[k + "uv"]) and np.array_equal(xyzv, g[k + "xyzv"])
        keep = rp.fov_velocity_filter(uv, xyzv, RADAR_DEMO["max_depth"], RADAR_DEMO["min_velocity"])
        assert np.array_equal(keep, g[k + "keep"])
        clusters, labels = rp.radar_dbscan(xyzv[keep], rp.CLUSTER_DTYPE, RADAR_DEMO["dbscan_weights"], RADAR_DEMO["dbscan_eps"])
        assert np.array_equal(np.asarray(labels), g[k + "labels"])
        for field in ("num_points", "center", "size", "avgV"):
            assert np.array_equal(clusters[field], g[k + field]), (f, field)
        big = clusters[clusters["num_points"] >= RADAR_DEMO["num_pts_filter"]]
        assert len(big) == int(g[k + "n_big"]) >= 2
        for ci, c in enumerate(big):   # the corner projection box_proposals relies on
            half = c["size"].astype(float) * np.array([1.0, 1.0, 0.0]) / 2
            u, v = rp.projection_xyr_to_uv(np.stack([c["center"] + half, c["center"] - half], 1), calib)
            assert np.array_equal(u, g[k + f"corner_u{ci}"]) and np.array_equal(v, g[k + f"corner_v{ci}"])
        if prev is not None:
            uo, un, (mo, mn) = rp.associate_clusters(prev, big)
            assert np.array_equal(mo, g[k + "matched_old"]) and np.array_equal(mn, g[k + "matched_new"])
            assert list(uo) == list(g[k + "unmatched_old"]) and list(un) == list(g[k + "unmatched_new"])
        prev = big


def test_empty_inputs():
    calib = np.array(RADAR_CALIB)
    uv, xyzv = rp.from_3d_to_2d(np.zeros((4, 0)), calib)
    assert uv.shape == (0, 2) and xyzv.shape == (0, 4)
    clusters, labels = rp.radar_dbscan(xyzv, rp.CLUSTER_DTYPE, [2, 1, 3, 1])
    assert len(clusters) == 0 and labels == []
    uo, un, (mo, mn) = rp.associate_clusters(np.zeros(0, rp.CLUSTER_DTYPE), np.zeros(2, rp.CLUSTER_DTYPE))
    assert len(uo) == 0 and list(un) == [0, 1] and len(mo) == 0
    gen = rp.RadarProposalGenerator(calib)
    boxes, cloud = gen([])
    assert boxes.shape == (0, 4) and cloud.shape == (0, 4)


def test_kalman_filter_is_the_textbook_recursion():
    """LinearKalmanFilter (unpinned restatement of filterpy's predict / update) against the recursion written with explicit
    matrices, on the tracker's own 9-state / 7-measurement model."""
    rng = np.random.RandomState(0)
    c = np.zeros(1, rp.CLUSTER_DTYPE)[0]
    c["center"], c["size"], c["avgV"], c["num_points"] = (300.0, 200.0, 4.0), (30.0, 60.0, 0.5), 0.8, 9
    trk = rp.KalmanClusterTracker(c.copy(), 1 / 20, 4)
    F, H, Q, R = trk.kf.F.copy(), trk.kf.H.copy(), trk.kf.Q.copy(), trk.kf.R.copy()
    assert Q[0, 0] == 0.03 and abs(Q[8, 8] - 0.0015) < 1e-12 and trk.kf.P[0, 0] == 10 and trk.kf.P[3, 3] == 1000 and trk.kf.P[2, 2] == 1
    x, P = trk.kf.x.copy(), trk.kf.P.copy()
    for step in range(6):
        x, P = F @ x, F @ P @ F.T + Q
        trk.predict()
        assert np.allclose(trk.kf.x, x, atol=1e-12) and np.allclose(trk.kf.P, P, atol=1e-9)
        z = ((H @ x).ravel() + rng.randn(7)).astype(np.float32).astype(np.float64)  # what a float32 cluster record carries
        S = H @ P @ H.T + R
        K = P @ H.T @ np.linalg.inv(S)
        x = x + K @ (z.reshape(7, 1) - H @ x)
        P = (np.eye(9) - K @ H) @ P @ (np.eye(9) - K @ H).T + K @ R @ K.T
        m = np.zeros(1, rp.CLUSTER_DTYPE)[0]
        m["center"], m["avgV"], m["size"], m["num_points"] = z[:3], z[3], z[4:], 11
        trk.update(m)
        assert np.allclose(trk.kf.x, x, rtol=1e-10, atol=1e-9) and np.allclose(trk.kf.P, P, rtol=1e-9, atol=1e-9)
        assert np.allclose(trk.cluster["center"], x[:3, 0], rtol=1e-5) and trk.cluster["num_points"] == 11
    assert trk.hit_streak == 6 and trk.time_since_update == 0


def test_tracker_life_cycle_and_proposals():
    """Tracks are reported during the first min_hits frames, then only once confirmed; an unmatched track coasts for max_age
    frames and is dropped; the whole generator turns the seeded frames into plausible pixel boxes."""
    calib = np.array(RADAR_CALIB)
    rp.KalmanClusterTracker.count = 0
    gen = rp.RadarProposalGenerator(calib, min_hits=3, max_age=2)
    counts = []
    for f in range(RADAR_FRAMES):
        boxes, cloud = gen([radar_points(f)])
        counts.append(len(boxes))
        assert cloud.shape[1] == 4 and np.all(cloud[:, 2] < 10) and np.all(np.abs(cloud[:, 3]) >= 0.1)
        for x1, y1, x2, y2 in boxes:
            assert x2 > x1 and y2 > y1 and -200 < x1 < 840 and -200 < y1 < 680
    assert counts[0] >= 2 and counts[-1] >= 2, counts
    n_tracks = len(gen.tracker.trackers)
    for _ in range(2):                       # nothing detected: the tracks coast ...
        gen([np.zeros((4, 0))])
    assert len(gen.tracker.trackers) == n_tracks
    gen([np.zeros((4, 0))])                  # ... and die after max_age missed frames
    assert len(gen.tracker.trackers) == 0
    # hand-computed compensation: a 2 x 1 (u x v metres) cluster straight ahead at 5 m
    c = np.zeros(1, rp.CLUSTER_DTYPE)
    c["center"], c["size"], c["num_points"] = (0.07, 0.05, 5.0), (2.0, 1.0, 0.3), 9   # cancels the radar->camera translation
    box = rp.box_proposals(c, calib, max_size=20)[0]
    cc = c["center"][0].astype(float)   # the record stores float32: 0.07 / 0.05 are not exact
    u, v = rp.projection_xyr_to_uv(np.array([[cc[0] + 1.0, cc[0] - 1.0], [cc[1] + 0.5, cc[1] - 0.5], [5.0, 5.0]]), calib)
    w, h = u[0] - u[1], v[0] - v[1]
    cx, cy = (u[0] + u[1]) / 2, (v[0] + v[1]) / 2 + 0.16 * h
    assert np.allclose(box, [cx - 0.6 * w, cy - 0.7 * h, cx + 0.6 * w, cy + 0.7 * h], rtol=1e-12)
    assert len(rp.box_proposals(c, calib, max_size=1.5)) == 0   # largest extent 2.0 >= max_size: skipped


def test_demo_glue_on_the_host():
    """The two host-only pieces of millieye_amd/demo.py: run_mp's mode selection (auto = fusion below a mean brightness of
    0.08) and the proposal -> radar_box normalisation (padding of the short side, division by the padded side, clamp, empty
    boxes dropped; run_mp.py:119-135, 204-212)."""
    import torch
    from millieye_amd.demo import mode_selection, radar_boxes_for_network
    assert [mode_selection(m, None) for m in (0, 1, 2)] == [0, 1, 2] and mode_selection(9, None) is None
    assert mode_selection(3, torch.full((1, 3, 8, 8), 0.05)) == 0 and mode_selection(3, torch.full((1, 3, 8, 8), 0.5)) == 1
    assert mode_selection(3, torch.full((1, 3, 8, 8), 0.09)) == 1 and mode_selection(3, torch.full((1, 3, 8, 8), 0.09), 0.1) == 0
    # landscape frame 480 x 640: 80 rows of padding above and below, side 640
    rb = radar_boxes_for_network([[10, 20, 110, 220], [700, 10, 650, 50], [-50, -50, 30, 30]], (480, 640))
    assert rb.shape == (2, 5)
    assert torch.allclose(rb[0], torch.tensor([0, 10 / 640, 100 / 640, 110 / 640, 300 / 640]))
    assert torch.allclose(rb[1], torch.tensor([0, 0.0, 30 / 640, 30 / 640, 110 / 640]))   # clamped; the reversed box is gone
    # portrait frame 640 x 480: the padding goes left / right
    rb = radar_boxes_for_network([[0, 0, 480, 640]], (640, 480))
    assert torch.allclose(rb[0], torch.tensor([0, 80 / 640, 0.0, 560 / 640, 1.0]))
    assert radar_boxes_for_network([], (480, 640)).shape == (0, 5)
