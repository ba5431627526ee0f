"""Picklable pieces for the two-process pipeline tests (the producer runs under the ``spawn`` start method)."""
import time

import numpy as np


class SyntheticSource:
    """``n`` dark 480x640 frames with the seeded radar frames of tests/golden/make_golden.radar_points; ``delay`` seconds
    between frames (a slow camera)."""

    def __init__(self, n, delay=0.0):
        self.n, self.delay = n, delay

    def __call__(self):
        from millieye_amd import synth
        from tests.golden.make_golden import radar_points
        frame = (synth.uniform("demo/frame", (480, 640, 3)) * 255 * 0.1).astype(np.uint8)
        for f in range(self.n):
            if self.delay:
                time.sleep(self.delay)
            yield frame, [radar_points(f)]


def fake_infer_slow_first(payload, _state={"n": 0}):
    """Stand-in for the device half: the first call is slow (like the first GPU inference), the rest cost 5 ms."""
    time.sleep(0.5 if _state["n"] == 0 else 0.005)
    _state["n"] += 1
    return payload["radar_box"].clone(), dict(points=payload["points"])


class BrokenSource:
    def __call__(self):
        raise RuntimeError("camera unplugged")


class KilledSource:
    """Two frames, then the producer process dies the hard way (no exception, no END payload): what the OOM killer or a
    segfault in the decode code looks like from the consumer's side."""

    def __call__(self):
        import os
        from millieye_amd import synth
        from tests.golden.make_golden import radar_points
        frame = (synth.uniform("demo/frame", (480, 640, 3)) * 255 * 0.1).astype(np.uint8)
        for f in range(2):
            yield frame, [radar_points(f)]
        time.sleep(0.3)   # let the queue's feeder thread flush what was queued
        os._exit(7)


class TaggedGenerator:
    """A caller-supplied generator (picklable): proposals that no default RadarProposalGenerator would produce."""

    def __init__(self, calib_param):
        self.calib_param = calib_param

    def __call__(self, radar_frames):
        return np.array([[11.0, 22.0, 133.0, 144.0]], dtype=np.float32), np.zeros((0, 4))
