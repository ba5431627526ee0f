"""CPU: the run_mp two-process hand-over (millieye_amd/pipeline.py; reference run_mp.py:42-160,289-338) with a stand-in for
the device half: frame order, the first-frame rendezvous, the newest-wins queue policy, clean termination, and that the
producer's host half equals running ``FrameFuser.prepare`` in-process on the same frames."""
import time

import numpy as np
import torch

from millieye_amd import radar_proposals as rp
from millieye_amd.demo import FrameFuser
from millieye_amd.pipeline import QUEUE_SIZE, FusionPipeline
from tests.golden.make_golden import RADAR_CALIB
from tests.pipeline_helpers import BrokenSource, SyntheticSource, TaggedGenerator, fake_infer_slow_first


def _fuser():
    rp.KalmanClusterTracker.count = 0
    return FrameFuser(None, RADAR_CALIB, model_mode=0, min_hits=2)


def test_every_frame_in_order_with_back_pressure():
    n = 8
    got = list(FusionPipeline(_fuser(), SyntheticSource(n), infer=lambda p: (p["radar_box"].clone(), dict(points=p["points"])),
                              drop_oldest=False))
    assert [info["frame_idx"] for _r, info in got] == list(range(n))
    # the producer's host half == the same frames through FrameFuser.prepare in this process (tracker state included)
    local = _fuser()
    for (rows, info), (frame, radar) in zip(got, SyntheticSource(n)()):
        want = local.prepare(frame, radar)
        assert torch.equal(rows, want["radar_box"]) and info["points"] == want["points"]
    assert any(len(r) for r, _ in got), "tracks must confirm and produce proposals within 8 frames"


def test_newest_wins_when_the_consumer_is_slow():
    """run_mp.py:146-149: the producer waits for the consumer's first inference, then never blocks - with more than two items
    waiting it drops the oldest.  A slow consumer therefore sees a strictly increasing subsequence that ends at the last
    frame, and never finds more than QUEUE_SIZE items waiting."""
    n = 40
    pipe = FusionPipeline(_fuser(), SyntheticSource(n), infer=slow_infer, drop_oldest=True)
    t0 = time.perf_counter()
    seen = [info["frame_idx"] for _r, info in pipe]
    assert seen[0] == 0 and seen == sorted(set(seen)) and seen[-1] == n - 1
    assert len(seen) < n and pipe.stats["dropped"] == n - len(seen), (len(seen), pipe.stats)
    assert time.perf_counter() - t0 < 30


def slow_infer(payload):
    time.sleep(0.05)
    return payload["radar_box"].clone(), dict(points=payload["points"])


def test_first_frame_rendezvous():
    """The producer does not run ahead while the first (slow) inference is in flight: the second payload is prepared only
    after the consumer signalled, so with a 0.5 s first inference the second frame cannot arrive before 0.5 s."""
    pipe = FusionPipeline(_fuser(), SyntheticSource(4), infer=fake_infer_slow_first, drop_oldest=False)
    stamps = []
    t0 = time.perf_counter()
    for _rows, info in pipe:
        stamps.append((info["frame_idx"], time.perf_counter() - t0))
    assert [i for i, _ in stamps] == [0, 1, 2, 3]
    assert stamps[1][1] - stamps[0][1] < 0.45, "after the rendezvous the pipeline flows"
    assert QUEUE_SIZE == 3


def test_producer_failure_is_reported_not_mistaken_for_the_end():
    """an exception in the source / prepare half must not look like exhaustion: the frames delivered before it
    arrive, then the iteration raises ProducerError with the producer's traceback."""
    import pytest
    from millieye_amd.pipeline import ProducerError
    pipe = FusionPipeline(_fuser(), BrokenSource(), infer=slow_infer, drop_oldest=False)
    got = []
    with pytest.raises(ProducerError) as err:
        for rows, info in pipe:
            got.append(info["frame_idx"])
    assert "BrokenSource" in str(err.value) or "Error" in str(err.value)
    assert got == list(range(len(got))) and pipe.stats["frames"] == len(got)


def test_producer_killed_hard_raises_instead_of_hanging():
    """a producer that dies without posting END (os._exit here; the OOM killer in production) ends the iteration
    with ProducerError naming the exit code - the consumer's q.get() must not wait forever."""
    import pytest
    from millieye_amd.pipeline import ProducerError
    from tests.pipeline_helpers import KilledSource
    pipe = FusionPipeline(_fuser(), KilledSource(), infer=slow_infer, drop_oldest=False)
    got = []
    with pytest.raises(ProducerError) as err:
        for rows, info in pipe:
            got.append(info["frame_idx"])
    assert "exit code 7" in str(err.value), str(err.value)
    assert got == [0, 1]


def test_custom_generator_reaches_the_producer_or_is_refused():
    import pytest
    fuser = FrameFuser(None, RADAR_CALIB, model_mode=0, generator=TaggedGenerator(RADAR_CALIB))
    got = list(FusionPipeline(fuser, SyntheticSource(3), infer=lambda p: (p["radar_box"].clone(), dict(p=p["proposals"])),
                              drop_oldest=False))
    assert len(got) == 3
    for rows, info in got:
        assert np.allclose(np.asarray(info["p"]), [[11.0, 22.0, 133.0, 144.0]]) and rows.shape == (1, 5)
    unpicklable = TaggedGenerator(RADAR_CALIB)
    unpicklable.fn = lambda x: x
    with pytest.raises(TypeError):
        FusionPipeline(FrameFuser(None, RADAR_CALIB, generator=unpicklable), SyntheticSource(1))
