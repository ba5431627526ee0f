"""The live demo's two-stage pipeline (``module3_our_dataset/run_mp.py:42-160,289-338``):

    producer process  (run_mp ``pre_process``)   frame + radar frames -> radar tracking / box proposals / staged inputs
         |  mp.Queue(maxsize=3), newest kept: when more than two items wait, the producer drops the oldest (:148-149)
         |  mp.Event: after the first frame the producer waits until the consumer finished its first - slow - inference (:146-147)
    consumer (this process, owns the GPU)         input kernels -> mode selection -> Network.forward -> NMS 0.3 -> rescale

Host work (~0.65 ms per frame: Kalman tracks, DBSCAN, proposal arithmetic) and device work overlap exactly like in the
reference; the device half is :meth:`millieye_amd.demo.FrameFuser.infer`, the host half :meth:`FrameFuser.prepare`.
Video decoding, the serial-port capture and the OpenCV window are not part of the path: frames come from a caller-supplied
*source factory* - a picklable callable returning an iterator of ``(frame_uint8_hwc, radar_frames)`` - which is called
inside the producer process (like the reference opens ``cv2.VideoCapture`` there).
"""
import multiprocessing as mp
import pickle
import queue as _queue
import time
import traceback

__all__ = ["FusionPipeline", "ProducerError", "QUEUE_SIZE"]


class ProducerError(RuntimeError):
    """The producer process failed (source iterator, radar tracking, staging); carries its traceback text."""

QUEUE_SIZE = 3      # run_mp.py:289
_END = "__end__"


def _producer(q, first_done, all_done, source_factory, prepare_factory, drop_oldest):
    """run_mp.py:42-160.  ``prepare_factory()`` builds the host half inside this process (tracker state lives here)."""
    dropped = 0
    error = None
    try:
        prepare = prepare_factory()
        for idx, (frame, radar_frames) in enumerate(source_factory()):
            payload = prepare(frame, radar_frames)
            payload["frame_idx"] = idx
            q.put(payload)
            if idx == 0:
                first_done.wait()          # inference of the first frame on the GPU is very slow (:146-147)
            if drop_oldest:
                try:
                    if q.qsize() > QUEUE_SIZE - 1:   # avoid blocking on a full queue: discard the oldest (:148-149)
                        q.get_nowait()
                        dropped += 1
                except (NotImplementedError, _queue.Empty):
                    pass
    except BaseException as exc:  # reported through the END payload: a crash must not look like the end of the stream
        error = f"{type(exc).__name__}: {exc}\n{traceback.format_exc()}"
    finally:
        q.put({"frame_idx": _END, "dropped": dropped, "error": error})
        # tensors travel as shared-memory handles served by THIS process: stay until the consumer has taken everything
        all_done.wait(timeout=120)


class _PrepareFactory:
    """Picklable recipe of the host half: builds a fresh ``FrameFuser`` (without a model) in the producer."""

    def __init__(self, calib_param, img_size, generator_kwargs, generator=None):
        self.calib_param, self.img_size, self.generator_kwargs = calib_param, img_size, generator_kwargs
        self.generator = generator   # a caller-supplied generator instance travels by pickle (checked by FusionPipeline)

    def __call__(self):
        from .demo import FrameFuser
        if self.generator is not None:
            return FrameFuser(None, self.calib_param, img_size=self.img_size, generator=self.generator).prepare
        return FrameFuser(None, self.calib_param, img_size=self.img_size, **self.generator_kwargs).prepare


class FusionPipeline:
    """``for rows, info in FusionPipeline(fuser, source_factory): ...`` - detections per frame, in frame order, frames the
    producer dropped are skipped (``info["frame_idx"]`` says which one this is).

    ``fuser``: a :class:`millieye_amd.demo.FrameFuser` (its model stays in this process).  A default
    ``RadarProposalGenerator`` is re-created from its parameters inside the producer; a generator the caller passed to
    ``FrameFuser(generator=...)`` is sent there by pickle as it is now, and refused (``TypeError``) when it cannot be
    pickled - never silently replaced.  ``infer``: override of the device half (tests).  ``drop_oldest=False`` turns the
    reference's newest-wins policy into back-pressure (every frame is processed).  A failure in the producer (source,
    tracking, staging) ends the iteration with :class:`ProducerError` carrying the producer's traceback."""

    def __init__(self, fuser, source_factory, infer=None, drop_oldest=True, prepare_factory=None, start_method="spawn"):
        self.fuser, self.source_factory, self.drop_oldest = fuser, source_factory, drop_oldest
        self.infer = infer or fuser.infer
        if prepare_factory is None:
            g = fuser.generator
            if getattr(fuser, "generator_is_default", False):
                prepare_factory = _PrepareFactory(g.calib_param, fuser.img_size, g.kwargs)
            else:
                try:
                    pickle.dumps(g)
                except Exception as exc:
                    raise TypeError("FusionPipeline: the FrameFuser was built with a custom generator= that cannot be "
                                    f"pickled into the producer process ({exc}); pass prepare_factory= instead") from exc
                prepare_factory = _PrepareFactory(getattr(g, "calib_param", None), fuser.img_size, {}, generator=g)
        self.prepare_factory = prepare_factory
        self.ctx = mp.get_context(start_method)      # run_mp.py:287: spawn
        self.stats = {}

    @staticmethod
    def _next_payload(q, proc, poll_s=0.5):
        """``q.get()`` that notices a producer killed hard (OOM killer, a segfault in the decode / radar code): such a
        process never posts its END payload, and a crash must not look like an endless wait."""
        import queue as _queue
        while True:
            try:
                return q.get(timeout=poll_s)
            except _queue.Empty:
                if not proc.is_alive():
                    try:   # whatever it managed to queue before dying (END included) still counts
                        return q.get(timeout=poll_s)
                    except _queue.Empty:
                        raise ProducerError(f"the producer process died without ending the stream (exit code "
                                            f"{proc.exitcode})") from None

    def __iter__(self):
        ctx = self.ctx
        q = ctx.Queue(maxsize=QUEUE_SIZE)
        first_done = ctx.Event()
        all_done = ctx.Event()
        proc = ctx.Process(target=_producer, args=(q, first_done, all_done, self.source_factory, self.prepare_factory,
                                                   self.drop_oldest), daemon=True)
        proc.start()
        t0, frames, dropped = time.perf_counter(), 0, None
        try:
            while True:
                payload = self._next_payload(q, proc)
                if payload["frame_idx"] == _END:
                    dropped = payload["dropped"]
                    if payload.get("error"):
                        raise ProducerError("the producer process failed:\n" + payload["error"])
                    break
                rows, info = self.infer(payload)
                first_done.set()               # run_mp.py:316
                info = dict(info, frame_idx=payload["frame_idx"])
                frames += 1
                yield rows, info
        finally:
            first_done.set()
            all_done.set()
            proc.join(timeout=10)
            if proc.is_alive():
                proc.terminate()               # run_mp.py:336
            self.stats = dict(dropped=dropped, frames=frames, seconds=time.perf_counter() - t0)
