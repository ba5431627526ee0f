"""Multi-GPU helpers: one process per GPU, frames are the independent units.

The reference is single-process, single-device (no DataParallel / distributed anywhere); scaling is new
in this build and deliberately minimal, shaped for xGMI:

* **Inference**: contiguous split of the batch by frame, weights replicated, NO collective; the caller
  concatenates the ``[m, 8]`` rows after re-offsetting their ``image_i`` column (:func:`merge_outputs`).
* **Stage-3 training**: every loss term is a *sum* over proposals (``reduction="sum"``, my_models.py:289,
  405-406,617,631), so data parallelism is a SUM all-reduce of the gradients - not a mean.  The trainable
  state is 100 153 parameters (0.40 MB fp32): a single flat bucket and ONE RCCL call per optimizer step
  (pure latency on xGMI; bucketing / overlap machinery would only add launches).  Negative sampling uses
  each rank's own ``random`` stream, BatchNorm statistics are per shard (like per-GPU BN in any DDP run);
  with eval-mode BatchNorm and no sub-sampling the summed shard gradients equal the 1-way step exactly
  (tests/test_parallel_cpu.py::test_dp_stage3_step_equals_one_way).  This holds for STAGE 3 ONLY: the detector's YOLO
  loss is a per-batch mean, see :class:`GradChunkReducer` (``average=True``).

* **Detector training** (row a6; the reference never trains the detector, ``train.py:98`` freezes it): 61.9 M parameters =
  247.8 MB of fp32 gradients next to a ~20 ms step.  One flat bucket behind the whole backward would put the entire
  exchange on the critical path, so :class:`GradChunkReducer` cuts the gradient stream where it is produced - the
  backward walks the layers in reverse, the deep 13 x 13 layers hold most of the bytes and come first - into chunks of
  ``chunk_bytes`` and all-reduces each chunk on its own HIP stream while the data-gradient convolutions of the shallower
  layers are still running.  xGMI is point-to-point, a ring step moves chunk / N bytes per link: 32 MB chunks keep every
  step in the bandwidth regime (4 MB per link at N = 8) while leaving eight exchanges to hide.

``torch.distributed`` (backend ``nccl`` = RCCL on ROCm, ``gloo`` in the CPU tests) is the transport.
"""
import os

import torch
import torch.distributed as dist

__all__ = ["shard_range", "shard_batch", "merge_outputs", "flatten_grads", "allreduce_gradients", "GradChunkReducer",
           "overlap_detector_allreduce", "begin_epoch"]


def begin_epoch(dataloader, epoch, device=None):
    """Top of every epoch of a data-parallel training loop (train.py, module2/train.py):

    * every rank must issue the same number of collectives - the optimizer step (and its all-reduce) is keyed on the local
      batch counter, so an uneven shard would hang the job: fail loudly instead;
    * ``DistributedSampler(shuffle=True)`` derives its permutation from ``seed + epoch``: without ``set_epoch`` every epoch
      replays epoch 0's order and every rank keeps the same fixed 1/N shard for the whole run (the single-process loop
      reshuffles through ``DataLoader(shuffle=True)``).

    No-op without a process group."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    on_gpu = dist.get_backend() == "nccl"
    lens = torch.tensor([len(dataloader), -len(dataloader)], dtype=torch.int64, device=device if on_gpu else "cpu")
    dist.all_reduce(lens, op=dist.ReduceOp.MAX)
    if int(lens[0]) != -int(lens[1]):
        raise RuntimeError(f"ranks see different batch counts per epoch (min {-int(lens[1])}, max {int(lens[0])}): "
                           "shard the dataset evenly (DistributedSampler pads to equal counts; main() does)")
    sampler = getattr(dataloader, "sampler", None)
    if hasattr(sampler, "set_epoch"):
        sampler.set_epoch(epoch)


def shard_range(n_frames, rank, world):
    """Contiguous frame range ``[lo, hi)`` of ``rank`` (first ``n % world`` ranks get one extra frame)."""
    base, extra = divmod(n_frames, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_batch(images, maps, radar_boxes, targets, rank, world):
    """Slice one global batch for ``rank``.  ``radar_boxes`` ``[r,5]`` and ``targets`` ``[q,6]`` carry the frame
    index in column 0: rows of other ranks are dropped and the index is re-based to the shard."""
    lo, hi = shard_range(images.shape[0], rank, world)

    def pick(rows):
        if rows is None:
            return None
        sel = (rows[:, 0] >= lo) & (rows[:, 0] < hi)
        out = rows[sel].clone()
        out[:, 0] -= lo
        return out

    return images[lo:hi], maps[lo:hi], pick(radar_boxes), pick(targets)


def merge_outputs(outputs, frames_per_rank):
    """Concatenate per-rank ``[m_i, 8]`` outputs (rank order) into global rows; ``image_i`` is re-offset by
    the number of frames of the preceding ranks.  Row order inside a rank is kept."""
    merged, offset = [], 0
    for out, frames in zip(outputs, frames_per_rank):
        out = out.clone()
        if out.numel():
            out[:, 0] += offset
        merged.append(out)
        offset += frames
    return torch.cat(merged, 0) if merged else torch.empty((0, 8))


def flatten_grads(params):
    """One flat fp32 bucket with every trainable parameter's gradient (zeros where ``.grad`` is None)."""
    params = [p for p in params if p.requires_grad]
    total = sum(p.numel() for p in params)
    if total == 0:
        return torch.zeros(0), params
    dev = params[0].device
    # one concatenation (a couple of batched-copy launches) instead of one slice assignment per parameter
    pieces = [p.grad.reshape(-1).to(torch.float32) if p.grad is not None else torch.zeros(p.numel(), device=dev)
              for p in params]
    bucket = torch.cat(pieces) if len(pieces) > 1 else pieces[0].clone()
    return bucket, params


_CHECK_EVERY = 64  # static_pattern: the reduced flags are still read (and checked) on every 64th step


class _PatternState:
    """Per-model bookkeeping of allreduce_gradients(static_pattern=True).  It hangs on the first Parameter object itself
    (``_me_dp_state``), so it lives and dies with the model - no id()-keyed global that a garbage-collected model's id
    could alias."""
    __slots__ = ("n", "union", "steps")

    def __init__(self, n):
        self.n, self.union, self.steps = n, None, 0


def _pattern_state(params):
    st = getattr(params[0], "_me_dp_state", None)
    if st is None or st.n != len(params):
        st = _PatternState(len(params))
        params[0]._me_dp_state = st
    return st


def allreduce_gradients(params, group=None, static_pattern=False):
    """SUM all-reduce of the gradients of ``params`` through ONE flat bucket = one collective per step.

    The bucket carries, behind the gradients, one flag per parameter ("this rank produced a gradient"), so the question
    "who has a gradient anywhere" (a rank whose shard yields no RoI returns ``None`` for everything; ``net1`` / ``net3`` /
    ``fusion_head`` never get one) rides in the same collective instead of a second MAX all-reduce.  Parameters whose
    gradient is ``None`` on every rank stay ``None``.  Reading the reduced flags back is a host sync (after the
    collective, local to the rank).  ``static_pattern=True`` (the training loops of this package): the set of parameters
    that can receive a gradient is a property of the model, and a rank produces either that whole set or - empty shard -
    nothing; then the flags are only read when this rank's own pattern is empty or differs from the union it has seen,
    i.e. in steady state the step issues one RCCL call and never waits on the device.  The promise is enforced: every
    ``_CHECK_EVERY``-th step (every step with ``MILLIEYE_DP_CHECK=1``) the reduced flags are read anyway and a rank that
    produced a gradient outside the assumed set raises instead of letting the replicas diverge.  Returns the gradient
    bytes."""
    params = [p for p in params if p.requires_grad]
    if not params:
        return 0
    distributed = dist.is_available() and dist.is_initialized()
    had = tuple(p.grad is not None for p in params)
    if not distributed:
        return sum(p.numel() for p in params) * 4  # single process: nothing to exchange, nothing to copy
    dev = next((p.grad.device for p in params if p.grad is not None), params[0].device)
    pieces = [p.grad.reshape(-1).to(torch.float32) if p.grad is not None else torch.zeros(p.numel(), device=dev)
              for p in params]
    pieces.append(torch.tensor([1.0 if h else 0.0 for h in had], device=dev))
    bucket = torch.cat(pieces)
    dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=group)
    n_grad = bucket.numel() - len(params)
    st = _pattern_state(params)
    st.steps += 1
    assumed = static_pattern and st.union == had and any(had)
    check = assumed and (st.steps % _CHECK_EVERY == 0 or os.environ.get("MILLIEYE_DP_CHECK") == "1")
    if assumed and not check:
        anywhere = had
    else:  # first step / empty shard / caller makes no promise / periodic check: look at the reduced flags (a host sync)
        anywhere = tuple(f > 0 for f in bucket[n_grad:].tolist())
        if check and anywhere != had:
            extra = [i for i, (a, h) in enumerate(zip(anywhere, had)) if a and not h]
            raise RuntimeError(f"allreduce_gradients(static_pattern=True): another rank produced gradients for parameters "
                               f"{extra} that this rank's static pattern does not contain; the replicas would diverge")
        st.union = anywhere if st.union is None else tuple(a or b for a, b in zip(anywhere, st.union))
    off = 0
    dst, src = [], []
    for p, h in zip(params, anywhere):
        n = p.numel()
        if h:
            g = bucket[off:off + n].view_as(p)
            if p.grad is None:
                p.grad = g.clone()
            else:
                dst.append(p.grad)
                src.append(g)
        off += n
    if dst:
        torch._foreach_copy_(dst, src)  # batched: one launch per ~100 tensors instead of one per parameter
    return n_grad * 4


class GradChunkReducer:
    """SUM all-reduce of a gradient stream in production order, one collective per ``chunk_bytes``, each on a dedicated
    communication stream that only waits for the streams its members were produced on.

    Scaling: the exchange SUMS the ranks' gradients.  Stage 3's loss terms are sums over proposals, so the summed shard
    gradients ARE the 1-way step.  The YOLO loss of the detector is a per-batch MEAN (yolov3/models.py:163-168 of the
    reference): summing N shard-mean gradients gives ~N times the gradient of the same global batch on one GPU.
    ``average=True`` divides every reduced chunk by the world size (the gradient of the mean over the global batch, up to
    the per-shard object counts in the means) - use it, or scale the learning rate, when a single-GPU schedule is kept.

    ``begin(dev)`` -> ``push(name, grad, stream)`` for every gradient as soon as its kernels are enqueued -> ``finish()``
    returns ``{name: reduced gradient}`` (views of the reduced flat chunks: no copy back) and makes the caller's stream
    wait for the exchanges.  The members of a chunk are concatenated on the communication stream (one batched copy), so
    the producers never wait for the pack.  ``chunks_last`` / ``bytes_last`` describe the last backward."""

    def __init__(self, chunk_bytes=32 << 20, group=None, average=False):
        self.chunk_bytes, self.group, self.average = int(chunk_bytes), group, bool(average)
        self._comm = {}
        self.chunks_last = self.bytes_last = 0
        self._cur, self._cur_bytes, self._streams, self._done, self._flats = [], 0, [], {}, []
        self._dev = None

    def _comm_stream(self):
        key = str(self._dev)
        if key not in self._comm:
            self._comm[key] = torch.cuda.Stream(device=self._dev) if self._dev.type == "cuda" else None
        return self._comm[key]

    def begin(self, dev):
        self._dev = torch.device(dev)
        self._cur, self._cur_bytes, self._streams, self._done, self._flats = [], 0, [], {}, []
        self.chunks_last = self.bytes_last = 0

    def push(self, name, grad, stream=None):
        """``grad`` was (or is being) produced on ``stream`` (default: the current one)."""
        if self._dev.type == "cuda":
            st = stream if stream is not None else torch.cuda.current_stream(self._dev)
            if all(st is not x for x in self._streams):
                self._streams.append(st)
        self._cur.append((name, grad))
        self._cur_bytes += grad.numel() * grad.element_size()
        if self._cur_bytes >= self.chunk_bytes:
            self.flush()

    def flush(self):
        if not self._cur:
            return
        members, self._cur = self._cur, []
        nbytes, self._cur_bytes = self._cur_bytes, 0
        comm = self._comm_stream()
        distributed = dist.is_available() and dist.is_initialized()

        def exchange():
            flat = torch.cat([g.reshape(-1) for _n, g in members]) if len(members) > 1 else members[0][1].reshape(-1).clone()
            if distributed:
                dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
                if self.average:
                    flat.div_(dist.get_world_size(self.group))
            off = 0
            for name, g in members:
                n = g.numel()
                self._done[name] = flat[off:off + n].view(g.shape)
                off += n
            return flat

        if comm is None:
            self._flats.append(exchange())
        else:
            for st in self._streams:  # everything enqueued so far on the producing streams
                comm.wait_stream(st)
            self._streams = []
            with torch.cuda.stream(comm):
                flat = exchange()
            for _n, g in members:
                g.record_stream(comm)
            self._flats.append(flat)
        self.chunks_last += 1
        self.bytes_last += nbytes

    def finish(self):
        self.flush()
        comm = self._comm_stream() if self._dev is not None else None
        if comm is not None:
            cur = torch.cuda.current_stream(self._dev)
            cur.wait_stream(comm)
            for flat in self._flats:
                flat.record_stream(cur)
        done, self._done, self._flats = self._done, {}, []
        return done


def overlap_detector_allreduce(model, chunk_bytes=32 << 20, group=None, average=False):
    """Attach a :class:`GradChunkReducer` to a ``Darknet``: ``loss.backward()`` of ``Darknet.forward(x, targets)`` then
    hands every layer's gradients to it as they are produced and returns already-reduced gradients (do NOT call
    ``allreduce_gradients`` on the detector parameters as well).  Without a process group there is nothing to exchange:
    nothing is attached and ``None`` is returned.  The gradients are SUMMED over the ranks; the detector's loss is a
    per-batch mean, so that is ~world_size times the 1-way gradient - pass ``average=True`` to divide by the world size
    (see :class:`GradChunkReducer`)."""
    if not (dist.is_available() and dist.is_initialized()):
        model.__dict__.pop("_grad_reducer", None)
        return None
    red = GradChunkReducer(chunk_bytes, group, average)
    model.__dict__["_grad_reducer"] = red
    return red
