"""Several frames of a stream of scenes in flight on one GPU.

Consecutive frames do not depend on each other, and one 160k-point frame does not fill an MI355X: its launches are a
serial chain of latency- and issue-bound kernels with ramp-up, tail and a dependent-launch gap each.  `FramePipeline`
deals the frames round-robin to `depth` HIP streams; the whole-frame call keeps one frame object (persistent workspace,
pinned status words, events) per stream (mssvt_amd/frame.py), so frames in flight share nothing they write and the
hardware interleaves their kernels.  Every frame computes exactly what `net(batch_dict)` computes
(tests/test_pipeline_gpu.py: bit-identical outputs, at the benchmark size too).

    pipe = FramePipeline(net)                 # depth from the batch size (auto_depth)
    with torch.cuda.stream(side):             # (consumers off the legacy default stream: "Which streams" below)
        for bd in loader:
            frame = pipe(bd)                  # returns right behind the enqueue: nothing waits here
            ...
            out = frame.get()                 # the frame's one host wait; the CURRENT stream now waits for that frame
            head(out["encoded_spconv_tensor"])  # and the outputs are recorded for it: use them like any tensor

The reference runs its frames one by one on the legacy default stream ("Threading / streams"); this is the
MI355X-side answer to the same loop (a detector's data loader hands over frame i + 1 while frame i is still running).

Which streams (`STREAMS`, or MSSVT_PIPE_STREAMS=cumask|priority|pooled).  Measured on one MI355X, one 160k-point scene per
step, four frames in flight (one frame at a time: 1 555 - 1 575 frames/s; profiles/r06_*_two_streams.txt, r06_*_pipe_trace.txt):

                                             bench.py loop, 20 / 50 steps      + a consumer kernel per frame on the DEFAULT / a side stream
    cumask    (own hardware queue each)          1 680 / 1 784                      592 / 1 646     (100-step loop: 1 878)
    priority  (framework pool, priority -1)      1 527 / 1 609                    1 609 / 1 608     (100-step loop: 1 865 in a process
    pooled    (framework pool, priority 0)             -                          1 746 / 1 749      that had made cumask streams first)

* The runtime multiplexes ordinary streams onto a few hardware queues (GPU_MAX_HW_QUEUES = 4 for normal priority) and two
  streams on one queue serialise.  Which streams collide is the luck of the process's stream-creation history: in a fresh
  process the four high-priority framework streams share TWO queues -- the completion times of tools/pipe_trace.py come in
  pairs 1.3 ms apart, i.e. no gain over one frame at a time -- while the same streams made after four CU-mask streams ran at
  the own-queue rate.  Only a stream created with a CU mask (`cumask`: hipExtStreamCreateWithCUMask, every CU enabled, nothing
  partitioned) is GIVEN a queue of its own: the default.
* Such a stream is a BLOCKING stream in the legacy sense (the call takes no flags): every operation on the NULL (default)
  stream -- an event record included -- waits for everything queued on it and holds back whatever follows.  Round 5's default
  call recorded an event on the default stream per frame (`wait_stream`) and ran 673 / 1 129 / 914 frames/s at depth 1 / 2 / 3
  for that reason (1.485 ms per frame at depth 1 against 0.641).  Now inputs that come from the default stream take NO event
  (the implicit ordering already covers them), and the default call runs at the deferred rate (1 552 / 1 850 at depth 1 / 4).
  What remains is the consumer side: kernels a caller launches on the DEFAULT stream join every frame in flight (592 frames/s).
  Run the stages around the backbone on a side stream (`with torch.cuda.stream(side):` -- 1 646), inside the pipeline
  (`FramePipeline(chain)`), or select `priority` / `pooled` streams (no legacy coupling, 1 6xx - 1 7xx either way); `get()`
  logs a warning once when it is called with the default stream current on `cumask` streams.
* A consumer waits for ITS frame (an event recorded right behind the frame), not for the frame's stream: `wait_stream`
  would also wait for the next frame already queued there, and the next submission would wait for the consumer -- the
  pipeline then runs at depth ~1.5 (measured: 995 frames/s).
"""
import ctypes
import logging
import os

import torch


STREAMS = "cumask"  # "cumask": a hardware queue of its own per stream (blocking w.r.t. the default stream); "priority"; "pooled"


def auto_depth(batch_size):
    """Frames in flight that pay at `batch_size` scenes per step (measured, DESIGN 5: one scene per step +15 - 20 % at four
    in flight, two scenes per step +3 % at two; from four scenes per step the launches fill the chip by themselves -- batch 4:
    1 875 / 1 881 / 1 856 frames/s at depth 1 / 2 / 4, batch 8: 1 894 / 1 931 / 1 909, profiles/r06_a_depth_by_batch.txt -- and more
    depth only adds workspaces)."""
    b = int(batch_size)
    return 4 if b < 2 else (2 if b < 4 else 1)


def _own_queue_streams(n, device):
    """n HIP streams with a hardware queue of their own each, as torch streams -- or None when the runtime cannot make them.

    The runtime multiplexes ordinary streams onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default), and two
    streams that land on the same queue serialise: with the framework's pooled streams three frames in flight ran
    anywhere between 1 633 and 1 858 frames/s and two frames in flight at the single-stream rate when the two shared a
    queue (tools/two_streams.py with GPU_MAX_HW_QUEUES = 2).  A stream created with a CU mask is given its own queue;
    the mask used here enables EVERY compute unit, so nothing is partitioned (quarter-of-the-chip masks measured the
    same: 1 884 against 1 894 frames/s at four frames in flight)."""
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        fn = hip.hipExtStreamCreateWithCUMask
    except (OSError, AttributeError):
        return None
    cus = int(torch.cuda.get_device_properties(device).multi_processor_count)
    words = (cus + 31) // 32
    mask = (ctypes.c_uint32 * words)()
    for cu in range(cus):
        mask[cu // 32] |= 1 << (cu % 32)
    out = []
    with torch.cuda.device(device):
        for _ in range(n):
            s = ctypes.c_void_p()
            if fn(ctypes.byref(s), ctypes.c_uint32(words), mask) != 0 or not s.value:
                return None
            out.append(torch.cuda.ExternalStream(s.value, device=device))
    return out


def _tensors_of(out):
    """The device tensors of an output dict that a consumer may read (what `record_stream` has to cover)."""
    seen = []
    for v in out.values():
        if isinstance(v, torch.Tensor):
            if v.is_cuda:
                seen.append(v)
        elif v is not None and hasattr(v, "features") and hasattr(v, "indices"):  # SparseTensor
            for name in ("features", "indices", "map_table", "v_bs_cnt"):
                t = getattr(v, name, None)
                if isinstance(t, torch.Tensor) and t.is_cuda:
                    seen.append(t)
    return seen


class FramePipeline(object):
    def __init__(self, net, depth=None, device=None, batch_size=1, pre=None, post=None):
        """`depth`: frames in flight; None = `auto_depth(batch_size)`.
        `pre` / `post`: stages either side of the backbone that run ON THE FRAME'S STREAM -- `pre(batch_dict) -> batch_dict`
        right before the backbone is enqueued (e.g. the DynamicVFE: points -> voxels), `post(batch_dict) -> batch_dict` right
        behind the frame's host wait (e.g. dense() / HeightCompression, which need the output row count).  Unlike a chain
        passed as `net`, this keeps the backbone's host wait DEFERRED: the host is held per frame only by what `pre` itself
        waits for (the voxelizer's voxel count), not until the backbone's level set-up has run."""
        if depth is None:
            depth = auto_depth(batch_size)
        assert depth >= 1
        self.net = net  # the backbone, or any callable batch_dict -> batch_dict (then pass `device`)
        self.pre, self.post = pre, post
        self.device = torch.device(device if device is not None else next(net.parameters()).device)
        kind = os.environ.get("MSSVT_PIPE_STREAMS", "pooled" if os.environ.get("MSSVT_PIPE_POOLED") == "1" else STREAMS)
        self.streams = _own_queue_streams(depth, self.device) if kind == "cumask" else None
        self.own_queues = self.streams is not None  # (blocking streams in the legacy sense: module docstring)
        if self.streams is None:
            prio = -1 if kind == "priority" else 0
            self.streams = [torch.cuda.Stream(self.device, priority=prio) for _ in range(depth)]
        self.stream_kind = "cumask" if self.own_queues else ("priority" if kind == "priority" else "pooled")
        self.turn = 0
        self.frame_events = True
        self.pending = [None] * depth  # per stream: the frame whose host wait has not happened yet

    @property
    def depth(self):
        return len(self.streams)

    def __call__(self, batch_dict, inputs_ready=None, defer=True):
        """Enqueue one forward on the next stream and return its `PendingFrame` right behind the enqueue; `get()` (or
        indexing it like the output dict) does the frame's one host wait -- the output row count -- and hands the outputs
        over to the caller's current stream.  The host never stands still between two submissions: a frame's wait happens
        when its result is asked for, or when its stream comes round again `depth` submissions later.

        The inputs may come from the caller's current stream: the frame's stream is ordered behind what is queued there
        now (`inputs_ready=None`), by an event -- or by nothing at all when that stream is the legacy default stream and the
        pipeline's streams are blocking streams (module docstring) -- unless `inputs_ready=True` says they are complete
        already.  The input tensors are recorded for the frame's stream (the caller may drop them right away).
        `defer=False`: return the output dict itself (= `pipe(bd).get()`)."""
        k = self.turn % len(self.streams)
        s = self.streams[k]
        self.turn += 1
        if self.pending[k] is not None:  # this stream's frame object is about to be reused: its frame is `depth` old
            self.pending[k]._finish()
        cur = torch.cuda.current_stream(self.device)
        if not inputs_ready and cur.cuda_stream != s.cuda_stream:
            if not (self.own_queues and cur.cuda_stream == 0):  # (implicit ordering, see above)
                s.wait_stream(cur)
        for v in batch_dict.values():  # allocated on the caller's stream, read on `s` for the whole frame
            if isinstance(v, torch.Tensor) and v.is_cuda and cur.cuda_stream != s.cuda_stream:
                v.record_stream(s)
        with torch.cuda.stream(s), torch.no_grad():
            if self.pre is not None:
                batch_dict = self.pre(batch_dict)
            p = PendingFrame(self, s, batch_dict)
            p._enqueue()
            if self.frame_events:  # what a consumer on another stream waits for: THIS frame, not whatever follows it on `s`
                p.done = torch.cuda.Event()
                p.done.record(s)
        if p.out is None:
            self.pending[k] = p
        return p if defer else p.get()

    @staticmethod
    def result(frame):
        """The output dict of what `__call__` returned (a `PendingFrame`, or already a dict)."""
        return frame.get() if isinstance(frame, PendingFrame) else frame

    def synchronize(self):
        """Every submitted frame finished (host side and device side)."""
        err = None
        for p in list(self.pending):
            if p is not None:
                try:
                    p._finish()
                except Exception as e:  # noqa: BLE001 -- finish the others, then report the first
                    err = err or e
        for s in self.streams:
            s.synchronize()
        if err is not None:
            raise err

    def close(self):
        """Finish everything and give the per-stream frame objects (workspaces) back."""
        if not self.streams:
            return
        try:
            self.synchronize()
        finally:
            from . import frame
            for s in self.streams:
                if hasattr(self.net, "backbone"):
                    frame.forget_stream(self.net, s.cuda_stream)
            # (the streams themselves are NOT destroyed: the framework's caching allocator keeps blocks and events tied to
            # every stream a tensor was allocated on -- destroying one under it crashed the process at exit; they live until
            # the process ends, as the framework's own pooled streams do)
            self.streams = []
            self.pending = []

    # (no __del__: at interpreter exit the HIP runtime may be torn down before this object)


class PendingFrame(object):
    """One submitted frame.  `get()` -> the output dict (as `net(batch_dict)`, plus "stream": the stream its tensors were
    produced on).  Also readable like that dict: `frame["encoded_spconv_tensor"]`."""

    def __init__(self, pipe, stream, batch_dict):
        self.pipe, self.stream, self.batch_dict = pipe, stream, batch_dict
        self.out = self.pend = self.error = self.done = None
        self._handed = set()

    def _enqueue(self):
        from . import frame
        net, bd = self.pipe.net, self.batch_dict
        feats, coords = bd.get('voxel_features'), bd.get('voxel_coords')
        ok = (hasattr(net, "backbone") and getattr(net, "_unsorted_skip", 0) == 0 and getattr(net, "assume_sorted", False) and
              feats is not None and feats.is_cuda and any(getattr(b, 'impl', None) == 'fused' for b in net.backbone))
        self.pend = frame.forward(net, feats, coords, bd['batch_size'], defer=True) if ok else None
        if self.pend is None:  # not the whole-frame call's case (or any other callable, e.g. VFE -> backbone -> BEV): run it here
            self.out = net(bd)
            if self.pipe.post is not None:
                self.out = self.pipe.post(self.out)
            self.out["stream"] = self.stream

    def _finish(self):
        """The frame's host wait, once; whatever it raises is raised again by every later call (and the pipeline's slot is
        free either way: one bad frame does not wedge its stream)."""
        if self.error is not None:
            raise self.error
        if self.out is None:
            from . import fused
            net = self.pipe.net
            try:
                with torch.cuda.stream(self.stream), torch.no_grad():
                    try:
                        sp = self.pend.finish()
                        self.batch_dict.update({'encoded_spconv_tensor': sp, 'encoded_spconv_tensor_stride': 1})
                        self.out = self.batch_dict
                    except fused.UnsortedVoxels:  # as MixedScaleSparseTransformer.forward: redo on the order-agnostic kernels
                        net._unsorted_skip = net._unsorted_backoff
                        self.out = net._forward(self.batch_dict, False)
                    if self.pipe.post is not None:  # enqueued behind the frame on its stream; what a consumer waits for moves with it
                        self.out = self.pipe.post(self.out)
                        if self.done is not None:
                            self.done = torch.cuda.Event()
                            self.done.record(self.stream)
                self.out["stream"] = self.stream
            except Exception as e:  # noqa: BLE001
                self.error = e
                raise
            finally:
                self.pend = None
                pend = self.pipe.pending
                for k in range(len(pend)):
                    if pend[k] is self:
                        pend[k] = None
        return self.out

    def get(self, sync=True):
        """The output dict.  `sync`: the caller's CURRENT stream waits for the frame's stream (device side; the host does
        not) and the outputs are recorded for it, so that they can be used -- and dropped -- there like any tensor; the
        framework's allocator would otherwise hand a dropped block back to the frame's stream while the consumer still
        reads it.  `sync=False`: the caller orders and records by hand (`cur.wait_stream(out["stream"])`,
        `t.record_stream(cur)`), or only wants host-side metadata."""
        out = self._finish()
        if sync:
            cur = torch.cuda.current_stream(self.pipe.device)
            if cur.cuda_stream != self.stream.cuda_stream and cur.cuda_stream not in self._handed:
                self._handed.add(cur.cuda_stream)
                if cur.cuda_stream == 0 and self.pipe.own_queues and not self.pipe.__dict__.get("_warned"):
                    self.pipe._warned = True
                    logging.getLogger("mssvt_amd.pipeline").warning(
                        "FramePipeline: the consumer runs on the legacy default stream, which joins every frame in flight on the "
                        "pipeline's (blocking) streams -- run it under torch.cuda.stream(side) or set MSSVT_PIPE_STREAMS=priority")
                if self.done is not None:
                    cur.wait_event(self.done)
                else:
                    cur.wait_stream(self.stream)
                for t in _tensors_of(out):
                    t.record_stream(cur)
        return out

    def __getitem__(self, key):
        return self.get()[key]

    def __contains__(self, key):
        return key in self.get()

    def keys(self):
        return self.get().keys()
