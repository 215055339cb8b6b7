"""Several frames of a stream of scenes in flight on one GPU.

Consecutive frames do not depend on each other, and one 160k-point frame does not fill an MI355X: its 27 launches are a
serial chain of latency- and issue-bound kernels with ramp-up, tail and a dependent-launch gap each.  `FramePipeline`
deals the frames round-robin to `depth` HIP streams; the whole-frame call keeps one frame object (persistent workspace,
pinned status words, events) per stream (mssvt_amd/frame.py), so frames in flight share nothing they write and the
hardware interleaves their kernels.  Measured on one MI355X (tools/two_streams.py, one scene per step, each stream on a
hardware queue of its own): 1 573 -> 1 779 (two frames in flight) -> 1 860 (three) -> 1 894 frames/s (four) -> 1 797 (six);
at four scenes per step +1 % -- those launches fill the chip by themselves.
Every frame computes exactly what `net(batch_dict)` computes (tests/test_pipeline_gpu.py: bit-identical outputs).

The reference runs its frames one by one on the legacy default stream ("Threading / streams"); this is the
MI355X-side answer to the same loop (a detector's data loader hands over frame i + 1 while frame i is still running).
"""
import ctypes
import os

import torch


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


class FramePipeline(object):
    def __init__(self, net, depth=4, device=None):
        assert depth >= 1
        self.net = net  # the backbone, or any callable batch_dict -> batch_dict (then pass `device`)
        self.device = torch.device(device if device is not None else next(net.parameters()).device)
        self.own_queues = True
        self.streams = None if os.environ.get("MSSVT_PIPE_POOLED") == "1" else _own_queue_streams(depth, self.device)
        if self.streams is None:
            self.own_queues = False
            self.streams = [torch.cuda.Stream(self.device) for _ in range(depth)]
        self.turn = 0
        self.pending = [None] * depth  # per stream: the deferred frame whose host wait has not happened yet

    @property
    def depth(self):
        return len(self.streams)

    def __call__(self, batch_dict, inputs_ready=False, defer=False):
        """Enqueue one forward on the next stream and return its output dict (as `net(batch_dict)`, plus "stream": the
        stream its tensors are produced on -- a consumer on another stream waits for it: `cur.wait_stream(out["stream"])`).
        The inputs may come from the caller's current stream: the frame's stream first waits for what is queued there now,
        unless `inputs_ready` says they are complete already (resident frames: the event pair on the default stream costs
        4 % of the frame rate at two frames in flight, tools/two_streams.py).
        `defer`: return a `PendingFrame` right behind the enqueue; its `get()` does the frame's one host wait (the output
        row count) and returns the dict.  The host then never stands still between two submissions: a frame's wait happens
        when its result is asked for, or when its stream comes round again `depth` submissions later."""
        k = self.turn % len(self.streams)
        s = self.streams[k]
        self.turn += 1
        if self.pending[k] is not None:  # this stream's frame object is about to be reused: its frame is `depth` old
            self.pending[k].get()
        if not inputs_ready:
            s.wait_stream(torch.cuda.current_stream(self.device))
        p = PendingFrame(self, s, batch_dict)
        with torch.cuda.stream(s), torch.no_grad():
            p._enqueue()
        if defer and p.out is None:
            self.pending[k] = p
            return p
        return p.get()

    def synchronize(self):
        """Every submitted frame finished (host side and device side)."""
        for p in self.pending:
            if p is not None:
                p.get()
        for s in self.streams:
            s.synchronize()

    def close(self):
        """Finish everything and give the per-stream frame objects (workspaces) back."""
        if not self.streams:
            return
        try:
            self.synchronize()
            from . import frame
            for s in self.streams:
                if hasattr(self.net, "backbone"):
                    frame.forget_stream(self.net, s.cuda_stream)
            # (the streams themselves are NOT destroyed: the framework's caching allocator keeps blocks and events tied to
            # every stream a tensor was allocated on -- destroying one under it crashed the process at exit; they live until
            # the process ends, as the framework's own pooled streams do)
        finally:
            self.streams = []
            self.pending = []

    # (no __del__: at interpreter exit the HIP runtime may be torn down before this object)


class PendingFrame(object):
    def __init__(self, pipe, stream, batch_dict):
        self.pipe, self.stream, self.batch_dict = pipe, stream, batch_dict
        self.out = self.pend = None

    def _enqueue(self):
        from . import frame, fused
        net, bd = self.pipe.net, self.batch_dict
        feats, coords = bd.get('voxel_features'), bd.get('voxel_coords')
        ok = (hasattr(net, "backbone") and getattr(net, "_unsorted_skip", 0) == 0 and getattr(net, "assume_sorted", False) and
              feats is not None and feats.is_cuda and any(getattr(b, 'impl', None) == 'fused' for b in net.backbone))
        self.pend = frame.forward(net, feats, coords, bd['batch_size'], defer=True) if ok else None
        if self.pend is None:  # not the whole-frame call's case (or any other callable, e.g. VFE -> backbone -> BEV): run it here
            self.out = net(bd)
            self.out["stream"] = self.stream

    def get(self):
        """The output dict of this frame (the host wait happens here, once)."""
        if self.out is None:
            from . import fused
            net = self.pipe.net
            with torch.cuda.stream(self.stream), torch.no_grad():
                try:
                    sp = self.pend.finish()
                    self.batch_dict.update({'encoded_spconv_tensor': sp, 'encoded_spconv_tensor_stride': 1})
                    self.out = self.batch_dict
                except fused.UnsortedVoxels:  # as MixedScaleSparseTransformer.forward: redo on the order-agnostic kernels
                    net._unsorted_skip = net._unsorted_backoff
                    self.out = net._forward(self.batch_dict, False)
            self.out["stream"] = self.stream
            self.pend = None
            k = self.pipe.streams.index(self.stream)
            if self.pipe.pending[k] is self:
                self.pipe.pending[k] = None
        return self.out
