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
        self.net = net
        self.device = torch.device(device if device is not None else next(net.parameters()).device)
        self.own_queues = True
        self.streams = _own_queue_streams(depth, self.device)
        if self.streams is None:
            self.own_queues = False
            self.streams = [torch.cuda.Stream(self.device) for _ in range(depth)]
        self.turn = 0

    @property
    def depth(self):
        return len(self.streams)

    def __call__(self, batch_dict, inputs_ready=False):
        """Enqueue one forward on the next stream and return its output dict (as `net(batch_dict)`, plus "stream": the
        stream its tensors are produced on -- a consumer on another stream waits for it: `cur.wait_stream(out["stream"])`).
        The inputs may come from the caller's current stream: the frame's stream first waits for what is queued there now,
        unless `inputs_ready` says they are complete already (resident frames: the event pair on the default stream costs
        4 % of the frame rate at two frames in flight, tools/two_streams.py)."""
        s = self.streams[self.turn % len(self.streams)]
        self.turn += 1
        if not inputs_ready:
            s.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(s), torch.no_grad():
            out = self.net(batch_dict)
        out["stream"] = s
        return out

    def synchronize(self):
        for s in self.streams:
            s.synchronize()
