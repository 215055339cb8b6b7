"""Several frames of a stream of scenes in flight on one GPU.

Consecutive frames do not depend on each other, and one 160k-point frame does not fill an MI355X: its 27 launches are a
serial chain of latency- and issue-bound kernels with ramp-up, tail and a dependent-launch gap each.  `FramePipeline`
deals the frames round-robin to `depth` HIP streams; the whole-frame call keeps one frame object (persistent workspace,
pinned status words, events) per stream (mssvt_amd/frame.py), so frames in flight share nothing they write and the
hardware interleaves their kernels.  Measured on one MI355X (tools/two_streams.py, one scene per step): 1 583 -> 1 778
(two streams) -> 1 850 frames/s (three); at four scenes per step +1 % -- those launches fill the chip by themselves.
Every frame computes exactly what `net(batch_dict)` computes (tests/test_pipeline_gpu.py: bit-identical outputs).

The reference runs its frames one by one on the legacy default stream ("Threading / streams"); this is the
MI355X-side answer to the same loop (a detector's data loader hands over frame i + 1 while frame i is still running).
"""
import torch


class FramePipeline(object):
    def __init__(self, net, depth=2, device=None):
        assert depth >= 1
        self.net = net
        self.device = device if device is not None else next(net.parameters()).device
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
