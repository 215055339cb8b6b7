#!/usr/bin/env python3
"""Frames of a stream of scenes are independent: how much throughput do two frames in flight (two HIP streams, one
persistent workspace each) buy over one?  python tools/two_streams.py [--batch 1] [--steps 100]"""
import argparse
import copy
import os
import sys
import time

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mssvt_amd import config  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--streams", type=int, default=2)
    ap.add_argument("--kinds", default="cumask,priority,pooled", help="FramePipeline stream kinds to compare (MSSVT_PIPE_STREAMS)")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    cfg = config.load_yaml(config.DEFAULT_CFG)
    net = config.build_backbone_from_cfg(cfg).to(dev).eval()
    nets = [net] + [copy.deepcopy(net) for _ in range(a.streams - 1)]
    streams = [torch.cuda.Stream() for _ in range(a.streams)]
    frames = [bench.make_inputs(160000, a.batch, 0, dev, frame=f) for f in range(4)]

    def run(n_streams, steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            f = frames[i % len(frames)]
            k = i % n_streams
            with torch.cuda.stream(streams[k]), torch.no_grad():
                nets[k](dict(voxel_features=f[3], voxel_coords=f[2], batch_size=a.batch))
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps

    for n in range(1, a.streams + 1):
        run(n, 20)
        t = min(run(n, a.steps) for _ in range(3))
        print("%d stream(s), one network copy each: %.3f ms per step, %.0f frames/s" % (n, t * 1e3, a.batch / t))
    from mssvt_amd.pipeline import FramePipeline
    side = torch.cuda.Stream()

    def run_pipe(pipe, steps, mode):
        """mode: default = pipe(bd) from the default stream; side = the caller works on a non-default stream (an event orders
        the frame behind it); ready = inputs_ready=True; consume_default / consume_side = every frame's output is read by a
        small kernel on the default / a side stream `depth` submissions later (get(): wait_stream + record_stream)."""
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        held = []
        ctx = torch.cuda.stream(side) if mode in ("side", "consume_side") else torch.cuda.stream(torch.cuda.default_stream())
        with ctx:
            for i in range(steps):
                f = frames[i % len(frames)]
                bd = dict(voxel_features=f[3], voxel_coords=f[2], batch_size=a.batch)
                held.append(pipe(bd, inputs_ready=True) if mode == "ready" else pipe(bd))
                if mode.startswith("consume") and len(held) > pipe.depth:
                    out = held.pop(0).get()
                    out["encoded_spconv_tensor"].features.sum()
        pipe.synchronize()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps

    for kind in a.kinds.split(","):
        os.environ["MSSVT_PIPE_STREAMS"] = kind
        for n in ([1, a.streams] if kind == "cumask" else [a.streams]):
            pipe = FramePipeline(net, depth=n)
            for mode in ("default", "side", "ready", "consume_default", "consume_side"):
                run_pipe(pipe, 20, mode)
                t = min(run_pipe(pipe, a.steps, mode) for _ in range(3))
                print("FramePipeline depth %d (%s streams, %s): %.3f ms per step, %.0f frames/s" %
                      (n, pipe.stream_kind, mode, t * 1e3, a.batch / t))
            pipe.close()


if __name__ == "__main__" and "--cu-mask" not in sys.argv:
    main()


def cu_masked_streams(n, cus=256, full=False):
    """n HIP streams, each confined to its own 1/n of the CUs (hipExtStreamCreateWithCUMask), as torch ExternalStreams."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    out = []
    words = (cus + 31) // 32
    per = cus // n
    for k in range(n):
        mask = (ctypes.c_uint32 * words)()
        # CUs are numbered across shader engines / XCDs: interleave so that every XCD contributes to every partition
        for cu in range(cus):
            if full or cu % n == k:
                mask[cu // 32] |= 1 << (cu % 32)
        s = ctypes.c_void_p()
        rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), ctypes.c_uint32(words), mask)
        assert rc == 0, rc
        out.append(torch.cuda.ExternalStream(s.value))
    return out


def main_cu():
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    cfg = config.load_yaml(config.DEFAULT_CFG)
    net = config.build_backbone_from_cfg(cfg).to(dev).eval()
    frames = [bench.make_inputs(160000, 1, 0, dev, frame=f) for f in range(4)]
    for n, masked in ((1, False), (2, "full"), (3, "full"), (4, "full"), (4, True), (6, "full"), (2, True)):
        streams = cu_masked_streams(n, full=masked == "full") if masked else [torch.cuda.Stream() for _ in range(n)]

        def run(steps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                f = frames[i % len(frames)]
                with torch.cuda.stream(streams[i % n]), torch.no_grad():
                    net(dict(voxel_features=f[3], voxel_coords=f[2], batch_size=1))
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / steps
        run(20)
        t = min(run(100) for _ in range(3))
        print("%d stream(s), one network, %s: %.3f ms per step, %.0f frames/s" %
              (n, "own hardware queue, all CUs each" if masked == "full" else "1/%d of the CUs each" % n if masked else "all CUs each", t * 1e3, 1 / t))


if __name__ == "__main__" and "--cu-mask" in sys.argv:
    main_cu()
