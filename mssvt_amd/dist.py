"""Batch sharding across the GPUs of one node.

The MsSVT forward has no exchange step: hash tables are per sample and windows never cross
samples (ref: pcdet/ops/mssvt/mssvt_ops.py:16,36), so scenes are simply dealt to ranks
(one process per GPU, ``torch.distributed`` with the ``nccl`` backend = RCCL over xGMI on
ROCm; ``gloo`` in the CPU tests).  The only collectives are a start/stop barrier and a MAX
reduction of the elapsed time -- nothing on the data path.
"""
import os
import time

import torch


def env_rank_world():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init(backend=None, device=None):
    """Join the process group described by the torchrun environment (no-op for world size 1)."""
    import torch.distributed as dist
    rank, world, _ = env_rank_world()
    if world == 1:
        return None
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
    kw = {}
    if backend == "nccl" and device is not None:
        kw["device_id"] = device
    dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return dist


def scene_seeds(rank, scenes_per_gpu, seed0=0):
    """Scenes of rank r: seeds [seed0 + r*S, seed0 + (r+1)*S) -- disjoint and gap-free over ranks."""
    return [seed0 + rank * scenes_per_gpu + i for i in range(scenes_per_gpu)]


def timed_steps(step, steps, dist=None, device=None, sync=None):
    """barrier + sync, `steps` calls of step(), sync + barrier; returns MAX elapsed over ranks."""
    sync = sync or (torch.cuda.synchronize if torch.cuda.is_available() else (lambda: None))
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    out = None
    for _ in range(steps):
        out = step()
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed, out
