"""N>1 path on CPU: two gloo ranks shard the scenes, time a stand-in step between barriers and
agree on the max-over-ranks time; no collective touches the data path."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import time
    from mssvt_amd import dist as mdist, synthetic
    d = mdist.init("gloo")
    seeds = mdist.scene_seeds(rank, 2)
    vc, _, _ = synthetic.voxelize_numpy(synthetic.make_batch_points(500, 2, seeds[0]))

    net = None
    if torch.cuda.is_available():  # the real (small) model where a device exists; the CPU box keeps the sleep below
        from mssvt_amd import config
        torch.manual_seed(0)
        net = config.build_backbone_from_cfg().cuda().eval()
        feats = torch.randn(vc.shape[0], 128).cuda()
        coords = torch.from_numpy(vc).cuda()

    def step():  # rank 1 is slower: the reported time must be ITS time on both ranks
        time.sleep(0.02 * (rank + 1))
        if net is not None:
            with torch.no_grad():
                return int(net(dict(voxel_features=feats, voxel_coords=coords, batch_size=2))["encoded_spconv_tensor"].features.shape[0])
        return vc.shape[0]

    elapsed, out = mdist.timed_steps(step, 3, d, torch.device("cpu"), sync=lambda: None)
    q.put((rank, seeds, elapsed, out))
    d.barrier()
    d.destroy_process_group()


def test_two_rank_sharding_and_timing():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, s0, t0, n0), (r1, s1, t1, n1) = res
    assert s0 == [0, 1] and s1 == [2, 3]  # disjoint, gap-free scene assignment
    assert abs(t0 - t1) < 1e-9 and t0 >= 3 * 0.04  # both ranks report the slowest rank's time
    assert n0 > 0 and n1 > 0 and n0 != n1  # different scenes on different ranks


def test_single_process_is_a_noop_group():
    from mssvt_amd import dist as mdist
    os.environ.pop("WORLD_SIZE", None)
    os.environ.pop("RANK", None)
    assert mdist.init("gloo") is None
    el, out = mdist.timed_steps(lambda: 7, 2, None, None, sync=lambda: None)
    assert out == 7 and el >= 0


def test_bench_launcher_starts_ranks_before_any_gpu_call_and_relays_failure():
    """`bench.py --gpus 2` without a torchrun environment starts the two ranks itself (torch.distributed.run);
    on this GPU-less box every rank stops at its "needs an MI355X" assert -- the parent must come back non-zero
    (and cannot have touched a GPU itself: there is none, and it would have failed before spawning)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    if torch.cuda.is_available():
        return  # on the GPU box tests/test_bench_gpu.py runs the launcher for real
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0
    assert "needs an MI355X" in r.stderr and r.stderr.count("needs an MI355X") >= 2  # both ranks ran
