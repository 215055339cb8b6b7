"""bench.py -- frames/s of the MsSVT backbone forward on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--points 160000] [--batch 1]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one backbone forward (voxel_coords, voxel_features -> encoded SparseTensor,
eval mode, fp32) over one batch of synthetic Waymo-shaped scenes that are already resident
in HBM.  Workload at N=1: BASELINE.json configs[1] -- one 160k-point scene, the full
mssvt.yaml backbone ("W": 4 Blocks + 1 CompressBlock, C=128, windows [3,3,5]/[7,7,7]).
Multi-GPU: scenes are sharded by rank (no data-path collective; weak scaling), timing is the
max over ranks between barriers.

Prints ONE JSON line on rank 0 with the driver's contract fields plus
  "roofline":     the dominant kernel (k_ffn_up, fp32 MFMA bound): algorithmic FLOP per launch / its
                  measured average duration (HIP events on the launching stream) vs the 157.3 TFLOP/s
                  dense fp32 matrix-core peak, HBM traffic from the committed PMC passes; the next two
                  kernels beside it (mssvt_amd/fused.py: roofline);
  "cpu_baseline": the CPU oracle (a port of the reference's CUDA semantics; the reference
                  has no CPU path) timed on this box's host on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8 TB/s (6.3 TB/s achievable)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--points", type=int, default=160000, help="LiDAR points per scene")
    ap.add_argument("--batch", type=int, default=1, help="scenes per GPU per step")
    ap.add_argument("--impl", default=None, choices=[None, "fused", "ops"])
    ap.add_argument("--cfg", default=None, help="backbone yaml (default: mssvt_amd/cfgs/mssvt.yaml = BASELINE configs[1])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    return ap.parse_args()


def make_inputs(points, batch, rank, device):
    from mssvt_amd import synthetic
    from mssvt_amd.dist import scene_seeds
    pts = synthetic.make_batch_points(points, batch, seed0=scene_seeds(rank, batch)[0])
    vc, _, _ = synthetic.voxelize_numpy(pts)
    g = torch.Generator().manual_seed(1000 + rank)
    feats = torch.randn(vc.shape[0], 128, generator=g)
    return vc, feats.numpy(), torch.from_numpy(vc).to(device), feats.to(device)


def event_time_ms(fn, iters, warm=3):
    """Average duration of fn() (which enqueues work on torch's current HIP stream)."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def cpu_baseline(net, vc_np, feats_np, batch):
    """Oracle on the host: Block 0 + the CompressBlock of the SAME frame(s), scaled to a whole
    forward as 4 x Block + Compress (the four Blocks do identical work)."""
    from mssvt_amd import config, synthetic
    from oracle import block_ref
    nthreads = torch.get_num_threads()
    torch.set_num_threads(1)  # the C oracle is scalar; keep the torch-CPU parts scalar too
    try:
        cfg = config.load_yaml(config.DEFAULT_CFG)
        params = [dict(p) for p in cfg.MODEL.BACKBONE_3D.PARAMS]
        sd = {k: v.detach().cpu().numpy() for k, v in net.state_dict().items()}
        sp = block_ref.SparseState(feats_np, vc_np, synthetic.GRID_SIZE, synthetic.VOXEL_SIZE,
                                   synthetic.POINT_CLOUD_RANGE, batch, 400000)
        p0, pc = params[0], params[-1]
        t0 = time.perf_counter()
        sp = block_ref.block_forward(sd, "backbone.0.", sp, p0["window_size"], p0["num_heads"],
                                     p0["max_num_win1"], p0["max_num_win2"], p0["cbs_pattern"],
                                     p0["key_num_sample"], p0["use_feature_interpolation"])
        t1 = time.perf_counter()
        block_ref.compress_forward(sd, "backbone.%d." % (len(params) - 1), sp, pc["window_size"],
                                   pc["num_heads"], pc["max_num_win1"])
        t2 = time.perf_counter()
    finally:
        torch.set_num_threads(nthreads)
    n_blocks = sum(p["name"].endswith("TransformerBlock") for p in params)
    est = n_blocks * (t1 - t0) + (t2 - t1)
    return {"value": batch / est, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "CPU oracle (C + torch-CPU port of the reference CUDA semantics; the reference has "
                      "no CPU path) on the same %d frame(s): Block 0 (%.1f s) + CompressBlock (%.1f s), "
                      "scaled as %d x Block + Compress" % (batch, t1 - t0, t2 - t1, n_blocks),
            "host_cpus": os.cpu_count()}


def main():
    args = parse()
    from mssvt_amd import dist as mdist
    rank, world, local_rank = mdist.env_rank_world()
    assert torch.cuda.is_available(), "bench.py needs an MI355X (the hot path has no CPU fallback)"
    # MSSVT_BENCH_ONE_DEVICE=1 (debugging the N > 1 code path on a single-GPU box): every rank on cuda:0, gloo
    one_dev = os.environ.get("MSSVT_BENCH_ONE_DEVICE", "0") == "1"
    dev = torch.device("cuda", 0 if one_dev else local_rank)
    torch.cuda.set_device(dev)
    dist = mdist.init("gloo" if one_dev else "nccl", dev)

    from mssvt_amd import config, roofline
    torch.manual_seed(0)
    cfg = config.load_yaml(args.cfg) if args.cfg else None
    net = config.build_backbone_from_cfg(cfg).to(dev).eval()
    if args.impl:
        net.set_impl(args.impl)
    impl = net.backbone[0].impl
    vc_np, feats_np, vc, feats = make_inputs(args.points, args.batch, rank, dev)

    def step():
        with torch.no_grad():
            return net(dict(voxel_features=feats, voxel_coords=vc, batch_size=args.batch))

    for _ in range(args.warmup):
        step()
    elapsed, out = mdist.timed_steps(step, args.steps, dist, dev)
    # live roofline: the same K steps once more on rank 0 with HIP events around every launch of the
    # dominant kernel (kept out of the timed region above: the event markers cost ~3 % of the frame rate)
    live = None
    if rank == 0 and impl == "fused" and not args.no_roofline and not args.cfg:
        from mssvt_amd import fused
        fused.FFN_TIMER = []
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        live = fused.ffn_timer_summary(fused.FFN_TIMER)
        fused.FFN_TIMER = None

    res = None
    if rank == 0:
        n_out = int(out["encoded_spconv_tensor"].features.shape[0])
        res = {
            "metric": "frames/sec (MsSVT backbone forward, synthetic Waymo-shaped scenes)",
            "value": world * args.batch * args.steps / elapsed,
            "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("BASELINE configs[1]: %d-point Waymo-shaped scene x batch %d per GPU, full "
                                    "mssvt.yaml backbone (4 Blocks [3,3,5]/[7,7,7] cbs 1,0,1,0 + CompressBlock "
                                    "[1,1,32], C=128, heads [4,4], HASH_SIZE 400000), fp32" % (args.points, args.batch))
                       if not args.cfg else "%s: %d-point scene x batch %d per GPU, fp32"
                       % (os.path.basename(args.cfg), args.points, args.batch),
                       "impl": impl, "voxels_per_gpu": int(vc.shape[0]), "output_voxels": n_out,
                       "parallelism": "scenes sharded over %d GPU(s), no data-path collective" % world},
        }
        if not args.no_roofline and not args.cfg:
            res["roofline"] = roofline.measure(net, vc, feats, args.batch, event_time_ms, HBM_PEAK_GBS, live=live)
        if not args.no_cpu_baseline and world == 1 and not args.cfg:
            res["cpu_baseline"] = cpu_baseline(net, vc_np, feats_np, args.batch)
        print(json.dumps(res), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
