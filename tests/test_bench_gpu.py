"""bench.py end to end on the GPU box: the --gpus N launcher (this process starts the ranks itself), the DDP
training step and the configs[2] flags.  A 1-GPU box runs the N > 1 code path with every rank on cuda:0 over gloo
(MSSVT_BENCH_ONE_DEVICE=1); on the 8-GPU node the driver runs the same code over RCCL."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*flags, env=None):
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--points",
                        "20000", "--no-cpu-baseline", "--no-roofline"] + list(flags), capture_output=True, text=True,
                       timeout=420, env=e, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-800:], r.stderr[-1500:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-800:]  # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_gpus_2_starts_two_ranks():
    res = _bench("--gpus", "2", "--batch", "2", env={"MSSVT_BENCH_ONE_DEVICE": "1"})
    assert res["n_gpus"] == 2 and res["config"]["process_group"]["world_size"] == 2
    assert res["value"] > 0 and res["scaling"] == "weak" and res["dtype"] == "f32"
    # whole-job aggregate: both ranks' scenes over the max-over-ranks time
    assert abs(res["value"] - 2 * 2 * 3 / (res["ms_per_step"] * 3e-3)) < 1e-6 * res["value"]


def test_bench_ddp_training_step_two_ranks():
    res = _bench("--gpus", "2", "--train", env={"MSSVT_BENCH_ONE_DEVICE": "1"})
    assert res["n_gpus"] == 2 and "DDP" in res["config"]["parallelism"] and res["value"] > 0


def test_bench_configs2_flags():
    res = _bench("--batch", "2", "--attn-dtype", "bf16")
    assert res["dtype"] == "bf16" and res["config"]["attn_dtype"] == "bf16" and res["n_gpus"] == 1
    assert res["timing"]["p10_ms"] <= res["timing"]["median_ms"] <= res["timing"]["p90_ms"]


def test_bench_ffn_arith_flag_and_roofline_object():
    """--ffn-arith selects the FFN / CompressBlock arithmetic and the JSON says which one ran; the default line carries the
    roofline of the dominant kernel (k_ffn_ws, HBM bound) with the whole-frame object."""
    res = _bench("--ffn-arith", "f32")
    assert "v_mfma_f32_16x16x4_f32" in res["config"]["ffn_arith"]
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--points", "20000",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=420, env=e, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-800:], r.stderr[-1500:])
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert res["config"]["ffn_arith"].startswith("split16") and "two fp16 halves" in res["config"]["ffn_arith"]
    # the arithmetic that RAN, per block (the outcome of the fp16-range guards on these parameters): 4 Blocks + the CompressBlock
    per = res["config"]["arith_per_block"]
    assert len(per) == 5 and all(b["ffn"] == "split16" and b["attn"].startswith("split16") for b in per)
    rf = res["roofline"]
    assert 0 < rf["frame"]["frac_design"] <= rf["frame"]["frac"]  # the design's own floor is the stricter figure
    assert rf["bound"] == "hbm" and "k_ffn_ws" in rf["kernel"] and 0 < rf["frac"] < 1 and rf["unit"] == "GB/s"
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and rf["frame"]["frac"] > 0
    assert any("mssvt_block_attention" in o["kernel"] for o in rf["other_kernels"])


def test_bench_detector_training_two_ranks_and_sync_bn():
    """ref tools/train.py:118-119,143-144: the WHOLE detector under DDP (two ranks, one device, gloo), and the SyncBatchNorm
    conversion of every BatchNorm layer before the wrap (one rank here: gloo cannot gather device tensors, and two RCCL ranks
    cannot share a device -- the driver's 8-GPU node runs the synchronised statistics)."""
    res = _bench("--gpus", "2", "--train", "--detector", "--batch", "1", env={"MSSVT_BENCH_ONE_DEVICE": "1"})
    d = res["config"]["detector"]
    assert res["n_gpus"] == 2 and d["batch_norm_layers"] > 20 and d["sync_batch_norm_layers"] == 0 and d["loss"] > 0
    assert "CenterPoint" in res["config"]["workload"] and "DDP" in res["config"]["parallelism"]
    res = _bench("--train", "--detector", "--sync-bn", "--batch", "2")
    d = res["config"]["detector"]
    assert d["sync_batch_norm_layers"] == d["batch_norm_layers"] > 20 and d["loss"] > 0 and res["value"] > 0


def test_bench_arith_f32_in_flight_and_from_points():
    res = _bench("--arith", "f32", "--in-flight", "1")
    assert res["config"]["arith"] == "f32" and "v_mfma_f32_16x16x4_f32" in res["config"]["ffn_arith"]
    assert res["config"]["attn_arith"] == "f32: v_mfma_f32_16x16x4_f32" and res["config"]["frames_in_flight"] == 1
    assert all(b["ffn"] == "f32" and b["attn"] == "f32" for b in res["config"]["arith_per_block"])
    res = _bench("--in-flight", "3")
    assert res["config"]["frames_in_flight"] == 3 and res["one_frame_in_flight"]["value"] > 0
    assert "mssvt_frame_forward" in res["config"]["host_path"]
    res = _bench("--from-points")
    fp = res["from_points"]
    assert fp["points"] == 20000 and fp["voxels"] > 5000 and fp["bev_shape"][1] == 128 and 0 < fp["frac"] < 1
    assert fp["algorithmic_bytes"] == fp["vfe_bytes"] + fp["backbone_bytes"] + fp["dense_bytes"]
