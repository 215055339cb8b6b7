"""Host logic of the drop-in module on CPU: the ``impl="ops"`` orchestration, state-dict
compatibility, query tables and config handling -- with the oracle standing in for the HIP
ops (tests/oracle_ops_shim.py).  The GPU tests run the same assertions on the real kernels."""
import json
import os

import numpy as np
import pytest
import torch

from tests import oracle_ops_shim
from tests.test_module_gpu import BLOCKS, _cfg, assert_feat_close, build_block, load


@pytest.fixture(autouse=True)
def _cpu_ops(monkeypatch):
    oracle_ops_shim.install(monkeypatch)
    import tests.test_module_gpu as T
    monkeypatch.setattr(T, "DEV", "cpu")


def _sp(d):
    import tests.test_module_gpu as T
    return T.make_sp(d)


@pytest.mark.parametrize("name", BLOCKS)
def test_block_ops_path_matches_reference_golden(golden_dir, name):
    d, sd = load(golden_dir, name)
    blk = build_block(d, sd, "block")
    blk.impl = "ops"
    with torch.no_grad():
        out = blk(_sp(d))
    assert_feat_close(out.features.numpy(), d["out_features"])


@pytest.mark.parametrize("name", ["compress_1x1x16", "compress_3x3x5", "compress_2x2x4", "compress_2x2x2_groups",
                                  "compress_empty_sample"])
def test_compress_ops_path_matches_reference_golden(golden_dir, name):
    d, sd = load(golden_dir, name)
    blk = build_block(d, sd, "compress")
    blk.impl = "ops"
    with torch.no_grad():
        out = blk(_sp(d))
    np.testing.assert_array_equal(out.indices.numpy(), d["out_indices"])
    assert_feat_close(out.features.numpy(), d["out_features"])


@pytest.mark.parametrize("name", ["backbone", "backbone_two_levels", "backbone_c128"])
def test_backbone_ops_path_and_state_dict_keys(golden_dir, name):
    from mssvt_amd.mssvt_backbone import MixedScaleSparseTransformer
    d, sd = load(golden_dir, name)
    params = json.loads(str(d["params_json"]))
    c_in, c_out = int(d.get("in_channels", 32)), int(d.get("out_features_dim", 48))
    net = MixedScaleSparseTransformer(_cfg(params, int(d["hash_size"]), c_out), c_in, d["grid_size"].tolist(),
                                      d["voxel_size"].tolist(), d["point_cloud_range"].tolist())
    assert sorted(net.state_dict().keys()) == sorted(sd.keys())  # the reference's key names, exactly
    for k, v in net.state_dict().items():
        assert tuple(v.shape) == tuple(sd[k].shape), k
    net.load_state_dict(sd, strict=True)
    net.eval().set_impl("ops")
    assert net.num_point_features == c_out
    with torch.no_grad():
        bd = net(dict(voxel_features=torch.from_numpy(d["voxel_features"]),
                      voxel_coords=torch.from_numpy(d["voxel_coords"]).float(), batch_size=int(d["batch_size"])))
    sp = bd["encoded_spconv_tensor"]
    np.testing.assert_array_equal(sp.indices.numpy(), d["out_indices"])
    assert_feat_close(sp.features.numpy(), d["out_features"])
    assert_feat_close(sp.dense()[0, :, 0].numpy(), d["dense_b0_z0"])


@pytest.mark.parametrize("name", ["w335_777", "w222_444", "w557_bbb", "w115", "w3316"])
def test_query_tables(golden_dir, name):
    from mssvt_amd import query_table
    from oracle import block_ref
    z = np.load(os.path.join(golden_dir, "query_tables.npz"))
    ws = z[name + ".window_size"].tolist()
    mine, n_odd, n_even = query_table.vox_query_table(ws[0], ws[1] if len(ws) == 2 else None)
    orc, o_odd, o_even = block_ref.vox_query_table(ws[0], ws[1] if len(ws) == 2 else None)
    assert (n_odd, n_even) == (o_odd, o_even)
    for k in mine:
        np.testing.assert_array_equal(mine[k], orc[k])  # product == oracle (stable tie order)
        ref = z["%s.%s" % (name, k)]  # == reference up to the order inside a Chebyshev shell
        assert sorted(map(tuple, mine[k])) == sorted(map(tuple, ref))
        np.testing.assert_array_equal(np.abs(mine[k]).max(1), np.abs(ref).max(1))


def test_yaml_config_builds_the_w_backbone():
    from mssvt_amd import config
    net = config.build_backbone_from_cfg()
    assert net.grid_size == [470, 470, 32] and net.hash_size == 400000
    assert len(net.backbone) == 5 and net.num_point_features == 128
    blk = net.backbone[0]
    assert (blk.max_num_odd, blk.max_num_even, blk.max_num_win1, blk.max_num_win2) == (20, 5, 45, 343)
    assert [b.cbs_pattern for b in net.backbone[:4]] == [1, 0, 1, 0]
    assert "pos_proj.2.weight" in net.backbone[4].state_dict() and "pos_proj.2.weight" not in blk.state_dict()


def test_drop_path_train_eval():
    from mssvt_amd.mssvt_backbone import DropPath
    dp = DropPath(0.5)
    x = torch.ones(1000, 4)
    dp.eval()
    assert torch.equal(dp(x), x)
    dp.train()
    torch.manual_seed(0)
    y = dp(x)
    assert set(y.unique().tolist()) <= {0.0, 2.0} and 300 < (y[:, 0] == 0).sum() < 700


def test_height_compression_mirror_state_dict_and_cfg_keys():
    """Drop-in surface of the MAP_TO_BEV consumer: reference config keys and checkpoint names."""
    from mssvt_amd.height_compression import HeightCompression
    m = HeightCompression(dict(NUM_BEV_FEATURES=8, COMPRESS_LAYER_NUMS=2, LAYER_STRIDES=[1, 1, 1],
                               LAYER_DIALATIONS=[1, 2, 2], LAYER_PADDINGS=[1, 2, 2]))
    keys = set(m.state_dict())
    assert {"compress_layers.0.weight", "compress_layers.1.weight", "compress_layers.1.running_mean",
            "compress_layers.3.weight", "compress_layers.4.bias"} <= keys
    assert m.compress_layers[3].dilation == (2, 2) and m.compress_layers[3].padding == (2, 2)
    assert HeightCompression(dict(NUM_BEV_FEATURES=8, COMPRESS_LAYER_NUMS=0)).compress_layers is None
    assert m.num_bev_features == 8 and m.use_amp is False
