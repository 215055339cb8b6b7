"""Known-answer check of the DynamicVFE restatement (oracle/vfe_ref.py): two voxels, hand-computed."""
import numpy as np

from oracle import vfe_ref


def test_vfe_restatement_known_answer():
    # grid 4x4x2 cells of 1 m from the origin; rows [b, x, y, z, intensity]
    pts = np.array([[0, 0.25, 0.25, 0.25, 1.0],   # voxel (0,0,0)
                    [0, 0.75, 0.25, 0.25, 3.0],   # voxel (0,0,0)
                    [0, 2.50, 1.50, 1.50, 2.0],   # voxel (2,1,1)
                    [0, 9.00, 0.00, 0.00, 7.0]],  # outside: dropped
                   np.float32)
    F_in = 4  # x, y, z, intensity
    C = F_in + 6
    sd = {"pfn.0.0.weight": np.eye(C, dtype=np.float32), "pfn.0.0.bias": np.zeros(C, np.float32),
          "pfn.0.1.weight": np.ones(C, np.float32), "pfn.0.1.bias": np.zeros(C, np.float32),
          "pfn.0.1.running_mean": np.zeros(C, np.float32), "pfn.0.1.running_var": np.ones(C, np.float32)}
    vf, vc = vfe_ref.dynamic_vfe_forward(sd, pts, F_in, [1.0, 1.0, 1.0], [4, 4, 2], [0, 0, 0, 4, 4, 2], 1, eps=1e-12)
    np.testing.assert_array_equal(vc, [[0, 0, 0, 0], [0, 1, 1, 2]])  # [b, z, y, x], sorted by (b, x, y, z)
    # voxel 0: mean x = 0.5 -> f_cluster x = (-0.25, +0.25); centre 0.5 -> f_center = (-0.25, +0.25, ...); relu + max
    want0 = [0.75, 0.25, 0.25, 3.0, 0.25, 0.0, 0.0, 0.25, 0.0, 0.0]
    want1 = [2.5, 1.5, 1.5, 2.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0]
    np.testing.assert_allclose(vf, [want0, want1], atol=1e-6)


import os  # noqa: E402

import pytest  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", ["dynamic_vfe_64_128", "dynamic_vfe_16"])
def test_vfe_restatement_matches_the_reference_run(name):
    """oracle/vfe_ref.py against the reference's own DynamicVFE.forward (oracle/gen_golden_vfe.py: the reference
    code on the CPU, torch_scatter's two reductions restated): voxel coordinates bit-exact, features 1e-5."""
    d = np.load(os.path.join(GOLDEN, name + ".npz"))
    sd = {k[3:]: d[k] for k in d.files if k.startswith("sd.")}
    vf, vc = vfe_ref.dynamic_vfe_forward(sd, d["points"], 5, d["voxel_size"].tolist(), d["grid_size"].tolist(),
                                         d["point_cloud_range"].tolist(), len(d["num_filters"]))
    np.testing.assert_array_equal(vc, d["voxel_coords"])
    np.testing.assert_allclose(vf, d["voxel_features"], rtol=1e-5, atol=1e-5)
