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
