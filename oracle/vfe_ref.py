"""CPU restatement of DynamicVFE.forward (test infrastructure only).

ref: pcdet/models/backbones_3d/vfe/dynamic_vfe.py:71-131.  The reference's VFE needs the un-vendored, un-pinned
``torch_scatter`` package (SURVEY.md F-list); this file restates the published semantics of the two functions it
calls -- ``scatter_mean`` = per-index arithmetic mean, ``scatter_max(...)[0]`` = per-index maximum -- with numpy.
Pinned: tests/golden/dynamic_vfe_*.npz are runs of the reference's own ``DynamicVFE.forward`` on the CPU with
exactly those two reductions restated (oracle/gen_golden_vfe.py); this restatement must reproduce them
(tests/test_oracle_vfe_cpu.py).  Parity against the real torch_scatter binaries stays UNPINNED.
"""
import numpy as np
import torch
import torch.nn.functional as F


def dynamic_vfe_forward(sd, points, num_point_features, voxel_size, grid_size, point_cloud_range, num_layers,
                        with_cluster_center=True, with_voxel_center=True, eps=1e-5):
    """sd: numpy state dict with keys pfn.{i}.0.{weight,bias}, pfn.{i}.1.{weight,bias,running_mean,running_var}.
    Returns (voxel_features (N,C) f32, voxel_coords (N,4) int32 [b,z,y,x])."""
    pts = np.asarray(points, np.float32)
    vs = np.asarray(voxel_size, np.float32)
    lo = np.asarray(point_cloud_range[:3], np.float32)
    gs = np.asarray(grid_size, np.int64)
    pc = np.floor((pts[:, 1:4] - lo) / vs).astype(np.int32)  # ref :83
    mask = ((pc >= 0) & (pc < gs)).all(1)
    pts, pc = pts[mask], pc[mask].astype(np.int64)
    merge = pts[:, 0].astype(np.int64) * gs[0] * gs[1] * gs[2] + pc[:, 0] * gs[1] * gs[2] + pc[:, 1] * gs[2] + pc[:, 2]
    unq, inv = np.unique(merge, return_inverse=True)  # ref :92 (sorted)
    n = unq.shape[0]
    xyz = pts[:, 1:4]
    feats = [pts[:, 1:num_point_features + 1]]
    if with_cluster_center:  # ref :96-100 (float64 accumulation, rounded once: the reference's order is undefined)
        s = np.zeros((n, 3), np.float64)
        np.add.at(s, inv, xyz.astype(np.float64))
        mean = (s / np.bincount(inv, minlength=n)[:, None]).astype(np.float32)
        feats.append(xyz - mean[inv])
    if with_voxel_center:  # ref :101-104
        off = (vs / 2 + lo).astype(np.float32)
        feats.append(xyz - (pc.astype(np.float32) * vs + off))
    x = torch.from_numpy(np.concatenate(feats, 1).astype(np.float32))

    def smax(t):
        out = np.full((n, t.shape[1]), -np.inf, np.float32)
        np.maximum.at(out, inv, t.numpy())
        return torch.from_numpy(out)

    for i in range(num_layers):  # ref get_points_fea :124-131
        p = "pfn.%d." % i
        x = F.linear(x, torch.from_numpy(sd[p + "0.weight"]), torch.from_numpy(sd[p + "0.bias"]))
        x = F.batch_norm(x, torch.from_numpy(sd[p + "1.running_mean"]), torch.from_numpy(sd[p + "1.running_var"]),
                         torch.from_numpy(sd[p + "1.weight"]), torch.from_numpy(sd[p + "1.bias"]), False, 0.0, eps)
        x = F.relu(x)
        if i < num_layers - 1:
            x = torch.cat((x, smax(x)[inv]), dim=-1)
    vf = smax(x).numpy()
    z = gs[2]
    coords = np.stack([unq // (gs[0] * gs[1] * z), unq % z, (unq % (gs[1] * z)) // z,
                       (unq % (gs[0] * gs[1] * z)) // (gs[1] * z)], 1).astype(np.int32)  # [b, z, y, x] (ref :113-118)
    return vf, coords
