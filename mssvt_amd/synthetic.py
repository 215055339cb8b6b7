"""Seeded synthetic Waymo-shaped LiDAR scenes.

There is no dataset access; every number in bench.py / tests is measured on
scenes from this generator: 64 beams with elevation linspace(-17.6 deg, +2.4 deg),
azimuth U(-pi, pi), sensor height 2.0 m, range = min(ground-hit range,
Gamma(k=2, theta=12 m)), points with r >= 75 m dropped, z += N(0, 0.02);
intensity = tanh(U(0,1)), elongation = U(0,1); exactly P points per scene.

Pure numpy (host side); the voxel grid constants are the "W" configuration of
SURVEY.md section 8 (range [-75.2,-75.2,-2, 75.2,75.2,4], voxel [0.32,0.32,0.1875]
-> grid [470,470,32]).
"""
import numpy as np

POINT_CLOUD_RANGE = [-75.2, -75.2, -2.0, 75.2, 75.2, 4.0]
VOXEL_SIZE = [0.32, 0.32, 0.1875]
GRID_SIZE = [470, 470, 32]


def make_scene(num_points, seed):
    """Return (P, 5) float32 [x, y, z, intensity, elongation]."""
    rng = np.random.default_rng(seed)
    sensor_h = 2.0
    elev = np.deg2rad(np.linspace(-17.6, 2.4, 64))
    chunks = []
    have = 0
    while have < num_points:
        n = max(1024, int((num_points - have) * 1.5))
        beam = rng.integers(0, 64, size=n)
        el = elev[beam]
        az = rng.uniform(-np.pi, np.pi, size=n)
        r_obj = rng.gamma(2.0, 12.0, size=n)
        with np.errstate(divide="ignore"):
            r_ground = np.where(el < 0, sensor_h / np.sin(-el), np.inf)
        r = np.minimum(r_ground, r_obj)
        keep = r < 75.0
        r, el, az = r[keep], el[keep], az[keep]
        x = r * np.cos(el) * np.cos(az)
        y = r * np.cos(el) * np.sin(az)
        z = sensor_h + r * np.sin(el) + rng.normal(0.0, 0.02, size=r.shape[0])
        inten = np.tanh(rng.uniform(0, 1, size=r.shape[0]))
        elong = rng.uniform(0, 1, size=r.shape[0])
        pts = np.stack([x, y, z, inten, elong], axis=1).astype(np.float32)
        chunks.append(pts)
        have += pts.shape[0]
    return np.ascontiguousarray(np.concatenate(chunks, axis=0)[:num_points])


def make_batch_points(num_points, batch_size, seed0=0):
    """(B*P, 6) float32 [b, x, y, z, intensity, elongation], scene b seeded seed0+b."""
    out = []
    for b in range(batch_size):
        pts = make_scene(num_points, seed0 + b)
        out.append(np.concatenate([np.full((pts.shape[0], 1), b, np.float32), pts], axis=1))
    return np.ascontiguousarray(np.concatenate(out, axis=0))


def voxelize_numpy(points, point_cloud_range=POINT_CLOUD_RANGE, voxel_size=VOXEL_SIZE,
                   grid_size=GRID_SIZE):
    """Host restatement of the index part of DynamicVFE.forward
    (pcdet/models/backbones_3d/vfe/dynamic_vfe.py:83-93,114-118): float division
    then floor, range mask, key ((b*X+x)*Y+y)*Z+z, sorted unique; coords [b,z,y,x].

    Returns (voxel_coords (N,4) int32, unq_inv (P_kept,) int64, kept_mask (P,) bool).
    Used to build inputs for tests / bench; the device voxelizer is
    mssvt_amd.voxelize (checked against this in tests).
    """
    pr = np.asarray(point_cloud_range, np.float32)
    vs = np.asarray(voxel_size, np.float32)
    gs = np.asarray(grid_size, np.int64)
    xyz = points[:, 1:4].astype(np.float32)
    coords = np.floor((xyz - pr[None, 0:3]) / vs[None, :]).astype(np.int64)
    mask = ((coords >= 0) & (coords < gs[None, :])).all(axis=1)
    coords = coords[mask]
    b = points[mask, 0].astype(np.int64)
    X, Y, Z = gs
    key = ((b * X + coords[:, 0]) * Y + coords[:, 1]) * Z + coords[:, 2]
    unq, inv = np.unique(key, return_inverse=True)
    z = unq % Z
    y = (unq // Z) % Y
    x = (unq // (Z * Y)) % X
    bb = unq // (Z * Y * X)
    vc = np.stack([bb, z, y, x], axis=1).astype(np.int32)
    return np.ascontiguousarray(vc), inv.astype(np.int64), mask
