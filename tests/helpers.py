"""Shared input builders for the parity tests (seeded, small enough for the CPU oracle)."""
import numpy as np

from corsair_amd import synth
from oracle import sparse


def make_batch(cloud_ids, n_points=10000, voxel=0.03, pose_ids=None):
    """Quantised + collated batch like CustomizeCADLib.collate_pair_fn (utils/Info/CADLib.py:148-178):
    returns coords int32 [N,4], feats f32 [N,1], origins f32 [N,3], offsets list."""
    grids, origins = [], []
    for j, cid in enumerate(cloud_ids):
        pc = synth.make_cloud(cid, 15000)[:n_points]
        if pose_ids is not None and pose_ids[j] is not None:
            pc = synth.apply_pose(pc, synth.random_pose(pose_ids[j], max_trans=0.0))
        xyz, grid, _ = sparse.quantize_cloud(pc, voxel)
        grids.append(grid)
        origins.append(xyz)
    coords = sparse.sparse_collate(grids)
    feats = np.ones((coords.shape[0], 1), np.float32)
    offsets = np.concatenate([[0], np.cumsum([len(g) for g in grids])]).tolist()
    return coords, feats, np.concatenate(origins, 0).astype(np.float32), offsets


LEGS = {4: np.array([[0.3, 0, 0.3], [-0.3, 0, 0.3], [-0.3, 0, -0.3], [0.3, 0, -0.3]]),
        2: np.array([[0.3, 0, 0.1], [-0.3, 0, -0.1]])}


def legged_object(seed, n_legs, n_per_leg=400):
    """A 2- or 4-legged object (legs parallel to y, 4 legs = 4-fold symmetric about y) whose 16-d
    'features' encode only the height of a point: the 50 feature-NN of an anchor are one slice of every
    leg, so symmetric_cut4's k-means finds the legs and its gate opens (utils/symmetry.py:232-243), while
    the vanilla 5-NN correspondences of find_kcorr land on a random leg (1/n_legs of them consistent with
    any one pose).  Returns (xyz f32 [n,3], feat f32 [n,16])."""
    rng = np.random.default_rng(seed)
    pts, feat = [], []
    for leg in LEGS[n_legs]:
        h = rng.uniform(-0.5, 0.5, n_per_leg)
        pts.append(leg + np.stack([rng.normal(0, 0.01, n_per_leg), h, rng.normal(0, 0.01, n_per_leg)], 1))
        f = np.zeros((n_per_leg, 16))
        f[:, 0] = h
        f[:, 1] = rng.normal(0, 1e-3, n_per_leg)
        feat.append(f)
    xyz = np.concatenate(pts).astype(np.float32)
    F = np.concatenate(feat).astype(np.float32)
    perm = rng.permutation(len(xyz))
    return xyz[perm], F[perm]


def rot_y_pose(deg, trans=(0.05, -0.02, 0.03)):
    t = np.deg2rad(deg)
    c, s = np.cos(t), np.sin(t)
    T = np.eye(4)
    T[:3, :3] = [[c, 0, s], [0, 1, 0], [-s, 0, c]]
    T[:3, 3] = trans
    return T
