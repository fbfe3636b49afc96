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
