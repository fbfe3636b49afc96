"""Counterpart of the hot part of the reference's utils/preprocess.py:27-70."""
import numpy as np
import torch

from .. import backend as B
from ._convert import to_dev


def load_raw_pc(path, samples):
    return np.load(path)[:samples, :]


def load_norm_pc(path, samples):
    """First `samples` points, centred, scaled to unit radius (utils/preprocess.py:32-36)."""
    pc0 = np.load(path)[:samples, :]
    pc0 -= pc0.mean(0)
    pc0 = pc0 / np.max(np.linalg.norm(pc0, 2, 1))
    return pc0


def apply_transform(pointcloud, T):
    """[N,3] x [4,4] -> [N,3] (utils/preprocess.py:39-48); tiny, host side."""
    pointcloud = np.asarray(pointcloud)
    T = np.asarray(T)
    homo = np.concatenate([pointcloud, np.ones([len(pointcloud), 1])], 1)
    return np.matmul(T, homo.T).T[:, :3]


def chamfer_1direction_transformed(pc0, T, pc1):
    """chamfer_kdtree_1direction(apply_transform(pc0, T), pc1) fused in one kernel (T as f32 4x4)."""
    s, t = to_dev(pc0), to_dev(pc1)
    Tt = to_dev(np.asarray(T, np.float32).reshape(1, 4, 4))
    out = B.chamfer_1dir(s, [0, s.shape[0]], t, [0, t.shape[0]], [0], [0], Tt)
    return float(out.cpu()[0])


def chamfer_kdtree_1direction(pc0, pc1):
    """Mean distance from every pc0 point to its nearest pc1 point (utils/preprocess.py:67-70)."""
    return chamfer_1direction_transformed(pc0, np.eye(4, dtype=np.float32), pc1)
