"""Counterpart of the reference's utils/eval_pose.py:48-128."""
import numpy as np
import torch

from .. import backend as B
from ._convert import to_dev
from .find_nn import find_knn


def find_kcorr(F0, F1, k=1, nn_max_n=500, subsample_size=-1):
    """Top-k matching pairs by local features (utils/eval_pose.py:48-79): returns
    (np.repeat(arange(N0), k), nn_inds.flatten()).  subsample_size > 0 and len(F0) > subsample_size (never on the
    evaluated path: sym_pose passes -1, utils/symmetry.py:267,311): both sides are cut to at most `subsample_size` rows
    drawn WITHOUT replacement from NumPy's global generator, F0's draw first -- the reference's two `np.random.choice`
    calls, so a caller that seeds `np.random` gets the reference's rows -- and the pair indices refer to the full sets."""
    sub = subsample_size > 0 and len(F0) > subsample_size
    if sub:
        inds0 = np.random.choice(len(F0), min(len(F0), subsample_size), replace=False)
        inds1 = np.random.choice(len(F1), min(len(F1), subsample_size), replace=False)
        if torch.is_tensor(F0):
            F0 = F0[torch.from_numpy(inds0).to(F0.device)]
        else:
            F0 = np.asarray(F0)[inds0]
        if torch.is_tensor(F1):
            F1 = F1[torch.from_numpy(inds1).to(F1.device)]
        else:
            F1 = np.asarray(F1)[inds1]
    nn_inds = find_knn(F0, F1, k).reshape(-1)
    if sub:
        return np.repeat(inds0, k), inds1[nn_inds]
    return np.repeat(np.arange(len(F0)), k), nn_inds


def registration_based_on_corr(source_pcd, target_pcd, max_corr_dist=0.03, seed=0,
                               max_iter=100000, confidence=0.999, ransac_n=10):
    """RANSAC over identity correspondences (utils/eval_pose.py:82-100); returns a 4x4 f64 array like
    Open3D's result.transformation."""
    s, t = to_dev(source_pcd), to_dev(target_pcd)
    T, _, _, _ = B.ransac_batch(s, t, [0, s.shape[0]], max_corr_dist, ransac_n, max_iter, confidence, seed)
    return T[0].cpu().numpy().astype(np.float64)


def rot_y(theta):
    """transforms3d.euler.euler2mat(0, theta, 0) (utils/eval_pose.py:114)."""
    c, s = np.cos(theta), np.sin(theta)
    return np.array([[c, 0.0, s], [0.0, 1.0, 0.0], [-s, 0.0, c]])


def eval_pose(T_est, T0, T1, axis_symmetry=1):
    """Rotation / translation error minimised over the y-axis symmetry group
    (utils/eval_pose.py:103-128).  Returns (t_loss, r_loss) of the element with the least r_loss.
    A few dozen flops per query: host side, NumPy, exactly the reference's expression order."""
    if isinstance(T_est, torch.Tensor):
        T_est = T_est.detach().cpu().numpy()
    T_est = np.asarray(T_est, np.float32)
    T0, T1 = np.asarray(T0), np.asarray(T1)
    t_loss_best, r_loss_best = np.inf, np.inf
    for i in range(int(axis_symmetry)):
        trans = np.eye(4)
        trans[:3, :3] = rot_y(i * (2 * np.pi / axis_symmetry))
        T_gt = np.matmul(T1, np.matmul(np.linalg.inv(trans), np.linalg.inv(T0))).astype(np.float32)
        tr = np.float64(np.trace(T_est[:3, :3].T @ T_gt[:3, :3]))
        r_loss = np.arccos(np.clip((tr - 1) / 2, -1, 1))
        t_loss = np.linalg.norm(T_est[:3, 3] - T_gt[:3, 3])
        if r_loss_best > r_loss:
            r_loss_best, t_loss_best = r_loss, t_loss
    return t_loss_best, r_loss_best
