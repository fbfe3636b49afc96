"""Counterpart of the reference's utils/symmetry.py:145-358 (split_corr, symmetric_cut4, sym_pose)."""
import numpy as np
import torch

from .. import backend as B
from .. import registration as R
from ._convert import to_dev


def symmetric_cut4(feat, raw_pc, K, max_sample=100, anchor_counter=0):
    """Cut a symmetric object along its planes of symmetry (utils/symmetry.py:182-259); returns the
    K boolean masks.  Raises AttributeError when no anchor passes the gate, ValueError when the cloud
    has fewer than max_sample voxels -- the two exceptions sym_pose catches in the reference."""
    f, x = to_dev(feat), to_dev(raw_pc)
    n = f.shape[0]
    anchors = R.draw_anchors(n, max_sample, anchor_counter)
    if anchors is None:
        raise ValueError("Cannot take a larger sample than population when 'replace=False'")
    a = torch.from_numpy(anchors[None]).to(f.device)
    c, cnt, mcd, mer = B.symcut_fit(f, x, [0, n], a, [K], 50, 10, 300, 0)
    sel = R.gate_and_order(c[0].cpu().numpy(), cnt[0].cpu().numpy(), mcd[0].cpu().numpy(),
                           mer[0].cpu().numpy(), n, K)
    if sel is None:
        raise AttributeError("'NoneType' object has no attribute 'cluster_centers_'")
    labels = B.symcut_labels(x, [0, n], [K], torch.from_numpy(sel[None]).to(f.device)).cpu().numpy()
    return [labels == i for i in range(K)]


def split_corr(pcsA, pcsB, featsA, featsB, knn, subsample_size=-1):
    """Per-part k-NN correspondences, concatenated (utils/symmetry.py:145-179)."""
    from .eval_pose import find_kcorr

    xa, xb = [], []
    for pcA, pcB, featA, featB in zip(pcsA, pcsB, featsA, featsB):
        if len(featA) == 0:
            continue
        idx_0, idx_1 = find_kcorr(featA, featB, k=knn, subsample_size=subsample_size)
        xa.append(np.asarray(pcA)[idx_0])
        xb.append(np.asarray(pcB)[idx_1])
    return np.concatenate(xa, axis=0), np.concatenate(xb, axis=0)


def sym_pose(baseF, xyz0, posF, xyz1, pos_sym, k_nn=5, max_corr=0.20, seed=0, anchor_ids=(0, 1),
             max_iter=100000, confidence=0.999):
    """Estimate pose with and without symmetry (utils/symmetry.py:262-358).  Returns
    (T_est_best, chamf_dist_best, T_est_ransac, chamf_dist_ransac, success) with the transforms as
    f32 torch tensors on the CPU, like the reference."""
    bf, x0, pf, x1 = to_dev(baseF), to_dev(xyz0), to_dev(posF), to_dev(xyz1)
    res = R.sym_pose_batch(bf, x0, [0, bf.shape[0]], pf, x1, [0, pf.shape[0]], [int(pos_sym)], k_nn,
                           max_corr, seed, [tuple(anchor_ids)], 100, max_iter, confidence)
    return (res.T_best[0].cpu(), float(res.cd_best[0].cpu()), res.T_ransac[0].cpu(),
            float(res.cd_ransac[0].cpu()), bool(res.ok[0]))
