"""CPU restatement of the CORSAIR post-processing path (retrieval, correspondences, RANSAC,
Chamfer, symmetry part cut, pose metrics).  TEST INFRASTRUCTURE ONLY.

Function names follow the reference so the parity tests read like it:
  scan2cad_retrieval_eval      utils/retrieval.py:139-177
  find_kcorr                   utils/eval_pose.py:48-79   (-> find_knn_cpu, utils/find_nn.py:43-49)
  registration_based_on_corr   utils/eval_pose.py:82-100  (-> Open3D RANSAC, restated in corsair_oracle.c)
  chamfer_kdtree_1direction    utils/preprocess.py:39-48,67-70
  symmetric_cut4 / split_corr / sym_pose   utils/symmetry.py:145-358
  eval_pose                    utils/eval_pose.py:103-128
Deliberate, documented differences from the reference (DESIGN.md "oracle vs reference"):
  * RANSAC: Open3D's global mt19937 + OpenMP schedule is replaced by a counter-based RNG and the
    single-thread iteration order (Open3D itself is run-to-run non-deterministic, README.md:260);
  * k-means: sklearn KMeans(random_state=0, n_init=10) is restated (same algorithm on the same RandomState(0)
    draws, f64 instead of f32; tests/test_pins_cpu.py compares with sklearn itself); the anchors come from
    an explicit seeded generator instead of NumPy's global RNG;
  * argsort ties (NumPy's default sort is unstable) are broken toward the smaller index.
"""
from __future__ import annotations

import numpy as np

from . import native


# ---- retrieval -------------------------------------------------------------------------------
def retrieval_rank(scan_feats, lib_feats):
    """argsort of the f64 Euclidean distance matrix, ties -> smaller index."""
    d2 = native.dist2_matrix(scan_feats, lib_feats)
    return np.argsort(d2, axis=1, kind="stable"), np.sqrt(d2)


def scan2cad_retrieval_eval_rank(pred_rank, table, best_match, pos_n):
    """utils/retrieval.py:139-167 given the predicted ranking."""
    table = np.asarray(table)
    gt_rank = np.argsort(table[best_match, :], 1, kind="stable")
    precision, top1_error, top1_predict, gt = [], [], [], []
    for g, p in zip(gt_rank, pred_rank):
        positive = np.isin(p[:pos_n], g[:pos_n]).astype(np.int32)
        precision.append(100.0 * np.sum(positive) / pos_n)
        top1_error.append(table[p[0], g[0]])
        top1_predict.append(int(p[0]))
        gt.append(int(g[0]))
    return {
        "precision": sum(precision) / len(precision),
        "top1_error": sum(top1_error) / len(top1_error),
        "top1_predict": top1_predict,
        "gt": gt,
    }


def scan2cad_retrieval_eval(scan_feats, lib_feats, best_match, table, pos_n):
    rank, _ = retrieval_rank(scan_feats, lib_feats)
    return scan2cad_retrieval_eval_rank(rank, table, best_match, pos_n)


# ---- correspondences ---------------------------------------------------------------------------
def find_kcorr(F0, F1, k=1):
    """k nearest F1 rows for every F0 row; returns (repeat(arange(N0), k), nn.flatten())."""
    nn = native.knn(F0, F1, k)
    return np.repeat(np.arange(len(F0)), k), nn.reshape(-1)


def registration_based_on_corr(source_pcd, target_pcd, max_corr_dist=0.03, seed=0,
                               max_iter=100000, confidence=0.999, ransac_n=10):
    T, inl, rmse, iters = native.ransac(source_pcd, target_pcd, max_corr_dist, ransac_n, max_iter,
                                        confidence, seed)
    return T


def apply_transform(pointcloud, T):
    pc = np.asarray(pointcloud, np.float64)
    T = np.asarray(T, np.float64)
    return pc @ T[:3, :3].T + T[:3, 3]


def chamfer_kdtree_1direction_T(xyz0, T, xyz1):
    """chamfer_kdtree_1direction(apply_transform(xyz0, T), xyz1) with T an f32 4x4."""
    return native.chamfer_1dir(xyz0, xyz1, np.asarray(T, np.float32))


# ---- symmetry part cut ----------------------------------------------------------------------------
def draw_anchors(n, n_anchor, rng_key, rng_counter):
    """np.random.choice(n, n_anchor, replace=False) from an explicit Philox stream."""
    if n < n_anchor:
        raise ValueError("Cannot take a larger sample than population when 'replace=False'")
    gen = np.random.Generator(np.random.Philox(key=rng_key, counter=rng_counter))
    return gen.choice(n, n_anchor, replace=False).astype(np.int32)


GATE_REFERENCE = (0.15, 0.15)          # utils/symmetry.py:232-257: dist.min() > 0.15 > max(error)
GATE_ANY = (0.0, float("inf"))         # any finite model passes (bench.py, random-init weights; corsair_amd/registration.py)


def gate_and_order(centers, counts, min_cdist, max_err, n, K, gate=GATE_REFERENCE):
    """The acceptance gate and centre ordering of symmetric_cut4 (utils/symmetry.py:232-257):
    accept anchors with dist.min() > 0.15 > max(error), keep the one with the smallest std of label
    fractions (first on ties); K=4: order = [0, nearest, farthest, middle] by distance from centre 0.
    Returns reordered centres [K,3] or raises AttributeError like the reference (no model found)."""
    counts = np.asarray(counts, np.float64)[:, :K]
    ratios = counts / float(n)
    std = np.sqrt(np.var(ratios, axis=1))
    valid = (min_cdist > gate[0]) & (gate[1] > max_err) & (std < 100)
    if not valid.any():
        raise AttributeError("'NoneType' object has no attribute 'cluster_centers_'")
    std_m = np.where(valid, std, np.inf)
    a = int(np.argmin(std_m))
    c = np.asarray(centers[a], np.float64)[:K]
    if K == 2:
        return c
    d = np.linalg.norm(c[0][None, :] - c[1:], axis=1)
    rank = np.argsort(d, kind="stable")
    order = [0, rank[0] + 1, rank[2] + 1, rank[1] + 1]
    return c[order]


def symmetric_cut4(feat, raw_pc, K, anchors, n_nn=50, n_init=10, max_iter=300, seed=0, force_gate=False):
    """Returns the integer part label of every voxel (part p == reference mask p)."""
    centers, counts, mcd, mer = native.symcut_fit(feat, raw_pc, anchors, K, n_nn, n_init, max_iter, seed)
    sel = gate_and_order(centers, counts, mcd, mer, len(raw_pc), K, GATE_ANY if force_gate else GATE_REFERENCE)
    return native.symcut_labels(raw_pc, K, sel)


def split_corr(xyz0, xyz1, F0, F1, lab0, lab1, perm, knn):
    """Per-part correspondences (utils/symmetry.py:145-179): part i of cloud 0 is matched against
    part perm[i] of cloud 1; parts concatenated in order, rows in original order inside a part.
    Returns (xyzA_corrs, xyzB_corrs) or None when a target part has fewer than knn voxels."""
    K = len(perm)
    nn = native.knn(F0, F1, knn, lab0, lab1, perm)
    order = np.argsort(lab0, kind="stable")
    order = order[(lab0[order] >= 0) & (lab0[order] < K)]
    idx = nn[order]
    if (idx < 0).any():
        return None
    return xyz0[np.repeat(order, knn)], xyz1[idx.reshape(-1)]


def part_configs(K, pos_sym):
    """Part assignments tried after the vanilla RANSAC: K cyclic shifts of pos_masks (utils/symmetry.py:303-324,
    `pos_masks = pos_masks[1:] + pos_masks[:1]`), then -- pos_sym >= 2 only -- 4 shifts of the mirrored order
    `[pos_masks[0], pos_masks[3], pos_masks[2], pos_masks[1]]` (utils/symmetry.py:326-356)."""
    configs = [[(i + s) % K for i in range(K)] for s in range(K)]
    if pos_sym >= 2:
        mirror = [0, 3, 2, 1]
        configs += [[mirror[(i + s) % 4] for i in range(4)] for s in range(4)]
    return configs


def sym_pose(baseF, xyz0, posF, xyz1, pos_sym, k_nn=5, max_corr=0.20, seed=0, anchors0=None,
             anchors1=None, max_iter=100000, confidence=0.999, force_gate=False, return_hyps=False):
    """utils/symmetry.py:262-358.  anchors0/anchors1: int32 anchor rows for the two clouds (None
    -> the cut fails like an exception in the reference).
    return_hyps: also return the list of every hypothesis evaluated, in evaluation order, as dicts
    {config (None = vanilla find_kcorr RANSAC), T f32[4,4], cd, iters, inliers, n_corr}, and the index
    of the one kept (first strict minimum of the Chamfer distance, utils/symmetry.py:322-324)."""
    idx_0, idx_1 = find_kcorr(baseF, posF, k=k_nn)
    T_ransac, inl, _, it = native.ransac(xyz0[idx_0], xyz1[idx_1], max_corr, 10, max_iter, confidence, seed)
    cd_ransac = chamfer_kdtree_1direction_T(xyz0, T_ransac, xyz1)
    T_best, cd_best = T_ransac, cd_ransac
    hyps = [dict(config=None, T=T_ransac, cd=cd_ransac, iters=it, inliers=inl, n_corr=len(idx_0))]
    chosen = 0

    def done(ok):
        out = (T_best, cd_best, T_ransac, cd_ransac, ok)
        return out + (hyps, chosen) if return_hyps else out

    K = 4 if pos_sym >= 2 else 2
    try:
        if anchors0 is None or anchors1 is None:
            raise ValueError("Cannot take a larger sample than population when 'replace=False'")
        lab0 = symmetric_cut4(baseF, xyz0, K, anchors0, seed=0, force_gate=force_gate)
        lab1 = symmetric_cut4(posF, xyz1, K, anchors1, seed=0, force_gate=force_gate)
    except (AttributeError, ValueError):
        return done(False)
    for perm in part_configs(K, pos_sym):
        corr = split_corr(xyz0, xyz1, baseF, posF, lab0, lab1, perm, k_nn)
        if corr is None:
            continue
        T, inl, _, it = native.ransac(corr[0], corr[1], max_corr, 10, max_iter, confidence, seed)
        cd = chamfer_kdtree_1direction_T(xyz0, T, xyz1)
        hyps.append(dict(config=list(perm), T=T, cd=cd, iters=it, inliers=inl, n_corr=len(corr[0])))
        if cd_best > cd:
            cd_best, T_best = cd, T
            chosen = len(hyps) - 1
    return done(True)


# ---- pose metrics --------------------------------------------------------------------------------
def rot_y(theta):
    """transforms3d.euler.euler2mat(0, theta, 0) (utils/eval_pose.py:114)."""
    c, s = np.cos(theta), np.sin(theta)
    return np.array([[c, 0.0, s], [0.0, 1.0, 0.0], [-s, 0.0, c]])


def eval_pose(T_est, T0, T1, axis_symmetry=1):
    """utils/eval_pose.py:103-128: (t_loss, r_loss) of the symmetry-group element with least RRE."""
    T_est = np.asarray(T_est, np.float32)
    t_best, r_best = np.inf, np.inf
    for i in range(int(axis_symmetry)):
        trans = np.eye(4)
        trans[:3, :3] = rot_y(i * (2 * np.pi / axis_symmetry))
        T_gt = np.matmul(T1, np.matmul(np.linalg.inv(trans), np.linalg.inv(T0))).astype(np.float32)
        # the f32 trace is promoted to f64 by "- 1" under the reference's NumPy 1.x scalar rules
        tr = np.float64(np.trace(T_est[:3, :3].T @ T_gt[:3, :3]))
        r_loss = np.arccos(np.clip((tr - 1) / 2, -1, 1))
        t_loss = np.linalg.norm(T_est[:3, 3] - T_gt[:3, 3])
        if r_best > r_loss:
            r_best, t_best = r_loss, t_loss
    return t_best, r_best
