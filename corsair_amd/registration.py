"""Batched symmetry-aided registration on the GPU.

Counterpart of ``sym_pose`` (utils/symmetry.py:262-358 of the reference) for a whole batch of
(query, CAD) pairs at once: the reference runs the pairs one after another in Python
(evaluation.py:297-331) and each stage on the CPU (SciPy KD-trees, Open3D RANSAC, sklearn k-means);
here every stage is one batched HIP launch over all pairs / hypotheses:

  1. feature 5-NN correspondences query -> CAD                  cs_knn_feat
  2. symmetry part cut of both clouds (100 anchors each)        cs_symcut_fit + host gate + cs_symcut_labels
  3. per-part correspondences for every cyclic / mirrored
     part assignment (K or K+4 hypotheses per pair)             cs_knn_feat (labelled)
  4. one RANSAC per hypothesis, all in one call                 cs_ransac_batch
  5. one-directional Chamfer of every estimate                  cs_chamfer_1dir
  6. keep the estimate with the smallest Chamfer (first wins)   argmin

torch is used for index plumbing on the device (stable sort of part labels, gathers).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import torch

from . import backend as B

ANCHOR_KEY = 0x5A11C0DE


@dataclass
class SymPoseResult:
    T_best: torch.Tensor        # f32 [P,4,4]
    cd_best: torch.Tensor       # f64 [P]
    T_ransac: torch.Tensor      # f32 [P,4,4]
    cd_ransac: torch.Tensor     # f64 [P]
    ok: np.ndarray              # bool [P]  (sym_ransac_success)
    iters: torch.Tensor         # int32 [n_problems] RANSAC iterations consumed
    n_problems: int


def draw_anchors(n, n_anchor, counter):
    """np.random.choice(n, n_anchor, replace=False) (utils/symmetry.py:193) from an explicit Philox
    stream keyed by `counter` instead of NumPy's global RNG."""
    if n < n_anchor:
        return None
    gen = np.random.Generator(np.random.Philox(key=ANCHOR_KEY, counter=counter))
    return gen.choice(n, n_anchor, replace=False).astype(np.int32)


def gate_and_order(centers, counts, min_cdist, max_err, n, K, force=False):
    """Acceptance gate + centre ordering of symmetric_cut4 (utils/symmetry.py:232-257) for one cloud.
    centers [A,4,3], counts [A,4], min_cdist/max_err [A] (NumPy).  Returns [4,3] centres ordered
    [0, nearest, farthest, middle] (K=4) or None when no anchor passes the gate."""
    ratios = counts[:, :K].astype(np.float64) / float(n)
    std = np.sqrt(np.var(ratios, axis=1))
    valid = (min_cdist > 0.15) & (0.15 > max_err) & (std < 100)
    if force:
        # bench-only: accept the best-balanced anchor whatever the gate says.  The thresholds are
        # tuned to trained features (the reference's caches report sym_ransac_success for every
        # query); with random-init weights they never pass and 2/3 of the registration work would
        # silently disappear from the timed region.
        valid = np.isfinite(max_err) & (min_cdist > 0)
    if not valid.any():
        return None
    a = int(np.argmin(np.where(valid, std, np.inf)))
    c = centers[a]
    out = np.zeros((4, 3), np.float64)
    if K == 2:
        out[:2] = c[:2]
        return out
    d = np.linalg.norm(c[0][None, :] - c[1:4], axis=1)
    rank = np.argsort(d, kind="stable")
    out[:] = c[[0, rank[0] + 1, rank[2] + 1, rank[1] + 1]]
    return out


def part_configs(K, pos_sym):
    """Part assignments tried by sym_pose: K cyclic shifts, plus 4 shifts of the mirrored order
    [0,3,2,1] when pos_sym >= 2 (utils/symmetry.py:303-356)."""
    cfgs = [[(i + s) % K for i in range(K)] for s in range(K)]
    if pos_sym >= 2:
        mirror = [0, 3, 2, 1]
        cfgs += [[mirror[(i + s) % 4] for i in range(4)] for s in range(4)]
    return cfgs


def sym_pose_batch(baseF, xyz0, off0, posF, xyz1, off1, pos_syms, k_nn=5, max_corr=0.20, seed=0,
                   anchor_ids=None, n_anchor=100, max_iter=100000, confidence=0.999,
                   use_symmetry=True, force_gate=False):
    """baseF f32 [N0,16], xyz0 f32 [N0,3] (query voxels of all pairs, segment p = off0[p]:off0[p+1]);
    posF/xyz1/off1 likewise for the CAD side; pos_syms: symmetry label per pair.
    anchor_ids[p] = (counter0, counter1) seeds the anchor draw of pair p (default (2p, 2p+1))."""
    dev = baseF.device
    P = len(off0) - 1
    off0 = [int(v) for v in off0]
    off1 = [int(v) for v in off1]
    n0 = [off0[p + 1] - off0[p] for p in range(P)]
    n1 = [off1[p + 1] - off1[p] for p in range(P)]
    k = k_nn

    # ---- 1. vanilla correspondences (find_kcorr, utils/eval_pose.py:48-79) -----------------
    nn = B.knn_feat(baseF, off0, posF, off1, k)                     # [N0, k] local CAD rows
    toff_rows = torch.repeat_interleave(
        torch.tensor(off1[:-1], device=dev, dtype=torch.int64),
        torch.tensor(n0, device=dev, dtype=torch.int64))            # CAD segment start per query row
    tgt_rows = (nn.to(torch.int64) + toff_rows[:, None]).reshape(-1)
    src_rows = torch.arange(off0[-1], device=dev, dtype=torch.int64).repeat_interleave(k)
    prob_src = [src_rows]
    prob_tgt = [tgt_rows]
    prob_len = [n0[p] * k for p in range(P)]
    prob_pair = list(range(P))
    ok = np.zeros(P, dtype=bool)

    # ---- 2./3. symmetry hypotheses ---------------------------------------------------------------
    if use_symmetry:
        Ks = [4 if int(pos_syms[p]) >= 2 else 2 for p in range(P)]
        if anchor_ids is None:
            anchor_ids = [(2 * p, 2 * p + 1) for p in range(P)]
        anc0 = [draw_anchors(n0[p], n_anchor, anchor_ids[p][0]) for p in range(P)]
        anc1 = [draw_anchors(n1[p], n_anchor, anchor_ids[p][1]) for p in range(P)]
        cand = [p for p in range(P) if anc0[p] is not None and anc1[p] is not None]
        sel0 = np.zeros((P, 4, 3))
        sel1 = np.zeros((P, 4, 3))
        if cand:
            def fit(feat, xyz, off, anc):
                a = torch.from_numpy(np.stack([anc[p] if anc[p] is not None
                                               else np.zeros(n_anchor, np.int32) for p in range(P)])).to(dev)
                c, cnt, mcd, mer = B.symcut_fit(feat, xyz, off, a, Ks, 50, 10, 300, 0)
                return c.cpu().numpy(), cnt.cpu().numpy(), mcd.cpu().numpy(), mer.cpu().numpy()

            c0, cnt0, mcd0, mer0 = fit(baseF, xyz0, off0, anc0)
            c1, cnt1, mcd1, mer1 = fit(posF, xyz1, off1, anc1)
            for p in cand:
                g0 = gate_and_order(c0[p], cnt0[p], mcd0[p], mer0[p], n0[p], Ks[p], force_gate)
                g1 = gate_and_order(c1[p], cnt1[p], mcd1[p], mer1[p], n1[p], Ks[p], force_gate)
                if g0 is not None and g1 is not None:
                    sel0[p], sel1[p] = g0, g1
                    ok[p] = True
        good = [p for p in range(P) if ok[p]]
        if good:
            lab0 = B.symcut_labels(xyz0, off0, Ks, torch.from_numpy(sel0).to(dev))
            lab1 = B.symcut_labels(xyz1, off1, Ks, torch.from_numpy(sel1).to(dev))
            qseg, tseg, perms, cfg_pair = [], [], [], []
            for p in good:
                for cfg in part_configs(Ks[p], int(pos_syms[p])):
                    qseg.append(p)
                    tseg.append(p)
                    perms.append(cfg + [-3] * (8 - len(cfg)))
                    cfg_pair.append(p)
            perm_t = torch.tensor(perms, dtype=torch.int32, device=dev)
            # stable partition of every query cloud by part label (split_corr concatenates the parts
            # in order, rows in original order inside a part): one stable sort of (pair, label) keys.
            # The labelled search runs on the partitioned rows, so a wave of 64 queries shares one
            # label and skips the targets of the other parts wholesale.
            seg_rows = torch.repeat_interleave(torch.arange(P, device=dev, dtype=torch.int64),
                                               torch.tensor(n0, device=dev, dtype=torch.int64))
            sorted_rows = torch.sort(seg_rows * 8 + lab0.to(torch.int64).clamp(0, 7), stable=True).indices
            nn_cfg = B.knn_feat(baseF[sorted_rows], off0, posF, off1, k, qseg=qseg, tseg=tseg,
                                qlabel=lab0[sorted_rows].contiguous(), tlabel=lab1, perm=perm_t)
            row = 0
            for j, p in enumerate(cfg_pair):
                idx = nn_cfg[row:row + n0[p]]          # rows already in (part, original row) order
                row += n0[p]
                if bool((idx < 0).any()):
                    continue  # a CAD part with fewer than k voxels: the reference cannot build it
                prob_src.append(sorted_rows[off0[p]:off0[p + 1]].repeat_interleave(k))
                prob_tgt.append((idx.to(torch.int64) + off1[p]).reshape(-1))
                prob_len.append(n0[p] * k)
                prob_pair.append(p)

    # ---- 4. RANSAC over all hypotheses (registration_based_on_corr, utils/eval_pose.py:82-100) ----
    src_idx = torch.cat(prob_src)
    tgt_idx = torch.cat(prob_tgt)
    src_pts = xyz0[src_idx]
    tgt_pts = xyz1[tgt_idx]
    offs = np.concatenate([[0], np.cumsum(prob_len)]).tolist()
    T, inl, rmse, iters = B.ransac_batch(src_pts, tgt_pts, offs, max_corr, 10, max_iter, confidence, seed)

    # ---- 5. Chamfer of every estimate (utils/preprocess.py:39-48,67-70) -----------------------------
    cd = B.chamfer_1dir(xyz0, off0, xyz1, off1, prob_pair, prob_pair, T)

    # ---- 6. best hypothesis per pair: first minimum, vanilla first (utils/symmetry.py:322-324) ----
    cd_h = cd.cpu().numpy()
    pair_h = np.asarray(prob_pair)
    best = np.arange(P)
    for j in range(P, len(pair_h)):
        p = pair_h[j]
        if cd_h[best[p]] > cd_h[j]:
            best[p] = j
    best_t = torch.from_numpy(best).to(dev)
    return SymPoseResult(T_best=T[best_t], cd_best=cd[best_t], T_ransac=T[:P], cd_ransac=cd[:P],
                         ok=ok, iters=iters, n_problems=len(prob_pair))
