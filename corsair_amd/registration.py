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

import os
from dataclasses import dataclass

import numpy as np
import torch

from . import _helper
from . import backend as B
from ._lib import to_host

ANCHOR_KEY = 0x5A11C0DE

# The vanilla RANSACs of a batch need nothing from the symmetry stages: they run on the helper thread
# (_helper.py) with its own stream while the calling thread goes through part cut, host gate and labelled
# 5-NN (a launch sequence with two host decisions in it, during which the GPU would otherwise idle), then
# the symmetric hypotheses get their own cs_ransac_batch call.  The draws of a problem depend on (seed,
# iteration) only, so splitting the call changes no result.  CORSAIR_SPLIT_RANSAC=0 keeps one call.


@dataclass
class SymPoseResult:
    T_best: torch.Tensor        # f32 [P,4,4]
    cd_best: torch.Tensor       # f64 [P]
    T_ransac: torch.Tensor      # f32 [P,4,4]
    cd_ransac: torch.Tensor     # f64 [P]
    ok: np.ndarray              # bool [P]  (sym_ransac_success)
    iters: torch.Tensor         # int32 [n_problems] RANSAC iterations consumed
    n_problems: int
    # every hypothesis evaluated (problem j): the first P are the vanilla find_kcorr RANSACs of the pairs,
    # then the part configurations of the pairs whose cut passed, pair-major, in the reference's order
    # (utils/symmetry.py:303-356)
    T_all: torch.Tensor = None      # f32 [n_problems,4,4]
    cd_all: torch.Tensor = None     # f64 [n_problems]
    inliers: torch.Tensor = None    # int32 [n_problems]
    prob_pair: list = None          # pair of problem j
    prob_cfg: list = None           # part assignment of problem j (None = vanilla)
    best: np.ndarray = None         # int64 [P] problem kept per pair (first strict Chamfer minimum)

    def hypotheses(self, p):
        """Problems of pair p in evaluation order."""
        return [j for j, q in enumerate(self.prob_pair) if q == p]


def draw_anchors(n, n_anchor, counter):
    """np.random.choice(n, n_anchor, replace=False) (utils/symmetry.py:193) from an explicit Philox
    stream keyed by `counter` instead of NumPy's global RNG."""
    if n < n_anchor:
        return None
    gen = np.random.Generator(np.random.Philox(key=ANCHOR_KEY, counter=counter))
    return gen.choice(n, n_anchor, replace=False).astype(np.int32)


# The acceptance test of symmetric_cut4 is `dist.min() > 0.15 > max(error)` (utils/symmetry.py:232-257): the two
# thresholds are hyper-parameters tuned to TRAINED features (the reference's caches report sym_ransac_success for every
# query).  They are an argument here, default = the reference's constants.  GATE_ANY = "any finite model passes" is what
# bench.py uses with its random-init weights, under which the reference's thresholds never pass and 2/3 of the
# registration work would silently leave the timed region (`force_gate=True` of the entry points is shorthand for it).
GATE_REFERENCE = (0.15, 0.15)
GATE_ANY = (0.0, float("inf"))


def gate_and_order(centers, counts, min_cdist, max_err, n, K, gate=GATE_REFERENCE):
    """Acceptance gate + centre ordering of symmetric_cut4 (utils/symmetry.py:232-257) for one cloud.
    centers [A,4,3], counts [A,4], min_cdist/max_err [A] (NumPy).  Returns [4,3] centres ordered
    [0, nearest, farthest, middle] (K=4) or None when no anchor passes the gate."""
    ratios = counts[:, :K].astype(np.float64) / float(n)
    std = np.sqrt(np.var(ratios, axis=1))
    valid = (min_cdist > gate[0]) & (gate[1] > max_err) & (std < 100)
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


def gate_and_order_batch(centers, counts, min_cdist, max_err, n, Ks, cand, gate=GATE_REFERENCE):
    """gate_and_order for the pairs in `cand` at once (same arithmetic, NumPy over the pair axis).
    centers [P,A,4,3], counts [P,A,4], min_cdist / max_err [P,A], n [P], Ks [P].
    Returns (sel [P,4,3], ok [P])."""
    P = centers.shape[0]
    sel = np.zeros((P, 4, 3), np.float64)
    ok = np.zeros(P, dtype=bool)
    n = np.asarray(n, dtype=np.float64)
    Ks = np.asarray(Ks)
    cand = np.asarray(cand, dtype=np.int64)
    for K in (2, 4):
        idx = cand[Ks[cand] == K]
        if idx.size == 0:
            continue
        ratios = counts[idx][:, :, :K].astype(np.float64) / n[idx][:, None, None]
        std = np.sqrt(np.var(ratios, axis=2))
        valid = (min_cdist[idx] > gate[0]) & (gate[1] > max_err[idx]) & (std < 100)
        has = valid.any(axis=1)
        a = np.argmin(np.where(valid, std, np.inf), axis=1)
        c = centers[idx, a]                                   # [G,4,3]
        out = np.zeros((idx.size, 4, 3), np.float64)
        if K == 2:
            out[:, :2] = c[:, :2]
        else:
            d = np.linalg.norm(c[:, 0:1] - c[:, 1:4], axis=2)  # [G,3]
            rank = np.argsort(d, axis=1, kind="stable")
            order = np.stack([np.zeros(idx.size, np.int64), rank[:, 0] + 1, rank[:, 2] + 1, rank[:, 1] + 1], 1)
            out = np.take_along_axis(c, order[:, :, None], axis=1)
        sel[idx[has]] = out[has]
        ok[idx[has]] = True
    return sel, ok


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
                   use_symmetry=True, force_gate=False, query_anchors=None):
    """baseF f32 [N0,16], xyz0 f32 [N0,3] (query voxels of all pairs, segment p = off0[p]:off0[p+1]);
    posF/xyz1/off1 likewise for the CAD side; pos_syms: symmetry label per pair.
    anchor_ids[p] = (counter0, counter1) seeds the anchor draw of pair p (default (2p, 2p+1)).
    query_anchors: the query-side draws (draw_anchors(n0[p], n_anchor, anchor_ids[p][0]) for every p)
    when the caller has already made them -- e.g. while the embedding kernels were still running; the
    ~1 ms of host work would otherwise sit between the 5-NN and the part-cut launches."""
    dev = baseF.device
    P = len(off0) - 1
    off0 = [int(v) for v in off0]
    off1 = [int(v) for v in off1]
    n0 = [off0[p + 1] - off0[p] for p in range(P)]
    n1 = [off1[p + 1] - off1[p] for p in range(P)]
    k = k_nn

    # ---- 1. vanilla correspondences (find_kcorr, utils/eval_pose.py:48-79) -----------------
    # A CAD cloud with fewer than k voxels has no k-th neighbour: SciPy's KD-tree returns the out-of-range index n
    # there and the reference's fancy indexing raises IndexError (utils/eval_pose.py:66-72).  Same here, before any
    # launch: the neighbour lists would hold -1 (k_corr_assemble must never read through one).
    short = [p for p in range(P) if n0[p] > 0 and n1[p] < k]
    if short:
        raise IndexError("sym_pose_batch: CAD cloud of pair %d has %d voxels, fewer than k_nn = %d (the reference's "
                         "find_kcorr fails on such a pair too)" % (short[0], n1[short[0]], k))
    nn = B.knn_feat(baseF, off0, posF, off1, k)                     # [N0, k] local CAD rows
    # correspondence lists straight from the neighbour lists (one launch; rounds 1-2 built index tensors with
    # repeat_interleave / arange / gathers): pair p = query rows off0[p]:off0[p+1] against the CAD cloud at off1[p]
    desc_v = np.asarray([[off0[p], off0[p], off1[p], n0[p], off0[p]] for p in range(P)], dtype=np.int64).reshape(P, 5)
    v_src, v_tgt = B.corr_assemble(xyz0, xyz1, None, nn, desc_v, off0[-1], max(n0) if P else 0)
    corr_src = [v_src]
    corr_tgt = [v_tgt]
    prob_len = [n0[p] * k for p in range(P)]
    prob_pair = list(range(P))
    prob_cfg = [None] * P
    ok = np.zeros(P, dtype=bool)
    vanilla = None
    if use_symmetry and dev.type == "cuda" and os.environ.get("CORSAIR_SPLIT_RANSAC", "1") != "0":
        v_offs = np.concatenate([[0], np.cumsum(prob_len)]).tolist()
        for t in (v_src, v_tgt):
            t.record_stream(_helper.stream(dev))
        vanilla = _helper.submit(dev, lambda: B.ransac_batch(v_src, v_tgt, v_offs, max_corr, 10, max_iter,
                                                            confidence, seed))

    try:
        # ---- 2./3. symmetry hypotheses ---------------------------------------------------------------
        if use_symmetry:
            Ks = [4 if int(pos_syms[p]) >= 2 else 2 for p in range(P)]
            if anchor_ids is None:
                anchor_ids = [(2 * p, 2 * p + 1) for p in range(P)]
            anc0 = query_anchors if query_anchors is not None else \
                [draw_anchors(n0[p], n_anchor, anchor_ids[p][0]) for p in range(P)]
            anc1 = [draw_anchors(n1[p], n_anchor, anchor_ids[p][1]) for p in range(P)]
            cand = [p for p in range(P) if anc0[p] is not None and anc1[p] is not None]
            sel0 = np.zeros((P, 4, 3))
            sel1 = np.zeros((P, 4, 3))
            if cand:
                # one launch set for both sides (query clouds, then CAD clouds): the k-means stage is
                # latency-bound (one thread per restart), twice the clouds cost the same time
                def anchors_of(anc):
                    return np.stack([anc[p] if anc[p] is not None else np.zeros(n_anchor, np.int32) for p in range(P)])

                a_all = torch.from_numpy(np.concatenate([anchors_of(anc0), anchors_of(anc1)])).to(dev)
                off_all = off0 + [off0[-1] + o for o in off1[1:]]
                c, cnt, mcd, mer = B.symcut_fit(torch.cat([baseF, posF]), torch.cat([xyz0, xyz1]), off_all, a_all,
                                                Ks + Ks, 50, 10, 300)
                c, cnt, mcd, mer = to_host(c, cnt, mcd, mer)
                c0, cnt0, mcd0, mer0 = c[:P], cnt[:P], mcd[:P], mer[:P]
                c1, cnt1, mcd1, mer1 = c[P:], cnt[P:], mcd[P:], mer[P:]
                gate = GATE_ANY if force_gate else GATE_REFERENCE
                g0, ok0 = gate_and_order_batch(c0, cnt0, mcd0, mer0, n0, Ks, cand, gate)
                g1, ok1 = gate_and_order_batch(c1, cnt1, mcd1, mer1, n1, Ks, cand, gate)
                ok = ok0 & ok1
                sel0[ok], sel1[ok] = g0[ok], g1[ok]
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
                # one small upload for everything the device needs from the host here: segment offsets,
                # configuration row ranges, part permutations
                lens = np.asarray([n0[p] for p in cfg_pair], dtype=np.int64)
                row_start = np.concatenate([[0], np.cumsum(lens)])
                n_cfg = len(cfg_pair)
                host_blk = np.concatenate([np.asarray(off0, np.int64), row_start,
                                           np.asarray(perms, np.int32).reshape(-1).view(np.int64)])
                dev_blk = torch.from_numpy(host_blk).to(dev)
                off0_t = dev_blk[:P + 1]
                first_t = dev_blk[P + 1:P + 2 + n_cfg]
                perm_t = dev_blk[P + 2 + n_cfg:].view(torch.int32).view(n_cfg, 8)
                # stable partition of every query cloud by part label (split_corr concatenates the parts
                # in order, rows in original order inside a part): one launch.  The labelled search runs on
                # the partitioned rows, so a wave of 64 queries shares one label and skips the targets of
                # the other parts wholesale.
                sorted_rows = B.partition_by_label(lab0, off0_t, P)
                nn_cfg = B.knn_feat(baseF[sorted_rows], off0, posF, off1, k, qseg=qseg, tseg=tseg,
                                    qlabel=lab0[sorted_rows].contiguous(), tlabel=lab1, perm=perm_t)
                # configuration j owns the query rows sorted_rows[off0[p] : off0[p+1]] and the result rows
                # nn_cfg[row_start[j] : row_start[j+1]].  A CAD part with fewer than k voxels leaves -1 entries:
                # the reference cannot build that configuration (one launch + one small copy decide all of them)
                bad = to_host(B.cfg_bad(nn_cfg, first_t, n_cfg))[0] > 0
                keep = [j for j in range(n_cfg) if not bad[j]]
                if keep:
                    out_first = np.concatenate([[0], np.cumsum(lens[keep])])
                    desc = np.asarray([[off0[cfg_pair[j]], row_start[j], off1[cfg_pair[j]], lens[j], out_first[i]]
                                       for i, j in enumerate(keep)], dtype=np.int64)
                    s_src, s_tgt = B.corr_assemble(xyz0, xyz1, sorted_rows, nn_cfg, desc, int(out_first[-1]),
                                                   int(lens[keep].max()))
                    corr_src.append(s_src)
                    corr_tgt.append(s_tgt)
                    for j in keep:
                        prob_len.append(n0[cfg_pair[j]] * k)
                        prob_pair.append(cfg_pair[j])
                        prob_cfg.append([c for c in perms[j] if c >= 0])
    except BaseException:
        if vanilla is not None:       # do not leave the helper's call running into freed tensors
            try:
                vanilla.result()
            except Exception:
                pass
        raise

    # ---- 4. RANSAC over all hypotheses (registration_based_on_corr, utils/eval_pose.py:82-100) ----
    if vanilla is None:
        src = corr_src[0] if len(corr_src) == 1 else torch.cat(corr_src)
        tgt = corr_tgt[0] if len(corr_tgt) == 1 else torch.cat(corr_tgt)
        offs = np.concatenate([[0], np.cumsum(prob_len)]).tolist()
        T, inl, rmse, iters = B.ransac_batch(src, tgt, offs, max_corr, 10, max_iter, confidence, seed)
    else:
        parts = []
        if len(corr_src) > 1:
            offs = np.concatenate([[0], np.cumsum(prob_len[P:])]).tolist()
            parts = [B.ransac_batch(corr_src[1], corr_tgt[1], offs, max_corr, 10, max_iter, confidence, seed)]
        first = vanilla.result()              # cs_ransac_batch returns with its stream drained
        cur = torch.cuda.current_stream(dev)
        for t in first:
            t.record_stream(cur)
        T, inl, rmse, iters = (torch.cat([a] + [q[i] for q in parts]) for i, a in enumerate(first))

    # ---- 5. Chamfer of every estimate (utils/preprocess.py:39-48,67-70) -----------------------------
    cd = B.chamfer_1dir(xyz0, off0, xyz1, off1, prob_pair, prob_pair, T)

    # ---- 6. best hypothesis per pair: first minimum, vanilla first (utils/symmetry.py:322-324) ----
    cd_h = to_host(cd)[0]
    pair_h = np.asarray(prob_pair)
    best = np.arange(P)
    for j in range(P, len(pair_h)):
        p = pair_h[j]
        if cd_h[best[p]] > cd_h[j]:
            best[p] = j
    best_t = torch.from_numpy(best).to(dev)
    return SymPoseResult(T_best=T[best_t], cd_best=cd[best_t], T_ransac=T[:P], cd_ransac=cd[:P],
                         ok=ok, iters=iters, n_problems=len(prob_pair), T_all=T, cd_all=cd, inliers=inl,
                         prob_pair=prob_pair, prob_cfg=prob_cfg, best=best)
