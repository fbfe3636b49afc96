"""Pairwise Chamfer table, counterpart of the reference's utils/pc_dist.py:45-99:
`chamfer(pc0, pc1)` = mean NN distance pc1->pc0 + mean NN distance pc0->pc1, `compute_dist(pcs)` =
symmetric [C,C] table with 200 on the diagonal (np.eye*100 then table += table.T; the consumer zeroes
the diagonal, datasets/ScannetDataset.py:65-66).  All C(C-1) directed problems run as batched
cs_chamfer_1dir launches (f64 distances; the reference's dense torch version is f32)."""
import numpy as np
import torch

from .. import backend as B
from ._convert import to_dev


def chamfer(pc0, pc1):
    """Two-directional Chamfer distance of two clouds."""
    a, b = to_dev(pc0), to_dev(pc1)
    X = torch.cat([a, b])
    off = [0, a.shape[0], a.shape[0] + b.shape[0]]
    I = torch.eye(4, dtype=torch.float32, device=a.device).repeat(2, 1, 1)
    d = B.chamfer_1dir(X, off, X, off, [0, 1], [1, 0], I)
    return float(d.sum().cpu())


def compute_dist(pcs, chunk=32768):
    """Pair-wise Chamfer distance matrix within a set of clouds (list of [n_i,3] arrays)."""
    C = len(pcs)
    dev_pcs = [to_dev(p) for p in pcs]
    X = torch.cat(dev_pcs)
    off = np.concatenate([[0], np.cumsum([p.shape[0] for p in dev_pcs])]).tolist()
    iu, ju = np.triu_indices(C, k=1)
    src = np.concatenate([iu, ju]).astype(np.int32)
    tgt = np.concatenate([ju, iu]).astype(np.int32)
    out = np.zeros(len(src), np.float64)
    I1 = torch.eye(4, dtype=torch.float32, device=X.device)
    for s in range(0, len(src), chunk):
        e = min(len(src), s + chunk)
        T = I1.repeat(e - s, 1, 1)
        out[s:e] = B.chamfer_1dir(X, off, X, off, src[s:e].tolist(), tgt[s:e].tolist(), T).cpu().numpy()
    table = np.eye(C) * 100
    half = len(iu)
    table[iu, ju] = out[:half] + out[half:]
    table += table.T
    return table
