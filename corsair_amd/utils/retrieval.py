"""Counterpart of the reference's utils/retrieval.py:139-177 (Scan2CAD retrieval metrics).

The f64 cdist + full argsort of every row is replaced by cs_l2_topk (exact f64 distances, ties to
the smaller index); only the first pos_n ranks are ever consumed by the metric."""
import numpy as np

from .. import backend as B
from ._convert import to_dev


def predicted_rank(scan_feats, lib_feats, top_n):
    """First top_n columns of np.argsort(cdist(scan_feats, lib_feats), 1)."""
    idx = B.l2_topk(to_dev(scan_feats), to_dev(lib_feats), int(top_n))
    return idx.cpu().numpy()


def scan2cad_retrieval_eval_rank(pred_rank, table, best_match, pos_n):
    """The statistics of scan2cad_retrieval_eval_dist (utils/retrieval.py:139-167) given the predicted ranking
    (only its first pos_n columns are consumed), all queries at once:
      precision   mean over queries of 100 * |pred[:pos_n] ∩ gt[:pos_n]| / pos_n,
      top1_error  mean of table[pred[0], gt[0]],   top1_predict / gt  the first entries,
    where gt = argsort(table[best_match]) (stable: ties -> smaller id).  The two means are LEFT-TO-RIGHT sums in
    query order divided by Q -- what Python's sum() over the reference's per-query lists evaluates -- so the
    figures equal the reference's to the last bit (np.cumsum adds sequentially; np.sum would add pairwise)."""
    table = np.asarray(table)
    pred = np.asarray(pred_rank)[:, :max(int(pos_n), 1)]
    best_match = np.asarray(best_match).astype(np.int64)
    n_q, n_lib = len(best_match), table.shape[1]
    gt_head = np.argsort(table[best_match, :], 1, kind="stable")[:, :max(int(pos_n), 1)]
    rows = np.arange(n_q)[:, None]
    member = np.zeros((n_q, n_lib), dtype=bool)          # member[q, c]: CAD c is among query q's pos_n GT neighbours
    member[rows, gt_head[:, :pos_n]] = True
    hits = member[rows, pred[:, :pos_n]].sum(axis=1)
    precision = 100.0 * hits / pos_n
    top1_error = table[pred[:, 0], gt_head[:, 0]]
    return {
        "precision": float(np.cumsum(precision)[-1]) / n_q,
        "top1_error": float(np.cumsum(top1_error)[-1]) / n_q,
        "top1_predict": [int(v) for v in pred[:, 0]],
        "gt": [int(v) for v in gt_head[:, 0]],
    }


def scan2cad_retrieval_eval(scan_feats, lib_feats, best_match, table, pos_n):
    """Scan2cad retrieval using descriptors (utils/retrieval.py:170-177)."""
    rank = predicted_rank(scan_feats, lib_feats, max(int(pos_n), 1))
    return scan2cad_retrieval_eval_rank(rank, table, best_match, pos_n)
