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
    """scan2cad_retrieval_eval_dist (utils/retrieval.py:139-167) given the predicted ranking."""
    table = np.asarray(table)
    best_match = np.asarray(best_match).astype(np.int64)
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
    """Scan2cad retrieval using descriptors (utils/retrieval.py:170-177)."""
    rank = predicted_rank(scan_feats, lib_feats, max(int(pos_n), 1))
    return scan2cad_retrieval_eval_rank(rank, table, best_match, pos_n)
