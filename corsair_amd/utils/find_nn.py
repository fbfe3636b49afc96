"""Counterpart of the reference's utils/find_nn.py:34-49: exact k-NN in feature space.
`find_nn_cpu` / `find_knn_cpu` keep the reference's names (they are the call sites of
utils/eval_pose.py:67,70) but run the brute-force f64 HIP kernel instead of a SciPy KD-tree."""
import numpy as np

from .. import backend as B
from ._convert import to_dev


def find_knn(feat0, feat1, k, return_distance=False):
    f0, f1 = to_dev(feat0), to_dev(feat1)
    res = B.knn_feat(f0, [0, f0.shape[0]], f1, [0, f1.shape[0]], int(k), return_distance=return_distance)
    if return_distance:
        return res[0].cpu().numpy().astype(np.int64), res[1].cpu().numpy()
    return res.cpu().numpy().astype(np.int64)


def find_knn_cpu(feat0, feat1, k, return_distance=False):
    return find_knn(feat0, feat1, k, return_distance)


def find_nn_cpu(feat0, feat1, return_distance=False):
    res = find_knn(feat0, feat1, 1, return_distance)
    if return_distance:
        return res[0][:, 0], res[1][:, 0]
    return res[:, 0]
