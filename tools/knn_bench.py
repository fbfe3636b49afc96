"""Time cs_knn_feat on the bench-like shape (32 problems of ~5.8 k x 5.8 k 16-d rows)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from corsair_amd import backend as B
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
P = int(sys.argv[1]) if len(sys.argv) > 1 else 32
n = [int(rng.integers(4500, 7200)) for _ in range(P)]
off = np.concatenate([[0], np.cumsum(n)]).tolist()
def feats(N):
    x = rng.normal(size=(N, 16)).astype(np.float32)
    return torch.from_numpy(x / np.linalg.norm(x, axis=1, keepdims=True)).to(dev)
Q, T = feats(off[-1]), feats(off[-1])
for rep in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5):
        idx = B.knn_feat(Q, off, T, off, 5)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
    pairs = sum(a * a for a in n)
    print("knn %.3f ms per call, %.2f Gpairs/s, checksum %d" % (dt * 1e3, pairs / dt / 1e9, int(idx.sum())))
import ctypes
from corsair_amd import _lib
os.environ["CS_KNN_STATS"] = "1"
st = (ctypes.c_uint64 * 2)()
_lib.load().cs_knn_shortlist_stats(st, 1)
B.knn_feat(Q, off, T, off, 5)
_lib.load().cs_knn_shortlist_stats(st, 0)
print("shortlist stats [queries, recomputed exhaustively]:", [int(v) for v in st])
