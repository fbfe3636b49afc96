"""Soak of the concurrent paths (split RANSAC calls + pipelined rounds): the same sym_pose batch 30 times,
every output must be identical to the single-stream reference run."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.test_gpu_post import _engine_features
from corsair_amd import registration as R
gpu = torch.device("cuda:0")
F, X, off = _engine_features(gpu, [30, 31, 32, 33, 30, 31, 32, 33], [7, 8, 9, 10, None, None, None, None])
off0, off1 = off[:5], [o - off[4] for o in off[4:]]
bF, x0 = F[:off[4]].contiguous(), X[:off[4]].contiguous()
pF, x1 = F[off[4]:].contiguous(), X[off[4]:].contiguous()
def run():
    r = R.sym_pose_batch(bF, x0, off0, pF, x1, off1, [1, 2, 4, 1], 5, 0.2, 0, None, 100, 60000, 0.999, force_gate=True)
    return [t.cpu().numpy() for t in (r.T_best, r.cd_best, r.T_ransac, r.cd_ransac, r.iters)]
os.environ["CORSAIR_SPLIT_RANSAC"] = "0"; os.environ["CS_RANSAC_OVERLAP"] = "0"
want = run()
os.environ["CORSAIR_SPLIT_RANSAC"] = "1"; os.environ["CS_RANSAC_OVERLAP"] = "1"
bad = 0
for i in range(30):
    got = run()
    ok = all(np.array_equal(a, b) for a, b in zip(want, got))
    bad += not ok
    if i % 10 == 9: print("iteration", i + 1, "mismatches so far", bad, flush=True)
print("SOAK", "OK" if bad == 0 else "FAILED", "iters", want[4].tolist())
