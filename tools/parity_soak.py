"""One-off randomized parity soak of the two stages whose arithmetic changed in round 2: batched RANSAC (f64
evaluation) and the part-cut fit (sklearn-faithful k-means), GPU against the CPU oracle, bit for bit, on more
and more ragged cases than the test suite holds.   python tools/parity_soak.py [n_ransac] [n_clouds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from corsair_amd import backend as B, synth
from oracle import native
from tests.test_pins_cpu import real_clouds, _features

native.load()
dev = torch.device("cuda:0")
n_r = int(sys.argv[1]) if len(sys.argv) > 1 else 48
n_c = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rng = np.random.default_rng(2024)

# ---- RANSAC
probs, specs = [], []
for i in range(n_r):
    m = int(rng.integers(12, 9000))
    frac = float(rng.uniform(0.02, 0.9))
    noise = float(rng.choice([0.0, 0.005, 0.02, 0.08]))
    src = rng.uniform(-0.8, 0.8, (m, 3)).astype(np.float32)
    T = synth.random_pose(500 + i, max_trans=float(rng.uniform(0, 1.0)))
    tgt = (synth.apply_pose(src, T) + rng.normal(0, noise, (m, 3))).astype(np.float32)
    bad = rng.random(m) > frac
    tgt[bad] = rng.uniform(-1.2, 1.2, (int(bad.sum()), 3)).astype(np.float32)
    probs.append((src, tgt))
bad_total = 0
for max_corr, n_s, max_iter, seed in [(0.2, 10, 7000, 0), (0.05, 10, 5000, 3), (0.4, 6, 3000, 9), (0.03, 10, 20000, 1)]:
    off = np.concatenate([[0], np.cumsum([len(p[0]) for p in probs])]).tolist()
    S = torch.from_numpy(np.concatenate([p[0] for p in probs])).to(dev)
    D = torch.from_numpy(np.concatenate([p[1] for p in probs])).to(dev)
    T, inl, rmse, iters = (t.cpu().numpy() for t in B.ransac_batch(S, D, off, max_corr, n_s, max_iter, 0.999, seed))
    wT, winl, wrm, wit = native.ransac_batch(np.concatenate([p[0] for p in probs]), np.concatenate([p[1] for p in probs]),
                                             off, max_corr, n_s, max_iter, 0.999, seed)
    bad = int((inl != winl).sum() + (iters != wit).sum() + (T.reshape(len(probs), 16) != wT.reshape(len(probs), 16)).any(1).sum())
    rel = np.abs(rmse - wrm) / np.maximum(wrm, 1e-300)
    print("ransac max_corr %.2f n %d iters %d seed %d: %d problems, mismatches %d, max rel rmse diff %.1e, early exits %d"
          % (max_corr, n_s, max_iter, seed, len(probs), bad, float(np.nanmax(np.where(wrm > 0, rel, 0))), int((wit < max_iter).sum())),
          flush=True)
    bad_total += bad

# ---- part cut
pcs, _ = real_clouds()
clouds = _features(native, pcs[:n_c])
for K in (2, 4):
    F = torch.from_numpy(np.concatenate([c[0] for c in clouds])).to(dev)
    X = torch.from_numpy(np.concatenate([c[1] for c in clouds])).to(dev)
    off = np.concatenate([[0], np.cumsum([len(c[0]) for c in clouds])]).tolist()
    anchors = np.stack([rng.choice(len(c[0]), 100, replace=False) for c in clouds]).astype(np.int32)
    c_, n_, d_, e_ = (t.cpu().numpy() for t in B.symcut_fit(F, X, off, torch.from_numpy(anchors).to(dev), [K] * len(clouds)))
    bad = 0
    for i, (Fc, Xc) in enumerate(clouds):
        wc, wn, wd, we = native.symcut_fit(Fc, Xc, anchors[i], K)
        bad += int((c_[i] != wc).any() or (n_[i] != wn).any() or (d_[i] != wd).any() or (e_[i] != we).any())
    print("symcut K=%d: %d clouds x 100 anchors, clouds with any mismatch: %d" % (K, len(clouds), bad), flush=True)
    bad_total += bad
print("PARITY SOAK", "OK" if bad_total == 0 else "FAILED (%d)" % bad_total)
