"""Randomized equivalence soak of the kernels that changed in round 5, each against its other implementation on the same
inputs (all selectable per call through the environment):
  * kernel maps: cs_coordmap_pyramid + k_level_maps (LDS, bucketised) vs chained create / stride + the global-table kernel
    (CS_PYRAMID=0 is read once per process, so the chained side calls create / stride directly; CS_KMAP_GLOBAL per call);
  * Chamfer: f16 matrix-core ranking (+ fallback) vs the f64 matrix-pipe kernel (CS_CHAMFER_F16 per call) -- bit-equal;
  * 16-d k-NN: threshold pass + shortlist vs shortlist alone (CS_KNN_TWOPASS per call), with and without labels;
  * descriptor top-k: f16 shortlist + re-score through LDS vs the exact f64 slab path (CS_TOPK_MFMA=0 per call).
python tools/soak_r5.py [n]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from corsair_amd import _lib, backend as B, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
dev = torch.device("cuda:0")
rng = np.random.default_rng(515)
bad = 0

# ---- kernel maps
def specs(c):
    return [(c[0], c[0]), (c[0], c[1]), (c[1], c[1]), (c[1], c[2]), (c[2], c[2]), (c[2], c[3]), (c[3], c[3]),
            (c[3], c[2], 3, True), (c[2], c[1], 3, True), (c[1], c[0], 3, True)]
for it in range(n):
    nb = int(rng.integers(1, 40))
    parts = []
    for b in range(nb):
        kind = rng.integers(0, 10)
        if kind == 0:
            g = np.zeros((0, 3), np.int64)                               # empty sample
        elif kind == 1 and it % 3 == 0:
            g = rng.integers(-30, 30, (int(rng.integers(16000, 30000)), 3))   # too many rows for the LDS table
        elif kind == 2:
            g = rng.integers(-600, 600, (int(rng.integers(2, 400)), 3)) * int(rng.choice([1, 3]))   # wide box
        else:
            pc = synth.make_cloud(int(rng.integers(0, 200)), 15000)[: int(rng.integers(300, 15000))]
            g = np.floor(pc / float(rng.choice([0.02, 0.03, 0.05]))).astype(np.int64)
        g = g[np.sort(np.unique(g, axis=0, return_index=True)[1])]
        parts.append(np.concatenate([np.full((len(g), 1), b), g], 1))
    coords = torch.from_numpy(np.concatenate(parts).astype(np.int32)).to(dev)
    if coords.shape[0] == 0:
        continue
    pyr = B.CoordMap.pyramid(coords, 4, nb if it % 2 == 0 else 0)
    c1 = B.CoordMap.create(coords, 1)
    chain = [c1, c1.stride(2)]
    chain.append(chain[-1].stride(2))
    chain.append(chain[-1].stride(2))
    fused = B.KernelMap.build_many(specs(pyr))
    os.environ["CS_KMAP_GLOBAL"] = "1"
    glob = B.KernelMap.build_many(specs(chain))
    del os.environ["CS_KMAP_GLOBAL"]
    ok = all(torch.equal(a.coords, b.coords) for a, b in zip(pyr, chain))
    ok = ok and all(a.num_pairs == b.num_pairs and torch.equal(a.table(), b.table()) for a, b in zip(fused, glob))
    if not ok:
        bad += 1
        print("kernel maps: batch", it, "differs")
print("kernel maps: %d random batches, %d mismatches so far" % (n, bad))

# ---- Chamfer
st = (ctypes.c_uint64 * 2)()
os.environ["CS_CHAMFER_STATS"] = "1"
_lib.load().cs_chamfer_f16_stats(st, 1)
for it in range(n):
    scale = float(rng.choice([0.05, 0.3, 1.0, 1.0, 5.0, 40.0]))
    clouds = []
    for c in range(int(rng.integers(2, 6))):
        pc = synth.make_cloud(int(rng.integers(0, 200)), 15000)[: int(rng.integers(1, 9000))] * scale
        if rng.random() < 0.3:   # near-duplicates
            pc = np.concatenate([pc, pc[: len(pc) // 3] + rng.normal(0, 1e-7 * scale, (len(pc) // 3, 3))])
        clouds.append(pc.astype(np.float32))
    off = np.concatenate([[0], np.cumsum([len(c) for c in clouds])]).tolist()
    X = torch.from_numpy(np.concatenate(clouds)).to(dev)
    P = int(rng.integers(1, 12))
    sseg = [int(v) for v in rng.integers(0, len(clouds), P)]
    tseg = [int(v) for v in rng.integers(0, len(clouds), P)]
    Ts = np.stack([synth.random_pose(1000 + 20 * it + p, max_trans=0.2 * scale).astype(np.float32) for p in range(P)])
    T = torch.from_numpy(Ts).to(dev)
    a = B.chamfer_1dir(X, off, X, off, sseg, tseg, T).cpu().numpy()
    os.environ["CS_CHAMFER_F16"] = "0"
    b = B.chamfer_1dir(X, off, X, off, sseg, tseg, T).cpu().numpy()
    del os.environ["CS_CHAMFER_F16"]
    if not np.array_equal(a, b):
        bad += 1
        print("chamfer: case", it, "differs", np.abs(a - b).max())
_lib.load().cs_chamfer_f16_stats(st, 0)
print("chamfer: %d random calls, f16 tiles %d, recomputed by the f64 kernel %d, %d mismatches so far" % (n, int(st[0]), int(st[1]), bad))

# ---- 16-d k-NN
for it in range(n):
    P = int(rng.integers(1, 6))
    nq = [int(rng.integers(1, 4000)) for _ in range(P)]
    nt = [int(rng.integers(6, 6000)) for _ in range(P)]
    qoff = np.concatenate([[0], np.cumsum(nq)]).tolist()
    toff = np.concatenate([[0], np.cumsum(nt)]).tolist()
    sc = float(rng.choice([0.01, 1.0, 1.0, 30.0]))
    def feats(N):
        x = rng.normal(size=(N, 16)).astype(np.float32)
        if it % 2 == 0:
            x /= np.linalg.norm(x, axis=1, keepdims=True)
        return x * sc
    Qn, Tn = feats(qoff[-1]), feats(toff[-1])
    if it % 4 == 1:
        Tn[: len(Tn) // 5] = Tn[len(Tn) // 5: 2 * (len(Tn) // 5)]     # exact duplicates: ties by row
    Q, T = torch.from_numpy(Qn).to(dev), torch.from_numpy(Tn).to(dev)
    k = int(rng.choice([1, 5, 6]))
    kw = {}
    if it % 3 == 2:
        kw = dict(qlabel=torch.from_numpy(rng.integers(0, 4, qoff[-1]).astype(np.int32)).to(dev),
                  tlabel=torch.from_numpy(rng.integers(0, 4, toff[-1]).astype(np.int32)).to(dev),
                  perm=torch.from_numpy(np.tile(np.arange(8, dtype=np.int32), (P, 1))).to(dev))
    try:
        ia, da = B.knn_feat(Q, qoff, T, toff, k, return_distance=True, **kw)
        os.environ["CS_KNN_TWOPASS"] = "0"
        ib, db = B.knn_feat(Q, qoff, T, toff, k, return_distance=True, **kw)
    finally:
        os.environ.pop("CS_KNN_TWOPASS", None)
    if not (torch.equal(ia, ib) and torch.equal(da, db)):
        bad += 1
        print("knn: case", it, "differs")
print("k-NN: %d random calls, %d mismatches in total" % (n, bad))

# ---- descriptor top-k: f16 shortlist + LDS re-score (round 5 rewrite) vs the exact f64 slab path (CS_TOPK_MFMA=0 per call)
st = (ctypes.c_uint64 * 2)()
_lib.load().cs_l2_topk_stats(st, 1)
for it in range(n):
    d = int(rng.choice([64, 128, 256, 256, 512]))
    nx = int(rng.choice([64, 65, 652, 830, 5000, int(rng.integers(6000, 120000))]))
    nq = int(rng.choice([1, 31, 32, 257, int(rng.integers(2, 3000))]))
    k = int(rng.integers(1, min(10, nx) + 1))
    X = synth.make_descriptors(nx, d, seed=7000 + it)
    Qd = synth.make_descriptors(nq, d, seed=8000 + it)
    if it % 3 == 1:
        X[: nx // 4] = X[nx // 4: 2 * (nx // 4)]                # exact duplicates: ties by row
    if it % 3 == 2:
        Qd[: min(nq, nx)] = X[: min(nq, nx)]                    # queries that ARE catalog rows (distance 0)
    sc = float(rng.choice([1.0, 1.0, 0.02, 20.0]))
    Xt, Qt = torch.from_numpy(X * sc).to(dev), torch.from_numpy(Qd * sc).to(dev)
    cat = B.TopkCatalog(Xt) if it % 2 == 0 else Xt
    try:
        os.environ["CS_TOPK_MFMA"] = "16"       # (unset, the f16 path is taken from nq * nx >= 2^24 on)
        ia, da = B.l2_topk(Qt, cat, k, True)
        os.environ["CS_TOPK_MFMA"] = "0"
        ib, db = B.l2_topk(Qt, Xt, k, True)
    finally:
        os.environ.pop("CS_TOPK_MFMA", None)
    if not (torch.equal(ia, ib) and torch.equal(da, db)):
        bad += 1
        print("topk: case", it, "differs (d %d nx %d nq %d k %d)" % (d, nx, nq, k))
_lib.load().cs_l2_topk_stats(st, 0)
print("top-k: %d random calls, f16 shortlist queries %d, re-done by the f64 path %d, %d mismatches in total" % (n, int(st[0]), int(st[1]), bad))
sys.exit(1 if bad else 0)
