"""Randomised agreement of the fast paths with the exhaustive / exact-only paths of the same library
(both through the C ABI; the exhaustive paths are the ones pinned against the oracle in test_gpu_post.py):
ragged segment sizes around the tile / stage boundaries of the kernels, unsorted part labels, parts
without targets, degenerate problems."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _feat(rng, n, scale=1.0):
    f = rng.standard_normal((n, 16)).astype(np.float32)
    return (scale * f / np.maximum(np.linalg.norm(f, axis=1, keepdims=True), 1e-6)).astype(np.float32)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_knn_f16_path_equals_exhaustive_on_ragged_labelled_problems(gpu, monkeypatch, seed):
    from corsair_amd import backend as B

    rng = np.random.default_rng(100 + seed)
    # segment sizes around the kernels' tile sizes (16/32-row tiles, 192/256-row stages, 256-query tiles)
    sizes = [(257, 193), (1, 1), (31, 33), (512, 191), (0, 50), (300, 0), (255, 385), (64, 5), (700, 1500)]
    qoff = np.concatenate([[0], np.cumsum([a for a, _ in sizes])]).tolist()
    toff = np.concatenate([[0], np.cumsum([b for _, b in sizes])]).tolist()
    scale = [1.0, 3.0, 0.05][seed]
    Q = torch.from_numpy(_feat(rng, qoff[-1], scale)).to(gpu)
    T = torch.from_numpy(_feat(rng, toff[-1], scale)).to(gpu)
    # labels: unsorted on the query side, some without a part, some parts without targets
    lq = rng.integers(-1, 5, qoff[-1]).astype(np.int32)
    lt = rng.integers(0, 4, toff[-1]).astype(np.int32)
    lt[toff[3]:toff[4]] = 2                      # segment 3: only part 2 exists
    lt[rng.random(toff[-1]) < 0.05] = 11         # a few targets without a part
    qseg = [0, 2, 3, 6, 8, 8, 7]
    tseg = [0, 2, 3, 6, 8, 8, 7]
    perms = [[(i + s) % 4 for i in range(4)] + [-3] * 4 for s in range(len(qseg))]
    perm = torch.tensor(perms, dtype=torch.int32, device=gpu)

    def run(k, labelled):
        if labelled:
            return B.knn_feat(Q, qoff, T, toff, k, qseg=qseg, tseg=tseg, qlabel=torch.from_numpy(lq).to(gpu),
                              tlabel=torch.from_numpy(lt).to(gpu), perm=perm, return_distance=True)
        return B.knn_feat(Q, qoff, T, toff, k, return_distance=True)

    for k in (1, 5, 6):
        for labelled in (False, True):
            monkeypatch.setenv("CS_KNN_MFMA", "0")
            wi, wd = run(k, labelled)
            for mode in ("1", "64"):
                monkeypatch.setenv("CS_KNN_MFMA", mode)
                gi, gd = run(k, labelled)
                assert torch.equal(gi, wi), (k, labelled, mode)
                assert torch.equal(gd, wd), (k, labelled, mode)


@pytest.mark.parametrize("seed", [0, 1])
def test_ransac_prefilter_equals_exact_on_ragged_batches(gpu, monkeypatch, seed):
    from corsair_amd import backend as B, synth

    rng = np.random.default_rng(7 + seed)
    ms = [191, 192, 193, 9, 10, 11, 3000, 0, 1537, 384, 6001]
    src, tgt = [], []
    for i, m in enumerate(ms):
        s = rng.uniform(-1, 1, (m, 3)).astype(np.float32)
        T = synth.random_pose(40 + i, max_trans=0.4)
        t = synth.apply_pose(s, T) + rng.normal(0, 0.01, (m, 3)).astype(np.float32)
        bad = rng.random(m) > [0.05, 0.3][seed]
        t[bad] = rng.uniform(-1.2, 1.2, (int(bad.sum()), 3)).astype(np.float32)
        if i == 6:
            t[:] = t[0]                           # degenerate: all targets identical
        src.append(s)
        tgt.append(t.astype(np.float32))
    off = np.concatenate([[0], np.cumsum(ms)]).tolist()
    S = torch.from_numpy(np.concatenate(src)).to(gpu)
    D = torch.from_numpy(np.concatenate(tgt)).to(gpu)

    def run(max_iter):
        return [t.cpu().numpy() for t in B.ransac_batch(S, D, off, 0.15, 10, max_iter, 0.999, 11 + seed)]

    for max_iter in (700, 5000, 33000):
        monkeypatch.setenv("CS_RANSAC_PREFILTER", "0")
        exact = run(max_iter)
        monkeypatch.setenv("CS_RANSAC_PREFILTER", "1")
        fast = run(max_iter)
        for a, b in zip(exact, fast):
            assert np.array_equal(a, b, equal_nan=True), max_iter


def test_chamfer_mfma_equals_exhaustive_on_ragged_problems(gpu, monkeypatch):
    from corsair_amd import backend as B, synth

    rng = np.random.default_rng(21)
    ns = [1, 15, 16, 17, 63, 64, 65, 511, 512, 513, 2000]
    clouds = [rng.uniform(-1, 1, (n, 3)).astype(np.float32) for n in ns]
    off = np.concatenate([[0], np.cumsum(ns)]).tolist()
    X = torch.from_numpy(np.concatenate(clouds)).to(gpu)
    P = 24
    a = rng.integers(0, len(ns), P).tolist()
    b = rng.integers(0, len(ns), P).tolist()
    Ts = torch.from_numpy(np.stack([synth.random_pose(60 + i, max_trans=0.3).astype(np.float32) for i in range(P)])).to(gpu)
    monkeypatch.setenv("CS_CHAMFER_MFMA", "0")
    want = B.chamfer_1dir(X, off, X, off, a, b, Ts).cpu().numpy()
    whd = B.hausdorff_1dir(X, off, X, off, a, b, Ts).cpu().numpy()
    monkeypatch.setenv("CS_CHAMFER_MFMA", "1")
    got = B.chamfer_1dir(X, off, X, off, a, b, Ts).cpu().numpy()
    ghd = B.hausdorff_1dir(X, off, X, off, a, b, Ts).cpu().numpy()
    assert np.allclose(got, want, rtol=1e-12, atol=0)
    assert np.array_equal(ghd, whd)              # a maximum of individually exact distances
