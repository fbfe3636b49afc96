"""SURVEY 8f 'next' rows on the GPU: geometric symmetry label + ShapeNet-style registration eval,
pairwise Chamfer table, both against SciPy KD-tree restatements of the reference code."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ring_cloud(n_fold, n=3000, seed=0, jitter=0.0):
    """Object with exact n_fold rotational symmetry about the y axis (one random blob, replicated)."""
    from corsair_amd.synth import euler2mat

    rng = np.random.default_rng(seed)
    blob = rng.normal(0, 0.05, (n // n_fold, 3)) + np.array([0.6, 0.0, 0.1])
    blob[:, 1] = rng.uniform(-0.4, 0.4, len(blob))
    parts = [blob @ euler2mat(0, 2 * np.pi * i / n_fold, 0).T for i in range(n_fold)]
    pc = np.concatenate(parts)
    return pc + rng.normal(0, jitter, pc.shape) if jitter else pc


def _ref_symmetry_label(pc, thr):
    """evaluation-shapenet.py:122-155 with SciPy KD-trees (vectorised queries)."""
    from scipy.spatial import KDTree

    from corsair_amd.synth import euler2mat

    tree = KDTree(pc)
    for s in [12, 8, 6, 4, 3, 2, 1]:
        ok = True
        for i in range(1, s // 2 + 1):
            rot = pc @ euler2mat(0, i * 2 * np.pi / s, 0).T
            err = max(KDTree(rot).query(pc)[0].max(), tree.query(rot)[0].max())
            if err > thr:
                ok = False
                break
        if ok:
            return s
    return 0


@pytest.mark.parametrize("n_fold", [1, 2, 3, 4, 6])
def test_get_symmetry_label(gpu, n_fold):
    from corsair_amd import shapenet_eval as S

    # an f32 file like the ShapeNetPC15k clouds: load_pc normalises it in place in f32 (evaluation-shapenet.py:70-76),
    # the reference's test then rotates with an f64 matrix and queries f64 KD-trees -- exactly _ref_symmetry_label(pc)
    pc = S.load_pc(_ring_cloud(n_fold, seed=n_fold).astype(np.float32))
    assert pc.dtype == np.float32
    want = _ref_symmetry_label(pc, 0.1)
    got = S.get_symmetry_label(pc, 0.1)
    assert got == want
    if n_fold in (2, 3, 4, 6):
        assert got % n_fold == 0 or got == n_fold


def test_hausdorff_matches_kdtree(gpu):
    from scipy.spatial import KDTree

    from corsair_amd import shapenet_eval as S

    rng = np.random.default_rng(1)
    a = rng.uniform(-1, 1, (1200, 3)).astype(np.float32)
    b = rng.uniform(-1, 1, (900, 3)).astype(np.float32)
    want = max(KDTree(b).query(a.astype(np.float64))[0].max(), KDTree(a).query(b.astype(np.float64))[0].max())
    assert S.chamfer_max(a, b) == pytest.approx(want, rel=1e-12)


def test_pairwise_chamfer_table(gpu):
    """utils/pc_dist.py:45-99: table[i,j] = two-directional Chamfer, 200 on the diagonal."""
    from scipy.spatial import KDTree

    from corsair_amd.utils import pc_dist

    rng = np.random.default_rng(2)
    pcs = [rng.uniform(-1, 1, (n, 3)).astype(np.float32) for n in (300, 257, 512, 64, 100)]
    table = pc_dist.compute_dist(pcs)
    assert table.shape == (5, 5) and np.allclose(np.diag(table), 200.0) and np.allclose(table, table.T)
    for i in range(5):
        for j in range(i + 1, 5):
            a, b = pcs[i].astype(np.float64), pcs[j].astype(np.float64)
            want = KDTree(a).query(b)[0].mean() + KDTree(b).query(a)[0].mean()
            assert table[i, j] == pytest.approx(want, rel=1e-10)
    assert pc_dist.chamfer(pcs[0], pcs[1]) == pytest.approx(table[0, 1], rel=1e-12)


def test_shapenet_style_registration_eval(gpu):
    """evaluation-shapenet.py:242-343 end to end on two synthetic models x 2 poses; every result
    equals the per-pair oracle sym_pose on the same features."""
    from corsair_amd import harness, registration as R, shapenet_eval as S, synth
    from oracle import post

    sd, emb = synth.make_state_dicts(31)
    pipe = harness.Pipeline(sd, emb, device=gpu)
    cfg = S.Config(n_poses_per_model=2, ransac_max_iter=2000, max_translation=0.5)
    clouds = [synth.make_cloud(50, 6000), _ring_cloud(4, n=6000, seed=3, jitter=0.002)]
    res = S.evaluate(pipe, clouds, cfg, pairs_per_batch=4)
    assert len(res) == 4
    assert [r["symmetry_label"] for r in res] == [S.get_symmetry_label(S.load_pc(clouds[0]), 0.1)] * 2 + \
        [S.get_symmetry_label(S.load_pc(clouds[1]), 0.1)] * 2
    assert res[2]["symmetry_label"] in (4, 8, 12)
    for r in res:
        assert r["chamfer_dist_sym"] <= r["chamfer_dist_ransac"]
        assert np.isfinite(r["rre_sym"]) and 0 <= r["rre_sym"] <= np.pi
    tab = S.threshold_table(res)
    assert set(tab) == {"ransac", "sym"} and 0.0 <= tab["sym"]["rre<=45"] <= 1.0
    # oracle re-run of the first pair on the GPU features
    rng = np.random.default_rng(cfg.random_seed)
    pc = S.load_pc(clouds[0])
    pose = S.generate_random_pose(cfg, rng)
    assert pc.dtype == np.float32                       # synth clouds are f32 like the ShapeNetPC15k files
    posed = pc @ pose[:3, :3].T + pose[:3, [3]].T       # f64, quantised in f64 (evaluation-shapenet.py:97-119)
    es = pipe.embed_groups([(torch.from_numpy(pc).to(gpu), [0, len(pc)]),
                            (torch.from_numpy(posed).to(gpu), [0, len(posed)])], cfg.voxel_size)
    o = es.offsets
    F, X = es.F.cpu().numpy(), es.origin.cpu().numpy()
    a0 = R.draw_anchors(o[1], 100, 0)
    a1 = R.draw_anchors(o[2] - o[1], 100, 1)
    Tb, cdb, Tr, cdr, ok = post.sym_pose(F[:o[1]], X[:o[1]], F[o[1]:], X[o[1]:], res[0]["symmetry_label"], 5,
                                         cfg.max_corr, 0, a0, a1, cfg.ransac_max_iter, cfg.ransac_confidence)
    assert np.array_equal(res[0]["T_est_ransac"], Tr) and np.array_equal(res[0]["T_est_sym"], Tb)
    assert res[0]["sym_success"] == ok


def test_sharding_exchange_runs_on_rccl(gpu):
    """The one exchange of the multi-GPU path (sharding.all_gather_counts / all_gather_embedded, bench.py's
    reduce / gather helpers) through the REAL backend: torch.distributed "nccl" = RCCL, here with a
    communicator of one rank on the one GPU of the test box (device tensors, int64 / f32 / f64 payloads,
    padded ragged all-gathers).  The multi-rank logic itself is covered on gloo (tests/test_sharding_gloo.py);
    this test catches what gloo cannot: a host tensor or an unsupported dtype handed to RCCL."""
    import os
    import socket

    import torch.distributed as dist

    from corsair_amd import sharding
    from corsair_amd.harness import EmbeddedSet

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=gpu)
    try:
        vox = sharding.all_gather_counts(dist, [0, 2, 1], [10, 30, 20], 3, 1)
        assert vox.tolist() == [10, 20, 30]
        F = torch.arange(7 * 16, dtype=torch.float32, device=gpu).reshape(7, 16)
        eset = EmbeddedSet(F, F[:, :3].contiguous(), [0, 3, 7], torch.ones((2, 256), device=gpu))
        (got,) = sharding.all_gather_embedded(dist, eset, 1)
        assert got.offsets == [0, 3, 7] and torch.equal(got.F, F) and torch.equal(got.desc, eset.desc)
        t = torch.tensor([1.5, 2.5], device=gpu, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.barrier()
        assert t.tolist() == [1.5, 2.5]
    finally:
        dist.destroy_process_group()


def test_bench_group_handle_runs_the_exchange_on_an_rccl_subgroup(gpu):
    """bench.py's process layout: the default group is gloo (control plane), the data collectives run on an
    RCCL group wrapped in bench.GroupDist -- the object corsair_amd.sharding is handed.  One rank, the one GPU."""
    import importlib.util
    import os
    import socket

    import torch.distributed as dist

    from corsair_amd import sharding
    from corsair_amd.harness import EmbeddedSet

    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        gd = bench.GroupDist(dist, dist.new_group(backend="nccl", device_id=gpu))
        assert gd.get_backend() == "nccl" and dist.get_backend() == "gloo"
        F = torch.arange(5 * 16, dtype=torch.float32, device=gpu).reshape(5, 16)
        eset = EmbeddedSet(F, F[:, :3].contiguous(), [0, 2, 5], torch.ones((2, 256), device=gpu))
        (got,) = sharding.all_gather_embedded(gd, eset, 1)
        assert got.offsets == [0, 2, 5] and torch.equal(got.F, F) and got.F.is_cuda
        assert sharding.all_gather_counts(gd, [1, 0], [7, 9], 2, 1).tolist() == [9, 7]
        t = torch.tensor([3.0], device=gpu, dtype=torch.float64)
        gd.all_reduce(t, op=gd.ReduceOp.MAX)
        gd.barrier()
        assert t.item() == 3.0
    finally:
        dist.destroy_process_group()


def _sharded_eval_inputs():
    from corsair_amd import synth

    C, Q, n = 12, 10, 3000
    catalog = [synth.make_cloud(c, 15000)[:n] for c in range(C)]
    queries = [synth.apply_pose(synth.make_cloud(q % C, 15000)[15000 - n:], synth.random_pose(q, max_trans=0.0), np.float64)
               for q in range(Q)]
    rng = np.random.default_rng(4)
    table = rng.random((C, C))
    table = table + table.T
    np.fill_diagonal(table, 0.0)
    syms = np.ones(C, np.int32)
    syms[[1, 4]] = [2, 4]
    return (catalog, queries, np.arange(Q) % C, table, np.stack([synth.random_pose(q, max_trans=0.0) for q in range(Q)]),
            np.stack([np.eye(4)] * C), syms)


def _sharded_eval_rank(rank, world, port, out_dir):
    import os
    import pickle

    import torch.distributed as dist

    from corsair_amd import harness, sharding, synth

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)     # two ranks share the box's one GPU: no RCCL
    try:
        sd, emb = synth.make_state_dicts(31)
        pipe = harness.Pipeline(sd, emb, device=torch.device("cuda:0"),
                                config=harness.Config(n_points=3000, ransac_max_iter=3000, batch_size=4))
        res = sharding.run_eval_sharded(pipe, dist, rank, world, *_sharded_eval_inputs(), "chair", True,
                                        os.path.join(out_dir, "cache"), True)
        with open(os.path.join(out_dir, f"r{rank}.pkl"), "wb") as f:
            pickle.dump((res.stat, res.per_query, res.report), f)
    finally:
        dist.destroy_process_group()


def test_sharded_evaluation_two_ranks_equals_single_rank_on_the_gpu(gpu, tmp_path):
    """sharding.run_eval_sharded with the REAL pipeline: two ranks (sharing the one GPU of the test box, exchange over
    gloo) produce the single-rank harness.run_eval result bit for bit -- descriptors, retrieval statistics, the nine
    per-query arrays in query order -- and rank 0 writes the result cache once (evaluation.py:334-383,421-441)."""
    import pickle
    import socket

    import torch.multiprocessing as mp

    from corsair_amd import cache, harness, synth

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_sharded_eval_rank, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    sd, emb = synth.make_state_dicts(31)
    pipe = harness.Pipeline(sd, emb, device=gpu, config=harness.Config(n_points=3000, ransac_max_iter=3000, batch_size=4))
    want = harness.run_eval(pipe, *_sharded_eval_inputs(), "chair", True, force_gate=True)
    for r in range(2):
        with open(tmp_path / f"r{r}.pkl", "rb") as f:
            stat, per_query, report = pickle.load(f)
        assert stat == want.stat and report == want.report
        for k in cache.NAMES:
            assert np.array_equal(per_query[k], want.per_query[k]), (r, k)
    loaded = cache.load_results(str(tmp_path / "cache"), "chair", True)
    assert all(np.array_equal(loaded[k], want.per_query[k]) for k in cache.NAMES)
