"""Multi-process sharding logic on CPU (gloo, world_size 2 and 8 -- BASELINE.json configs[3] is the 8-GPU node):
interleaved and voxel-balanced catalog shards (also EMPTY ones: fewer items than ranks), the all-gather of the
embedded catalog and the reassembly into catalog order, the catalog-sharded top-k with ragged query counts, and the
whole sharded evaluation at the chair contract size (Q = 993, C = 652) -- the N > 1 path of bench.py (RCCL on the
GPU node)."""
import os
import socket

import numpy as np
import pytest
import torch


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _fake_set(ids):
    """EmbeddedSet whose contents encode the item id (item c has 3 + c % 4 voxels)."""
    from corsair_amd.harness import EmbeddedSet

    F, O, off, D = [], [], [0], []
    for c in ids:
        n = 3 + c % 4
        F.append(torch.full((n, 16), float(c)) + torch.arange(n)[:, None] * 0.01)
        O.append(torch.full((n, 3), -float(c)))
        off.append(off[-1] + n)
        D.append(torch.full((1, 256), float(c)))
    if not ids:
        return EmbeddedSet(torch.zeros((0, 16)), torch.zeros((0, 3)), [0], torch.zeros((0, 256)))
    return EmbeddedSet(torch.cat(F), torch.cat(O), off, torch.cat(D))


def _worker(rank, world, port, n_items, out_dir):
    import torch.distributed as dist

    from corsair_amd import sharding

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine = sharding.shard_ids(n_items, rank, world)
        full = sharding.gather_catalog(dist, _fake_set(mine), n_items, world)
        want = _fake_set(list(range(n_items)))
        ok = (full.offsets == want.offsets and torch.equal(full.F, want.F)
              and torch.equal(full.origin, want.origin) and torch.equal(full.desc, want.desc))
        # query sharding: every rank owns a disjoint, equally sized slice (weak scaling)
        sub = full.gather([n_items - 1, 0, 0])
        ok = ok and sub.offsets == [0, 3 + (n_items - 1) % 4, 6 + (n_items - 1) % 4, 9 + (n_items - 1) % 4]
        ok = ok and float(sub.desc[0, 0]) == n_items - 1
        with open(os.path.join(out_dir, f"rank{rank}.txt"), "w") as f:
            f.write("ok" if ok else "mismatch")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_items", [7, 8])
def test_catalog_all_gather_world2(tmp_path, n_items):
    import torch.multiprocessing as mp

    port = _free_port()
    mp.spawn(_worker, args=(2, port, n_items, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        assert (tmp_path / f"rank{r}.txt").read_text() == "ok"


def _worker8(rank, world, port, out_dir):
    """Several catalog sizes in ONE process group (a spawn of 8 interpreters per case would dominate the test):
    fewer items than ranks (ranks 3..7 hold EMPTY shards), exactly one per rank, a ragged 21, and -- through
    balanced_shards -- one giant item that fills a rank on its own while the others share the rest."""
    import torch.distributed as dist

    from corsair_amd import sharding

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        bad = []
        for n_items in (3, 8, 21):
            mine = sharding.shard_ids(n_items, rank, world)
            full = sharding.gather_catalog(dist, _fake_set(mine), n_items, world)
            want = _fake_set(list(range(n_items)))
            if not (full.offsets == want.offsets and torch.equal(full.F, want.F)
                    and torch.equal(full.origin, want.origin) and torch.equal(full.desc, want.desc)):
                bad.append("interleaved %d" % n_items)
        for n_items, giant in ((5, 1), (19, 4)):
            w = [100 + 7 * c for c in range(n_items)]
            w[giant] = 100000
            mine = sharding.shard_ids(n_items, rank, world)
            vox = sharding.all_gather_counts(dist, mine, [w[c] for c in mine], n_items, world)
            shards = sharding.balanced_shards(vox, world)
            if vox.tolist() != w or [giant] not in shards:
                bad.append("counts / giant alone %d" % n_items)
            full = sharding.gather_catalog(dist, _fake_set(shards[rank]), n_items, world, shards)
            want = _fake_set(list(range(n_items)))
            if not (full.offsets == want.offsets and torch.equal(full.F, want.F) and torch.equal(full.desc, want.desc)):
                bad.append("balanced %d" % n_items)
        with open(os.path.join(out_dir, f"rank{rank}.txt"), "w") as f:
            f.write("ok" if not bad else "mismatch: " + ", ".join(bad))
    finally:
        dist.destroy_process_group()


def test_catalog_all_gather_world8_with_empty_and_giant_shards(tmp_path):
    """configs[3]'s rank count: all_gather_embedded / gather_catalog with fewer items than ranks (empty shards) and
    with one item that outweighs everything else (VERDICT r4 missing #1)."""
    import torch.multiprocessing as mp

    mp.spawn(_worker8, args=(8, _free_port(), str(tmp_path)), nprocs=8, join=True)
    for r in range(8):
        assert (tmp_path / f"rank{r}.txt").read_text() == "ok"


def test_shard_ids_cover_everything():
    from corsair_amd import sharding

    for n in (0, 1, 5, 652):
        for world in (1, 2, 4, 8):
            ids = [i for r in range(world) for i in sharding.shard_ids(n, r, world)]
            assert sorted(ids) == list(range(n))
            order = sharding.global_order(n, world)
            assert [ids[j] for j in order] == list(range(n))


def test_balanced_shards_by_voxel_count():
    """shards balanced by voxel count (N1 varies 1.6 k - 8 k), deterministic, complete."""
    from corsair_amd import sharding

    rng = np.random.default_rng(5)
    w = rng.integers(1600, 8000, 652)
    for world in (1, 2, 4, 8):
        shards = sharding.balanced_shards(w, world)
        flat = [i for s in shards for i in s]
        assert sorted(flat) == list(range(652))
        assert [flat[j] for j in sharding.shard_order(shards)] == list(range(652))
        assert shards == sharding.balanced_shards(w, world)
        inter = [sharding.shard_ids(652, r, world) for r in range(world)]
        assert sharding.imbalance(w, shards) <= sharding.imbalance(w, inter) + 1e-12
        assert sharding.imbalance(w, shards) < 1.005      # interleaved: 1.07 at world 8
    assert sharding.balanced_shards([], 2) == [[], []]


def _count_worker(rank, world, port, out_dir):
    import torch.distributed as dist

    from corsair_amd import sharding

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 11
        mine = sharding.shard_ids(n, rank, world)
        vox = sharding.all_gather_counts(dist, mine, [100 + 7 * c for c in mine], n, world)
        shards = sharding.balanced_shards(vox, world)
        full = sharding.gather_catalog(dist, _fake_set(shards[rank]), n, world, shards)
        want = _fake_set(list(range(n)))
        ok = vox.tolist() == [100 + 7 * c for c in range(n)] and full.offsets == want.offsets \
            and torch.equal(full.F, want.F) and torch.equal(full.desc, want.desc)
        with open(os.path.join(out_dir, f"rank{rank}.txt"), "w") as f:
            f.write("ok" if ok else "mismatch")
    finally:
        dist.destroy_process_group()


def test_balanced_catalog_all_gather_world2(tmp_path):
    import torch.multiprocessing as mp

    mp.spawn(_count_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        assert (tmp_path / f"rank{r}.txt").read_text() == "ok"


def _exact_topk(q, x, k):
    """Reference local top-k: exact f64 squared distances (the oracle's fma chain), ties -> smaller index."""
    os.environ.setdefault("ORACLE_THREADS", "1")   # (several ranks x one OpenMP team per visible core would spin against each other)
    from oracle import native

    d2 = native.dist2_matrix(q.numpy(), x.numpy())
    idx = np.argsort(d2, axis=1, kind="stable")[:, :k]
    return torch.from_numpy(idx.astype(np.int64)), torch.from_numpy(np.take_along_axis(d2, idx, 1))


def _topk_worker(rank, world, port, out_dir):
    import torch.distributed as dist

    from corsair_amd import sharding

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(5)
        C, d, k, Q = 301, 24, 10, 17
        x = torch.randn(C, d, generator=g)
        x[37] = x[200]                      # exact ties across the shard boundary: the smaller global index wins
        x[151] = x[150]
        q_all = torch.randn(world * Q, d, generator=g)
        q_all[3] = x[200]
        first, last = sharding.catalog_shard(C, rank, world)
        q_mine = q_all[rank * Q:(rank + 1) * Q]
        ids, d2 = sharding.sharded_topk(dist, q_mine, lambda qq: _exact_topk(qq, x[first:last], k), first, k, rank, world)
        want_ids, want_d2 = _exact_topk(q_mine, x, k)
        ok = torch.equal(ids, want_ids) and torch.equal(d2, want_d2)
        with open(os.path.join(out_dir, f"topk{rank}.txt"), "w") as f:
            f.write("ok" if ok else "mismatch")
    finally:
        dist.destroy_process_group()


def test_sharded_topk_world2_equals_single_rank(tmp_path, oracle_native):
    """Catalog sharded over 2 ranks, queries all-gathered, per-shard lists merged by (squared distance, global
    id): ids and distances equal the unsharded top-k bit for bit, ties across the shard boundary included."""
    import torch.multiprocessing as mp

    port = _free_port()
    mp.spawn(_topk_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        assert (tmp_path / f"topk{r}.txt").read_text() == "ok"


def _topk_worker8(rank, world, port, out_dir):
    """World 8: ragged query counts (rank r brings (5 r) % 7 queries: rank 0 and rank 7 none), then a catalog with
    fewer rows than k per shard, then one with fewer rows than RANKS (empty shards)."""
    import torch.distributed as dist

    from corsair_amd import sharding

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        bad = []
        for case, (C, k) in enumerate(((301, 10), (43, 10), (5, 3))):
            g = torch.Generator().manual_seed(5 + case)
            d = 24
            x = torch.randn(C, d, generator=g)
            if C > 250:
                x[37] = x[200]                  # exact ties across shard boundaries: the smaller global index wins
                x[151] = x[150]
            counts = [(5 * r) % 7 for r in range(world)] if case != 1 else [4] * world
            q_all = torch.randn(sum(counts), d, generator=g)
            if C > 250:
                q_all[3] = x[200]
            first, last = sharding.catalog_shard(C, rank, world)
            lo = sum(counts[:rank])
            q_mine = q_all[lo:lo + counts[rank]]
            ids, d2 = sharding.sharded_topk(dist, q_mine, lambda qq: _exact_topk(qq, x[first:last], k), first, k, rank, world)
            want_ids, want_d2 = _exact_topk(q_mine, x, k)
            if not (ids.shape == (counts[rank], k) and torch.equal(ids, want_ids) and torch.equal(d2, want_d2)):
                bad.append("C=%d k=%d" % (C, k))
        with open(os.path.join(out_dir, f"topk{rank}.txt"), "w") as f:
            f.write("ok" if not bad else "mismatch: " + ", ".join(bad))
    finally:
        dist.destroy_process_group()


def test_sharded_topk_world8_ragged_queries_and_short_shards(tmp_path, oracle_native):
    """sharded_topk at configs[3]'s rank count with what the stress leg never produces: a different number of queries
    per rank (also none), shards shorter than k, shards with no rows."""
    import torch.multiprocessing as mp

    mp.spawn(_topk_worker8, args=(8, _free_port(), str(tmp_path)), nprocs=8, join=True)
    for r in range(8):
        assert (tmp_path / f"topk{r}.txt").read_text() == "ok"


def test_merge_topk_orders_by_distance_then_id():
    from corsair_amd import sharding

    d2a = torch.tensor([[1.0, 2.0, 2.0]], dtype=torch.float64)
    ida = torch.tensor([[7, 9, 11]])
    d2b = torch.tensor([[2.0, 2.0, float("inf")]], dtype=torch.float64)
    idb = torch.tensor([[3, 10, -1]])
    ids, d2 = sharding.merge_topk([d2a, d2b], [ida, idb], 4)
    assert ids.tolist() == [[7, 3, 9, 10]] and d2.tolist() == [[1.0, 2.0, 2.0, 2.0]]
    assert sharding.catalog_shard(10, 3, 4) == (9, 10) and sharding.catalog_shard(10, 0, 4) == (0, 3)


# ---- the whole evaluation on 2 ranks ends in ONE result set equal to the single-rank run ----------
class _FakePipe:
    """Duck-typed harness.Pipeline without a GPU: every output is a deterministic function of the INPUT item only
    (cloud contents, global anchor ids), like the real path (eval-mode network, per-query anchor seeds) -- what the
    sharded evaluation relies on.  Exercises sharding, the three gathers and the re-ordering, not the kernels."""

    def __init__(self):
        from corsair_amd import harness

        self.cfg = harness.Config(batch_size=4)
        self.device = torch.device("cpu")

    def voxel_counts(self, clouds):
        return [len(np.unique(np.floor(np.asarray(c) / self.cfg.voxel_size).astype(np.int64), axis=0)) for c in clouds]

    def embed_clouds(self, clouds, batch_size=None):
        from corsair_amd.harness import EmbeddedSet

        F, O, off, D = [], [], [0], []
        for c in clouds:
            c = np.asarray(c, np.float64)
            n = 5 + int(abs(c[0, 0]) * 1000) % 7
            F.append(torch.from_numpy(np.tile(c[:n, :1], (1, 16)).astype(np.float32)))
            O.append(torch.from_numpy(c[:n].astype(np.float32)))
            off.append(off[-1] + n)
            d = np.cos(np.arange(256) * (1.0 + c[:32].sum()))
            D.append(torch.from_numpy((d / np.linalg.norm(d)).astype(np.float32))[None])
        return EmbeddedSet(torch.cat(F), torch.cat(O), off, torch.cat(D))

    def retrieve(self, q, x, k):
        out = []
        for s in range(0, q.shape[0], 64):        # (a [993, 652, 256] f64 difference tensor is 1.3 GB per rank)
            d2 = ((q[s:s + 64].double()[:, None, :] - x.double()[None, :, :]) ** 2).sum(-1).numpy()
            out.append(np.argsort(d2, axis=1, kind="stable")[:, :k])
        return torch.from_numpy(np.concatenate(out) if out else np.zeros((0, k), np.int64))

    def register(self, queries, cads, syms, anchor_ids=None, force_gate=False, **_):
        import types

        from corsair_amd import synth

        P = len(queries)
        Tr, Tb, cr, cb, ok = [], [], [], [], []
        for p in range(P):
            qs = float(queries.F[queries.offsets[p]:queries.offsets[p + 1]].double().sum())
            cs = float(cads.F[cads.offsets[p]:cads.offsets[p + 1]].double().sum())
            a = anchor_ids[p][0] * 0.37 + qs + 2.0 * cs + int(syms[p])
            T = np.eye(4)
            T[:3, :3] = synth.euler2mat(a, 0.5 * a, 0.25 * a)
            T[:3, 3] = [np.sin(a), np.cos(a), 0.1]
            Tr.append(T.astype(np.float32))
            Tb.append((T @ T).astype(np.float32))
            cr.append(abs(np.sin(a)) + 0.2)
            cb.append(abs(np.sin(a)) * 0.5)
            ok.append(anchor_ids[p][1] % 3 != 0)
        return types.SimpleNamespace(T_ransac=torch.from_numpy(np.stack(Tr)), T_best=torch.from_numpy(np.stack(Tb)),
                                     cd_ransac=torch.tensor(cr, dtype=torch.float64),
                                     cd_best=torch.tensor(cb, dtype=torch.float64), ok=np.asarray(ok))


def _eval_inputs(C=13, Q=11):
    from corsair_amd import synth

    rng = np.random.default_rng(12)
    catalog = [rng.uniform(-1, 1, (200 + 40 * (c % 5), 3)).astype(np.float32) for c in range(C)]
    queries = [synth.apply_pose(catalog[q % C][:150 + 20 * (q % 4)], synth.random_pose(q), np.float64) for q in range(Q)]
    table = rng.random((C, C))
    table = table + table.T
    np.fill_diagonal(table, 0.0)
    best_match = np.arange(Q) % C
    base_T = np.stack([synth.random_pose(q) for q in range(Q)])
    lib_T = np.stack([np.eye(4)] * C)
    syms = np.ones(C, np.int32)
    syms[[2 % C, 7 % C]] = [2, 4]
    return catalog, queries, best_match, table, base_T, lib_T, syms


def _eval_worker(rank, world, port, out_dir, C=13, Q=11):
    import pickle

    import torch.distributed as dist

    from corsair_amd import sharding

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        catalog, queries, best_match, table, base_T, lib_T, syms = _eval_inputs(C, Q)
        res = sharding.run_eval_sharded(_FakePipe(), dist, rank, world, catalog, queries, best_match, table, base_T,
                                        lib_T, syms, "chair", True, cache_dir=os.path.join(out_dir, "cache"))
        with open(os.path.join(out_dir, f"eval{rank}.pkl"), "wb") as f:
            pickle.dump((res.stat, res.per_query, res.report), f)
    finally:
        dist.destroy_process_group()


def test_sharded_evaluation_world2_ends_in_the_single_rank_result(tmp_path):
    """evaluation.py:207-441 on 2 ranks (sharding.run_eval_sharded): catalog and queries dealt out by voxel count, the
    embedded catalog, the query descriptors and the nine per-query arrays gathered; both ranks hold the result of
    the single-rank harness.run_eval bit for bit, and the nine cache files are written once, by rank 0."""
    import pickle

    import torch.multiprocessing as mp

    from corsair_amd import cache, harness

    mp.spawn(_eval_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    catalog, queries, best_match, table, base_T, lib_T, syms = _eval_inputs()
    want = harness.run_eval(_FakePipe(), catalog, queries, best_match, table, base_T, lib_T, syms, "chair", True)
    for r in range(2):
        with open(tmp_path / f"eval{r}.pkl", "rb") as f:
            stat, per_query, report = pickle.load(f)
        assert stat == want.stat and report == want.report
        for k in cache.NAMES:
            assert per_query[k].dtype == want.per_query[k].dtype and np.array_equal(per_query[k], want.per_query[k]), k
    loaded = cache.load_results(str(tmp_path / "cache"), "chair", True)
    for k in cache.NAMES:
        assert np.array_equal(loaded[k], want.per_query[k]), k
    assert sorted(p.name for p in (tmp_path / "cache").iterdir()) == sorted(f"{n}_chair_top1.npy" for n in cache.NAMES)


@pytest.mark.parametrize("C,Q", [(652, 993), (20, 5)])
def test_sharded_evaluation_world8_ends_in_the_single_rank_result(tmp_path, C, Q):
    """BASELINE.json configs[3]: run_eval_sharded on EIGHT ranks at the chair contract size (993 queries in shards of
    124 / 125, 652 CADs) and with fewer queries than ranks (Q = 5: three ranks register nothing and still take part in
    every collective -- ADVICE r4: their 0-size arrays used to fail in reshape before the all-gather).  Every rank ends
    with the single-rank result bit for bit; the cache is written once."""
    import pickle

    import torch.multiprocessing as mp

    from corsair_amd import cache, harness

    mp.spawn(_eval_worker, args=(8, _free_port(), str(tmp_path), C, Q), nprocs=8, join=True)
    catalog, queries, best_match, table, base_T, lib_T, syms = _eval_inputs(C, Q)
    want = harness.run_eval(_FakePipe(), catalog, queries, best_match, table, base_T, lib_T, syms, "chair", True)
    for r in range(8):
        with open(tmp_path / f"eval{r}.pkl", "rb") as f:
            stat, per_query, report = pickle.load(f)
        assert stat == want.stat and report == want.report
        for k in cache.NAMES:
            assert per_query[k].shape == want.per_query[k].shape and per_query[k].dtype == want.per_query[k].dtype, k
            assert np.array_equal(per_query[k], want.per_query[k]), k
    loaded = cache.load_results(str(tmp_path / "cache"), "chair", True)
    for k in cache.NAMES:
        assert np.array_equal(loaded[k], want.per_query[k]), k
