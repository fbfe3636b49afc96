"""GPU parity: coordinate maps, kernel maps and sparse convolution vs the CPU oracle (bit-exact)."""
import numpy as np
import pytest
import torch

from tests.helpers import make_batch

pytestmark = pytest.mark.gpu


def _maps(gpu, coords):
    from corsair_amd import engine

    return engine.BatchMaps(torch.from_numpy(coords).to(gpu))


def test_coordmaps_and_kernelmaps_bit_exact(gpu):
    from oracle import resunet as oref
    from oracle import sparse as osp

    coords, _, _, _ = make_batch([0, 1, 2], n_points=6000)
    m = _maps(gpu, coords)
    omaps, okm = oref.build_maps(coords)
    for name in ("c1", "c2", "c4", "c8"):
        got = getattr(m, name).coords.cpu().numpy()
        assert np.array_equal(got, omaps[name]), name
    for name, nbr in okm.items():
        km = getattr(m, name)
        assert np.array_equal(km.table().cpu().numpy(), nbr), name
        k, i, o = (t.cpu().numpy() for t in km.export())
        ok, oi, oo = osp.kernel_map_triples(nbr)
        assert km.num_pairs == len(ok)
        assert np.array_equal(k, ok) and np.array_equal(i, oi) and np.array_equal(o, oo), name


@pytest.mark.parametrize("cin,cout", [(1, 32), (32, 32), (32, 64), (64, 64), (64, 128), (256, 128),
                                      (96, 64), (64, 16)])
def test_conv_bit_exact(gpu, oracle_native, cin, cout):
    from corsair_amd import backend as B
    from oracle import resunet as oref

    coords, _, _, _ = make_batch([3, 4], n_points=3000)
    m = _maps(gpu, coords)
    _, okm = oref.build_maps(coords)
    rng = np.random.default_rng(cin * 1000 + cout)
    n = coords.shape[0]
    x = rng.standard_normal((n, cin)).astype(np.float32)
    w = (rng.standard_normal((27, cin, cout)) * 0.1).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    shift = rng.standard_normal(cout).astype(np.float32)
    res = rng.standard_normal((n, cout)).astype(np.float32)
    want = oracle_native.conv_fwd(okm["s1"], x, w, scale, shift, res, True)
    got = B.conv_fwd(m.s1, torch.from_numpy(x).to(gpu), torch.from_numpy(w).to(gpu),
                     torch.from_numpy(scale).to(gpu), torch.from_numpy(shift).to(gpu),
                     torch.from_numpy(res).to(gpu), True).cpu().numpy()
    assert np.array_equal(got, want), f"max diff {np.abs(got - want).max()}"
    # strided and transposed maps, no epilogue
    n2 = m.c2.n
    w2 = (rng.standard_normal((27, cin, cout)) * 0.1).astype(np.float32)
    want = oracle_native.conv_fwd(okm["s1_s2"], x, w2)
    got = B.conv_fwd(m.s1_s2, torch.from_numpy(x).to(gpu), torch.from_numpy(w2).to(gpu)).cpu().numpy()
    assert got.shape == (n2, cout) and np.array_equal(got, want)
    x2 = rng.standard_normal((n2, cin)).astype(np.float32)
    want = oracle_native.conv_fwd(okm["s2_s1_T"], x2, w2)
    got = B.conv_fwd(m.s2_s1_T, torch.from_numpy(x2).to(gpu), torch.from_numpy(w2).to(gpu)).cpu().numpy()
    assert got.shape == (n, cout) and np.array_equal(got, want)


@pytest.mark.parametrize("cfg", ["411", "412", "414", "221", "222", "141", "old", "front"])
def test_conv_every_tile_shape_bit_exact(gpu, oracle_native, monkeypatch, cfg):
    """Every tile shape of the LDS-DMA kernel (CS_CONV_CFG=<row groups><column groups><32-column
    accumulators per wave>; shapes a layer's Cout does not allow fall back to the default choice) and
    the round-1/2 register-staged kernel (CS_CONV_DMA=0) compute the same fma chains: the whole network,
    1x1 / strided / transposed layers included, is bit-exact against the oracle.  The batch has a ragged
    last tile at every level and an empty sample."""
    from corsair_amd import engine, synth
    from oracle import resunet as oref

    if cfg == "old":
        monkeypatch.setenv("CS_CONV_DMA", "0")
    elif cfg == "front":                     # tiles taken from the front of the tiling order (default: from the end)
        monkeypatch.setenv("CS_CONV_FWD_ORDER", "1")
    else:
        monkeypatch.setenv("CS_CONV_CFG", cfg)
    coords, feats, _, _ = make_batch([5, 6, 7], n_points=5000)
    coords[coords[:, 0] == 2, 0] = 3          # sample 2 is empty
    sd, emb = synth.make_state_dicts(31)
    eng = engine.ResUNetEngine(sd, emb, device=gpu)
    out, feat, maps = eng.forward(torch.from_numpy(coords).to(gpu), torch.from_numpy(feats).to(gpu))
    g = eng.embed(feat, maps, 4)
    want_out, want_feat, omaps = oref.resunet_forward(sd, coords, feats)
    want_g = oref.embedding_forward(emb, want_feat, omaps["c8"][:, 0], 4)
    assert np.array_equal(out.cpu().numpy(), want_out)
    assert np.array_equal(feat.cpu().numpy(), want_feat)
    assert np.array_equal(g[[0, 1, 3]].cpu().numpy(), want_g[[0, 1, 3]])


def test_resunet_forward_bit_exact(gpu, oracle_native):
    from corsair_amd import engine, synth
    from oracle import resunet as oref

    coords, feats, _, offsets = make_batch([5, 6, 7], n_points=5000)
    sd, emb = synth.make_state_dicts(31)
    eng = engine.ResUNetEngine(sd, emb, device=gpu)
    out, feat, maps = eng.forward(torch.from_numpy(coords).to(gpu), torch.from_numpy(feats).to(gpu))
    g = eng.embed(feat, maps, 3)
    want_out, want_feat, omaps = oref.resunet_forward(sd, coords, feats)
    want_g = oref.embedding_forward(emb, want_feat, omaps["c8"][:, 0], 3)
    assert np.array_equal(feat.cpu().numpy(), want_feat)
    assert np.array_equal(out.cpu().numpy(), want_out)
    assert np.array_equal(g.cpu().numpy(), want_g)


def test_kernel_maps_of_a_large_batch_equal_the_global_table_build(gpu, monkeypatch):
    """A stress-sized stride-1 level (>= 200 000 rows, 24 clouds at 2 cm; the 144-KB LDS table configuration): the map the
    level kernel builds from its per-sample LDS tables equals the map probed in the global hash table (CS_KMAP_GLOBAL=1),
    entry for entry; a submanifold map has every row as its own centre neighbour."""
    from corsair_amd import backend as B, synth

    clouds = [synth.make_cloud(c, 15000) for c in range(24)]
    xyz = torch.from_numpy(np.concatenate(clouds)).to(gpu)
    off = np.concatenate([[0], np.cumsum([len(c) for c in clouds])]).tolist()
    _, grid, _ = B.voxelize(xyz, off, 0.02)
    c1 = B.CoordMap.create(grid, 1)
    assert c1.n >= 200000
    lds = B.KernelMap.build(c1, c1)
    monkeypatch.setenv("CS_KMAP_GLOBAL", "1")
    glob = B.KernelMap.build(c1, c1)
    monkeypatch.delenv("CS_KMAP_GLOBAL")
    assert lds.num_pairs == glob.num_pairs
    assert torch.equal(lds.table(), glob.table())
    t = lds.table()
    assert bool((t[:, 13] == torch.arange(c1.n, device=gpu, dtype=torch.int32)).all())


def test_kernel_maps_build_many_equals_single_builds(gpu):
    """cs_kernelmap_build_many (ten independent chains on three streams inside one call) returns exactly the maps of ten
    cs_kernelmap_build calls: same neighbour tables, same pair counts; twice in a row (the second call reuses the
    scratch the first handed back after its join)."""
    from corsair_amd import backend as B

    coords, _, _, _ = make_batch([11, 12, 13, 14], n_points=6000)
    g = torch.from_numpy(coords).to(gpu)
    c1 = B.CoordMap.create(g, 1)
    c2 = c1.stride(2)
    c4 = c2.stride(2)
    c8 = c4.stride(2)
    specs = [(c1, c1), (c1, c2), (c2, c2), (c2, c4), (c4, c4), (c4, c8), (c8, c8),
             (c8, c4, 3, True), (c4, c2, 3, True), (c2, c1, 3, True)]
    single = [B.KernelMap.build(*sp) for sp in specs]
    for _ in range(2):
        many = B.KernelMap.build_many(specs)
        assert len(many) == len(single)
        for a, b in zip(single, many):
            assert a.n_out == b.n_out and a.n_in == b.n_in and a.transposed == b.transposed
            assert a.num_pairs == b.num_pairs
            assert torch.equal(a.table(), b.table())
    assert B.KernelMap.build_many([]) == []


@pytest.mark.parametrize("hint", [0, 5])
def test_kernel_maps_lds_path_with_fallback_samples(gpu, monkeypatch, hint):
    """LDS-built kernel maps: a batch mixing an ordinary sample, one too large for the LDS table
    (> 15 360 voxels), one with a bounding box wider than 1023 cells and an empty batch index; the
    workgroups of the samples that do not fit probe the level's global table instead.  Also: rows not grouped by
    sample (the global kernel for the whole map).  hint: batch size announced to the coordinate pyramid or not."""
    from oracle import resunet as oref
    from oracle import sparse as osp

    rng = np.random.default_rng(9)

    def cloud(n, span):
        g = rng.integers(-span, span, (n, 3))
        _, first = np.unique(g, axis=0, return_index=True)
        return g[np.sort(first)]

    parts = [cloud(3000, 12), cloud(40000, 20), np.array([[0, 0, 0], [3000, 1, 2], [3001, 1, 2], [1, 0, 0]]),
             np.zeros((0, 3), np.int64), cloud(500, 6)]
    assert len(parts[1]) > 15360
    coords = np.concatenate([np.concatenate([np.full((len(p), 1), b), p], 1) for b, p in enumerate(parts)]).astype(np.int32)
    from corsair_amd import engine

    m = engine.BatchMaps(torch.from_numpy(coords).to(gpu), hint)
    omaps, okm = oref.build_maps(coords)
    for name, nbr in okm.items():
        km = getattr(m, name)
        assert np.array_equal(km.table().cpu().numpy(), nbr), name
        assert km.num_pairs == int((nbr >= 0).sum()), name
    # interleaved batch indices (not grouped): whole map falls back to the global table
    perm = rng.permutation(len(coords))
    shuffled = coords[perm]
    m2 = engine.BatchMaps(torch.from_numpy(shuffled).to(gpu), hint)
    _, okm2 = oref.build_maps(shuffled)
    for name in ("s1", "s1_s2", "s2_s1_T"):
        assert np.array_equal(getattr(m2, name).table().cpu().numpy(), okm2[name]), name


@pytest.mark.parametrize("ns,cfg", [(3, "0"), (3, "411"), (3, "412"), (3, "221"), (3, "222"), (3, "141"), (2, "0")])
def test_conv_split_experiment_close_to_exact_chain(gpu, oracle_native, monkeypatch, ns, cfg):
    """CS_CONV_SPLIT=3 / 2 (off by default): the bf16-piece kernel on the matrix cores.  It is NOT
    bit-identical to the oracle's fma chain and no parity claim rests on it; this test pins how far it may drift:
    per output |diff| <= tol * sum_k |x||w| (the scale of one output's terms) with tol = 2^-19 for three pieces (kept
    products cover everything above 2^-23 |a||b|) and 2^-13 for two -- every tile shape, gathered / strided / transposed maps, ragged last tile, epilogue."""
    from corsair_amd import backend as B
    from oracle import resunet as oref

    coords, _, _, _ = make_batch([3, 4], n_points=3000)
    m = _maps(gpu, coords)
    _, okm = oref.build_maps(coords)
    tol = 2.0 ** -19 if ns == 3 else 2.0 ** -13
    for cin, cout in [(32, 32), (64, 128), (128, 64), (256, 128)]:
        rng = np.random.default_rng(cin * 1000 + cout)
        n = coords.shape[0]
        x = rng.standard_normal((n, cin)).astype(np.float32)
        w = (rng.standard_normal((27, cin, cout)) * 0.1).astype(np.float32)
        scale = rng.uniform(0.5, 1.5, cout).astype(np.float32)
        shift = rng.standard_normal(cout).astype(np.float32)
        res = rng.standard_normal((n, cout)).astype(np.float32)
        xd, wd = torch.from_numpy(x).to(gpu), torch.from_numpy(w).to(gpu)
        args = (torch.from_numpy(scale).to(gpu), torch.from_numpy(shift).to(gpu), torch.from_numpy(res).to(gpu), True)
        n2 = m.c2.n
        x2 = torch.from_numpy(rng.standard_normal((n2, cin)).astype(np.float32)).to(gpu)
        cases = [(m.s1, xd, args), (m.s1_s2, xd, ()), (m.s2_s1_T, x2, ())]   # (1x1 layers stay on the exact kernel)
        exact = [B.conv_fwd(km, xi, wd if km is not None else wd[13], *a).cpu().numpy() for km, xi, a in cases]
        # magnitude of an output's terms: the same convolution of |x| with |w| (exact path, no epilogue)
        mag = [B.conv_fwd(km, xi.abs(), (wd if km is not None else wd[13]).abs()).cpu().numpy() for km, xi, a in cases]
        monkeypatch.setenv("CS_CONV_SPLIT", str(ns))
        monkeypatch.setenv("CS_CONV_SPLIT_CFG", cfg)
        got = [B.conv_fwd(km, xi, wd if km is not None else wd[13], *a).cpu().numpy() for km, xi, a in cases]
        monkeypatch.delenv("CS_CONV_SPLIT")
        monkeypatch.delenv("CS_CONV_SPLIT_CFG")
        for i, (g, e, mg) in enumerate(zip(got, exact, mag)):
            bound = tol * (mg * (1.5 if i == 0 else 1.0) + 1e-6)
            assert g.shape == e.shape
            worst = float((np.abs(g - e) / bound).max())
            assert worst <= 1.0, f"cin {cin} cout {cout} case {i}: diff / bound = {worst}"
        assert any(not np.array_equal(g, e) for g, e in zip(got, exact))   # the experiment really ran


@pytest.mark.parametrize("hint", [0, 5, 3, 70000])
def test_coordmap_pyramid_equals_chained_calls(gpu, hint):
    """cs_coordmap_pyramid (all four coordinate levels from the stride-1 rows in one pass, one host wait) returns the maps of
    cs_coordmap_create + three chained cs_coordmap_stride(2) calls: same coordinates in the same (first-occurrence) order,
    same tables (the kernel maps built on them are identical, entry for entry).  hint = announced batch size: 0 (none), the
    true one, one that is too small (the announcement is violated: only the LDS path is lost) and one beyond the limit.
    The batch has an empty sample; a second batch has its rows shuffled (not grouped by sample)."""
    from corsair_amd import backend as B

    coords, _, _, _ = make_batch([21, 22, 23, 24], n_points=5000)
    coords[coords[:, 0] == 3, 0] = 4          # sample 3 is empty, batch indices 0, 1, 2, 4
    rng = np.random.default_rng(4)
    for rows in (coords, coords[rng.permutation(len(coords))]):
        g = torch.from_numpy(np.ascontiguousarray(rows)).to(gpu)
        c1 = B.CoordMap.create(g, 1)
        chain = [c1, c1.stride(2)]
        chain.append(chain[-1].stride(2))
        chain.append(chain[-1].stride(2))
        pyr = B.CoordMap.pyramid(g, 4, hint)
        for a, b in zip(chain, pyr):
            assert a.n == b.n and a.tensor_stride == b.tensor_stride
            assert torch.equal(a.coords, b.coords)
        specs = lambda c: [(c[0], c[0]), (c[0], c[1]), (c[1], c[1]), (c[1], c[2]), (c[2], c[2]), (c[2], c[3]), (c[3], c[3]),
                           (c[3], c[2], 3, True), (c[2], c[1], 3, True), (c[1], c[0], 3, True)]
        for ka, kb in zip(B.KernelMap.build_many(specs(chain)), B.KernelMap.build_many(specs(pyr))):
            assert ka.num_pairs == kb.num_pairs and torch.equal(ka.table(), kb.table())
    two = B.CoordMap.pyramid(torch.from_numpy(coords).to(gpu), 2, 5)
    assert len(two) == 2 and two[1].n == chain[1].n


def test_coordmap_pyramid_refuses_what_create_refuses(gpu):
    from corsair_amd import _lib, backend as B

    dup = torch.tensor([[0, 1, 2, 3], [0, 4, 5, 6], [0, 1, 2, 3]], dtype=torch.int32, device=gpu)
    with pytest.raises(_lib.CorsairHipError, match="duplicate"):
        B.CoordMap.pyramid(dup, 4, 1)
    far = torch.tensor([[0, 1, 2, 3], [0, 40000, 5, 6]], dtype=torch.int32, device=gpu)
    with pytest.raises(_lib.CorsairHipError, match="range"):
        B.CoordMap.pyramid(far, 4, 1)
    empty = B.CoordMap.pyramid(torch.zeros((0, 4), dtype=torch.int32, device=gpu), 4, 0)
    assert [m.n for m in empty] == [0, 0, 0, 0]


def test_coordmap_pyramid_beyond_its_packed_scan_limit(gpu):
    """2^21 rows and more do not fit the 21-bit fields of the pyramid's packed scan: cs_coordmap_pyramid then takes the
    chained create / stride path itself and must return the same maps."""
    from corsair_amd import backend as B

    side = 130                                            # 130^3 = 2 197 000 >= 2^21
    idx = np.arange(side ** 3, dtype=np.int64)
    rows = np.empty((len(idx), 4), np.int32)
    rows[:, 0] = idx % 3
    rows[:, 1], rows[:, 2], rows[:, 3] = np.unravel_index(idx, (side, side, side))
    rows[:, 1:] -= side // 2
    assert len(rows) >= 1 << 21
    g = torch.from_numpy(rows).to(gpu)
    c1 = B.CoordMap.create(g, 1)
    chain = [c1, c1.stride(2)]
    chain.append(chain[-1].stride(2))
    chain.append(chain[-1].stride(2))
    pyr = B.CoordMap.pyramid(g, 4, 3)
    for a, b in zip(chain, pyr):
        assert a.n == b.n and a.tensor_stride == b.tensor_stride
        assert torch.equal(a.coords, b.coords)
    ka = B.KernelMap.build(chain[3], chain[2], 3, True)
    kb = B.KernelMap.build(pyr[3], pyr[2], 3, True)
    assert ka.num_pairs == kb.num_pairs and torch.equal(ka.table(), kb.table())


def test_maps_of_a_tensor_stride_that_is_not_a_power_of_two(gpu):
    """The LDS level kernel and the pyramid kernels count cells with shifts and masks; a tensor stride of 3 must take the
    global-table / chained paths and still give the oracle's maps (coordinates that are multiples of 3, strided to 6)."""
    from corsair_amd import backend as B
    from oracle import sparse as osp

    coords, _, _, _ = make_batch([31, 32, 33], n_points=4000)
    c3 = coords.copy()
    c3[:, 1:] *= 3                                            # a stride-3 tensor's coordinates
    g = torch.from_numpy(c3).to(gpu)
    pyr = B.CoordMap.pyramid(g, 2, 3, tensor_stride=3)
    want6, cell = osp.coordmap_stride(c3, 3, 2)
    assert cell == 6 and pyr[0].tensor_stride == 3 and pyr[1].tensor_stride == 6
    assert np.array_equal(pyr[0].coords.cpu().numpy(), c3)
    assert np.array_equal(pyr[1].coords.cpu().numpy(), want6)
    specs = [(pyr[0], pyr[0]), (pyr[0], pyr[1]), (pyr[1], pyr[1]), (pyr[1], pyr[0], 3, True)]
    wants = [osp.kernel_map(c3, 3, c3, 3), osp.kernel_map(c3, 3, want6, 6), osp.kernel_map(want6, 6, want6, 6),
             osp.kernel_map(want6, 6, c3, 3, transposed=True)]
    for km, want in zip(B.KernelMap.build_many(specs), wants):
        assert np.array_equal(km.table().cpu().numpy(), want)
