"""The end-to-end evaluation entry harness.run_eval -- the counterpart of evaluation.py:207-441 -- on a
64-CAD x 32-query synthetic Scan2CAD-shaped set: retrieval statistics against the oracle, per-query
outputs against the stage-by-stage calls, the result cache round trip (the reference's nine files)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def small_eval(gpu):
    from corsair_amd import harness, synth

    cfg = harness.Config(n_points=4000, ransac_max_iter=4000)
    sd, emb = synth.make_state_dicts(31)
    pipe = harness.Pipeline(sd, emb, device=gpu, config=cfg)
    data = harness.SyntheticScan2CAD(n_catalog=64, n_query=32, n_points=4000).build()
    data.sym[[5, 9]] = [2, 4]
    return pipe, data, data.table()


def test_run_eval_outputs_and_cache_round_trip(gpu, oracle_native, small_eval, tmp_path, monkeypatch):
    from corsair_amd import cache, harness, registration
    from corsair_amd.utils.eval_pose import eval_pose
    from oracle import post

    pipe, data, table = small_eval
    catalog, queries, best_match, base_T, lib_T, syms = data.eval_inputs()
    cat = pipe.embed_clouds(catalog)
    qs = pipe.embed_clouds(queries)
    res = harness.run_eval(pipe, cat, qs, best_match, table, base_T, lib_T, syms, "chair", True,
                           cache_dir=str(tmp_path), force_gate=True)
    assert not res.from_cache
    # --- retrieval block (evaluation.py:272-283) against the oracle on the same descriptors ---
    want = post.scan2cad_retrieval_eval(qs.desc.cpu().numpy(), cat.desc.cpu().numpy(), best_match, table,
                                        int(0.1 * 64))
    assert res.stat["top1_predict"] == want["top1_predict"] and res.stat["gt"] == want["gt"]
    assert res.stat["precision"] == want["precision"] and res.stat["top1_error"] == want["top1_error"]
    assert res.stat["gt"] == best_match.tolist()        # table diagonal is 0: the GT CAD ranks itself first
    # --- registration loop (evaluation.py:297-331): same numbers as the stage-by-stage calls ---
    pq = res.per_query
    assert set(pq) == set(cache.NAMES)
    top = np.asarray(res.stat["top1_predict"])
    direct = pipe.register(qs, cat.gather(top), syms[top], anchor_ids=[(2 * i, 2 * i + 1) for i in range(32)],
                           force_gate=True)
    assert np.array_equal(pq["Ts_est_best"], direct.T_best.cpu().numpy())
    assert np.array_equal(pq["Ts_est_ransac"], direct.T_ransac.cpu().numpy())
    assert np.array_equal(pq["chamfer_dist_sym"], direct.cd_best.cpu().numpy())
    assert pq["sym_ransac_success"].dtype == bool and pq["sym_ransac_success"].all()
    assert (pq["chamfer_dist_sym"] <= pq["chamfer_dist_ransac"]).all()
    for i in range(32):
        t, r = eval_pose(pq["Ts_est_best"][i], base_T[i], lib_T[top[i]], int(syms[top[i]]))
        assert pq["t_losses_sym"][i] == t and pq["r_losses_sym"][i] == r
    # --- aggregation + log block (evaluation.py:334-383) ---
    assert res.sym["rre_15"] == float(np.sum(np.rad2deg(pq["r_losses_sym"]) <= 15) / 32)
    assert res.ransac["chamfer_mean"] == float(np.mean(pq["chamfer_dist_ransac"]))
    assert "vanilla ransac:" in res.report and "sym ransac:" in res.report and "sym success rate: 1.0" in res.report
    # --- cache (evaluation.py:421-441): nine files, reference layout, reload skips the registration ---
    files = sorted(p.name for p in tmp_path.iterdir())
    assert files == sorted(f"{n}_chair_top1.npy" for n in cache.NAMES)
    assert np.load(tmp_path / "Ts_est_best_chair_top1.npy").shape == (32, 16)
    loaded = cache.load_results(str(tmp_path), "chair", True)
    for k in cache.NAMES:
        assert np.array_equal(loaded[k], pq[k]), k
    monkeypatch.setattr(registration, "sym_pose_batch", lambda *a, **k: pytest.fail("cache must be used"))
    again = harness.run_eval(pipe, cat, qs, best_match, table, base_T, lib_T, syms, "chair", True,
                             cache_dir=str(tmp_path))
    assert again.from_cache and again.report == res.report and again.stat == res.stat


def test_run_eval_registers_against_gt_when_asked(gpu, small_eval):
    """register_top1=False (evaluation.py:303): the annotated CAD is registered; the queries are posed
    re-samplings of exactly that CAD."""
    from corsair_amd import harness

    pipe, data, table = small_eval
    catalog, queries, best_match, base_T, lib_T, syms = data.eval_inputs()
    res = harness.run_eval(pipe, catalog[:16], queries[:8], best_match[:8] % 16, table[:16, :16], base_T[:8],
                           lib_T[:16], syms[:16], "chair", False, force_gate=True)
    assert len(res.per_query["r_losses_sym"]) == 8 and not res.from_cache
    assert np.isfinite(res.per_query["chamfer_dist_sym"]).all()


def test_run_eval_with_batches_in_flight_equals_the_sequential_loop(gpu, small_eval):
    """register_queries(in_flight=3): three registration batches at a time on three host threads / HIP streams (what
    `python -m corsair_amd.harness` and `bench.py --scaling strong` use) -- the nine per-query arrays equal the sequential
    loop's bit for bit, whatever the batch size (here 5: seven batches, the last one ragged)."""
    from corsair_amd import cache, harness

    pipe, data, table = small_eval
    catalog, queries, best_match, base_T, lib_T, syms = data.eval_inputs()
    cat = pipe.embed_clouds(catalog)
    qs = pipe.embed_clouds(queries)
    seq = harness.run_eval(pipe, cat, qs, best_match, table, base_T, lib_T, syms, "chair", True, force_gate=True,
                           batch_size=5)
    par = harness.run_eval(pipe, cat, qs, best_match, table, base_T, lib_T, syms, "chair", True, force_gate=True,
                           batch_size=5, in_flight=3)
    assert seq.stat == par.stat and seq.report == par.report
    for k in cache.NAMES:
        assert seq.per_query[k].dtype == par.per_query[k].dtype and np.array_equal(seq.per_query[k], par.per_query[k]), k


def test_bench_grouped_forward_equals_one_forward_per_step(gpu, monkeypatch):
    """bench.py's sequential pass sends several steps' query batches through ONE forward of the network
    (RegistrationWorkload.step_group, --embed-group): retrieval ids, transforms, Chamfer distances and iteration counts of
    every step equal those of one forward per step bit for bit (batch composition changes no row)."""
    import sys

    import bench

    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--catalog", "24", "--steps", "3", "--warmup", "0"])
    args = bench.parse()
    ctx = bench.Ctx(args)
    wl = bench.RegistrationWorkload(ctx, "chair")
    wl.setup()
    for b in range(3):
        wl.step(b)
    single = {r[0]: r for r in wl.results}
    wl.results.clear()
    wl.step_group([0, 1, 2])
    assert sorted(r[0] for r in wl.results) == [0, 1, 2]
    for r in wl.results:
        s = single[r[0]]
        assert np.array_equal(r[1], s[1])                       # retrieved CAD ids
        for i in (2, 3, 4, 6):                                  # T_best, T_ransac, Chamfer, iterations
            assert np.array_equal(r[i], s[i]), i
        assert wl.same_results(r, s)


def test_bench_converging_leg_recovers_the_poses(gpu, monkeypatch):
    """bench.py's `converging` leg: the chair shapes with pose-invariant stand-in features -- retrieval
    finds the CAD every query was sampled from, sym_pose returns its pose (RRE <= 15 deg for all, mean of a few degrees) and the
    RANSACs leave through the confidence bound instead of running 100 000 iterations."""
    import sys

    import bench

    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--catalog", "40", "--steps", "2", "--warmup", "0"])
    args = bench.parse()
    ctx = bench.Ctx(args)
    wl = bench.RegistrationWorkload(ctx, "chair", converging=True)
    wl.setup()
    for b in range(2):
        wl.step(b)
    cfg = wl.config(2)
    assert cfg["top1_hit_rate"] == 1.0
    assert cfg["rre_15"] >= 0.95 and cfg["rre_mean_deg"] < 5.0
    assert cfg["ransac_early_exit_share"] > 0.8 and cfg["ransac_mean_iters"] < 30000
    assert "CONVERGING" in cfg["workload"]
