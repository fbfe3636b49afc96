"""File-level entries: a checkpoint FILE in the reference's format and directories of .npy clouds go
through `python -m corsair_amd.harness` / `python -m corsair_amd.shapenet_eval` (in-process `main(argv)`), and the results
equal the in-memory run -- what reproducing README.md:262-267 needs the day the blobs exist (evaluation.py:68-129,195-201,
evaluation-shapenet.py:277-343)."""
import csv
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _raw_real_clouds():
    z = np.load(os.path.join(GOLD, "real_clouds.npz"))
    z10 = np.load(os.path.join(GOLD, "real_clouds10.npz"))
    return [z["chair"].astype(np.float32), z["table"].astype(np.float32)] + list(z10["clouds"].astype(np.float32))


def _norm(pc):
    pc = pc - pc.mean(0)
    return (pc / np.max(np.linalg.norm(pc, 2, 1))).astype(np.float32)   # utils/preprocess.py:32-36


@pytest.fixture(scope="module")
def ckpt_file(gpu, tmp_path_factory):
    """synth.make_state_dicts(31) saved with the reference's save_checkpoint through the reference-named model classes."""
    from corsair_amd import synth
    from corsair_amd.model import fc, load_model
    from corsair_amd.utils import ckpts

    sd, emb = synth.make_state_dicts(31)
    model = load_model("ResUNetBN2C")(1, 16, bn_momentum=0.05, normalize_feature=True, conv1_kernel_size=3, D=3)
    head = fc.conv1_max_embedding(1024, 512, 256)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    head.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in emb.items()})
    opt = torch.optim.SGD(list(model.parameters()) + list(head.parameters()), lr=0.1)
    sched = torch.optim.lr_scheduler.ExponentialLR(opt, 0.9)
    d = tmp_path_factory.mktemp("ckpt")
    ckpts.save_checkpoint(model, head, opt, sched, 3, str(d), "scannet_pose_chair_best")
    return str(d / "scannet_pose_chair_best"), sd, emb


def test_harness_cli_on_files_equals_the_in_memory_run(gpu, ckpt_file, tmp_path):
    from corsair_amd import cache, harness, synth

    path, sd, emb = ckpt_file
    clouds = [_norm(c) for c in _raw_real_clouds()]                     # 12 bundled ShapeNet clouds, normalised
    cat_dir, q_dir = tmp_path / "cads", tmp_path / "scans"
    cat_dir.mkdir()
    q_dir.mkdir()
    for i, c in enumerate(clouds):
        np.save(cat_dir / f"cad_{i:03d}.npy", c)
    Q = 8
    poses = np.stack([[synth.random_pose(50 + q, max_trans=0.0)] * 3 for q in range(Q)])     # fix_trans layout [Q,3,4,4]
    for q in range(Q):
        np.save(q_dir / f"scan_{q:03d}.npy", clouds[(q * 5) % 12][::-1].copy())
    np.save(tmp_path / "fix_trans.npy", poses)
    best = (np.arange(Q) * 5) % 12
    np.save(tmp_path / "best.npy", best)
    with open(tmp_path / "sym.txt", "w") as f:
        for i in range(12):
            f.write(f"/scannet/x/cad_{i:03d}.npy {[1, 2, 4][i % 3]}\n")
    argv = ["--ckpt", path, "--catalog-dir", str(cat_dir), "--query-dir", str(q_dir), "--category", "chair",
            "--query-poses", str(tmp_path / "fix_trans.npy"), "--best-match", str(tmp_path / "best.npy"),
            "--sym-labels", str(tmp_path / "sym.txt"), "--cache-dir", str(tmp_path / "cache"), "--n-points", "6000",
            "--ransac-max-iter", "3000", "--batch-size", "5"]
    res = harness.main(argv)
    # the in-memory run on the same arrays
    cfg = harness.Config(n_points=6000, batch_size=5, ransac_max_iter=3000)
    pipe = harness.Pipeline(sd, emb, device=gpu, config=cfg)
    catalog = [c[:6000] for c in clouds]
    queries = [synth.apply_pose(clouds[(q * 5) % 12][::-1][:6000], poses[q, 0], np.float64) for q in range(Q)]
    from corsair_amd.utils import pc_dist
    table = pc_dist.compute_dist([c[:2000] for c in catalog])
    np.fill_diagonal(table, 0.0)
    syms = np.asarray([[1, 2, 4][i % 3] for i in range(12)], np.int32)
    want = harness.run_eval(pipe, catalog, queries, best, table, poses[:, 0], np.stack([np.eye(4)] * 12), syms, "chair", True)
    assert res.stat == want.stat and res.report == want.report
    for k in cache.NAMES:
        assert np.array_equal(res.per_query[k], want.per_query[k]), k
    files = sorted(p.name for p in (tmp_path / "cache").iterdir())
    assert files == sorted(f"{n}_chair_top1.npy" for n in cache.NAMES)
    again = harness.main(argv)                                            # second call: the cache is used
    assert again.from_cache and again.report == res.report


def test_shapenet_eval_cli_writes_the_reference_files(gpu, ckpt_file, tmp_path):
    from corsair_amd import harness, shapenet_eval as S

    path, sd, emb = ckpt_file
    raw = _raw_real_clouds()[:3]
    d = tmp_path / "03001627" / "test"
    d.mkdir(parents=True)
    for i, c in enumerate(raw):
        np.save(d / f"m{i}.npy", c[:5000])
    results, csv_file, npz_file = S.main(["--shapenet-root", str(tmp_path), "--category", "chair", "--n-models", "0",
                                          "--n-poses-per-model", "2", "--ckpt", path, "--random-seed", "4",
                                          "--ransac-max-iter", "3000", "--out-dir", str(tmp_path / "out")])
    assert os.path.basename(csv_file) == "results-shapenet-seed4-chair-3-2.csv"
    rows = list(csv.reader(open(csv_file)))
    assert tuple(rows[0]) == S.CSV_COLUMNS and len(rows) == 1 + 6            # evaluation-shapenet.py:323-334
    assert [r[0] for r in rows[1:]] == ["m0.npy", "m0.npy", "m1.npy", "m1.npy", "m2.npy", "m2.npy"]
    z = np.load(npz_file)
    assert z["poses_gt"].shape == (6, 4, 4) and z["poses_pred_sym"].shape == (6, 4, 4)
    # same numbers as the library call on the same arrays
    pipe = harness.Pipeline(sd, emb, device=gpu)
    cfg = S.Config(random_seed=4, n_poses_per_model=2, max_roll_deg=360, max_pitch_deg=360, max_yaw_deg=360,
                   ransac_max_iter=3000)
    want = S.evaluate(pipe, [c[:5000] for c in raw], cfg)
    for r, w, row in zip(results, want, rows[1:]):
        assert np.array_equal(r["T_est_sym"], w["T_est_sym"]) and r["rre_sym"] == w["rre_sym"]
        assert float(row[5]) == w["rre_sym"] and float(row[6]) == w["chamfer_dist_sym"] and int(row[2]) == w["symmetry_label"]
