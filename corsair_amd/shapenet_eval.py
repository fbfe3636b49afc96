"""Registration-only evaluation on clouds with synthetic poses: counterpart of the reference's
evaluation-shapenet.py:70-343.  Same stages as the Scan2CAD path without
retrieval: geometric symmetry label -> batch-of-2 ResUNet forward (model + randomly posed copy) ->
sym_pose(max_corr 0.4) -> eval_pose(T, I, pose_gt, label).  The reference fans the registrations out
to joblib worker processes (:341-343); here pairs are batched on the GPU.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import torch

from . import backend as B
from . import registration as R
from .synth import euler2mat
from .utils.eval_pose import eval_pose


@dataclass
class Config:
    """Defaults of evaluation-shapenet.py:38-67."""
    voxel_size: float = 0.03
    k_nn: int = 5
    max_corr: float = 0.4
    random_seed: int = 31
    n_poses_per_model: int = 1
    max_roll_deg: float = 180.0
    max_pitch_deg: float = 180.0
    max_yaw_deg: float = 180.0
    max_translation: float = 1.0
    symmetry_cd_threshold: float = 0.1
    ransac_max_iter: int = 100000
    ransac_confidence: float = 0.999


def load_pc(pc):
    """Centre and scale to unit radius, IN PLACE in the array's own type like evaluation-shapenet.py:70-76
    (`pc -= t; pc /= r` on what np.load returned: the ShapeNetPC15k files are f32, so everything up to the
    pose product is f32 there); takes an array, works on a copy."""
    pc = np.array(pc)
    if pc.dtype not in (np.float32, np.float64):
        pc = pc.astype(np.float64)
    t = pc.mean(axis=0, keepdims=True)
    pc -= t
    r = np.linalg.norm(pc, axis=1).max()
    pc /= r
    return pc


def generate_random_pose(cfg, rng):
    """evaluation-shapenet.py:79-94 with an explicit generator instead of NumPy's global RNG."""
    r, p, y = (np.deg2rad(rng.uniform(-m, m)) for m in (cfg.max_roll_deg, cfg.max_pitch_deg, cfg.max_yaw_deg))
    pose = np.eye(4)
    pose[:3, :3] = euler2mat(r, p, y)
    pose[:3, 3] = rng.uniform(-cfg.max_translation, cfg.max_translation, 3)
    return pose


def chamfer_max(pc0, pc1):
    """Two-sided Hausdorff distance (evaluation-shapenet.py:122-135), two cs_hausdorff_1dir problems."""
    dev = torch.device("cuda")
    a = torch.from_numpy(np.ascontiguousarray(pc0, np.float32)).to(dev)
    b = torch.from_numpy(np.ascontiguousarray(pc1, np.float32)).to(dev)
    X = torch.cat([a, b])
    off = [0, len(a), len(a) + len(b)]
    I = torch.eye(4, dtype=torch.float32, device=dev).repeat(2, 1, 1)
    return float(B.hausdorff_1dir(X, off, X, off, [0, 1], [1, 0], I).max().cpu())


def _hausdorff_many(pc, Rs):
    """max(H(pc -> R pc), H(R pc -> pc)) for every rotation in Rs (4x4), one batched launch.
    H(R pc -> pc) = H over source points transformed by R against the unrotated cloud;
    H(pc -> R pc) = H over source points transformed by R^-1 against the unrotated cloud (distances
    are rotation invariant), so both directions use the same resident cloud."""
    dev = torch.device("cuda")
    x = torch.from_numpy(np.ascontiguousarray(pc, np.float32)).to(dev)
    Ts = []
    for Rm in Rs:
        Ts.append(np.asarray(Rm, np.float64))
        Ts.append(np.linalg.inv(np.asarray(Rm, np.float64)))
    T = torch.from_numpy(np.stack(Ts).astype(np.float32)).to(dev)
    n = len(Ts)
    d = B.hausdorff_1dir(x, [0, len(x)], x, [0, len(x)], [0] * n, [0] * n, T).cpu().numpy()
    return d.reshape(-1, 2).max(axis=1)


def test_symmetry_label(sym_label, pc, cd_threshold):
    """True iff pc maps onto itself (Hausdorff <= threshold) under every rotation i*2pi/sym about y,
    i = 1..sym//2 (evaluation-shapenet.py:138-148)."""
    Rs = []
    for i in range(1, int(sym_label / 2) + 1):
        T = np.eye(4)
        T[:3, :3] = euler2mat(0, i * (2 * np.pi) / sym_label, 0)
        Rs.append(T)
    if not Rs:
        return True
    return bool((_hausdorff_many(pc, Rs) <= cd_threshold).all())


def get_symmetry_label(pc, cd_threshold):
    """Largest of (12, 8, 6, 4, 3, 2, 1) under which the cloud is symmetric; 1 = none
    (evaluation-shapenet.py:151-155)."""
    for sym_label in [12, 8, 6, 4, 3, 2, 1]:
        if test_symmetry_label(sym_label, pc, cd_threshold):
            return sym_label
    return 0


def evaluate(pipe, clouds, cfg=None, pairs_per_batch=16, seed=None, force_gate=False):
    """clouds: list of raw [n,3] arrays.  Returns a list of result dicts (one per model x pose) with
    the keys of registration_worker (evaluation-shapenet.py:242-275)."""
    cfg = cfg or Config()
    rng = np.random.default_rng(cfg.random_seed if seed is None else seed)
    dev = pipe.device
    jobs = []
    for mi, raw in enumerate(clouds):
        pc = load_pc(raw)
        label = get_symmetry_label(pc, cfg.symmetry_cd_threshold)
        for pi in range(cfg.n_poses_per_model):
            pose = generate_random_pose(cfg, rng)
            # generate_test_pc_pair (:115-119): the model stays in its own type, the posed copy is the f64
            # product with the f64 pose; quantize_pc (:97-107) floors each in its type and narrows afterwards
            jobs.append((mi, pi, pc, pc @ pose[:3, :3].T + pose[:3, [3]].T, pose, label))
    results = []
    for s in range(0, len(jobs), pairs_per_batch):
        chunk = jobs[s:s + pairs_per_batch]
        # one forward over [all models | all posed copies] of the chunk, every cloud quantised in its own type
        groups = []
        for col in (2, 3):
            xyz = torch.from_numpy(np.concatenate([j[col] for j in chunk])).to(dev)
            groups.append((xyz, np.concatenate([[0], np.cumsum([len(j[col]) for j in chunk])]).tolist()))
        es = pipe.embed_groups(groups, cfg.voxel_size)
        P = len(chunk)
        base = es.gather(list(range(P)))
        posed = es.gather(list(range(P, 2 * P)))
        labels = [j[5] for j in chunk]
        res = R.sym_pose_batch(base.F, base.origin, base.offsets, posed.F, posed.origin, posed.offsets,
                               labels, cfg.k_nn, cfg.max_corr, 0,
                               [(2 * (s + i), 2 * (s + i) + 1) for i in range(P)], 100,
                               cfg.ransac_max_iter, cfg.ransac_confidence, True, force_gate)
        Tb, Tr = res.T_best.cpu().numpy(), res.T_ransac.cpu().numpy()
        cdb, cdr = res.cd_best.cpu().numpy(), res.cd_ransac.cpu().numpy()
        for i, (mi, pi, _, _, pose, label) in enumerate(chunk):
            rte_s, rre_s = eval_pose(Tb[i], np.eye(4), pose, axis_symmetry=label)
            rte_r, rre_r = eval_pose(Tr[i], np.eye(4), pose, axis_symmetry=label)
            results.append(dict(model=mi, pose_idx=pi, symmetry_label=label, sym_success=bool(res.ok[i]),
                                T_est_sym=Tb[i], chamfer_dist_sym=float(cdb[i]), T_est_ransac=Tr[i],
                                chamfer_dist_ransac=float(cdr[i]), rte_sym=float(rte_s), rre_sym=float(rre_s),
                                rte_ransac=float(rte_r), rre_ransac=float(rre_r), pose_gt=pose))
    return results


def threshold_table(results, rre_deg=(5, 15, 45), rte=(0.02, 0.05, 0.10, 0.15)):
    """compute_metrics_shapenet.py-style summary: fraction of cases under each threshold."""
    out = {}
    for tag in ("ransac", "sym"):
        r = np.array([x[f"rre_{tag}"] for x in results])
        t = np.array([x[f"rte_{tag}"] for x in results])
        out[tag] = {**{f"rre<={d}": float(np.mean(r <= np.deg2rad(d))) for d in rre_deg},
                    **{f"rte<={v}": float(np.mean(t <= v)) for v in rte}}
    return out


# ---- file-level entry: `python -m corsair_amd.shapenet_eval` (evaluation-shapenet.py:158-240,277-380) ---------------------
CSV_COLUMNS = ("model", "pose_idx", "symmetry_label", "sym_success", "rte_sym", "rre_sym", "cd_sym", "rte_ransac",
               "rre_ransac", "cd_ransac")            # evaluation-shapenet.py:323-334


def write_results(results, names, csv_file, npz_file):
    """results-*.csv with the reference's columns and poses-*.npz with poses_gt / poses_pred_sym / poses_pred_ransac
    (evaluation-shapenet.py:345-380)."""
    import csv

    with open(csv_file, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(CSV_COLUMNS)
        for r in results:
            w.writerow([names[r["model"]], r["pose_idx"], r["symmetry_label"], r["sym_success"], r["rte_sym"], r["rre_sym"],
                        r["chamfer_dist_sym"], r["rte_ransac"], r["rre_ransac"], r["chamfer_dist_ransac"]])
    with open(npz_file, "wb") as f:
        np.savez(f, poses_gt=np.stack([r["pose_gt"] for r in results]),
                 poses_pred_sym=np.stack([r["T_est_sym"] for r in results]),
                 poses_pred_ransac=np.stack([r["T_est_ransac"] for r in results]))


def summary(results):
    """The three lines evaluation-shapenet.py:226-234 prints."""
    r = {k: np.asarray([x[k] for x in results]) for k in ("rte_sym", "rte_ransac", "rre_sym", "rre_ransac")}
    n, five = len(results), np.deg2rad(5)
    lines = []
    for title, fs, fr in (("RTE <= 0.02", r["rte_sym"] <= 0.02, r["rte_ransac"] <= 0.02),
                          ("RRE <= 5 deg", r["rre_sym"] <= five, r["rre_ransac"] <= five),
                          ("RTE <= 0.02 & RRE <= 5 deg", (r["rte_sym"] <= 0.02) & (r["rre_sym"] <= five),
                           (r["rte_ransac"] <= 0.02) & (r["rre_ransac"] <= five))):
        lines.append(f"{title}: sym: {fs.sum() / n:.4f}, ransac: {fr.sum() / n:.4f}")
    return "\n".join(lines)


def main(argv=None):
    import argparse
    import os

    from . import harness
    from .utils import ckpts

    ap = argparse.ArgumentParser(prog="python -m corsair_amd.shapenet_eval",
                                 description="Registration evaluation with synthetic poses on a directory of .npy clouds "
                                             "(evaluation-shapenet.py:158-240): writes results-*.csv / poses-*.npz")
    ap.add_argument("--data-dir", help="directory of [n,3] .npy clouds; or --shapenet-root + --category as in the reference")
    ap.add_argument("--shapenet-root")
    ap.add_argument("--category", default="chair", choices=["chair", "table"])
    ap.add_argument("--n-models", type=int, default=1)
    ap.add_argument("--n-poses-per-model", type=int, default=10)
    ap.add_argument("--max-roll-deg", type=float, default=360)
    ap.add_argument("--max-pitch-deg", type=float, default=360)
    ap.add_argument("--max-yaw-deg", type=float, default=360)
    ap.add_argument("--max-translation", type=float, default=1.0)
    ap.add_argument("--model-ckpt", "--ckpt", required=True, dest="ckpt")
    ap.add_argument("--random-seed", type=int, default=0)
    ap.add_argument("--ransac-max-iter", type=int, default=100000)
    ap.add_argument("--out-dir", default=".")
    ap.add_argument("--device", default="cuda", choices=["cuda"])
    a = ap.parse_args(argv)
    if not a.data_dir:
        if not a.shapenet_root:
            raise SystemExit("give --data-dir, or --shapenet-root with --category")
        a.data_dir = os.path.join(a.shapenet_root, {"chair": "03001627", "table": "04379243"}[a.category], "test")
    files = sorted(f for f in os.listdir(a.data_dir) if f.endswith(".npy"))
    rng = np.random.default_rng(a.random_seed)
    if 0 < a.n_models < len(files):                                   # evaluation-shapenet.py:200-203
        files = sorted(rng.choice(files, a.n_models, replace=False).tolist())
    clouds = [np.load(os.path.join(a.data_dir, f)) for f in files]
    sd, esd = ckpts.load_state_dicts(a.ckpt)
    pipe = harness.Pipeline(sd, esd, device=a.device)
    cfg = Config(random_seed=a.random_seed, n_poses_per_model=a.n_poses_per_model, max_roll_deg=a.max_roll_deg,
                 max_pitch_deg=a.max_pitch_deg, max_yaw_deg=a.max_yaw_deg, max_translation=a.max_translation,
                 ransac_max_iter=a.ransac_max_iter)
    results = evaluate(pipe, clouds, cfg)
    postfix = f"shapenet-seed{a.random_seed}-{a.category}-{len(files)}-{a.n_poses_per_model}"
    os.makedirs(a.out_dir, exist_ok=True)
    csv_file = os.path.join(a.out_dir, f"results-{postfix}.csv")
    npz_file = os.path.join(a.out_dir, f"poses-{postfix}.npz")
    write_results(results, files, csv_file, npz_file)
    print(summary(results))
    return results, csv_file, npz_file


if __name__ == "__main__":
    main()
