"""Result-cache format of the reference (evaluation.py:390-441): nine .npy files per
(category, top1|gt) in the cache directory.  Files written here load in the reference's `_load_data`
and vice versa (the shipped data/cache_* directories load with `load_results`)."""
import os

import numpy as np

NAMES = ("Ts_est_ransac", "Ts_est_best", "t_losses_ransac", "t_losses_sym", "r_losses_ransac",
         "r_losses_sym", "sym_ransac_success", "chamfer_dist_ransac", "chamfer_dist_sym")
# element types of the nine arrays as the registration loop produces them (eval_pose's t_loss is the norm of an f32
# vector, hence f32; its r_loss goes through np.float64): what an EMPTY result set carries, and what the sharded
# evaluation puts on the wire -- fixed by NAME so that ranks with and without queries agree on every collective
DTYPES = {"Ts_est_ransac": np.float32, "Ts_est_best": np.float32, "t_losses_ransac": np.float32,
          "t_losses_sym": np.float32, "r_losses_ransac": np.float64, "r_losses_sym": np.float64,
          "sym_ransac_success": np.bool_, "chamfer_dist_ransac": np.float64, "chamfer_dist_sym": np.float64}


def _suffix(register_top1):
    return "_top1.npy" if register_top1 else "_gt.npy"


def save_results(cache_dir, category, register_top1, results):
    """results: dict with the NAMES keys; transforms as [Q,4,4] (stored flattened [Q,16] f32)."""
    os.makedirs(cache_dir, exist_ok=True)
    suf = _suffix(register_top1)
    for name in NAMES:
        data = np.asarray(results[name])
        if name.startswith("Ts_est"):
            data = data.reshape(len(data), 16).astype(np.float32)
        np.save(os.path.join(cache_dir, f"{name}_{category}{suf}"), data)


def load_results(cache_dir, category, register_top1):
    """Returns the dict, or None when any of the nine files is missing (the reference then
    recomputes, evaluation.py:287)."""
    suf = _suffix(register_top1)
    out = {}
    for name in NAMES:
        path = os.path.join(cache_dir, f"{name}_{category}{suf}")
        if not os.path.exists(path):
            return None
        data = np.load(path)
        if name.startswith("Ts_est"):
            data = data.reshape(-1, 4, 4)
        out[name] = data
    return out
