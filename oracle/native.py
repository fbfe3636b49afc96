"""Build and bind oracle/corsair_oracle.c (CPU restatement; TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from ctypes import POINTER, c_double, c_float, c_int, c_int32, c_int64, c_uint32, c_uint64, c_void_p

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "corsair_oracle.c")
BUILD_DIR = os.path.join(_HERE, "_build")
# ORACLE_SANITIZE=1: the AddressSanitizer + UndefinedBehaviorSanitizer build of the same source (CPU only,
# never on the GPU box).  The C oracle is the arbiter of every parity test and is 800 lines of hand-rolled index
# arithmetic: tests/test_oracle_sanitized_cpu.py re-runs the oracle's CPU tests on this build in a child interpreter
# started with LD_PRELOAD=libasan (an instrumented .so cannot be loaded into an uninstrumented python otherwise).
SANITIZE = os.environ.get("ORACLE_SANITIZE", "0") == "1"
LIB = os.path.join(BUILD_DIR, "libcorsair_oracle_san.so" if SANITIZE else "libcorsair_oracle.so")
SAN_FLAGS = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer"]

_lib = None


def sanitizer_runtime():
    """Path of gcc's libasan.so (what the child interpreter preloads)."""
    return subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()


def build(force=False):
    os.makedirs(BUILD_DIR, exist_ok=True)
    newest = max(os.path.getmtime(SRC), os.path.getmtime(os.path.join(_HERE, "kmeans_draws.h")))
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= newest:
        return LIB
    opt = SAN_FLAGS if SANITIZE else ["-O3"]
    cmd = ["gcc"] + opt + ["-mavx2", "-mfma", "-ffp-contract=off", "-fopenmp", "-shared", "-fPIC",
                           "-std=c11", "-I", _HERE, SRC, "-o", LIB, "-lm"]
    subprocess.check_call(cmd)
    return LIB


def num_threads():
    return int(load().oc_get_threads())


def cores_visible():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def cgroup_cpu_quota():
    """CPUs the cgroup may use (cpu.max / cfs quota), or None when unlimited / unreadable."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()[:2]
        if q != "max":
            return max(1, int(int(q) / int(p)))
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            q = int(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            p = int(f.read())
        if q > 0:
            return max(1, q // p)
    except (OSError, ValueError):
        pass
    return None


def cpu_share():
    n = cores_visible()
    q = cgroup_cpu_quota()
    if q is not None:
        n = min(n, q)
    cap = int(os.environ.get("ORACLE_THREADS", "16") or 16)
    return max(1, min(n, cap))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return None if a is None else a.ctypes.data_as(c_void_p)


def load():
    global _lib
    if _lib is None:
        build()
        lib = ctypes.CDLL(LIB)
        vp = c_void_p
        lib.oc_conv_fwd.argtypes = [vp, c_int, c_int64, vp, c_int, c_int, vp, c_int, vp, vp, vp,
                                    c_int, c_int, vp, c_int]
        lib.oc_affine_act.argtypes = [c_int64, c_int, vp, c_int, vp, vp, vp, c_int, c_int, vp, c_int]
        lib.oc_row_l2_normalize.argtypes = [c_int64, c_int, vp, c_int, c_float, vp, c_int]
        lib.oc_dist2_matrix.argtypes = [vp, c_int64, vp, c_int64, c_int, vp]
        lib.oc_knn.argtypes = [vp, c_int64, vp, c_int64, c_int, c_int, vp, vp, vp, vp, vp]
        lib.oc_chamfer_1dir.argtypes = [vp, c_int64, vp, c_int64, vp]
        lib.oc_chamfer_1dir.restype = c_double
        lib.oc_rng_indices.argtypes = [c_uint64, c_uint64, c_int, c_uint32, vp]
        lib.oc_rigid_fit.argtypes = [vp, vp, c_int, vp, vp]
        lib.oc_rigid_fit2.argtypes = [vp, vp, c_int, vp, vp]
        lib.oc_rigid_fit2.restype = c_int
        lib.oc_rigid_fit_force_jacobi.argtypes = [c_int]
        lib.oc_ransac.argtypes = [vp, vp, c_int64, c_double, c_int, c_int, c_double, c_uint64, vp,
                                  POINTER(c_int32), POINTER(c_double), POINTER(c_int32)]
        lib.oc_ransac_batch.argtypes = [vp, vp, vp, c_int, c_double, c_int, c_int, c_double,
                                        c_uint64, vp, vp, vp, vp]
        lib.oc_symcut_fit_one.argtypes = [vp, c_int, vp, c_int, c_int, c_int, c_int, c_int, c_int,
                                          c_uint64, vp, vp, POINTER(c_double), POINTER(c_double), vp]
        lib.oc_symcut_fit.argtypes = [vp, c_int, vp, c_int, vp, c_int, c_int, c_int, c_int, c_int,
                                      c_uint64, vp, vp, vp, vp]
        lib.oc_symcut_labels.argtypes = [vp, c_int, c_int, vp, vp]
        lib.oc_voxel_index.argtypes = [vp, c_int64, c_float, vp]
        lib.oc_set_threads.argtypes = [c_int]
        lib.oc_get_threads.restype = c_int
        # OpenMP threads = the CPU SHARE of the process, not the logical CPUs it can see: a one-GPU box of the
        # pool shows every CPU of its host in the affinity mask but is entitled to 16 of them, and spinning
        # OpenMP threads beyond the share stall each other (a round-3 run with one thread per visible CPU sat
        # > 7 minutes in one oracle call).  Share = min(affinity, cgroup quota if any, ORACLE_THREADS or 16);
        # bench.py prints threads used next to cores_visible.
        lib.oc_set_threads(cpu_share())
        _lib = lib
    return _lib


# ---- numpy front ends ------------------------------------------------------------------------
def conv_fwd(nbr, feats, weight, scale=None, shift=None, residual=None, relu=False, n_out=None):
    """feats f32 [n_in,cin]; weight f32 [kvol,cin,cout] (or [cin,cout]); nbr int32 [n_out,kvol] or None."""
    lib = load()
    feats = _f32(feats)
    weight = _f32(weight)
    if weight.ndim == 2:
        weight = weight[None]
    kvol, cin, cout = weight.shape
    if nbr is not None:
        nbr = np.ascontiguousarray(nbr, dtype=np.int32)
        assert nbr.shape[1] == kvol
        n_out = nbr.shape[0]
    else:
        assert kvol == 1
        n_out = feats.shape[0]
    out = np.empty((n_out, cout), dtype=np.float32)
    scale = None if scale is None else _f32(scale).reshape(-1)
    shift = None if shift is None else _f32(shift).reshape(-1)
    residual = None if residual is None else _f32(residual)
    lib.oc_conv_fwd(_p(nbr), kvol, n_out, _p(feats), feats.shape[1], cin, _p(weight), cout,
                    _p(scale), _p(shift), _p(residual), 0 if residual is None else residual.shape[1],
                    int(bool(relu)), _p(out), cout)
    return out


def affine_act(x, scale=None, shift=None, residual=None, relu=False):
    lib = load()
    x = _f32(x)
    n, c = x.shape
    out = np.empty_like(x)
    scale = None if scale is None else _f32(scale).reshape(-1)
    shift = None if shift is None else _f32(shift).reshape(-1)
    residual = None if residual is None else _f32(residual)
    lib.oc_affine_act(n, c, _p(x), c, _p(scale), _p(shift), _p(residual),
                      0 if residual is None else residual.shape[1], int(bool(relu)), _p(out), c)
    return out


def row_l2_normalize(x, eps=0.0):
    lib = load()
    x = _f32(x)
    out = np.empty_like(x)
    lib.oc_row_l2_normalize(x.shape[0], x.shape[1], _p(x), x.shape[1], float(eps), _p(out), x.shape[1])
    return out


def dist2_matrix(q, x):
    lib = load()
    q, x = _f32(q), _f32(x)
    out = np.empty((q.shape[0], x.shape[0]), dtype=np.float64)
    lib.oc_dist2_matrix(_p(q), q.shape[0], _p(x), x.shape[0], q.shape[1], _p(out))
    return out


def knn(qf, tf, k, qlabel=None, tlabel=None, perm=None, return_distance=False):
    lib = load()
    qf, tf = _f32(qf), _f32(tf)
    idx = np.empty((qf.shape[0], k), dtype=np.int32)
    dist = np.empty((qf.shape[0], k), dtype=np.float64)
    if qlabel is not None:
        qlabel = np.ascontiguousarray(qlabel, dtype=np.int32)
        tlabel = np.ascontiguousarray(tlabel, dtype=np.int32)
        p8 = np.full(8, -3, dtype=np.int32)
        p8[: len(perm)] = perm
        perm = p8
    lib.oc_knn(_p(qf), qf.shape[0], _p(tf), tf.shape[0], qf.shape[1], k, _p(qlabel), _p(tlabel),
               _p(perm), _p(idx), _p(dist))
    return (idx, dist) if return_distance else idx


def chamfer_1dir(src, tgt, T):
    lib = load()
    src, tgt = _f32(src), _f32(tgt)
    T = _f32(np.asarray(T).reshape(4, 4))
    return float(lib.oc_chamfer_1dir(_p(src), src.shape[0], _p(tgt), tgt.shape[0], _p(T)))


def rng_indices(seed, itr, n, m):
    lib = load()
    out = np.empty(n, dtype=np.int32)
    lib.oc_rng_indices(seed, itr, n, m, _p(out))
    return out


def rigid_fit(ps, pt, return_path=False, force_jacobi=False):
    """oc_rigid_fit2.  return_path: also the eigen-solver used (0 = characteristic polynomial + adjugate,
    1 = Jacobi fallback); force_jacobi: the fallback alone (tests)."""
    lib = load()
    ps = np.ascontiguousarray(ps, dtype=np.float64)
    pt = np.ascontiguousarray(pt, dtype=np.float64)
    R = np.empty((3, 3), dtype=np.float64)
    t = np.empty(3, dtype=np.float64)
    lib.oc_rigid_fit_force_jacobi(1 if force_jacobi else 0)
    try:
        path = lib.oc_rigid_fit2(_p(ps), _p(pt), ps.shape[0], _p(R), _p(t))
    finally:
        lib.oc_rigid_fit_force_jacobi(0)
    return (R, t, path) if return_path else (R, t)


def ransac(src, tgt, max_corr, ransac_n=10, max_iter=100000, confidence=0.999, seed=0, force_jacobi=False):
    lib = load()
    src, tgt = _f32(src), _f32(tgt)
    T = np.empty(16, dtype=np.float32)
    inl, it = c_int32(0), c_int32(0)
    rm = c_double(0.0)
    lib.oc_rigid_fit_force_jacobi(1 if force_jacobi else 0)
    try:
        lib.oc_ransac(_p(src), _p(tgt), src.shape[0], float(max_corr), ransac_n, max_iter,
                      float(confidence), seed, _p(T), ctypes.byref(inl), ctypes.byref(rm),
                      ctypes.byref(it))
    finally:
        lib.oc_rigid_fit_force_jacobi(0)
    return T.reshape(4, 4), inl.value, rm.value, it.value


def ransac_batch(src, tgt, offsets, max_corr, ransac_n=10, max_iter=100000, confidence=0.999, seed=0):
    lib = load()
    src, tgt = _f32(src), _f32(tgt)
    off = np.ascontiguousarray(offsets, dtype=np.int64)
    n = len(off) - 1
    T = np.empty((n, 16), dtype=np.float32)
    inl = np.empty(n, dtype=np.int32)
    rm = np.empty(n, dtype=np.float64)
    it = np.empty(n, dtype=np.int32)
    lib.oc_ransac_batch(_p(src), _p(tgt), _p(off), n, float(max_corr), ransac_n, max_iter,
                        float(confidence), seed, _p(T), _p(inl), _p(rm), _p(it))
    return T.reshape(n, 4, 4), inl, rm, it


def symcut_fit(feat, xyz, anchors, K, n_nn=50, n_init=10, max_iter=300, seed=0):
    lib = load()
    feat, xyz = _f32(feat), _f32(xyz)
    anchors = np.ascontiguousarray(anchors, dtype=np.int32)
    na = len(anchors)
    centers = np.empty((na, 4, 3), dtype=np.float64)
    counts = np.empty((na, 4), dtype=np.int32)
    mcd = np.empty(na, dtype=np.float64)
    mer = np.empty(na, dtype=np.float64)
    lib.oc_symcut_fit(_p(feat), feat.shape[1], _p(xyz), feat.shape[0], _p(anchors), na, K, n_nn,
                      n_init, max_iter, seed, _p(centers), _p(counts), _p(mcd), _p(mer))
    return centers, counts, mcd, mer


def symcut_nn_rows(feat, xyz, anchor, K, n_nn=50, n_init=10, max_iter=300, seed=0):
    """Rows of the n_nn feature-nearest voxels of one anchor (ascending row order)."""
    lib = load()
    feat, xyz = _f32(feat), _f32(xyz)
    centers = np.empty(12, dtype=np.float64)
    counts = np.empty(4, dtype=np.int32)
    a, b = c_double(0), c_double(0)
    rows = np.empty(n_nn, dtype=np.int32)
    lib.oc_symcut_fit_one(_p(feat), feat.shape[1], _p(xyz), feat.shape[0], int(anchor), K, n_nn,
                          n_init, max_iter, seed, _p(centers), _p(counts), ctypes.byref(a),
                          ctypes.byref(b), _p(rows))
    return rows


def symcut_labels(xyz, K, centers):
    lib = load()
    xyz = _f32(xyz)
    centers = np.ascontiguousarray(centers, dtype=np.float64).reshape(-1)
    c12 = np.zeros(12, dtype=np.float64)
    c12[: len(centers)] = centers
    labels = np.empty(xyz.shape[0], dtype=np.int32)
    lib.oc_symcut_labels(_p(xyz), xyz.shape[0], K, _p(c12), _p(labels))
    return labels


def voxel_index(xyz, voxel):
    lib = load()
    xyz = _f32(xyz)
    out = np.empty(xyz.shape, dtype=np.int32)
    lib.oc_voxel_index(_p(xyz), xyz.shape[0], float(np.float32(voxel)), _p(out))
    return out
