"""Synthetic workload of SURVEY.md section 8(d): primitive-union point clouds, seeded SE(3) query
poses, random-init ResUNetBN2C / embedding weights (the reference checkpoints are not available:
.MISSING_LARGE_BLOBS of the reference).  Pure NumPy, deterministic across machines (Philox).
"""
from __future__ import annotations

import numpy as np

CLOUD_KEY = 0xC0125A12


def _rng(key, counter):
    return np.random.Generator(np.random.Philox(key=key, counter=counter))


def make_cloud(cloud_id, n_points=15000):
    """Union of 3..8 primitives (box surfaces, thin cylinders), sampled by area, then the
    reference normalisation (utils/preprocess.py:32-36: centre, divide by max radius). f32 [n,3]."""
    rng = _rng(CLOUD_KEY, cloud_id)
    n_parts = int(rng.integers(3, 9))
    prims = []
    for _ in range(n_parts):
        centre = rng.uniform(-0.4, 0.4, 3)
        if rng.random() < 0.6:
            half = rng.uniform(0.05, 0.5, 3)
            area = 8.0 * (half[0] * half[1] + half[1] * half[2] + half[0] * half[2])
            prims.append(("box", centre, half, area))
        else:
            r = rng.uniform(0.01, 0.05)
            h = rng.uniform(0.2, 0.8)
            axis = int(rng.integers(0, 3))
            area = 2.0 * np.pi * r * h
            prims.append(("cyl", centre, (r, h, axis), area))
    areas = np.array([p[3] for p in prims])
    counts = rng.multinomial(n_points, areas / areas.sum())
    pts = []
    for (kind, centre, par, _), cnt in zip(prims, counts):
        if cnt == 0:
            continue
        if kind == "box":
            half = par
            face_area = np.array([half[1] * half[2], half[1] * half[2], half[0] * half[2],
                                  half[0] * half[2], half[0] * half[1], half[0] * half[1]])
            face = rng.choice(6, size=cnt, p=face_area / face_area.sum())
            p = rng.uniform(-1.0, 1.0, (cnt, 3)) * half
            ax = face // 2
            sign = np.where(face % 2 == 0, -1.0, 1.0)
            p[np.arange(cnt), ax] = sign * half[ax]
        else:
            r, h, axis = par
            th = rng.uniform(0, 2 * np.pi, cnt)
            z = rng.uniform(-h / 2, h / 2, cnt)
            p = np.zeros((cnt, 3))
            o = [a for a in range(3) if a != axis]
            p[:, o[0]] = r * np.cos(th)
            p[:, o[1]] = r * np.sin(th)
            p[:, axis] = z
        pts.append(p + centre)
    pc = np.concatenate(pts, 0)
    pc = pc[rng.permutation(len(pc))].astype(np.float32)
    pc -= pc.mean(0)
    pc = pc / np.max(np.linalg.norm(pc, 2, 1))
    return pc.astype(np.float32)


def euler2mat(ai, aj, ak):
    """transforms3d.euler.euler2mat(ai, aj, ak) with the default 'sxyz' axes:
    R = Rz(ak) @ Ry(aj) @ Rx(ai)  (used by evaluation-shapenet.py:79-94, utils/eval_pose.py:114)."""
    ci, si = np.cos(ai), np.sin(ai)
    cj, sj = np.cos(aj), np.sin(aj)
    ck, sk = np.cos(ak), np.sin(ak)
    Rx = np.array([[1, 0, 0], [0, ci, -si], [0, si, ci]])
    Ry = np.array([[cj, 0, sj], [0, 1, 0], [-sj, 0, cj]])
    Rz = np.array([[ck, -sk, 0], [sk, ck, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def random_pose(pose_id, max_angle=np.pi, max_trans=1.0):
    """Seeded SE(3): rotation from 3 Euler angles U(-max_angle, max_angle), translation
    U(-max_trans, max_trans)^3 (evaluation-shapenet.py:79-94)."""
    rng = _rng(CLOUD_KEY ^ 0x5EED, pose_id)
    a = rng.uniform(-max_angle, max_angle, 3)
    T = np.eye(4)
    T[:3, :3] = euler2mat(a[0], a[1], a[2])
    T[:3, 3] = rng.uniform(-max_trans, max_trans, 3)
    return T


def apply_pose(pc, T, dtype=np.float32):
    """pc @ R.T + t in f64.  dtype=np.float64 keeps what the reference's apply_transform
    (utils/preprocess.py:39-48) and generate_test_pc_pair (evaluation-shapenet.py:115-119) hand to the
    quantiser: an f64 cloud, floored in f64, cast to f32 only after the selection."""
    return (pc.astype(np.float64) @ T[:3, :3].T + T[:3, 3]).astype(dtype)


# ---- weights -----------------------------------------------------------------------------------
_CONVS = [
    ("conv1", 27, 1, 32), ("conv2", 27, 32, 64), ("conv3", 27, 64, 128), ("conv4", 27, 128, 256),
    ("conv4_tr", 27, 256, 128), ("conv3_tr", 27, 256, 64), ("conv2_tr", 27, 128, 64),
]
_BLOCKS = [("block1", 32), ("block2", 64), ("block3", 128), ("block4", 256), ("block4_tr", 128),
           ("block3_tr", 64), ("block2_tr", 64)]
_NORMS = [("norm1", 32), ("norm2", 64), ("norm3", 128), ("norm4", 256), ("norm4_tr", 128),
          ("norm3_tr", 64), ("norm2_tr", 64)]


def make_state_dicts(seed=31):
    """Random-init weights with the reference's state-dict names and shapes:
    Kaiming-normal conv kernels (std = sqrt(2 / (kvol * cin))), BN running_mean ~ N(0, 0.1),
    running_var ~ U(0.5, 1.5), gamma = 1, beta = 0.  Returns (state_dict, embedding_state_dict) of
    NumPy f32 arrays."""
    rng = _rng(0xBEEF, seed)
    sd = {}

    def conv(name, kvol, cin, cout):
        std = np.sqrt(2.0 / (kvol * cin))
        shape = (kvol, cin, cout) if kvol > 1 else (cin, cout)
        sd[name + ".kernel"] = (rng.standard_normal(shape) * std).astype(np.float32)

    def bn(name, c):
        sd[name + ".bn.weight"] = np.ones(c, np.float32)
        sd[name + ".bn.bias"] = np.zeros(c, np.float32)
        sd[name + ".bn.running_mean"] = (rng.standard_normal(c) * 0.1).astype(np.float32)
        sd[name + ".bn.running_var"] = rng.uniform(0.5, 1.5, c).astype(np.float32)
        sd[name + ".bn.num_batches_tracked"] = np.zeros((), np.int64)

    for name, kvol, cin, cout in _CONVS:
        conv(name, kvol, cin, cout)
    for name, c in _NORMS:
        bn(name, c)
    for name, c in _BLOCKS:
        conv(name + ".conv1", 27, c, c)
        bn(name + ".norm1", c)
        conv(name + ".conv2", 27, c, c)
        bn(name + ".norm2", c)
    conv("conv1_tr", 1, 96, 64)
    conv("final", 1, 64, 16)
    sd["final.bias"] = (rng.standard_normal((1, 16)) * 0.01).astype(np.float32)

    emb = {}
    emb["final.final.kernel"] = (rng.standard_normal((256, 1024)) * np.sqrt(2.0 / 256)).astype(np.float32)
    emb["final.final.bias"] = (rng.standard_normal((1, 1024)) * 0.01).astype(np.float32)
    emb["fc1.weight"] = (rng.standard_normal((512, 1024)) * np.sqrt(2.0 / 1024)).astype(np.float32)
    emb["fc1.bias"] = (rng.standard_normal(512) * 0.01).astype(np.float32)
    emb["bn1.weight"] = np.ones(512, np.float32)
    emb["bn1.bias"] = np.zeros(512, np.float32)
    emb["bn1.running_mean"] = (rng.standard_normal(512) * 0.1).astype(np.float32)
    emb["bn1.running_var"] = rng.uniform(0.5, 1.5, 512).astype(np.float32)
    emb["bn1.num_batches_tracked"] = np.zeros((), np.int64)
    emb["fc2.weight"] = (rng.standard_normal((256, 512)) * np.sqrt(2.0 / 512)).astype(np.float32)
    emb["fc2.bias"] = (rng.standard_normal(256) * 0.01).astype(np.float32)
    return sd, emb


def make_descriptors(n, d=256, seed=1234):
    """Row-normalised standard-normal descriptors (config C5 of BASELINE.json)."""
    rng = _rng(0xD35C, seed)
    x = rng.standard_normal((n, d)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return x.astype(np.float32)
