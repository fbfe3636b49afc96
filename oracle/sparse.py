"""NumPy restatement of the coordinate-side semantics of MinkowskiEngine 0.5.5 that the reference
relies on.  TEST INFRASTRUCTURE ONLY.

MinkowskiEngine is an un-vendored submodule of the reference (.gitmodules:1-3; deps/MinkowskiEngine
is empty), so these follow its published behaviour and are anchored on the reference's call sites:
  sparse_quantize   utils/Info/CADLib.py:106-121, datasets/CategoryDataset.py:179-197
  sparse_collate    utils/Info/CADLib.py:166-168, datasets/ChairDataset.py:204-208
  strided maps      model/resunet.py:64-72,80-87,95-103 (MinkowskiConvolution stride=2)
  kernel maps       every MinkowskiConvolution / MinkowskiConvolutionTranspose in model/resunet.py
Parity with real ME is unpinned (no ME in the container); the independent dense conv3d cross-check
in tests/test_oracle_sparse.py pins the kernel-map + convolution semantics themselves.
"""
from __future__ import annotations

import numpy as np

KERNEL_VOLUME = 27


def kernel_offsets():
    """delta_k for k = (dx+1) + 3(dy+1) + 9(dz+1): first spatial axis fastest (A.1 item 2)."""
    offs = np.zeros((27, 3), dtype=np.int64)
    for k in range(27):
        offs[k] = (k % 3 - 1, (k // 3) % 3 - 1, k // 9 - 1)
    return offs


def _pack(coords):
    c = np.asarray(coords, dtype=np.int64)
    assert c.ndim == 2 and c.shape[1] == 4
    if c.size:
        assert (c[:, 0] >= 0).all() and (c[:, 0] < 65536).all(), "batch index out of range"
        assert (np.abs(c[:, 1:]) < 32768).all(), "coordinate out of range"
    return (c[:, 0] << 48) | ((c[:, 1] + 32768) << 32) | ((c[:, 2] + 32768) << 16) | (c[:, 3] + 32768)


def sparse_quantize(grid_coords):
    """Indices of the first point of every voxel, ascending (return_index=True, maps only)."""
    g = np.asarray(grid_coords)
    g = np.floor(g).astype(np.int64)
    _, first = np.unique(g, axis=0, return_index=True)
    return np.sort(first)


def quantize_cloud(xyz, voxel_size):
    """`quant` of the reference (utils/Info/CADLib.py:106-121, datasets/CategoryDataset.py:179-197) and
    `quantize_pc` (evaluation-shapenet.py:97-107): `np.floor(xyz / voxel_size)` in the cloud's OWN type
    -- NumPy divides an f32 array by the Python float in f32 (the catalog clouds), an f64 array in f64
    (posed queries: the output of apply_transform / `pc @ R.T + t`) -- first point of every voxel kept.
    Returns (kept xyz in the input type, int32 grid coords [n,3], kept indices)."""
    xyz = np.asarray(xyz)
    if xyz.dtype not in (np.float32, np.float64):
        raise TypeError("quantize_cloud: f32 or f64 clouds only, got %s" % xyz.dtype)
    grid = np.floor(xyz / voxel_size)
    assert grid.dtype == xyz.dtype
    keep = sparse_quantize(grid)
    return xyz[keep], grid[keep].astype(np.int32), keep


def sparse_collate(coords_list, feats_list=None):
    """Prepend the batch index and concatenate (ME.utils.sparse_collate)."""
    out = []
    for b, c in enumerate(coords_list):
        c = np.asarray(c)
        bc = np.empty((c.shape[0], 4), dtype=np.int32)
        bc[:, 0] = b
        bc[:, 1:] = np.floor(c).astype(np.int32)
        out.append(bc)
    coords = np.concatenate(out, 0) if out else np.zeros((0, 4), np.int32)
    if feats_list is None:
        return coords
    feats = np.concatenate([np.asarray(f) for f in feats_list], 0)
    return coords, feats


def coordmap_stride(coords, tensor_stride, stride=2):
    """Coordinates of a stride-`stride` convolution output: unique rows of
    floor(c / (stride*ts)) * (stride*ts), ordered by first occurrence in input-row order."""
    c = np.asarray(coords, dtype=np.int64)
    cell = tensor_stride * stride
    s = c.copy()
    s[:, 1:] = np.floor_divide(c[:, 1:], cell) * cell
    keys = _pack(s)
    _, first = np.unique(keys, return_index=True)
    first = np.sort(first)
    return s[first].astype(np.int32), cell


def _lookup(table_keys_sorted, table_rows_sorted, query_keys):
    pos = np.searchsorted(table_keys_sorted, query_keys)
    pos = np.clip(pos, 0, len(table_keys_sorted) - 1) if len(table_keys_sorted) else pos
    if len(table_keys_sorted) == 0:
        return np.full(query_keys.shape, -1, dtype=np.int32)
    hit = table_keys_sorted[pos] == query_keys
    return np.where(hit, table_rows_sorted[pos], -1).astype(np.int32)


def kernel_map(in_coords, in_stride, out_coords, out_stride, kernel_size=3, transposed=False):
    """Neighbour table int32 [n_out, kvol]: in-row feeding out-row o through offset k, or -1.

    regular:    in coordinate = o + delta_k * in_stride          (out_stride in {in, 2 in})
    transposed: in coordinate = o - delta_k * out_stride         (in_stride == 2 out_stride)
    """
    ic = np.asarray(in_coords, dtype=np.int64)
    oc = np.asarray(out_coords, dtype=np.int64)
    if kernel_size == 1:
        assert in_stride == out_stride
        keys = _pack(ic)
        order = np.argsort(keys, kind="stable")
        return _lookup(keys[order], order.astype(np.int32), _pack(oc)).reshape(-1, 1)
    assert kernel_size == 3
    if transposed:
        assert in_stride == 2 * out_stride
        step, sign = out_stride, -1
    else:
        assert out_stride in (in_stride, 2 * in_stride)
        step, sign = in_stride, 1
    keys = _pack(ic)
    order = np.argsort(keys, kind="stable")
    ks, rows = keys[order], order.astype(np.int32)
    offs = kernel_offsets()
    nbr = np.full((oc.shape[0], 27), -1, dtype=np.int32)
    for k in range(27):
        q = oc.copy()
        q[:, 1:] += sign * offs[k] * step
        ok = (np.abs(q[:, 1:]) < 32768).all(axis=1)
        res = np.full(oc.shape[0], -1, dtype=np.int32)
        if ok.any():
            res[ok] = _lookup(ks, rows, _pack(q[ok]))
        nbr[:, k] = res
    return nbr


def kernel_map_triples(nbr):
    """Canonical (k, in_row, out_row) triples sorted by (k, out_row)."""
    kk, oo = np.nonzero((nbr >= 0).T)
    return kk.astype(np.int32), nbr[oo, kk].astype(np.int32), oo.astype(np.int32)


INORM_CHUNK = 256


def instance_norm(feats, seg, weight=None, bias=None, eps=1e-8):
    """ME.MinkowskiInstanceNorm over the rows of every sample (model/common.py:23-24; IN network variants):
    (x - mean) / sqrt(var + eps) * weight + bias, biased variance, eps 1e-8 inside the root
    [ME-knowledge, unpinned].  seg: row offsets of the samples.  Arithmetic as the library defines it: f64
    sums over chunks of 256 consecutive rows of a sample (sequential inside a chunk, chunk sums added in
    order), mean / var rounded to f32, 1/sqrt in f64 rounded to f32, affine part in f32."""
    x = np.asarray(feats, dtype=np.float32)
    out = np.empty_like(x)
    f32 = np.float32

    def chunked_sum(v):                      # v f64 [rows, c]: sequential over rows inside 256-row chunks
        tot = np.zeros(v.shape[1], np.float64)
        for a in range(0, len(v), INORM_CHUNK):
            acc = np.zeros(v.shape[1], np.float64)
            for row in v[a:a + INORM_CHUNK]:
                acc = acc + row
            tot = tot + acc
        return tot

    for b in range(len(seg) - 1):
        r0, r1 = int(seg[b]), int(seg[b + 1])
        if r1 == r0:
            continue
        xs = x[r0:r1]
        mean = (chunked_sum(xs.astype(np.float64)) / float(r1 - r0)).astype(f32)
        d = (xs - mean[None, :]).astype(f32)
        var = (chunked_sum(d.astype(np.float64) * d.astype(np.float64)) / float(r1 - r0)).astype(f32)
        inv = (1.0 / np.sqrt(var.astype(np.float64) + np.float64(f32(eps)))).astype(f32)
        v = (d * inv[None, :]).astype(f32)
        if weight is not None:
            v = (v * np.asarray(weight, f32).reshape(1, -1)).astype(f32)
        if bias is not None:
            v = (v + np.asarray(bias, f32).reshape(1, -1)).astype(f32)
        out[r0:r1] = v
    return out


def segmented_max(feats, batch_index, n_batch):
    """Per-sample column-wise max (model/fc.py:23-29,124-125)."""
    feats = np.asarray(feats, dtype=np.float32)
    out = np.full((n_batch, feats.shape[1]), -np.inf, dtype=np.float32)
    for b in range(n_batch):
        m = batch_index == b
        if m.any():
            out[b] = feats[m].max(0)
    return out
