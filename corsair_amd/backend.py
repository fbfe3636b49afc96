"""Thin torch-tensor front ends of the C ABI (include/corsair_hip.h).

Everything here launches hand-written HIP kernels through libcorsair_hip.so; torch only owns the
device buffers and the stream.  No CPU fallback: inputs must be CUDA(HIP) tensors.
"""
from __future__ import annotations

import ctypes
from ctypes import c_void_p

import numpy as np
import torch

from . import _lib
from ._lib import check, i32_array, i64_array, ptr, stream_ptr

KERNEL_VOLUME = 27


def _dev(t, dtype, what):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.CorsairHipError(f"{what} must be a device tensor (the HIP path has no CPU fallback)")
    if t.dtype != dtype:
        raise TypeError(f"{what} must be {dtype}, got {t.dtype}")
    return t


def _rows(t, what):
    """(tensor, leading dimension) of a 2-D tensor whose rows are contiguous."""
    if t.dim() != 2 or t.stride(1) != 1:
        raise ValueError(f"{what} must be 2-D with unit inner stride")
    return t, t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])


class CoordMap:
    """Owns one cs_coordmap handle (coordinates of one tensor stride + hash index)."""

    def __init__(self, handle, device):
        self._h = c_void_p(handle)
        self.device = device
        lib = _lib.load()
        self.n = int(lib.cs_coordmap_size(self._h))
        self.tensor_stride = int(lib.cs_coordmap_tensor_stride(self._h))
        self._coords = None

    @classmethod
    def create(cls, coords, tensor_stride=1):
        _lib.require_gpu()
        coords = _dev(coords, torch.int32, "coordinates").contiguous()
        if coords.dim() != 2 or coords.shape[1] != 4:
            raise ValueError("coordinates must be int32 [N, 4] (batch, x, y, z)")
        out = c_void_p()
        check(_lib.load().cs_coordmap_create(ptr(coords), coords.shape[0], tensor_stride,
                                             stream_ptr(), ctypes.byref(out)))
        return cls(out.value, coords.device)

    @classmethod
    def pyramid(cls, coords, n_levels=4, n_batch=0, tensor_stride=1):
        """The maps of tensor strides ts, 2 ts, 4 ts, ... in ONE library call (cs_coordmap_pyramid): the same maps as
        create() followed by chained stride(2) calls, with one host wait instead of one per level.  n_batch > 0 announces
        sparse_collate's order (rows grouped by sample, batch indices < n_batch)."""
        _lib.require_gpu()
        coords = _dev(coords, torch.int32, "coordinates").contiguous()
        if coords.dim() != 2 or coords.shape[1] != 4:
            raise ValueError("coordinates must be int32 [N, 4] (batch, x, y, z)")
        outs = (c_void_p * n_levels)()
        check(_lib.load().cs_coordmap_pyramid(ptr(coords), coords.shape[0], tensor_stride, n_levels, int(n_batch or 0),
                                              stream_ptr(), outs))
        return [cls(h, coords.device) for h in outs]

    def stride(self, s=2):
        out = c_void_p()
        check(_lib.load().cs_coordmap_stride(self._h, s, stream_ptr(), ctypes.byref(out)))
        return CoordMap(out.value, self.device)

    @property
    def coords(self):
        """int32 [n,4] device tensor (a copy owned by torch)."""
        if self._coords is None:
            t = torch.empty((self.n, 4), dtype=torch.int32, device=self.device)
            if self.n:
                src = _lib.load().cs_coordmap_coords(self._h)
                _hip_memcpy_d2d(t.data_ptr(), src, self.n * 16)
            self._coords = t
        return self._coords

    def __del__(self):
        try:
            if self._h:
                _lib.load().cs_coordmap_free(self._h)
                self._h = None
        except Exception:
            pass


def _hip_memcpy_d2d(dst, src, nbytes):
    # tiny helper through torch: wrap the library-owned memory without taking ownership
    import ctypes as C

    hip = _hip_runtime()
    rc = hip.hipMemcpyAsync(c_void_p(dst), c_void_p(src), C.c_size_t(nbytes), 3,
                            c_void_p(torch.cuda.current_stream().cuda_stream))
    if rc != 0:
        raise _lib.CorsairHipError(f"hipMemcpyAsync failed with {rc}")


_hip = None


def _hip_runtime():
    global _hip
    if _hip is None:
        _hip = ctypes.CDLL("libamdhip64.so")
    return _hip


class KernelMap:
    """Owns one cs_kernelmap handle (int32 [n_out,27] neighbour table on the device)."""

    def __init__(self, handle, in_map, out_map, transposed):
        self._h = c_void_p(handle)
        lib = _lib.load()
        self.n_out = int(lib.cs_kernelmap_rows(self._h))
        self.n_in = in_map.n
        self._num_pairs = None
        self.transposed = transposed
        self.device = in_map.device

    @property
    def num_pairs(self):
        """Resolved on first use: building a map does not wait for its pair count."""
        if self._num_pairs is None:
            self._num_pairs = int(_lib.load().cs_kernelmap_num_pairs(self._h))
        return self._num_pairs

    @classmethod
    def build(cls, in_map, out_map, kernel_size=3, transposed=False):
        out = c_void_p()
        check(_lib.load().cs_kernelmap_build(in_map._h, out_map._h, kernel_size,
                                             1 if transposed else 0, stream_ptr(),
                                             ctypes.byref(out)))
        return cls(out.value, in_map, out_map, transposed)

    @classmethod
    def build_many(cls, specs):
        """The kernel maps of one batch in one library call (cs_kernelmap_build_many: independent chains on up to three
        streams).  specs: (in_map, out_map[, kernel_size[, transposed]]) tuples; returns the maps in that order."""
        full = [(sp[0], sp[1], sp[2] if len(sp) > 2 else 3, bool(sp[3]) if len(sp) > 3 else False) for sp in specs]
        n = len(full)
        if n == 0:
            return []
        VP, CI = c_void_p * n, ctypes.c_int * n
        ins = VP(*[sp[0]._h.value for sp in full])
        outs = VP(*[sp[1]._h.value for sp in full])
        ks = CI(*[int(sp[2]) for sp in full])
        tr = CI(*[1 if sp[3] else 0 for sp in full])
        kms = VP()
        check(_lib.load().cs_kernelmap_build_many(n, ins, outs, ks, tr, stream_ptr(), kms))
        return [cls(kms[i], full[i][0], full[i][1], full[i][3]) for i in range(n)]

    def export(self):
        """Canonical (k, in_row, out_row) int32 triples sorted by (k, out_row)."""
        n = max(self.num_pairs, 1)
        k = torch.empty(n, dtype=torch.int32, device=self.device)
        i = torch.empty(n, dtype=torch.int32, device=self.device)
        o = torch.empty(n, dtype=torch.int32, device=self.device)
        got = check(_lib.load().cs_kernelmap_export(self._h, ptr(k), ptr(i), ptr(o), n, stream_ptr()))
        return k[:got], i[:got], o[:got]

    def table(self):
        t = torch.empty((self.n_out, KERNEL_VOLUME), dtype=torch.int32, device=self.device)
        if self.n_out:
            _hip_memcpy_d2d(t.data_ptr(), _lib.load().cs_kernelmap_table(self._h), self.n_out * 27 * 4)
        return t

    def __del__(self):
        try:
            if self._h:
                _lib.load().cs_kernelmap_free(self._h)
                self._h = None
        except Exception:
            pass


def conv_fwd(kmap, x, weight, scale=None, shift=None, residual=None, relu=False, out=None,
             n_out=None):
    """Sparse convolution forward with fused epilogue (cs_conv_fwd).  x/out/residual may be column
    slices of wider row-major buffers."""
    x, ld_in = _rows(_dev(x, torch.float32, "input features"), "input features")
    weight = _dev(weight, torch.float32, "kernel").contiguous()
    if weight.dim() == 2:
        cin, cout = weight.shape
    else:
        cin, cout = weight.shape[1], weight.shape[2]
    if x.shape[1] != cin:
        raise ValueError(f"input has {x.shape[1]} channels, kernel expects {cin}")
    n_in = x.shape[0]
    if kmap is None:
        n_out = n_in
    else:
        n_out = kmap.n_out
    if out is None:
        out = torch.empty((n_out, cout), dtype=torch.float32, device=x.device)
    out, ld_out = _rows(out, "output")
    if out.shape[0] != n_out or out.shape[1] != cout:
        raise ValueError("output buffer has the wrong shape")
    ld_res = 0
    if residual is not None:
        residual, ld_res = _rows(_dev(residual, torch.float32, "residual"), "residual")
    check(_lib.load().cs_conv_fwd(kmap._h if kmap is not None else None, n_in, n_out, ptr(x), ld_in,
                                  cin, ptr(weight), cout, ptr(scale), ptr(shift), ptr(residual),
                                  ld_res, 1 if relu else 0, ptr(out), ld_out, stream_ptr()))
    return out


def affine_act(x, scale=None, shift=None, residual=None, relu=False, out=None):
    x, ld_in = _rows(_dev(x, torch.float32, "input"), "input")
    if out is None:
        out = torch.empty((x.shape[0], x.shape[1]), dtype=torch.float32, device=x.device)
    out, ld_out = _rows(out, "output")
    ld_res = 0
    if residual is not None:
        residual, ld_res = _rows(residual, "residual")
    check(_lib.load().cs_affine_act(x.shape[0], x.shape[1], ptr(x), ld_in, ptr(scale), ptr(shift),
                                    ptr(residual), ld_res, 1 if relu else 0, ptr(out), ld_out,
                                    stream_ptr()))
    return out


def row_l2_normalize(x, eps=0.0, out=None):
    x, ld_in = _rows(_dev(x, torch.float32, "input"), "input")
    if out is None:
        out = torch.empty((x.shape[0], x.shape[1]), dtype=torch.float32, device=x.device)
    out, ld_out = _rows(out, "output")
    check(_lib.load().cs_row_l2_normalize(x.shape[0], x.shape[1], ptr(x), ld_in, float(eps), ptr(out),
                                          ld_out, stream_ptr()))
    return out


def segmented_max(x, coords, n_batch):
    """Per-sample column max; `coords` is the int32 [n,4] coordinate tensor (batch in column 0)."""
    x, ld_in = _rows(_dev(x, torch.float32, "input"), "input")
    coords = _dev(coords, torch.int32, "coords")
    out = torch.empty((n_batch, x.shape[1]), dtype=torch.float32, device=x.device)
    check(_lib.load().cs_segmented_max(x.shape[0], x.shape[1], ptr(x), ld_in, ptr(coords),
                                       coords.stride(0), n_batch, ptr(out), stream_ptr()))
    return out


def instance_norm(x, seg, weight=None, bias=None, eps=1e-8):
    """cs_instance_norm: x f32 [n,c] rows grouped by sample, seg int32 [n_batch+1] device row offsets."""
    x, ld_in = _rows(_dev(x, torch.float32, "input"), "input")
    seg = _dev(seg, torch.int32, "segment offsets").contiguous()
    out = torch.empty((x.shape[0], x.shape[1]), dtype=torch.float32, device=x.device)
    w = _dev(weight, torch.float32, "weight").contiguous().reshape(-1) if weight is not None else None
    b = _dev(bias, torch.float32, "bias").contiguous().reshape(-1) if bias is not None else None
    for t, name in ((w, "weight"), (b, "bias")):
        if t is not None and t.numel() != x.shape[1]:
            raise ValueError("instance_norm: %s must have %d entries" % (name, x.shape[1]))
    check(_lib.load().cs_instance_norm(x.shape[0], x.shape[1], ptr(x), ld_in, ptr(seg), seg.numel() - 1,
                                       ptr(w) if w is not None else None, ptr(b) if b is not None else None,
                                       float(eps), ptr(out), out.stride(0), stream_ptr()))
    return out


def voxelize(xyz, offsets, voxel_size):
    """cs_voxelize / cs_voxelize_f64: xyz f32 or f64 [n,3] device, offsets host list.  The grid index is
    floor(x / voxel) in the cloud's own type, like the reference's NumPy expression: f32 for the catalog
    clouds (utils/Info/CADLib.py:106-121), f64 for posed queries (datasets/CategoryDataset.py:179-197,
    evaluation-shapenet.py:97-119).  Returns (keep_idx int64 [m], grid int32 [m,4], out_offsets list)."""
    f64 = torch.is_tensor(xyz) and xyz.dtype == torch.float64
    xyz = _dev(xyz, torch.float64 if f64 else torch.float32, "xyz").contiguous()
    n = xyz.shape[0]
    nseg = len(offsets) - 1
    keep = torch.empty(max(n, 1), dtype=torch.int64, device=xyz.device)
    grid = torch.empty((max(n, 1), 4), dtype=torch.int32, device=xyz.device)
    h_off = i64_array(offsets)
    h_out = (ctypes.c_int64 * (nseg + 1))()
    fn = _lib.load().cs_voxelize_f64 if f64 else _lib.load().cs_voxelize
    check(fn(ptr(xyz), h_off, nseg, float(voxel_size), ptr(keep), ptr(grid), h_out, stream_ptr()))
    out_off = [int(v) for v in h_out]
    m = out_off[-1]
    return keep[:m], grid[:m], out_off


class TopkCatalog:
    """A fixed retrieval library (cs_topk_catalog): what the matrix-core top-k derives from the catalog rows is made once
    and reused by every l2_topk against it.  Holds a reference to the descriptor tensor."""

    def __init__(self, x):
        self.x = _dev(x, torch.float32, "catalog").contiguous()
        out = c_void_p()
        check(_lib.load().cs_topk_catalog_create(ptr(self.x), self.x.shape[0], self.x.shape[1], stream_ptr(),
                                                 ctypes.byref(out)))
        self._h = out
        self.shape = self.x.shape

    def __del__(self):
        try:
            if self._h:
                _lib.load().cs_topk_catalog_free(self._h)
                self._h = None
        except Exception:
            pass


def l2_topk(q, x, k, return_distance=False, squared=False):
    """x: catalog tensor or TopkCatalog.  squared=True (with return_distance): the squared distances the ranking was
    made on (shard merges)."""
    q = _dev(q, torch.float32, "queries").contiguous()
    idx = torch.empty((q.shape[0], k), dtype=torch.int64, device=q.device)
    dist = torch.empty((q.shape[0], k), dtype=torch.float64, device=q.device) if return_distance else None
    if isinstance(x, TopkCatalog):
        if x.shape[1] != q.shape[1]:
            raise ValueError("query and catalog dimensions differ")
        check(_lib.load().cs_l2_topk_catalog(ptr(q), q.shape[0], x._h, k, ptr(idx), ptr(dist), 1 if squared else 0,
                                             stream_ptr()))
        return (idx, dist) if return_distance else idx
    x = _dev(x, torch.float32, "catalog").contiguous()
    fn = _lib.load().cs_l2_topk_sq if squared else _lib.load().cs_l2_topk
    check(fn(ptr(q), q.shape[0], ptr(x), x.shape[0], q.shape[1], k, ptr(idx), ptr(dist), stream_ptr()))
    return (idx, dist) if return_distance else idx


def knn_feat(qf, qoff, tf, toff, k, qseg=None, tseg=None, qlabel=None, tlabel=None, perm=None,
             return_distance=False):
    """Batched feature k-NN.  qoff/toff: host offset lists of the segment tables; qseg/tseg: per
    problem segment ids (default: problem p uses segment p of both).  perm: int32 [n_prob,8] device
    tensor when labels are used.  Output rows are problem-major."""
    qf = _dev(qf, torch.float32, "query features").contiguous()
    tf = _dev(tf, torch.float32, "target features").contiguous()
    if qseg is None:
        qseg = list(range(len(qoff) - 1))
        tseg = list(range(len(toff) - 1))
    n_prob = len(qseg)
    total = sum(int(qoff[s + 1]) - int(qoff[s]) for s in qseg)
    idx = torch.empty((total, k), dtype=torch.int32, device=qf.device)
    dist = torch.empty((total, k), dtype=torch.float64, device=qf.device) if return_distance else None
    check(_lib.load().cs_knn_feat(ptr(qf), i64_array(qoff), ptr(tf), i64_array(toff), i32_array(qseg),
                                  i32_array(tseg), n_prob, qf.shape[1], k, ptr(qlabel), ptr(tlabel),
                                  ptr(perm), ptr(idx), ptr(dist), stream_ptr()))
    return (idx, dist) if return_distance else idx


def chamfer_1dir(src, soff, tgt, toff, src_seg, tgt_seg, T):
    """Batched one-directional Chamfer; T f32 [n_prob,4,4] device.  Returns f64 [n_prob]."""
    src = _dev(src, torch.float32, "source").contiguous()
    tgt = _dev(tgt, torch.float32, "target").contiguous()
    T = _dev(T, torch.float32, "transforms").contiguous()
    n_prob = len(src_seg)
    out = torch.empty(n_prob, dtype=torch.float64, device=src.device)
    check(_lib.load().cs_chamfer_1dir(ptr(src), i64_array(soff), ptr(tgt), i64_array(toff),
                                      i32_array(src_seg), i32_array(tgt_seg), n_prob, ptr(T), ptr(out),
                                      stream_ptr()))
    return out


def hausdorff_1dir(src, soff, tgt, toff, src_seg, tgt_seg, T):
    """Batched directed Hausdorff distance (max over source points of the nearest-target distance)."""
    src = _dev(src, torch.float32, "source").contiguous()
    tgt = _dev(tgt, torch.float32, "target").contiguous()
    T = _dev(T, torch.float32, "transforms").contiguous()
    n_prob = len(src_seg)
    out = torch.empty(n_prob, dtype=torch.float64, device=src.device)
    check(_lib.load().cs_hausdorff_1dir(ptr(src), i64_array(soff), ptr(tgt), i64_array(toff),
                                        i32_array(src_seg), i32_array(tgt_seg), n_prob, ptr(T), ptr(out),
                                        stream_ptr()))
    return out


def ransac_batch(src, tgt, offsets, max_corr, ransac_n=10, max_iter=100000, confidence=0.999, seed=0):
    """Batched correspondence RANSAC.  Returns (T f32 [n,4,4], inliers int32, rmse f64, iters int32)."""
    src = _dev(src, torch.float32, "source correspondences").contiguous()
    tgt = _dev(tgt, torch.float32, "target correspondences").contiguous()
    n_prob = len(offsets) - 1
    dev = src.device
    T = torch.empty((n_prob, 4, 4), dtype=torch.float32, device=dev)
    inl = torch.empty(n_prob, dtype=torch.int32, device=dev)
    rmse = torch.empty(n_prob, dtype=torch.float64, device=dev)
    iters = torch.empty(n_prob, dtype=torch.int32, device=dev)
    check(_lib.load().cs_ransac_batch(ptr(src), ptr(tgt), i64_array(offsets), n_prob, float(max_corr),
                                      ransac_n, max_iter, float(confidence), int(seed), ptr(T),
                                      ptr(inl), ptr(rmse), ptr(iters), stream_ptr()))
    return T, inl, rmse, iters


def partition_by_label(label, d_off, n_cloud):
    """Rows of every cloud stably partitioned by part label (split_corr's part order).  d_off: int64 device
    tensor [n_cloud + 1].  Returns int64 [N] global row indices."""
    label = _dev(label, torch.int32, "labels").contiguous()
    order = torch.empty(label.shape[0], dtype=torch.int64, device=label.device)
    check(_lib.load().cs_partition_by_label(ptr(label), ptr(d_off), n_cloud, ptr(order), stream_ptr()))
    return order


def cfg_bad(nn, d_first, n_cfg):
    """int32 [n_cfg]: 1 where rows [first[j], first[j+1]) of the neighbour lists hold a negative entry."""
    nn = _dev(nn, torch.int32, "neighbour lists").contiguous()
    bad = torch.empty(n_cfg, dtype=torch.int32, device=nn.device)
    check(_lib.load().cs_cfg_bad(ptr(nn), nn.shape[1], ptr(d_first), n_cfg, ptr(bad), stream_ptr()))
    return bad


def corr_assemble(xyz0, xyz1, rows, nn, desc, total, max_len):
    """Correspondence lists of the configurations desc (host int64 [n_cfg, 5] = q_first, n_first, t_first, len,
    out_first): returns (src f32 [total * k, 3], tgt f32 [total * k, 3])."""
    xyz0 = _dev(xyz0, torch.float32, "query xyz").contiguous()
    xyz1 = _dev(xyz1, torch.float32, "CAD xyz").contiguous()
    nn = _dev(nn, torch.int32, "neighbour lists").contiguous()
    k = nn.shape[1]
    dev = xyz0.device
    src = torch.empty((total * k, 3), dtype=torch.float32, device=dev)
    tgt = torch.empty((total * k, 3), dtype=torch.float32, device=dev)
    d_desc = torch.from_numpy(np.ascontiguousarray(desc, dtype=np.int64)).to(dev, non_blocking=True)
    check(_lib.load().cs_corr_assemble(ptr(xyz0), ptr(xyz1), ptr(rows) if rows is not None else None, ptr(nn), k,
                                       ptr(d_desc), len(desc), int(max_len), ptr(src), ptr(tgt), stream_ptr()))
    return src, tgt


def symcut_fit(feat, xyz, offsets, anchors, Ks, n_nn=50, n_init=10, max_iter=300):
    """anchors int32 [n_cloud, n_anchor] device; Ks host list.  Returns centers f64 [c,a,4,3],
    counts int32 [c,a,4], min centre distance f64 [c,a], max error f64 [c,a]."""
    feat = _dev(feat, torch.float32, "features").contiguous()
    xyz = _dev(xyz, torch.float32, "xyz").contiguous()
    anchors = _dev(anchors, torch.int32, "anchors").contiguous()
    nc, na = anchors.shape
    dev = feat.device
    centers = torch.empty((nc, na, 4, 3), dtype=torch.float64, device=dev)
    counts = torch.empty((nc, na, 4), dtype=torch.int32, device=dev)
    mcd = torch.empty((nc, na), dtype=torch.float64, device=dev)
    mer = torch.empty((nc, na), dtype=torch.float64, device=dev)
    check(_lib.load().cs_symcut_fit(ptr(feat), feat.shape[1], ptr(xyz), i64_array(offsets), nc,
                                    ptr(anchors), na, i32_array(Ks), n_nn, n_init, max_iter,
                                    ptr(centers), ptr(counts), ptr(mcd), ptr(mer), stream_ptr()))
    return centers, counts, mcd, mer


def symcut_labels(xyz, offsets, Ks, sel_centers):
    xyz = _dev(xyz, torch.float32, "xyz").contiguous()
    sel_centers = _dev(sel_centers, torch.float64, "centres").contiguous()
    labels = torch.empty(xyz.shape[0], dtype=torch.int32, device=xyz.device)
    check(_lib.load().cs_symcut_labels(ptr(xyz), i64_array(offsets), len(Ks), i32_array(Ks),
                                       ptr(sel_centers), ptr(labels), stream_ptr()))
    return labels
