"""MinkowskiEngine-compatible operator surface on libcorsair_hip.so (inference only).

The reference's model code (model/resunet.py, model/residual_block.py, model/common.py, model/fc.py)
and evaluation.py talk to MinkowskiEngine 0.5.5 through a small Python API.  This module
implements exactly that subset, so those files run unmodified when ``shim/`` is on PYTHONPATH
(``import MinkowskiEngine as ME`` resolves to ``shim/MinkowskiEngine`` which re-exports this module).

  ME.SparseTensor(feat, coords)              evaluation.py:215-218          -> cs_coordmap_create
  ME.MinkowskiConvolution / ...Transpose     model/resunet.py:49-193        -> cs_kernelmap_build, cs_conv_fwd
  ME.MinkowskiBatchNorm (eval)               model/common.py:22             -> cs_affine_act
  MEF.relu, SparseTensor.__iadd__, ME.cat    model/resunet.py:212-255, residual_block.py:70
  ME.utils.sparse_quantize / sparse_collate  utils/Info/CADLib.py:106-121,166-168 (CPU, DataLoader workers)

Everything that touches features runs a HIP kernel; there is no CPU fallback and no training path
(backward is not implemented: the reference's evaluated path is inference under torch.no_grad()).
The fused production path is corsair_amd.engine.ResUNetEngine; this op-by-op surface is the drop-in.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from .. import backend as B

__version__ = "0.5.5"  # model/resunet.py:55 compares ME.__version__ >= "0.5.4" as strings


# ---- coordinate manager ---------------------------------------------------------------------------
class CoordinateMapKey:
    def __init__(self, tensor_stride, uid=0):
        self.tensor_stride = int(tensor_stride)
        self.uid = uid

    def get_tensor_stride(self):
        return [self.tensor_stride] * 3

    def __eq__(self, other):
        return isinstance(other, CoordinateMapKey) and (self.tensor_stride, self.uid) == (other.tensor_stride, other.uid)

    def __hash__(self):
        return hash((self.tensor_stride, self.uid))

    def __repr__(self):
        return f"CoordinateMapKey(tensor_stride={self.tensor_stride})"


class CoordinateManager:
    """Owns the coordinate maps of one input batch and caches kernel maps, like ME's manager
    (maps keyed by tensor stride; kernel maps by (in key, out key, kernel size, transposed))."""

    def __init__(self, D=3):
        self.D = D
        self._maps = {}
        self._kmaps = {}

    def insert(self, coords, tensor_stride=1):
        key = CoordinateMapKey(tensor_stride)
        self._maps[key] = B.CoordMap.create(coords, tensor_stride)
        return key

    def get(self, key):
        return self._maps[key]

    def stride(self, key, stride):
        out_key = CoordinateMapKey(key.tensor_stride * stride)
        if out_key not in self._maps:
            self._maps[out_key] = self._maps[key].stride(stride)
        return out_key

    def kernel_map(self, in_key, out_key, kernel_size, transposed=False):
        k = (in_key, out_key, kernel_size, transposed)
        if k not in self._kmaps:
            self._kmaps[k] = B.KernelMap.build(self._maps[in_key], self._maps[out_key], kernel_size, transposed)
        return self._kmaps[k]

    def get_coordinates(self, key):
        return self._maps[key].coords


class SparseTensor:
    def __init__(self, features, coordinates=None, *, coordinate_map_key=None, coordinate_manager=None,
                 tensor_stride=1, device=None, **unused):
        if not isinstance(features, torch.Tensor):
            raise TypeError("features must be a torch.Tensor")
        if device is not None:
            features = features.to(device)
        if coordinates is not None:
            if coordinate_manager is None:
                coordinate_manager = CoordinateManager()
            coordinates = coordinates.to(device=features.device, dtype=torch.int32)
            if coordinates.shape[0] != features.shape[0]:
                raise RuntimeError("SparseTensor: features and coordinates have different row counts")
            coordinate_map_key = coordinate_manager.insert(coordinates, tensor_stride)
        elif coordinate_map_key is None or coordinate_manager is None:
            raise ValueError("SparseTensor needs coordinates or (coordinate_map_key, coordinate_manager)")
        n = coordinate_manager.get(coordinate_map_key).n
        if features.shape[0] != n:
            raise RuntimeError(f"SparseTensor: {features.shape[0]} feature rows for {n} coordinates")
        self._F = features.float()
        self.coordinate_map_key = coordinate_map_key
        self.coordinate_manager = coordinate_manager

    # attribute names of ME 0.5.x used by the reference
    @property
    def F(self):
        return self._F

    @property
    def C(self):
        return self.coordinate_manager.get_coordinates(self.coordinate_map_key)

    @property
    def feats(self):
        return self._F

    @property
    def coords(self):
        return self.C

    @property
    def tensor_stride(self):
        return self.coordinate_map_key.get_tensor_stride()

    @property
    def device(self):
        return self._F.device

    @property
    def shape(self):
        return self._F.shape

    def _same(self, other):
        if self.coordinate_map_key != other.coordinate_map_key or self.coordinate_manager is not other.coordinate_manager:
            raise RuntimeError("SparseTensor: operands live on different coordinate maps")

    def __iadd__(self, other):
        """out += residual (model/residual_block.py:70)."""
        self._same(other)
        B.affine_act(self._F, None, None, other.F, False, out=self._F)
        return self

    def __add__(self, other):
        self._same(other)
        return SparseTensor(B.affine_act(self._F, None, None, other.F, False),
                            coordinate_map_key=self.coordinate_map_key,
                            coordinate_manager=self.coordinate_manager)

    def to(self, *a, **k):
        return self  # already on the device; `.to("cuda")` in model/fc.py:93 is a no-op here

    def __repr__(self):
        return f"SparseTensor(F={tuple(self._F.shape)}, {self.coordinate_map_key})"


def cat(*tensors):
    """Channel concatenation of tensors sharing a coordinate map (model/resunet.py:239,246,253)."""
    for t in tensors[1:]:
        tensors[0]._same(t)
    return SparseTensor(torch.cat([t.F for t in tensors], dim=1),
                        coordinate_map_key=tensors[0].coordinate_map_key,
                        coordinate_manager=tensors[0].coordinate_manager)


# ---- modules ------------------------------------------------------------------------------------------
class MinkowskiNetwork(nn.Module):
    def __init__(self, D):
        super().__init__()
        self.D = D


class _ConvBase(nn.Module):
    transposed = False

    def __init__(self, in_channels, out_channels, kernel_size=-1, stride=1, dilation=1, bias=False,
                 kernel_generator=None, expand_coordinates=False, convolution_mode=None, dimension=None,
                 **legacy):
        super().__init__()
        if "has_bias" in legacy:  # ME < 0.5.4 spelling, still produced by the reference for old versions
            bias = legacy.pop("has_bias")
        if dimension not in (None, 3) or dilation != 1 or kernel_size not in (1, 3) or stride not in (1, 2):
            raise NotImplementedError("only 3-D, dilation 1, kernel 1|3, stride 1|2 are on the CORSAIR path")
        if expand_coordinates:
            raise NotImplementedError("expand_coordinates is not used by the reference")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride = kernel_size, stride
        kvol = kernel_size ** 3
        shape = (kvol, in_channels, out_channels) if kvol > 1 else (in_channels, out_channels)
        self.kernel = nn.Parameter(torch.empty(shape))
        self.bias = nn.Parameter(torch.zeros(1, out_channels)) if bias else None
        with torch.no_grad():  # ME's reset_parameters: U(-s, s), s = 1/sqrt(in_channels * kernel_volume)
            s = 1.0 / np.sqrt(in_channels * kvol)
            self.kernel.uniform_(-s, s)
            if self.bias is not None:
                self.bias.uniform_(-s, s)

    def forward(self, x):
        if torch.is_grad_enabled() and (self.kernel.requires_grad and x.F.requires_grad):
            raise NotImplementedError("corsair_amd implements inference only (wrap in torch.no_grad())")
        cm = x.coordinate_manager
        in_key = x.coordinate_map_key
        if self.kernel_size == 1 and self.stride == 1:
            kmap, out_key = None, in_key
        elif not self.transposed:
            out_key = cm.stride(in_key, self.stride) if self.stride > 1 else in_key
            kmap = cm.kernel_map(in_key, out_key, self.kernel_size, False)
        else:
            if in_key.tensor_stride % self.stride:
                raise RuntimeError("transposed convolution below tensor stride 1")
            out_key = CoordinateMapKey(in_key.tensor_stride // self.stride)
            if out_key not in cm._maps:
                raise RuntimeError("transposed convolution needs the finer coordinate map to exist "
                                   "(expand_coordinates=False semantics)")
            kmap = cm.kernel_map(in_key, out_key, self.kernel_size, True)
        bias = self.bias.detach().reshape(-1) if self.bias is not None else None
        out = B.conv_fwd(kmap, x.F, self.kernel.detach(), None, bias, None, False)
        return SparseTensor(out, coordinate_map_key=out_key, coordinate_manager=cm)


class MinkowskiConvolution(_ConvBase):
    transposed = False


class MinkowskiConvolutionTranspose(_ConvBase):
    transposed = True


class MinkowskiBatchNorm(nn.Module):
    """BatchNorm1d over rows (state-dict prefix `.bn.`), eval mode only."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True):
        super().__init__()
        self.bn = nn.BatchNorm1d(num_features, eps=eps, momentum=momentum, affine=affine,
                                 track_running_stats=track_running_stats)
        self._folded = None

    def _fold(self):
        bn = self.bn
        ver = tuple(int(t._version) for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var))
        if self._folded is None or self._folded[0] != ver or self._folded[1].device != bn.weight.device:
            g = bn.weight.detach().cpu().numpy().astype(np.float32)
            b = bn.bias.detach().cpu().numpy().astype(np.float32)
            m = bn.running_mean.detach().cpu().numpy().astype(np.float32)
            v = bn.running_var.detach().cpu().numpy().astype(np.float32)
            scale = (g / np.sqrt(v + np.float32(bn.eps))).astype(np.float32)
            shift = (b - m * scale).astype(np.float32)
            dev = bn.weight.device
            self._folded = (ver, torch.from_numpy(scale).to(dev), torch.from_numpy(shift).to(dev))
        return self._folded[1], self._folded[2]

    def forward(self, x):
        if self.training:
            raise NotImplementedError("MinkowskiBatchNorm: training mode is not implemented (call .eval())")
        scale, shift = self._fold()
        return SparseTensor(B.affine_act(x.F, scale, shift, None, False),
                            coordinate_map_key=x.coordinate_map_key, coordinate_manager=x.coordinate_manager)


class MinkowskiInstanceNorm(nn.Module):
    """Per-sample, per-channel normalisation of the rows (the IN variants of the network build their
    residual blocks with it, model/common.py:23-24, model/resunet.py:311-333).  Parameters `weight` /
    `bias` of shape [1, C] as in MinkowskiEngine; eps 1e-8 inside the root, biased variance."""

    def __init__(self, num_features, dimension=None):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(1, num_features))
        self.bias = nn.Parameter(torch.zeros(1, num_features))

    def forward(self, x):
        batch = x.C[:, 0].contiguous()
        n_batch = int(batch[-1].item()) + 1 if batch.numel() else 0      # rows are grouped by sample
        seg = torch.searchsorted(batch, torch.arange(n_batch + 1, device=batch.device, dtype=batch.dtype))
        out = B.instance_norm(x.F, seg.to(torch.int32), self.weight.detach(), self.bias.detach(), 1e-8)
        return SparseTensor(out, coordinate_map_key=x.coordinate_map_key, coordinate_manager=x.coordinate_manager)


class MinkowskiReLU(nn.Module):
    def __init__(self, inplace=False):
        super().__init__()

    def forward(self, x):
        return MinkowskiFunctional.relu(x)


class _Functional:
    @staticmethod
    def relu(x):
        return SparseTensor(B.affine_act(x.F, None, None, None, True),
                            coordinate_map_key=x.coordinate_map_key, coordinate_manager=x.coordinate_manager)


MinkowskiFunctional = _Functional()


def _not_on_path(name):
    class _Stub(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()
            raise NotImplementedError(f"{name} is not on the CORSAIR inference path (SURVEY 2 #7)")

    _Stub.__name__ = name
    return _Stub


MinkowskiAvgPooling = _not_on_path("MinkowskiAvgPooling")
MinkowskiGlobalMaxPooling = _not_on_path("MinkowskiGlobalMaxPooling")
MinkowskiLinear = _not_on_path("MinkowskiLinear")
MinkowskiSumPooling = _not_on_path("MinkowskiSumPooling")
MinkowskiGlobalPooling = _not_on_path("MinkowskiGlobalPooling")
MinkowskiBroadcastMultiplication = _not_on_path("MinkowskiBroadcastMultiplication")


# ---- ME.utils (CPU, safe in forked DataLoader workers: no GPU context is created) -----------------
class _Utils:
    @staticmethod
    def sparse_quantize(coordinates, features=None, labels=None, ignore_label=-100, return_index=False,
                        return_inverse=False, return_maps_only=False, quantization_size=None, device="cpu"):
        """First point of every voxel, indices ascending (utils/Info/CADLib.py:108-112 calls it with
        return_index=True, return_maps_only=True).  Runs on the host like the reference's DataLoader
        workers do; the GPU voxeliser of the production path is cs_voxelize."""
        c = coordinates.numpy() if isinstance(coordinates, torch.Tensor) else np.asarray(coordinates)
        if quantization_size is not None:
            c = c / quantization_size
        g = np.floor(c).astype(np.int64)
        _, first, inverse = np.unique(g, axis=0, return_index=True, return_inverse=True)
        order = np.sort(first)
        # np.unique numbers the voxels lexicographically; the kept points are in input order: renumber the
        # inverse map so that  g[order][inverse] == g  (voxel of point i = row inverse[i] of the output)
        rank = np.empty(len(first), dtype=np.int64)
        rank[np.argsort(first, kind="stable")] = np.arange(len(first))
        inverse = rank[np.asarray(inverse).reshape(-1)]
        if return_maps_only:
            return (order, inverse) if return_inverse else order
        out = [g[order].astype(np.int32)]
        if features is not None:
            out.append(np.asarray(features)[order])
        if return_index:
            out.append(order)
        if return_inverse:
            out.append(inverse)
        return out[0] if len(out) == 1 else tuple(out)

    @staticmethod
    def batched_coordinates(coords, dtype=torch.int32, device=None):
        rows = []
        for b, c in enumerate(coords):
            c = torch.as_tensor(np.asarray(c)) if not isinstance(c, torch.Tensor) else c
            c = torch.floor(c.double()).to(dtype)
            rows.append(torch.cat([torch.full((c.shape[0], 1), b, dtype=dtype), c], 1))
        return torch.cat(rows, 0)

    @staticmethod
    def sparse_collate(coords, feats, labels=None, dtype=torch.int32, device=None):
        """Prepend the batch index, concatenate (utils/Info/CADLib.py:166-168)."""
        bc = _Utils.batched_coordinates(coords, dtype)
        f = torch.cat([torch.as_tensor(np.asarray(x)) if not isinstance(x, torch.Tensor) else x for x in feats], 0)
        if labels is not None:
            return bc, f, torch.cat([torch.as_tensor(l) for l in labels], 0)
        return bc, f

    @staticmethod
    def kaiming_normal_(tensor, mode="fan_out", nonlinearity="relu"):
        raise NotImplementedError("training-time initialisation is out of scope (SURVEY 2 #7)")


utils = _Utils()
