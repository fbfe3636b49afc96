"""Fused ResUNetBN2C + global-embedding forward on libcorsair_hip.so.

This is the production feature extractor: the same graph as the reference's
``ResUNet2.forward`` (model/resunet.py:207-280), ``BasicBlockBN.forward``
(model/residual_block.py:60-73) and ``conv1_max_embedding.forward`` (model/fc.py:114-128), but
  * eval-mode batch-norm, bias, residual add and ReLU are folded into the epilogue of the
    producing convolution (one kernel per conv, 26 launches per forward),
  * ``ME.cat`` is free: decoder convs write straight into the left columns of the concat buffer
    whose right columns were written by the encoder,
  * the 7 distinct kernel maps of a batch are built once and shared (the 3 transposed convs use
    the swapped strided maps),
  * the two Linear layers of the embedding head run through the same MFMA kernel (a 1x1 conv is a
    plain matmul), so the whole forward is hand-written HIP and bit-comparable with the oracle.
It consumes a state dict with the reference's parameter names, i.e. the
``state_dict`` / ``embedding_state_dict`` entries of a reference checkpoint (utils/ckpts.py:44-61).
"""
from __future__ import annotations

import numpy as np
import torch

from . import backend as B

CHANNELS = [None, 32, 64, 128, 256]
TR_CHANNELS = [None, 64, 64, 64, 128]
BN_EPS = 1e-5


def _np32(t):
    if isinstance(t, torch.Tensor):
        t = t.detach().cpu().numpy()
    return np.asarray(t, dtype=np.float32)


def fold_bn(sd, prefix, eps=BN_EPS):
    """BatchNorm1d(eval) -> (scale, shift) in f32 on the host: scale = g / sqrt(var + eps),
    shift = b - mean * scale (model/common.py:22 wraps nn.BatchNorm1d as `.bn`)."""
    g = _np32(sd[prefix + ".bn.weight"])
    b = _np32(sd[prefix + ".bn.bias"])
    m = _np32(sd[prefix + ".bn.running_mean"])
    v = _np32(sd[prefix + ".bn.running_var"])
    scale = (g / np.sqrt(v + np.float32(eps))).astype(np.float32)
    shift = (b - m * scale).astype(np.float32)
    return scale, shift


class BatchMaps:
    """Coordinate maps (tensor strides 1, 2, 4, 8) and the kernel maps of one input batch."""

    def __init__(self, coords, n_batch=0):
        # all four coordinate levels in one library call (one host wait); n_batch = the collated batch size when the
        # caller knows it (rows grouped by sample): the per-sample segments then come out of the same pass
        self.c1, self.c2, self.c4, self.c8 = B.CoordMap.pyramid(coords, 4, n_batch)
        c1, c2, c4, c8 = self.c1, self.c2, self.c4, self.c8
        names = ("s1", "s1_s2", "s2", "s2_s4", "s4", "s4_s8", "s8", "s8_s4_T", "s4_s2_T", "s2_s1_T")
        specs = [(c1, c1), (c1, c2), (c2, c2), (c2, c4), (c4, c4), (c4, c8), (c8, c8),
                 (c8, c4, 3, True), (c4, c2, 3, True), (c2, c1, 3, True)]
        # one call: the ten maps are independent chains of small launches, built on three streams inside the library
        for name, km in zip(names, B.KernelMap.build_many(specs)):
            setattr(self, name, km)

    def total_pairs(self):
        return {n: getattr(self, n).num_pairs for n in
                ("s1", "s1_s2", "s2", "s2_s4", "s4", "s4_s8", "s8", "s8_s4_T", "s4_s2_T", "s2_s1_T")}


class ResUNetEngine:
    """ResUNetBN2C(in=1, out=16, normalize_feature=True, conv1_kernel_size=3, D=3) + embedding."""

    def __init__(self, state_dict, embedding_state_dict=None, device="cuda"):
        self.device = torch.device(device)
        dev = self.device

        def up(a):
            return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)

        self.w = {}
        self.bn = {}
        for name in ["conv1", "conv2", "conv3", "conv4", "conv4_tr", "conv3_tr", "conv2_tr",
                     "conv1_tr", "final"]:
            self.w[name] = up(_np32(state_dict[name + ".kernel"]))
        for blk in ["block1", "block2", "block3", "block4", "block4_tr", "block3_tr", "block2_tr"]:
            for c in ("conv1", "conv2"):
                self.w[f"{blk}.{c}"] = up(_np32(state_dict[f"{blk}.{c}.kernel"]))
            for nrm in ("norm1", "norm2"):
                s, b = fold_bn(state_dict, f"{blk}.{nrm}")
                self.bn[f"{blk}.{nrm}"] = (up(s), up(b))
        for nrm in ["norm1", "norm2", "norm3", "norm4", "norm4_tr", "norm3_tr", "norm2_tr"]:
            s, b = fold_bn(state_dict, nrm)
            self.bn[nrm] = (up(s), up(b))
        self.final_bias = up(_np32(state_dict["final.bias"]).reshape(-1))
        self.emb = None
        if embedding_state_dict is not None:
            e = embedding_state_dict
            g, b = _np32(e["bn1.weight"]), _np32(e["bn1.bias"])
            m, v = _np32(e["bn1.running_mean"]), _np32(e["bn1.running_var"])
            scale = (g / np.sqrt(v + np.float32(BN_EPS))).astype(np.float32)
            shift = (b - m * scale).astype(np.float32)
            shift = (_np32(e["fc1.bias"]) * scale + shift).astype(np.float32)
            self.emb = {
                "conv": up(_np32(e["final.final.kernel"])),
                "conv_bias": up(_np32(e["final.final.bias"]).reshape(-1)),
                "w1": up(np.ascontiguousarray(_np32(e["fc1.weight"]).T)),
                "s1": up(scale),
                "b1": up(shift),
                "w2": up(np.ascontiguousarray(_np32(e["fc2.weight"]).T)),
                "b2": up(_np32(e["fc2.bias"])),
            }

    # -- building blocks -------------------------------------------------------------------
    def _block(self, name, x, km, out=None):
        s1, b1 = self.bn[name + ".norm1"]
        s2, b2 = self.bn[name + ".norm2"]
        y = B.conv_fwd(km, x, self.w[name + ".conv1"], s1, b1, None, True)
        return B.conv_fwd(km, y, self.w[name + ".conv2"], s2, b2, x, True, out=out)

    def _conv_bn(self, conv, norm, x, km):
        s, b = self.bn[norm]
        return B.conv_fwd(km, x, self.w[conv], s, b, None, False)

    def forward(self, coords, feats, maps=None, n_batch=0):
        """coords int32 [N,4] (batch,x,y,z) unique; feats f32 [N,1]; n_batch: number of samples of a collated batch (rows
        grouped by sample) when known.  Returns (out [N,16] unit rows, feat [N8,256], maps)."""
        if maps is None:
            maps = BatchMaps(coords, n_batch)
        m = maps
        dev = feats.device
        n1, n2, n4 = m.c1.n, m.c2.n, m.c4.n
        C, T = CHANNELS, TR_CHANNELS
        # concat buffers: [decoder | encoder skip]
        cat1 = torch.empty((n1, T[2] + C[1]), dtype=torch.float32, device=dev)
        cat2 = torch.empty((n2, T[3] + C[2]), dtype=torch.float32, device=dev)
        cat4 = torch.empty((n4, T[4] + C[3]), dtype=torch.float32, device=dev)

        x = self._conv_bn("conv1", "norm1", feats, m.s1)
        out_s1 = self._block("block1", x, m.s1, out=cat1[:, T[2]:])
        x = self._conv_bn("conv2", "norm2", out_s1, m.s1_s2)
        out_s2 = self._block("block2", x, m.s2, out=cat2[:, T[3]:])
        x = self._conv_bn("conv3", "norm3", out_s2, m.s2_s4)
        out_s4 = self._block("block3", x, m.s4, out=cat4[:, T[4]:])
        x = self._conv_bn("conv4", "norm4", out_s4, m.s4_s8)
        out_s8 = self._block("block4", x, m.s8)

        x = self._conv_bn("conv4_tr", "norm4_tr", out_s8, m.s8_s4_T)
        self._block("block4_tr", x, m.s4, out=cat4[:, :T[4]])
        x = self._conv_bn("conv3_tr", "norm3_tr", cat4, m.s4_s2_T)
        self._block("block3_tr", x, m.s2, out=cat2[:, :T[3]])
        x = self._conv_bn("conv2_tr", "norm2_tr", cat2, m.s2_s1_T)
        self._block("block2_tr", x, m.s1, out=cat1[:, :T[2]])
        x = B.conv_fwd(None, cat1, self.w["conv1_tr"], None, None, None, True)
        x = B.conv_fwd(None, x, self.w["final"], None, self.final_bias, None, False)
        out = B.row_l2_normalize(x, 0.0)
        return out, out_s8, maps

    def embed(self, feat, maps, n_batch, normalize=True):
        """conv1_max_embedding + F.normalize: feat [N8,256] -> [n_batch,256]."""
        e = self.emb
        y = B.conv_fwd(None, feat, e["conv"], None, e["conv_bias"], None, False)
        pooled = B.segmented_max(y, maps.c8.coords, n_batch)
        h = B.conv_fwd(None, pooled, e["w1"], e["s1"], e["b1"], None, True)
        g = B.conv_fwd(None, h, e["w2"], None, e["b2"], None, False)
        if normalize:
            g = B.row_l2_normalize(g, 1e-12)
        return g
