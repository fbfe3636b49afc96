"""CPU restatement of the sparse ResUNetBN2C forward (and of the instance-norm block variants,
model/resunet.py:311-333, chosen by the state dict) + global embedding head.
TEST INFRASTRUCTURE ONLY.

Follows the graph of the reference's model code, read as text:
  ResUNet2.forward            model/resunet.py:207-280  (layer construction 49-193)
  BasicBlockBN.forward        model/residual_block.py:60-73
  get_norm("BN")              model/common.py:20-26  (MinkowskiBatchNorm == BatchNorm1d over rows, eval)
  conv1_max_embedding.forward model/fc.py:114-128 (+ split_batch 23-29, conv1_chamfer 60-75)
  row normalisation           model/resunet.py:260-262 (no eps) and evaluation.py:231 (eps 1e-12)
Parameter names are the reference's state-dict names, so a real checkpoint
dict can be fed unchanged.  Sparse-conv semantics (kernel offset order, strided / transposed maps)
are those of oracle/sparse.py.
"""
from __future__ import annotations

import numpy as np

from . import native, sparse

CHANNELS = [None, 32, 64, 128, 256]
TR_CHANNELS = [None, 64, 64, 64, 128]
BN_EPS = 1e-5


def fold_bn(w, prefix, eps=BN_EPS):
    """Eval-mode BatchNorm1d as y = x*scale + shift (f32):  scale = g / sqrt(var + eps),
    shift = b - mean * scale."""
    g = np.asarray(w[prefix + ".bn.weight"], dtype=np.float32)
    b = np.asarray(w[prefix + ".bn.bias"], dtype=np.float32)
    m = np.asarray(w[prefix + ".bn.running_mean"], dtype=np.float32)
    v = np.asarray(w[prefix + ".bn.running_var"], dtype=np.float32)
    scale = (g / np.sqrt(v + np.float32(eps))).astype(np.float32)
    shift = (b - m * scale).astype(np.float32)
    return scale, shift


def build_maps(coords):
    """The 4 coordinate maps and 10 neighbour tables one batch needs (7 distinct kernel maps; the 3
    transposed tables are the strided maps with roles swapped)."""
    c1 = np.asarray(coords, dtype=np.int32)
    c2, _ = sparse.coordmap_stride(c1, 1)
    c4, _ = sparse.coordmap_stride(c2, 2)
    c8, _ = sparse.coordmap_stride(c4, 4)
    km = {
        "s1": sparse.kernel_map(c1, 1, c1, 1),
        "s1_s2": sparse.kernel_map(c1, 1, c2, 2),
        "s2": sparse.kernel_map(c2, 2, c2, 2),
        "s2_s4": sparse.kernel_map(c2, 2, c4, 4),
        "s4": sparse.kernel_map(c4, 4, c4, 4),
        "s4_s8": sparse.kernel_map(c4, 4, c8, 8),
        "s8": sparse.kernel_map(c8, 8, c8, 8),
        "s8_s4_T": sparse.kernel_map(c8, 8, c4, 4, transposed=True),
        "s4_s2_T": sparse.kernel_map(c4, 4, c2, 2, transposed=True),
        "s2_s1_T": sparse.kernel_map(c2, 2, c1, 1, transposed=True),
    }
    return {"c1": c1, "c2": c2, "c4": c4, "c8": c8}, km


def _seg_of(coords):
    """Row offsets of the samples of a coordinate map (rows grouped by batch index)."""
    b = np.asarray(coords)[:, 0]
    n_batch = int(b[-1]) + 1 if len(b) else 0
    return np.searchsorted(b, np.arange(n_batch + 1))


def _block_in(w, prefix, x, nbr, seg):
    """BasicBlockIN (model/residual_block.py:60-73 with NORM_TYPE "IN"): conv, instance norm, ReLU, conv,
    instance norm, + input, ReLU -- op by op, as the MinkowskiEngine modules run it."""
    y = native.conv_fwd(nbr, x, w[prefix + ".conv1.kernel"], None, None, None, False)
    y = sparse.instance_norm(y, seg, w[prefix + ".norm1.weight"], w[prefix + ".norm1.bias"])
    y = np.maximum(y, np.float32(0))
    y = native.conv_fwd(nbr, y, w[prefix + ".conv2.kernel"], None, None, None, False)
    y = sparse.instance_norm(y, seg, w[prefix + ".norm2.weight"], w[prefix + ".norm2.bias"])
    return np.maximum((y + np.asarray(x, np.float32)).astype(np.float32), np.float32(0))


def _block(w, prefix, x, nbr, seg=None):
    if prefix + ".norm1.weight" in w:       # instance-norm block (the IN network variants)
        return _block_in(w, prefix, x, nbr, seg)
    s1, b1 = fold_bn(w, prefix + ".norm1")
    s2, b2 = fold_bn(w, prefix + ".norm2")
    y = native.conv_fwd(nbr, x, w[prefix + ".conv1.kernel"], s1, b1, None, True)
    return native.conv_fwd(nbr, y, w[prefix + ".conv2.kernel"], s2, b2, x, True)


def resunet_forward(w, coords, feats, normalize_feature=True):
    """Returns (out [N1,16] row-normalised, feat [N8,256], maps dict)."""
    maps, km = build_maps(coords)
    x = np.asarray(feats, dtype=np.float32)

    def conv_bn(name, norm, x, nbr):
        s, b = fold_bn(w, norm)
        return native.conv_fwd(nbr, x, w[name + ".kernel"], s, b, None, False)

    g1, g2, g4, g8 = (_seg_of(maps[k]) for k in ("c1", "c2", "c4", "c8"))
    out_s1 = _block(w, "block1", conv_bn("conv1", "norm1", x, km["s1"]), km["s1"], g1)
    out_s2 = _block(w, "block2", conv_bn("conv2", "norm2", out_s1, km["s1_s2"]), km["s2"], g2)
    out_s4 = _block(w, "block3", conv_bn("conv3", "norm3", out_s2, km["s2_s4"]), km["s4"], g4)
    out_s8 = _block(w, "block4", conv_bn("conv4", "norm4", out_s4, km["s4_s8"]), km["s8"], g8)
    feat = out_s8  # block output is already >= 0, MEF.relu at resunet.py:227 is idempotent

    out = _block(w, "block4_tr", conv_bn("conv4_tr", "norm4_tr", out_s8, km["s8_s4_T"]), km["s4"], g4)
    out = np.concatenate([out, out_s4], 1)
    out = _block(w, "block3_tr", conv_bn("conv3_tr", "norm3_tr", out, km["s4_s2_T"]), km["s2"], g2)
    out = np.concatenate([out, out_s2], 1)
    out = _block(w, "block2_tr", conv_bn("conv2_tr", "norm2_tr", out, km["s2_s1_T"]), km["s1"], g1)
    out = np.concatenate([out, out_s1], 1)
    out = native.conv_fwd(None, out, w["conv1_tr.kernel"], None, None, None, True)
    out = native.conv_fwd(None, out, w["final.kernel"], None,
                          np.asarray(w["final.bias"], np.float32).reshape(-1), None, False)
    if normalize_feature:
        out = native.row_l2_normalize(out, 0.0)
    return out, feat, maps


def embedding_forward(ew, feat, batch_index, n_batch, normalize=True):
    """conv1_max_embedding (model/fc.py:114-128) + F.normalize (evaluation.py:231).
    The two Linear layers run through the same fma-chain matmul as the 1x1 convs
    (weight [out,in] transposed to [in,out]); BatchNorm1d(eval) and the Linear bias fold into the
    epilogue:  bn(xW + b) = (xW) * s + (b * s + t)."""
    y = native.conv_fwd(None, feat, ew["final.final.kernel"], None,
                        np.asarray(ew["final.final.bias"], np.float32).reshape(-1), None, False)
    pooled = sparse.segmented_max(y, np.asarray(batch_index), n_batch)
    g = np.asarray(ew["bn1.weight"], np.float32)
    b = np.asarray(ew["bn1.bias"], np.float32)
    m = np.asarray(ew["bn1.running_mean"], np.float32)
    v = np.asarray(ew["bn1.running_var"], np.float32)
    scale = (g / np.sqrt(v + np.float32(BN_EPS))).astype(np.float32)
    shift = (b - m * scale).astype(np.float32)
    shift = (np.asarray(ew["fc1.bias"], np.float32) * scale + shift).astype(np.float32)
    w1 = np.ascontiguousarray(np.asarray(ew["fc1.weight"], np.float32).T)
    w2 = np.ascontiguousarray(np.asarray(ew["fc2.weight"], np.float32).T)
    h = native.conv_fwd(None, pooled, w1, scale, shift, None, True)
    out = native.conv_fwd(None, h, w2, None, np.asarray(ew["fc2.bias"], np.float32), None, False)
    if normalize:
        out = native.row_l2_normalize(out, 1e-12)
    return out
