"""The MinkowskiEngine-compatible operator surface (drop-in boundary) on the GPU:
reference-style model code built from ME modules == fused engine == CPU oracle."""
import os
import sys

import numpy as np
import pytest
import torch

from tests.helpers import make_batch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _torch_sd(sd):
    return {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}


def test_model_code_on_shim_matches_engine_and_oracle(gpu, oracle_native):
    sys.path.insert(0, os.path.join(ROOT, "shim"))
    import MinkowskiEngine as ME  # noqa: the shim package

    from corsair_amd import engine, synth
    from corsair_amd.model import load_model
    from corsair_amd.model import fc
    from oracle import resunet as oref

    assert ME.__version__ >= "0.5.4"  # string compare at model/resunet.py:55
    coords, feats, _, offsets = make_batch([11, 12], n_points=4000)
    sd, emb = synth.make_state_dicts(31)

    Model = load_model("ResUNetBN2C")
    model = Model(1, 16, bn_momentum=0.05, normalize_feature=True, conv1_kernel_size=3, D=3).to(gpu)
    head = fc.conv1_max_embedding(1024, 512, 256).to(gpu)
    # reference checkpoints are loaded exactly like this (evaluation.py:195-201)
    model.load_state_dict(_torch_sd(sd))
    head.load_state_dict(_torch_sd(emb))
    model.eval()
    head.eval()
    with torch.no_grad():
        x = ME.SparseTensor(torch.from_numpy(feats).to(gpu), torch.from_numpy(coords).to(gpu))
        out, feat = model(x)
        g = torch.nn.functional.normalize(head(feat), dim=1)
    # row order of the input is preserved (evaluation.py:227-229 masks origins with out.C[:,0] == i)
    assert np.array_equal(out.C.cpu().numpy(), coords)

    eng = engine.ResUNetEngine(sd, emb, device=gpu)
    e_out, e_feat, maps = eng.forward(torch.from_numpy(coords).to(gpu), torch.from_numpy(feats).to(gpu))
    e_g = eng.embed(e_feat, maps, 2)
    assert torch.equal(out.F, e_out) and torch.equal(feat.F, e_feat)   # op-by-op == fused
    assert torch.allclose(g, e_g, atol=2e-6)                           # dense head: torch Linear vs MFMA
    want_out, want_feat, _ = oref.resunet_forward(sd, coords, feats)
    assert np.array_equal(out.F.cpu().numpy(), want_out)
    assert np.array_equal(feat.F.cpu().numpy(), want_feat)
    # state-dict names are the reference's: 129 tensors + embedding
    names = set(model.state_dict().keys())
    assert names == set(sd.keys()) and len(names) == 129
    assert set(head.state_dict().keys()) == set(emb.keys())


@pytest.mark.parametrize("c,ld_pad", [(32, 0), (300, 4), (1, 0)])
def test_instance_norm_bit_exact(gpu, c, ld_pad):
    """cs_instance_norm against the oracle: ragged samples around the 256-row chunk size, an empty sample,
    more than 256 channels, a strided input view."""
    from corsair_amd import backend as B
    from oracle import sparse

    rng = np.random.default_rng(c)
    seg = [0, 255, 511, 511, 1024, 1025, 1900]
    full = (rng.normal(size=(seg[-1], c + ld_pad)) * 3 + 1).astype(np.float32)
    x = torch.from_numpy(full).to(gpu)[:, :c]                       # ld = c + ld_pad
    w = rng.normal(size=(1, c)).astype(np.float32)
    b = rng.normal(size=(1, c)).astype(np.float32)
    seg_t = torch.tensor(seg, dtype=torch.int32, device=gpu)
    got = B.instance_norm(x, seg_t, torch.from_numpy(w).to(gpu), torch.from_numpy(b).to(gpu)).cpu().numpy()
    assert np.array_equal(got, sparse.instance_norm(full[:, :c], seg, w, b))
    got = B.instance_norm(x, seg_t).cpu().numpy()
    assert np.array_equal(got, sparse.instance_norm(full[:, :c], seg))


def test_instance_norm_network_variant_matches_oracle(gpu, oracle_native):
    """ResUNetIN2C (instance-norm residual blocks, model/resunet.py:323-325) built from the ME-compatible
    modules == the oracle's op-by-op restatement, bit for bit."""
    sys.path.insert(0, os.path.join(ROOT, "shim"))
    import MinkowskiEngine as ME  # noqa: the shim package

    from corsair_amd.model import load_model
    from oracle import resunet as oref

    coords, feats, _, _ = make_batch([11, 12, 13], n_points=3000)
    torch.manual_seed(5)
    model = load_model("ResUNetIN2C")(1, 16, bn_momentum=0.05, normalize_feature=True, conv1_kernel_size=3, D=3)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if ".norm" in name and name.startswith("block"):        # affine part of the instance norms
                p.copy_(torch.randn_like(p) * 0.3 + (1.0 if name.endswith("weight") else 0.0))
    model = model.to(gpu).eval()
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    assert "block1.norm1.weight" in sd and "block1.norm1.bn.weight" not in sd and "norm1.bn.weight" in sd
    with torch.no_grad():
        out, feat = model(ME.SparseTensor(torch.from_numpy(feats).to(gpu), torch.from_numpy(coords).to(gpu)))
    want_out, want_feat, _ = oref.resunet_forward(sd, coords, feats)
    assert np.array_equal(feat.F.cpu().numpy(), want_feat)
    assert np.array_equal(out.F.cpu().numpy(), want_out)


def test_shim_sparse_tensor_semantics(gpu):
    sys.path.insert(0, os.path.join(ROOT, "shim"))
    import MinkowskiEngine as ME
    import MinkowskiEngine.MinkowskiFunctional as MEF

    coords, feats, _, _ = make_batch([13], n_points=1500)
    c = torch.from_numpy(coords).to(gpu)
    x = ME.SparseTensor(torch.randn(len(coords), 8, device=gpu), c)
    y = ME.SparseTensor(torch.randn(len(coords), 8, device=gpu), coordinate_map_key=x.coordinate_map_key,
                        coordinate_manager=x.coordinate_manager)
    z = ME.cat(x, y)
    assert z.F.shape[1] == 16 and torch.equal(z.F[:, :8], x.F)
    want = x.F + y.F
    x += y
    assert torch.equal(x.F, want)
    assert (MEF.relu(x).F >= 0).all()
    conv = ME.MinkowskiConvolution(8, 8, kernel_size=3, stride=2, dimension=3).to(gpu)
    up = ME.MinkowskiConvolutionTranspose(8, 4, kernel_size=3, stride=2, dimension=3).to(gpu)
    with torch.no_grad():
        d = conv(x)
        u = up(d)
    assert d.tensor_stride == [2, 2, 2] and u.tensor_stride == [1, 1, 1]
    assert u.coordinate_map_key == x.coordinate_map_key and u.F.shape == (len(coords), 4)
    # mismatched maps / duplicate coordinates raise like ME
    with pytest.raises(RuntimeError):
        ME.cat(x, d)
    with pytest.raises(RuntimeError):
        ME.SparseTensor(torch.ones(2, 1, device=gpu), torch.zeros((2, 4), dtype=torch.int32, device=gpu))
    # ME.utils runs on the host (DataLoader workers) and keeps the first point per voxel
    pts = np.random.default_rng(0).uniform(-1, 1, (500, 3))
    idx = ME.utils.sparse_quantize(np.floor(pts / 0.2), return_index=True, return_maps_only=True)
    from oracle import sparse

    assert np.array_equal(idx, sparse.sparse_quantize(np.floor(pts / 0.2)))
    bc, bf = ME.utils.sparse_collate([np.floor(pts[idx] / 0.2)], [np.ones((len(idx), 1))])
    assert bc.dtype == torch.int32 and bc.shape == (len(idx), 4) and (bc[:, 0] == 0).all()


def test_gpu_voxelize_matches_sparse_quantize(gpu):
    """cs_voxelize == ME.utils.sparse_quantize + sparse_collate semantics."""
    from corsair_amd import backend as B, synth
    from oracle import sparse

    clouds = [synth.make_cloud(i, 15000)[:n] for i, n in ((40, 10000), (41, 3000), (42, 1))]
    off = np.concatenate([[0], np.cumsum([len(c) for c in clouds])]).tolist()
    xyz = torch.from_numpy(np.concatenate(clouds)).to(gpu)
    for voxel in (0.03, 0.02):
        keep, grid, out_off = B.voxelize(xyz, off, voxel)
        keep, grid = keep.cpu().numpy(), grid.cpu().numpy()
        grids, keeps = [], []
        for b, c in enumerate(clouds):
            _, g, k = sparse.quantize_cloud(c, voxel)
            grids.append(g)
            keeps.append(k + off[b])
        assert np.array_equal(keep, np.concatenate(keeps))
        assert np.array_equal(grid, sparse.sparse_collate(grids))
        assert out_off == np.concatenate([[0], np.cumsum([len(g) for g in grids])]).tolist()


def test_gpu_voxelize_f64_floors_what_the_reference_floors(gpu):
    """the QUERY side of the reference quantises f64 clouds -- datasets/CategoryDataset.py:179-197 floors
    the f64 output of apply_transform, evaluation-shapenet.py:97-119 floors `pc @ R.T + t` and narrows the KEPT points
    afterwards.  cs_voxelize_f64 on 200 f64-posed clouds: kept indices and grids equal np.floor(p64 / voxel) + first
    occurrence exactly (plain NumPy here, not the oracle), and the set contains clouds on which narrowing to f32 first
    (what the f32 entry would see) puts a point into another voxel."""
    from corsair_amd import backend as B, synth
    from oracle import sparse

    voxel = 0.03
    clouds = [synth.apply_pose(synth.make_cloud(c, 15000)[:10000], synth.random_pose(c), np.float64) for c in range(200)]
    assert all(c.dtype == np.float64 for c in clouds)
    moved, other_keep = 0, 0
    for s in range(0, 200, 40):
        chunk = clouds[s:s + 40]
        off = np.concatenate([[0], np.cumsum([len(c) for c in chunk])]).tolist()
        xyz = torch.from_numpy(np.concatenate(chunk)).to(gpu)
        assert xyz.dtype == torch.float64
        keep, grid, out_off = B.voxelize(xyz, off, voxel)
        keep, grid = keep.cpu().numpy(), grid.cpu().numpy()
        k32, g32, _ = (t.cpu().numpy() if torch.is_tensor(t) else t for t in B.voxelize(xyz.to(torch.float32), off, voxel))
        for b, c in enumerate(chunk):
            g = np.floor(c / voxel)                                    # the reference's expression, f64
            _, first = np.unique(g.astype(np.int64), axis=0, return_index=True)
            first = np.sort(first)
            lo, hi = out_off[b], out_off[b + 1]
            assert np.array_equal(keep[lo:hi] - off[b], first), (s + b)
            assert np.array_equal(grid[lo:hi, 1:], g[first].astype(np.int32)) and (grid[lo:hi, 0] == b).all()
            # the oracle restates the same thing, type-preserving
            _, og, ok = sparse.quantize_cloud(c, voxel)
            assert np.array_equal(ok, first) and np.array_equal(og, grid[lo:hi, 1:])
            g_narrow = np.floor(c.astype(np.float32) / np.float32(voxel))
            if not np.array_equal(g_narrow.astype(np.float64), g):
                moved += 1
                sel = (k32 >= off[b]) & (k32 < off[b + 1])
                if not (np.array_equal(k32[sel] - off[b], first) and np.array_equal(g32[sel][:, 1:], g[first].astype(np.int32))):
                    other_keep += 1
    assert moved >= 3 and other_keep >= 1, (moved, other_keep)


def test_embed_groups_quantises_every_cloud_in_its_own_type(gpu):
    """Pipeline.embed_groups = evaluation-shapenet.py:299-310: an f32 model and its f64 posed copy in one forward; the
    result equals two separate embeds (eval-mode BN: samples are independent), origins are the kept points narrowed
    AFTER the selection."""
    from corsair_amd import harness, synth

    sd, emb = synth.make_state_dicts(31)
    pipe = harness.Pipeline(sd, emb, device=gpu)
    pc = synth.make_cloud(110, 15000)[:10000]
    posed = synth.apply_pose(pc, synth.random_pose(110), np.float64)
    a = torch.from_numpy(pc).to(gpu)
    b = torch.from_numpy(posed).to(gpu)
    both = pipe.embed_groups([(a, [0, len(pc)]), (b, [0, len(posed)])])
    ea = pipe.embed_batch(a, [0, len(pc)])
    eb = pipe.embed_batch(b, [0, len(posed)])
    o = both.offsets
    assert o == [0, ea.offsets[1], ea.offsets[1] + eb.offsets[1]]
    assert torch.equal(both.F[:o[1]], ea.F) and torch.equal(both.F[o[1]:], eb.F)
    assert torch.equal(both.origin[o[1]:], eb.origin) and both.origin.dtype == torch.float32
    assert torch.equal(both.desc, torch.cat([ea.desc, eb.desc]))
    g = np.floor(posed / 0.03)
    _, first = np.unique(g.astype(np.int64), axis=0, return_index=True)
    assert np.array_equal(eb.origin.cpu().numpy(), posed[np.sort(first)].astype(np.float32))
    with pytest.raises(TypeError):
        pipe.embed_clouds([pc, posed])


def test_fused_bn_epilogue_matches_torch_batchnorm1d(gpu):
    """MinkowskiBatchNorm is nn.BatchNorm1d over the rows (model/common.py:22).  The product folds it into the
    convolution epilogue; here the un-normalised GPU convolution goes through the torch module itself (eval
    mode, running statistics) and must agree with the fused kernel at 1e-6 -- a check of the fold that does not
    share it with the oracle."""
    from corsair_amd import backend as B, engine
    from tests.helpers import make_batch

    coords, feats, _, _ = make_batch([8, 9], n_points=3000)
    m = engine.BatchMaps(torch.from_numpy(coords).to(gpu))
    rng = np.random.default_rng(5)
    C = 64
    x = torch.from_numpy(rng.standard_normal((coords.shape[0], 32)).astype(np.float32)).to(gpu)
    w = torch.from_numpy((rng.standard_normal((27, 32, C)) * 0.1).astype(np.float32)).to(gpu)
    sd = {"n.bn.weight": rng.uniform(0.5, 2.0, C).astype(np.float32), "n.bn.bias": rng.standard_normal(C).astype(np.float32),
          "n.bn.running_mean": rng.standard_normal(C).astype(np.float32),
          "n.bn.running_var": rng.uniform(0.05, 3.0, C).astype(np.float32)}
    bn = torch.nn.BatchNorm1d(C, eps=1e-5, momentum=0.05)
    with torch.no_grad():
        bn.weight.copy_(torch.from_numpy(sd["n.bn.weight"]))
        bn.bias.copy_(torch.from_numpy(sd["n.bn.bias"]))
        bn.running_mean.copy_(torch.from_numpy(sd["n.bn.running_mean"]))
        bn.running_var.copy_(torch.from_numpy(sd["n.bn.running_var"]))
    bn.eval()
    raw = B.conv_fwd(m.s1, x, w)
    with torch.no_grad():
        want = torch.relu(bn(raw.cpu())).numpy()
    s, b = engine.fold_bn(sd, "n")
    got = B.conv_fwd(m.s1, x, w, torch.from_numpy(s).to(gpu), torch.from_numpy(b).to(gpu), None, True).cpu().numpy()
    assert np.allclose(got, want, rtol=1e-6, atol=1e-6 * np.abs(want).max()), np.abs(got - want).max()
