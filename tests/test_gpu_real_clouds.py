"""GPU parity on REAL occupancy: the bundled ShapeNet PC15k clouds of the reference
(docker/data/ShapeNetCore.v2.PC15k/*/test, 12 of them as fixtures) through voxeliser, coordinate /
kernel maps and the whole ResUNetBN2C + embedding forward -- bit-exact against the oracle -- and an
oracle-free end-to-end check that the registration pipeline recovers a known pose."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _real_clouds():
    out = []
    z = np.load(os.path.join(GOLD, "real_clouds.npz"))
    raw = [z["chair"], z["table"]]
    z10 = np.load(os.path.join(GOLD, "real_clouds10.npz"))
    raw += list(z10["clouds"].astype(np.float32))
    for pc in raw:
        pc = pc.astype(np.float32)
        pc = pc - pc.mean(0)
        out.append((pc / np.max(np.linalg.norm(pc, 2, 1))).astype(np.float32))   # utils/preprocess.py:32-36
    return out


@pytest.mark.parametrize("voxel", [0.03, 0.02])
def test_real_clouds_voxelize_maps_forward_bit_exact(gpu, oracle_native, voxel):
    from corsair_amd import backend as B, engine, synth
    from oracle import resunet as oref, sparse as osp

    clouds = _real_clouds()
    # --- voxeliser: first point per voxel, input order (ME.utils.sparse_quantize) ---
    off = np.concatenate([[0], np.cumsum([len(c) for c in clouds])]).tolist()
    xyz = torch.from_numpy(np.concatenate(clouds)).to(gpu)
    keep, grid, out_off = B.voxelize(xyz, off, voxel)
    grids = []
    for i, pc in enumerate(clouds):
        _, g, idx = osp.quantize_cloud(pc, voxel)
        grids.append(g)
        assert np.array_equal(keep[out_off[i]:out_off[i + 1]].cpu().numpy() - off[i], idx), i
    coords = osp.sparse_collate(grids)
    assert np.array_equal(grid.cpu().numpy(), coords)
    # --- coordinate maps and all 10 neighbour tables ---
    m = engine.BatchMaps(grid)
    omaps, okm = oref.build_maps(coords)
    for name in ("c1", "c2", "c4", "c8"):
        assert np.array_equal(getattr(m, name).coords.cpu().numpy(), omaps[name]), name
    for name, nbr in okm.items():
        km = getattr(m, name)
        assert np.array_equal(km.table().cpu().numpy(), nbr), name
        assert km.num_pairs == int((nbr >= 0).sum())
    # occupancy as SURVEY Appendix B reports it for these clouds
    per_point = (okm["s1"] >= 0).sum() / len(coords)
    assert 5.0 < per_point < 16.0
    # --- forward + embedding ---
    sd, emb = synth.make_state_dicts(31)
    eng = engine.ResUNetEngine(sd, emb, device=gpu)
    feats = np.ones((len(coords), 1), np.float32)
    out, feat8, _ = eng.forward(grid, torch.from_numpy(feats).to(gpu), maps=m)
    g = eng.embed(feat8, m, len(clouds))
    want_out, want_feat, _ = oref.resunet_forward(sd, coords, feats)
    want_g = oref.embedding_forward(emb, want_feat, omaps["c8"][:, 0], len(clouds))
    assert np.array_equal(out.cpu().numpy(), want_out)
    assert np.array_equal(feat8.cpu().numpy(), want_feat)
    assert np.array_equal(g.cpu().numpy(), want_g)


@pytest.mark.parametrize("force_gate", [False, True])
def test_pipeline_recovers_a_known_pose(gpu, force_gate):
    """Oracle-free: the query is a posed re-sampling of a CAD cloud and both sides carry pose-invariant
    'features' (the CAD-frame coordinates, zero-padded to 16-d), so feature 5-NN = spatial 5-NN in the
    CAD frame.  sym_pose_batch must return the pose that maps the query onto the CAD -- in the
    convention eval_pose expects with T0 = query pose, T1 = I (the bench's) -- and RANSAC must leave
    through its confidence bound.  A transposed / inverted transform, swapped source and target or a
    wrong correspondence order would fail here even if product and oracle shared it."""
    from corsair_amd import backend as B, registration as R, synth
    from corsair_amd.utils.eval_pose import eval_pose

    voxel, syms = 0.03, [1, 1, 2, 4]
    cad_xyz, q_xyz, cad_F, q_F, Ts = [], [], [], [], []
    for p, cid in enumerate([11, 12, 13, 14]):
        full = synth.make_cloud(cid, 15000)
        T = synth.random_pose(100 + p, max_trans=0.5)
        cad, q_cad_frame = full[:10000], full[5000:]
        c_keep, _, _ = B.voxelize(torch.from_numpy(cad).to(gpu), [0, len(cad)], voxel)
        q_posed = synth.apply_pose(q_cad_frame, T)
        q_keep, _, _ = B.voxelize(torch.from_numpy(q_posed).to(gpu), [0, len(q_posed)], voxel)
        c_keep, q_keep = c_keep.cpu().numpy(), q_keep.cpu().numpy()
        cad_xyz.append(cad[c_keep])
        q_xyz.append(q_posed[q_keep])
        cad_F.append(np.pad(cad[c_keep], ((0, 0), (0, 13))))
        q_F.append(np.pad(q_cad_frame[q_keep], ((0, 0), (0, 13))))
        Ts.append(T)
    off0 = np.concatenate([[0], np.cumsum([len(x) for x in q_xyz])]).tolist()
    off1 = np.concatenate([[0], np.cumsum([len(x) for x in cad_xyz])]).tolist()
    dev = lambda a: torch.from_numpy(np.concatenate(a).astype(np.float32)).to(gpu)
    max_iter = 100000
    res = R.sym_pose_batch(dev(q_F), dev(q_xyz), off0, dev(cad_F), dev(cad_xyz), off1, syms, 5, 0.2, 0,
                           None, 100, max_iter, 0.999, True, force_gate)
    iters = res.iters.cpu().numpy()
    assert (iters[:4] < max_iter // 10).all(), iters      # confidence exit on clean correspondences
    for name, T_est, cd in (("ransac", res.T_ransac, res.cd_ransac), ("best", res.T_best, res.cd_best)):
        T_est, cd = T_est.cpu().numpy(), cd.cpu().numpy()
        for p in range(4):
            rte, rre = eval_pose(T_est[p], Ts[p], np.eye(4), syms[p])
            assert rre < np.deg2rad(5.0) and rte < 0.05, (name, p, np.rad2deg(rre), rte)
            # the estimate maps query voxels onto the CAD surface: Chamfer well under a voxel
            assert cd[p] < voxel, (name, p, cd[p])
            moved = q_xyz[p].astype(np.float64) @ T_est[p][:3, :3].T.astype(np.float64) + T_est[p][:3, 3]
            back = synth.apply_pose(q_xyz[p], np.linalg.inv(Ts[p]))
            assert np.abs(moved - back).max() < 0.05
    if force_gate:
        assert res.n_problems > 4 and res.ok.all()
    assert (res.cd_best.cpu().numpy() <= res.cd_ransac.cpu().numpy()).all()   # utils/symmetry.py:322-324
