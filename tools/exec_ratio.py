"""Executed / useful MFMA work of the tiling order on a real batch: rows sorted by the Gray rank of their neighbour mask,
32-row groups, a group executes every offset any of its rows has.  `python tools/exec_ratio.py [batch points voxel]`"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 32
NP = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
VOX = float(sys.argv[3]) if len(sys.argv) > 3 else 0.03
import numpy as np, torch
from corsair_amd import backend as B, engine, synth
dev = torch.device('cuda:0')
clouds = [synth.make_cloud(c, 15000)[:NP] for c in range(NB)]
xyz = torch.from_numpy(np.concatenate(clouds)).to(dev); off = np.concatenate([[0], np.cumsum([len(c) for c in clouds])]).tolist()
keep, grid, out_off = B.voxelize(xyz, off, VOX)
maps = engine.BatchMaps(grid)
def inv_gray(m):
    r = m.copy(); s = 1
    while s < 32:
        r ^= r >> s; s *= 2
    return r
for name in ("s1", "s1_s2", "s2", "s2_s4", "s4", "s4_s8", "s8", "s8_s4_T", "s4_s2_T", "s2_s1_T"):
    km = getattr(maps, name)
    nbr = km.table().cpu().numpy()                      # [n_out, kvol]
    has = nbr >= 0
    mask = (has.astype(np.uint32) << np.arange(nbr.shape[1], dtype=np.uint32)).sum(1).astype(np.uint32)
    order = np.argsort(inv_gray(mask), kind="stable")
    m = mask[order]
    pad = (-len(m)) % 32
    g = np.bitwise_or.reduce(np.concatenate([m, np.zeros(pad, np.uint32)]).reshape(-1, 32), axis=1)
    pop = np.array([bin(int(x)).count("1") for x in g])
    pairs = int(has.sum())
    print(f"{name:8s} rows {len(m):7d} pairs {pairs:8d}  executed/useful {pop.sum() * 32 / pairs:5.2f}  offsets per group: mean {pop.mean():5.2f} "
          f"first decile {pop[:len(pop)//10+1].mean():5.2f} last decile {pop[-(len(pop)//10+1):].mean():5.2f}")
