"""Executed / useful matrix work of the convolution's tiling order for 32-row and 16-row groups, on the CPU (oracle maps):
rows sorted by the Gray rank of their neighbour mask, a group executes every offset any of its rows has (k_conv_dma skips the
others).  Per kernel map and weighted by the channels of the ResUNetBN2C layers that run on it (build a
v_mfma_f32_16x16x4_f32 variant only if the weighted executed work drops >= 8 %).
usage: python tools/exec_ratio_cpu.py [clouds points voxel]   (stress batch: 64 15000 0.02; chair forward: 128 10000 0.03)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from corsair_amd import synth
from oracle import resunet as oref, sparse as osp

NB = int(sys.argv[1]) if len(sys.argv) > 1 else 64
NP = int(sys.argv[2]) if len(sys.argv) > 2 else 15000
VOX = float(sys.argv[3]) if len(sys.argv) > 3 else 0.02
# (Cin x Cout summed over the gathered layers that use each map: model/resunet.py:207-280, CHANNELS / TR_CHANNELS)
LAYERS = {"s1": 2 * 32 * 32 + 2 * 64 * 64, "s1_s2": 32 * 64, "s2": 2 * 64 * 64 + 2 * 64 * 64, "s2_s4": 64 * 128,
          "s4": 2 * 128 * 128 + 2 * 128 * 128, "s4_s8": 128 * 256, "s8": 2 * 256 * 256, "s8_s4_T": 256 * 128,
          "s4_s2_T": 256 * 64, "s2_s1_T": 128 * 64}
t0 = time.time()
grids = [osp.quantize_cloud(synth.make_cloud(c, 15000)[:NP], VOX)[1] for c in range(NB)]
coords = osp.sparse_collate(grids)
_, okm = oref.build_maps(coords)
print("# %d clouds x %d points @ %.3f: %d voxels, maps built in %.1f s" % (NB, NP, VOX, len(coords), time.time() - t0))


def inv_gray(m):
    r = m.copy()
    s = 1
    while s < 32:
        r ^= r >> s
        s *= 2
    return r


# the tiling key of coordmap.hip (round 5): bit j of the mask = kernel offset KORDER[j] (centre, faces, edges, corners),
# then the Gray rank; ORDER=k on the command line (4th argument) evaluates the plain-k order of rounds 3-4
OFFS = [(k % 3 - 1, (k // 3) % 3 - 1, k // 9 - 1) for k in range(27)]
KORDER = sorted(range(27), key=lambda k: (sum(abs(v) for v in OFFS[k]), k))
PLAIN = len(sys.argv) > 4 and sys.argv[4] == "k"


def tiling_key(mask):
    if PLAIN:
        return inv_gray(mask)
    pm = np.zeros_like(mask)
    for j, k in enumerate(KORDER):
        pm |= ((mask >> np.uint32(k)) & np.uint32(1)) << np.uint32(j)
    return inv_gray(pm)


tot = {32: 0.0, 16: 0.0, "useful": 0.0}
print("%-8s %8s %9s | executed / useful at 32-row groups | at 16-row groups | share of the network's useful work" % ("map", "rows", "pairs"))
rows_out = []
for name, w in LAYERS.items():
    nbr = okm[name]
    has = nbr >= 0
    mask = (has.astype(np.uint32) << np.arange(27, dtype=np.uint32)).sum(1).astype(np.uint32)
    m = mask[np.argsort(tiling_key(mask), kind="stable")]
    pairs = int(has.sum())
    ratio = {}
    for G in (32, 16):
        pad = (-len(m)) % G
        g = np.bitwise_or.reduce(np.concatenate([m, np.zeros(pad, np.uint32)]).reshape(-1, G), axis=1)
        pop = np.array([bin(int(x)).count("1") for x in g], dtype=np.int64)
        ratio[G] = pop.sum() * G / max(pairs, 1)
        tot[G] += pop.sum() * G * w
    tot["useful"] += pairs * w
    rows_out.append((name, len(m), pairs, ratio[32], ratio[16], pairs * w))
for name, n, pairs, r32, r16, uw in rows_out:
    print("%-8s %8d %9d | %33.3f | %16.3f | %5.1f %%" % (name, n, pairs, r32, r16, 100.0 * uw / tot["useful"]))
e32, e16 = tot[32] / tot["useful"], tot[16] / tot["useful"]
print("weighted over the gathered layers: executed / useful %.3f (32-row groups), %.3f (16-row groups): the executed work drops %.1f %%"
      % (e32, e16, 100.0 * (1.0 - e16 / e32)))
