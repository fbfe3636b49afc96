"""Wall time of the host-serial front of one embedding step (voxelise -> coordinate maps -> kernel maps), phase by phase with
a device synchronisation after each (so a phase = host latency + its GPU work), and the same front without the
synchronisations.  `python tools/embed_phases.py [batch points voxel]`"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 32
NP = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
VOX = float(sys.argv[3]) if len(sys.argv) > 3 else 0.03
import numpy as np, torch
from corsair_amd import backend as B, engine, synth
dev = torch.device('cuda:0')
clouds = [synth.make_cloud(c, 15000)[:NP] for c in range(NB)]
xyz = torch.from_numpy(np.concatenate(clouds)).to(dev); off = np.concatenate([[0], np.cumsum([len(c) for c in clouds])]).tolist()
def front(sync):
    t = [time.perf_counter()]; names = []
    def mark(n):
        if sync: torch.cuda.synchronize()
        t.append(time.perf_counter()); names.append(n)
    keep, grid, out_off = B.voxelize(xyz, off, VOX); mark("voxelize (returns the counts: one sync inside)")
    origin = xyz[keep]; feats = torch.ones((grid.shape[0], 1), device=dev); mark("origin gather + ones")
    c1 = B.CoordMap.create(grid, 1); mark("coordmap create")
    c2 = c1.stride(2); mark("stride 2")
    c4 = c2.stride(2); mark("stride 4")
    c8 = c4.stride(2); mark("stride 8")
    specs = [(c1, c1), (c1, c2), (c2, c2), (c2, c4), (c4, c4), (c4, c8), (c8, c8), (c8, c4, 3, True), (c4, c2, 3, True), (c2, c1, 3, True)]
    maps = B.KernelMap.build_many(specs); mark("ten kernel maps")
    torch.cuda.synchronize(); t.append(time.perf_counter()); names.append("drain")
    return names, np.diff(t) * 1e3
for _ in range(3): front(False)
acc = None
for _ in range(10):
    names, d = front(True); acc = d if acc is None else acc + d
print("with a synchronisation after every phase (ms):")
for n, v in zip(names, acc / 10): print("  %-52s %.3f" % (n, v))
print("  total %.3f" % (acc.sum() / 10))
tot = 0
for _ in range(10):
    torch.cuda.synchronize(); t0 = time.perf_counter(); front(False); tot += time.perf_counter() - t0
print("without: %.3f ms" % (tot * 100))
