"""Time the kernel-map construction of one batch (coordinate maps of strides 1/2/4/8 + the ten kernel maps with their
tiling order) and one forward of the network on it.  usage: kmap_bench.py [stress|chair] [reps]
Env switches are read by the library: CS_PYRAMID=0 (chained level calls), CS_KMAP_GLOBAL=1 (global-table probes), CS_KMAP_TRACE=1 (phase report)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from corsair_amd import backend as B, engine, synth, _lib

kind = sys.argv[1] if len(sys.argv) > 1 else "stress"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
n_cl, n_pts, vox = (64, 15000, 0.02) if kind == "stress" else (128, 10000, 0.03)
clouds = [synth.make_cloud(c, 15000)[:n_pts] for c in range(n_cl)]
xyz = torch.from_numpy(np.concatenate(clouds)).to(dev)
off = np.concatenate([[0], np.cumsum([len(c) for c in clouds])]).tolist()
_, grid, _ = B.voxelize(xyz, off, vox)
sd, emb = synth.make_state_dicts(31)
eng = engine.ResUNetEngine(sd, emb, device=dev)
feats = torch.ones((grid.shape[0], 1), device=dev)

def timed(fn, n):
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    t = time.perf_counter(); e0.record()
    for _ in range(n):
        r = fn()
    e1.record(); torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3, e0.elapsed_time(e1) / n, r

maps = engine.BatchMaps(grid, n_cl)
pairs = maps.total_pairs()
print("%s batch: %d clouds, rows s1/s2/s4/s8 = %d / %d / %d / %d, pairs %d" % (
    kind, n_cl, maps.c1.n, maps.c2.n, maps.c4.n, maps.c8.n, sum(pairs.values())))
chk = sum(int(getattr(maps, k).table().to(torch.int64).sum()) for k in pairs)
for rep in range(3):
    wall, gpu, _ = timed(lambda: engine.BatchMaps(grid, n_cl), reps)
    print("maps: %.3f ms wall, %.3f ms event time per batch" % (wall, gpu))
if len(sys.argv) > 3 and sys.argv[3] == "maps-only":      # (under rocprofv3: only the map construction in the trace)
    sys.exit(0)
_lib.prof_reset(); _lib.prof_enable(True)
wall, gpu, _ = timed(lambda: eng.forward(grid, feats, n_batch=n_cl), reps)
_lib.prof_enable(False)
print("forward incl. maps: %.3f ms wall, %.3f ms event time; families (ms per forward): kmap %.3f, conv %.3f" % (
    wall, gpu, _lib.prof_get("kmap")[0] / reps, _lib.prof_get("conv")[0] / reps))
wall, gpu, _ = timed(lambda: eng.forward(grid, feats, maps), reps)
print("forward on ready maps: %.3f ms wall, %.3f ms event time" % (wall, gpu))
print("table checksum", chk, "pairs", sum(pairs.values()))
