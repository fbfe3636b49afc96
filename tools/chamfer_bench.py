"""Time cs_chamfer_1dir on the chair step's shape: 97 hypotheses of ~4.5 k x 4.5 k voxelised clouds.
usage: chamfer_bench.py [reps]   (CS_CHAMFER_F16=0: the f64 matrix-pipe kernel; CS_CHAMFER_STATS=1 prints the fallbacks)"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from corsair_amd import _lib, backend as B, synth
dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
keep = lambda pc: pc[np.unique(np.floor(pc / 0.03).astype(np.int64), axis=0, return_index=True)[1]]
srcs, tgts, Ts, sseg, tseg = [], [], [], [], []
for q in range(32):
    full = synth.make_cloud(q, 15000)
    srcs.append(keep(full[5000:]).astype(np.float32))
    tgts.append(keep(full[:10000]).astype(np.float32))
    for h in range(3):
        Ts.append(synth.random_pose(10 * q + h, max_trans=0.02 * h).astype(np.float32))
        sseg.append(q)
        tseg.append(q)
off_s = np.concatenate([[0], np.cumsum([len(c) for c in srcs])]).tolist()
off_t = np.concatenate([[0], np.cumsum([len(c) for c in tgts])]).tolist()
S, Tg, TT = (torch.from_numpy(np.concatenate(a)).to(dev) for a in (srcs, tgts, [t.reshape(1, 16) for t in Ts]))
TT = TT.reshape(-1, 4, 4)
st = (ctypes.c_uint64 * 2)()
_lib.load().cs_chamfer_f16_stats(st, 1)
out = B.chamfer_1dir(S, off_s, Tg, off_t, sseg, tseg, TT)
pairs = sum(len(srcs[a]) * len(tgts[b]) for a, b in zip(sseg, tseg))
for rep in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps):
        out = B.chamfer_1dir(S, off_s, Tg, off_t, sseg, tseg, TT)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / reps
    print("chamfer: %.3f ms per call (%d problems, %.2f G pairs, %.2f T pairs/s), checksum %.17g" % (
        dt * 1e3, len(sseg), pairs / 1e9, pairs / dt / 1e12, float(out.sum())))
_lib.load().cs_chamfer_f16_stats(st, 0)
print("f16 ranking stats [tiles, recomputed by the f64 kernel]:", [int(v) for v in st])
