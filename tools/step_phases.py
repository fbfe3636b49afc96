"""Host-side wall time of the phases of one sequential bench step (no profiler): where the GPU waits for Python.
Each phase is closed with a device synchronisation, so the numbers are phase latencies, not overlapped times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from corsair_amd import harness, synth, _lib, registration, backend as B
dev = torch.device("cuda:0")
cfg = harness.Config(); sd, emb = synth.make_state_dicts(31)
pipe = harness.Pipeline(sd, emb, device=dev, config=cfg)
C = 652
catalog = pipe.embed_clouds([synth.make_cloud(c, 15000)[:10000] for c in range(C)])
qs_clouds = [synth.apply_pose(synth.make_cloud(q % C, 15000)[5000:], synth.random_pose(q, max_trans=0.0)) for q in range(32)]
xyz = torch.from_numpy(np.concatenate(qs_clouds)).to(dev)
off = np.concatenate([[0], np.cumsum([len(c) for c in qs_clouds])]).tolist()
sym = np.ones(C, np.int32)
T = {}
def mark(name, t0):
    torch.cuda.synchronize(); T[name] = T.get(name, 0.0) + time.perf_counter() - t0
def step(timed):
    t = time.perf_counter(); qs = pipe.embed_batch(xyz, off)
    th = time.perf_counter() - t
    if timed: T["embed host return"] = T.get("embed host return", 0.0) + th; mark("embed (synced)", t)
    t = time.perf_counter(); top = _lib.to_host(pipe.retrieve(qs.desc, catalog.desc, 1)[:, 0])[0]
    if timed: mark("retrieve + download", t)
    t = time.perf_counter(); cads = catalog.gather(top)
    th = time.perf_counter() - t
    if timed: T["gather host return"] = T.get("gather host return", 0.0) + th; mark("catalog.gather (synced)", t)
    t = time.perf_counter(); res = pipe.register(qs, cads, sym[top], force_gate=True)
    out = _lib.to_host(res.T_best, res.cd_best)
    if timed: mark("register + download", t)
for _ in range(2): step(False)
N = 10
for _ in range(N): step(True)
for k, v in T.items(): print("%-28s %7.3f ms" % (k, v / N * 1e3))
