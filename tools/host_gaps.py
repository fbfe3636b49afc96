"""Where does the host spend time between GPU launches in one bench step? (line-level timers)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from corsair_amd import harness, synth, registration as R, backend as B
dev = torch.device('cuda:0')
cfg = harness.Config(); sd, emb = synth.make_state_dicts(31)
pipe = harness.Pipeline(sd, emb, device=dev, config=cfg)
C = 128
catalog = pipe.embed_clouds([synth.make_cloud(c,15000)[:10000] for c in range(C)])
qs_clouds=[synth.apply_pose(synth.make_cloud(q%C,15000)[5000:], synth.random_pose(q, max_trans=0.0)) for q in range(32)]
xyz = torch.from_numpy(np.concatenate(qs_clouds)).to(dev); off = np.concatenate([[0],np.cumsum([len(c) for c in qs_clouds])]).tolist()
sym = np.ones(C, np.int32)
import cProfile, pstats, io
def step():
    qs = pipe.embed_batch(xyz, off)
    top = pipe.retrieve(qs.desc, catalog.desc, 1)[:,0].cpu().numpy()
    cads = catalog.gather(top)
    res = pipe.register(qs, cads, sym[top], force_gate=True)
    return res.T_best.cpu()
step(); torch.cuda.synchronize()
# wrap backend calls to measure time NOT inside them
inside = [0.0]
def wrap(mod, name):
    f = getattr(mod, name)
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); inside[0] += time.perf_counter() - t; return r
    setattr(mod, name, g)
for n in ("knn_feat","ransac_batch","chamfer_1dir","symcut_fit","symcut_labels","l2_topk","voxelize","conv_fwd","row_l2_normalize","segmented_max"):
    wrap(B, n)
import corsair_amd.engine as E
t0 = time.perf_counter()
for _ in range(3): step()
torch.cuda.synchronize(); tot = time.perf_counter() - t0
print("per step ms", tot/3*1e3, "inside backend calls", inside[0]/3*1e3, "python outside", (tot-inside[0])/3*1e3)
pr = cProfile.Profile(); pr.enable(); step(); torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(40); print(s.getvalue()[:9000])
