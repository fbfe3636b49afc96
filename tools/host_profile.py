"""cProfile of one bench step (host-side overhead hunting)."""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from corsair_amd import harness, synth, _lib
dev = torch.device('cuda:0')
cfg = harness.Config(); sd, emb = synth.make_state_dicts(31)
pipe = harness.Pipeline(sd, emb, device=dev, config=cfg)
C = 128
catalog = pipe.embed_clouds([synth.make_cloud(c,15000)[:10000] for c in range(C)])
qs_clouds=[synth.apply_pose(synth.make_cloud(q%C,15000)[5000:], synth.random_pose(q, max_trans=0.0)) for q in range(32)]
xyz = torch.from_numpy(np.concatenate(qs_clouds)).to(dev); off = np.concatenate([[0],np.cumsum([len(c) for c in qs_clouds])]).tolist()
sym = np.ones(C, np.int32)
def step():
    qs = pipe.embed_batch(xyz, off)
    top = pipe.retrieve(qs.desc, catalog.desc, 1)[:,0].cpu().numpy()
    cads = catalog.gather(top)
    res = pipe.register(qs, cads, sym[top], force_gate=True)
    return res.T_best.cpu()
step(); torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(3): step()
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(28); print(s.getvalue()[:6000])
