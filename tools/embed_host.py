import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from corsair_amd import _lib, harness, synth
dev = torch.device('cuda:0')
cfg = harness.Config()
sd, emb = synth.make_state_dicts(31)
pipe = harness.Pipeline(sd, emb, device=dev, config=cfg)
qs=[synth.make_cloud(q,15000)[5000:] for q in range(32)]
xyz = torch.from_numpy(np.concatenate(qs)).to(dev); off = np.concatenate([[0],np.cumsum([len(c) for c in qs])]).tolist()
for rep in range(3):
    q = pipe.embed_batch(xyz, off); torch.cuda.synchronize()
import cProfile, pstats
for rep in range(3):
    torch.cuda.synchronize(); t0=time.perf_counter()
    q = pipe.embed_batch(xyz, off)
    t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
    print("embed_batch: host returns after %.2f ms, GPU done after %.2f ms"%((t1-t0)*1e3,(t2-t0)*1e3))
pr=cProfile.Profile(); pr.enable()
for rep in range(5):
    q = pipe.embed_batch(xyz, off)
torch.cuda.synchronize()
pr.disable()
st=pstats.Stats(pr); st.sort_stats('cumulative').print_stats(28)
