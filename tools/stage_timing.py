import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from corsair_amd import _lib, harness, synth, backend as B, registration as R
def log(*a):
    print(*a, flush=True)
dev = torch.device('cuda:0')
cfg = harness.Config()
sd, emb = synth.make_state_dicts(31)
pipe = harness.Pipeline(sd, emb, device=dev, config=cfg)
NB = int(sys.argv[1]); C = int(sys.argv[2]); MAXIT = int(sys.argv[3])
cfg.ransac_max_iter = MAXIT
t=time.time()
cat = [synth.make_cloud(c,15000)[:10000] for c in range(C)]
log('gen', time.time()-t)
t=time.time(); catalog = pipe.embed_clouds(cat); torch.cuda.synchronize(); log('catalog embed', time.time()-t, catalog.F.shape)
t=time.time(); catalog = pipe.embed_clouds(cat); torch.cuda.synchronize(); log('catalog embed again', time.time()-t)
qs_clouds=[]; 
for q in range(NB):
    T = synth.random_pose(q, max_trans=0.0)
    qs_clouds.append(synth.apply_pose(synth.make_cloud(q%C,15000)[5000:], T))
xyz = torch.from_numpy(np.concatenate(qs_clouds)).to(dev); off = np.concatenate([[0],np.cumsum([len(c) for c in qs_clouds])]).tolist()
_lib.prof_enable(True); _lib.prof_reset()
for rep in range(2):
    t=time.time(); qs = pipe.embed_batch(xyz, off); torch.cuda.synchronize(); log('query embed', time.time()-t)
    t=time.time(); top = pipe.retrieve(qs.desc, catalog.desc, 1)[:,0].cpu().numpy(); log('retrieve', time.time()-t, top[:8])
    t=time.time(); cads = catalog.gather(top); torch.cuda.synchronize(); log('gather', time.time()-t)
    sym = np.ones(C, np.int32); sym[::326]=4
    t=time.time(); res = pipe.register(qs, cads, sym[top], use_symmetry=False); torch.cuda.synchronize(); log('register nosym', time.time()-t, res.iters.cpu().numpy()[:8])
    t=time.time(); res = pipe.register(qs, cads, sym[top]); torch.cuda.synchronize(); log('register sym', time.time()-t, res.n_problems, res.ok[:8], res.iters.cpu().numpy()[:12])
for name in ("conv","ransac_eval","ransac_pre","ransac_hyp","knn","chamfer","topk","symcut","kmap"):
    log(name, _lib.prof_get(name))
import ctypes
st = (ctypes.c_uint64 * 5)(); _lib.load().cs_ransac_prefilter_stats(st, 0)
log('prefilter stats [viol, checked, slack, survivors, generated]', [int(v) for v in st])
