"""Per-layer timing of one fused ResUNet forward: `python tools/conv_layers.py [batch points voxel]`
(default: 32 eval clouds, 10 000 points at 0.03; the stress shape is `64 15000 0.02`)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 32
NP = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
VOX = float(sys.argv[3]) if len(sys.argv) > 3 else 0.03
import numpy as np, torch
from corsair_amd import backend as B, engine, synth, harness
dev = torch.device('cuda:0')
sd, emb = synth.make_state_dicts(31)
pipe = harness.Pipeline(sd, emb, device=dev)
clouds = [synth.make_cloud(c, 15000)[:NP] for c in range(NB)]
xyz = torch.from_numpy(np.concatenate(clouds)).to(dev); off = np.concatenate([[0], np.cumsum([len(c) for c in clouds])]).tolist()
keep, grid, out_off = B.voxelize(xyz, off, VOX)
feats = torch.ones((grid.shape[0], 1), device=dev)
maps = engine.BatchMaps(grid)
print("rows", maps.c1.n, maps.c2.n, maps.c4.n, maps.c8.n, "pairs", maps.total_pairs())
orig = B.conv_fwd
log = []
def timed(kmap, x, weight, *a, **k):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = orig(kmap, x, weight, *a, **k); e1.record(); torch.cuda.synchronize()
    cin, cout = weight.shape[-2], weight.shape[-1]
    pairs = kmap.num_pairs if kmap is not None else x.shape[0]
    nout = kmap.n_out if kmap is not None else x.shape[0]
    log.append((nout, cin, cout, pairs, e0.elapsed_time(e1)))
    return r
pipe.engine.forward(grid, feats, maps); torch.cuda.synchronize()
B.conv_fwd = timed; engine.B.conv_fwd = timed
out, f8, _ = pipe.engine.forward(grid, feats, maps); pipe.engine.embed(f8, maps, NB)
tot = 0
for nout, cin, cout, pairs, ms in log:
    fl = 2.0 * pairs * cin * cout
    dense = 2.0 * nout * 27 * cin * cout
    tot += ms
    print(f"n_out={nout:7d} {cin:4d}->{cout:4d} pairs={pairs:8d} {ms*1e3:8.1f} us  useful {fl/ms/1e9:7.2f} TF  density {pairs/(nout*27.0):.2f}")
print("total conv ms", tot, " useful TF", sum(2.0 * l[3] * l[1] * l[2] for l in log) / tot / 1e9)
# wall time of whole forwards (maps prebuilt), back to back
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
B.conv_fwd = orig; engine.B.conv_fwd = orig
e0.record()
for _ in range(5):
    pipe.engine.forward(grid, feats, maps)
e1.record(); torch.cuda.synchronize()
print("forward (maps prebuilt) ms", e0.elapsed_time(e1) / 5)
