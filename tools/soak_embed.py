"""Soak of the multi-stream kernel-map call: the same batch embedded N times (voxelise, coordinate maps,
cs_kernelmap_build_many on four streams, forward, embedding), alternating with a different batch so the pool's
scratch changes hands in between; every output must equal the first run's bit for bit, and the maps must equal maps
built one by one.  `python tools/soak_embed.py [n]`"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from corsair_amd import backend as B, harness, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda:0")
sd, emb = synth.make_state_dicts(31)
pipe = harness.Pipeline(sd, emb, device=dev)
def batch(first, nb, npts):
    clouds = [synth.make_cloud(first + c, 15000)[:npts] for c in range(nb)]
    return (torch.from_numpy(np.concatenate(clouds)).to(dev), np.concatenate([[0], np.cumsum([len(c) for c in clouds])]).tolist())
A, Bb = batch(0, 32, 10000), batch(100, 24, 7000)
def run(x):
    s = pipe.embed_batch(*x)
    return [t.cpu().numpy() for t in (s.F, s.origin, s.desc)] + [np.asarray(s.offsets)]
want_a, want_b = run(A), run(Bb)
bad = 0
for i in range(n):
    for x, want in ((A, want_a), (Bb, want_b)):
        got = run(x)
        if not all(np.array_equal(g, w) for g, w in zip(got, want)):
            bad += 1
            print("run", i, "differs")
print("embedding soak: %d double runs, %d mismatches" % (n, bad))
sys.exit(1 if bad else 0)
