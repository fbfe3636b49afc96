"""BASELINE.json configs[4] ("synthetic stress"): batch-64 sparse ResUNet forward on 15k-point clouds
at 2 cm voxels + 1M x 1M x 256-d descriptor top-10, with roofline fractions.  Sizes are scaled by
--clouds / --queries so the default run finishes in about a minute; per-unit rates are what the full
configuration (100k clouds, 10^6 queries) would see since both stages are embarrassingly batched.

  python tools/stress.py [--clouds 1024] [--queries 65536] [--catalog 1000000]
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from corsair_amd import _lib, backend as B, engine, harness, synth

ap = argparse.ArgumentParser()
ap.add_argument("--clouds", type=int, default=1024)
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--queries", type=int, default=65536)
ap.add_argument("--catalog", type=int, default=1000000)
args = ap.parse_args()
dev = torch.device("cuda:0")
cfg = harness.Config(voxel_size=0.02, n_points=15000, batch_size=args.batch)
sd, emb = synth.make_state_dicts(31)
pipe = harness.Pipeline(sd, emb, device=dev, config=cfg)
out = {}

# ---- forward -------------------------------------------------------------------------------------
n_unique = min(args.clouds, 256)
clouds = [synth.make_cloud(c, 15000) for c in range(n_unique)]
batches = []
for b in range(0, n_unique, args.batch):
    chunk = clouds[b:b + args.batch]
    batches.append((torch.from_numpy(np.concatenate(chunk)).to(dev),
                    np.concatenate([[0], np.cumsum([len(c) for c in chunk])]).tolist()))
pipe.embed_batch(*batches[0]); torch.cuda.synchronize()
_lib.prof_enable(True); _lib.prof_reset()
t0 = time.time(); done = 0; vox = 0
while done < args.clouds:
    for xyz, off in batches:
        es = pipe.embed_batch(xyz, off); done += len(off) - 1; vox += es.F.shape[0]
        if done >= args.clouds: break
torch.cuda.synchronize(); dt = time.time() - t0
_lib.prof_enable(False)
ms, n, flop = _lib.prof_get("conv")
kms, kn, _ = _lib.prof_get("kmap")
out["forward"] = {"clouds": done, "batch": args.batch, "clouds_per_s": done / dt, "voxels_per_cloud": vox / done,
                  "conv_ms_per_batch": ms / (done / args.batch), "kmap_ms_per_batch": kms / (done / args.batch),
                  "conv_useful_tflops": flop / (ms * 1e-3) / 1e12, "conv_frac_of_f32_mfma_peak": flop / (ms * 1e-3) / 1e12 / 157.3,
                  "gflop_per_cloud": flop / done / 1e9,
                  "est_100k_clouds_s": 100000 / (done / dt)}

# ---- top-10 ------------------------------------------------------------------------------------------
q = torch.from_numpy(synth.make_descriptors(args.queries, 256, seed=1234)).to(dev)
x = torch.from_numpy(synth.make_descriptors(args.catalog, 256, seed=4321)).to(dev)
B.l2_topk(q[:4096].contiguous(), x, 10); torch.cuda.synchronize()
t0 = time.time(); idx, dist = B.l2_topk(q, x, 10, True); torch.cuda.synchronize(); dt = time.time() - t0
fl = 2.0 * args.queries * args.catalog * 256
import ctypes
from corsair_amd import _lib
st = (ctypes.c_uint64 * 2)(); _lib.load().cs_l2_topk_stats(st, 0)
# the f16 shortlist executes 3 products per algorithmic multiply-add (x_hi v_hi + x_lo v_hi + x_hi v_lo)
out["top10"] = {"queries": args.queries, "catalog": args.catalog, "seconds": dt,
                "algorithmic_tflops": fl / dt / 1e12, "f16_mfma_tflops": 3 * fl / dt / 1e12,
                "frac_of_f16_mfma_peak": 3 * fl / dt / 1e12 / 2516.6,
                "est_1M_x_1M_s": dt * (1e6 / args.queries) * (1e6 / args.catalog),
                "f16_shortlist_queries": int(st[0]), "recomputed_by_f64_path": int(st[1]),
                "sorted": bool((dist[:, 1:] >= dist[:, :-1]).all())}
print(json.dumps(out, indent=1))
