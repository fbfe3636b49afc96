"""BASELINE.json configs[4], second half at FULL contract size: top-10 of 10^6 queries against a 10^6
catalog of row-normalised descriptors (Philox standard normal), 256-d (the reference's
descriptor width) and 512-d (the width BASELINE.json names).  Queries go through cs_l2_topk_catalog in 16 slabs
of 65 536 against ONE catalog handle (image and norms made once); the catalog stays resident.  Prints one JSON object.

  python tools/topk_full.py [--dims 256 512] [--n 1000000] [--slab 65536]
"""
import argparse, ctypes, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from corsair_amd import _lib, backend as B, synth

ap = argparse.ArgumentParser()
ap.add_argument("--dims", type=int, nargs="+", default=[256, 512])
ap.add_argument("--n", type=int, default=1000000)
ap.add_argument("--slab", type=int, default=65536)
args = ap.parse_args()
dev = torch.device("cuda:0")
_lib.require_gpu()
out = {}
for d in args.dims:
    t0 = time.time()
    x = torch.empty((args.n, d), dtype=torch.float32, device=dev)
    q = torch.empty((args.n, d), dtype=torch.float32, device=dev)
    for i, s in enumerate(range(0, args.n, 131072)):
        m = min(131072, args.n - s)
        x[s:s + m] = torch.from_numpy(synth.make_descriptors(m, d, seed=4321 + i)).to(dev)
        q[s:s + m] = torch.from_numpy(synth.make_descriptors(m, d, seed=1234 + i)).to(dev)
    print("[topk_full] d=%d: descriptors generated in %.1fs" % (d, time.time() - t0), file=sys.stderr, flush=True)
    cat = B.TopkCatalog(x)
    B.l2_topk(q[:4096], cat, 10)
    torch.cuda.synchronize()
    st = (ctypes.c_uint64 * 2)()
    _lib.load().cs_l2_topk_stats(st, 1)
    idx = torch.empty((args.n, 10), dtype=torch.int64, device=dev)
    first = torch.empty(args.n, dtype=torch.float64, device=dev)
    ok_sorted = True
    t0 = time.time()
    for s in range(0, args.n, args.slab):
        i, dist = B.l2_topk(q[s:s + args.slab], cat, 10, True)
        idx[s:s + args.slab] = i
        first[s:s + args.slab] = dist[:, 0]
        ok_sorted = ok_sorted and bool((dist[:, 1:] >= dist[:, :-1]).all())
    torch.cuda.synchronize()
    dt = time.time() - t0
    _lib.load().cs_l2_topk_stats(st, 0)
    terms = 2 if d == 512 else 3
    fl = 2.0 * args.n * args.n * d
    # spot check against the exact slab path (f64 distances of every pair) on 64 queries
    os.environ["CS_TOPK_MFMA"] = "0"
    sample = torch.arange(0, args.n, args.n // 64, device=dev)[:64]
    ref = B.l2_topk(q[sample].contiguous(), x, 10)
    del os.environ["CS_TOPK_MFMA"]
    out["d%d" % d] = {"queries": args.n, "catalog": args.n, "k": 10, "slabs": (args.n + args.slab - 1) // args.slab,
                      "seconds": dt, "algorithmic_tflops": fl / dt / 1e12,
                      "executed_f16_mfma_tflops": terms * fl / dt / 1e12,
                      "frac_of_f16_mfma_peak_algorithmic": fl / dt / 1e12 / 2516.6,
                      "frac_of_f16_mfma_peak_executed": terms * fl / dt / 1e12 / 2516.6,
                      "f16_shortlist_queries": int(st[0]), "recomputed_by_f64_path": int(st[1]),
                      "sorted": ok_sorted, "ids_in_range": bool((idx >= 0).all() and (idx < args.n).all()),
                      "spot_check_64_queries_equal_exact_path": bool(torch.equal(idx[sample], ref)),
                      "mean_nearest_distance": float(first.mean())}
    del x, q, idx
    _lib.load().cs_pool_trim()
    torch.cuda.empty_cache()
print(json.dumps(out, indent=1))
