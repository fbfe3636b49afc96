"""A/B of one environment switch of cs_l2_topk_catalog on the stress shapes (10^6-row catalog, slabs of queries):
milliseconds per call, alternating, results compared.

  python tools/topk_ab.py [--var CS_TOPK_XCD] [--values 1 0] [--dims 256] [--queries 10240 65536] [--catalog 1000000]
"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from corsair_amd import _lib, backend as B, synth

ap = argparse.ArgumentParser()
ap.add_argument("--var", default="CS_TOPK_XCD")
ap.add_argument("--values", nargs="+", default=["1", "0"])
ap.add_argument("--dims", type=int, nargs="+", default=[256])
ap.add_argument("--queries", type=int, nargs="+", default=[10240, 65536])
ap.add_argument("--catalog", type=int, default=1000000)
ap.add_argument("--reps", type=int, default=5)
args = ap.parse_args()
dev = torch.device("cuda:0")
_lib.require_gpu()
for d in args.dims:
    x = torch.empty((args.catalog, d), dtype=torch.float32, device=dev)
    for i, s in enumerate(range(0, args.catalog, 131072)):
        m = min(131072, args.catalog - s)
        x[s:s + m] = torch.from_numpy(synth.make_descriptors(m, d, seed=4321 + i)).to(dev)
    cat = B.TopkCatalog(x)
    for nq in args.queries:
        q = torch.from_numpy(synth.make_descriptors(nq, d, seed=99)).to(dev)
        ref = None
        for rnd in range(2):
            for v in args.values:
                os.environ[args.var] = v
                B.l2_topk(q, cat, 10)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(args.reps):
                    idx, dist = B.l2_topk(q, cat, 10, True)
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) / args.reps * 1e3
                same = True if ref is None else bool(torch.equal(idx, ref[0]) and torch.equal(dist, ref[1]))
                if ref is None:
                    ref = (idx.clone(), dist.clone())
                tf = 2.0 * nq * args.catalog * d / ms / 1e9
                print(f"d={d} nq={nq} {args.var}={v}: {ms:8.3f} ms per call, {tf:6.1f} TFLOP/s algorithmic, identical={same}", flush=True)
    del cat, x
