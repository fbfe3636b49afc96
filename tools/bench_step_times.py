"""Per-work-item wall times of bench.py's sequential pass: python tools/bench_step_times.py [--steps K --warmup W --embed-group N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

args = bench.parse()
ctx = bench.Ctx(args)
wl = bench.RegistrationWorkload(ctx, args.workload)
wl.setup()
g = max(1, args.embed_group)
n = args.warmup + args.steps
times = []
if os.environ.get("BENCH_STEP_PROF"):
    from corsair_amd import _lib
    _lib.prof_enable(True)
    _lib.prof_reset()
for b in range(0, n, g):
    item = list(range(b, min(b + g, n)))
    torch.cuda.synchronize(); t0 = time.time()
    if len(item) > 1:
        sets = wl.embed_steps(item)
        torch.cuda.synchronize(); t1 = time.time()
        for bb, qs in sets.items():
            wl.register_step(bb, qs)
        torch.cuda.synchronize(); t2 = time.time()
        times.append((item[0], len(item), (t1 - t0) * 1e3, (t2 - t1) * 1e3))
    else:
        wl.step(item[0])
        torch.cuda.synchronize(); t2 = time.time()
        times.append((item[0], 1, 0.0, (t2 - t0) * 1e3))
for b, k, te, tr in times:
    print("steps %2d..%2d  embed %7.2f ms  rest %7.2f ms  per step %6.2f ms" % (b, b + k - 1, te, tr, (te + tr) / k))
print("stats", torch.cuda.memory_stats().get("num_alloc_retries"), "device mallocs", torch.cuda.memory_stats().get("num_device_alloc"), "reserved GB", torch.cuda.memory_reserved() / 1e9)
