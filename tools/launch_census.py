"""Per-step launch census from the two rocprofv3 --stats runs of tools/launch_census.sh."""
import csv, sys
out = sys.argv[1]
def load(n):
    return {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(f"{out}/kernel_stats_{n}.csv"))}
a, b = load(2), load(10)
rows = []
for k in b:
    c0, t0 = a.get(k, (0, 0.0))
    c1, t1 = b[k]
    rows.append(((c1 - c0) / 8.0, (t1 - t0) / 8.0 / 1e6, k))
rows.sort(reverse=True)
tot_c = sum(r[0] for r in rows); tot_t = sum(r[1] for r in rows)
glue = sum(r[1] for r in rows if not ("cs::" in r[2] or "_ZN2cs" in r[2]))
with open(f"{out}/census.txt", "w") as f:
    f.write(f"# launches per timed step {tot_c:.1f}, GPU ms per step {tot_t:.2f}; not the library's kernels (memset / copy / torch / rocprim): "
            f"{sum(r[0] for r in rows if not ('cs::' in r[2] or '_ZN2cs' in r[2])):.1f} launches, {glue:.2f} ms = {100 * glue / tot_t:.1f} %\n")
    for c, t, k in rows:
        if c > 0:
            f.write(f"{c:8.1f} {t:8.3f} ms  {k[:120]}\n")
print(open(f"{out}/census.txt").read())
