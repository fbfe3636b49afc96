"""GPU idle gaps from a rocprofv3 --kernel-trace csv: total idle, and the gaps attributed to the kernel
that ran before each gap (what the host was doing after that kernel)."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5   # analyse the last fraction of the run
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
cut = t1 - (t1 - t0) * frac
anchor = sys.argv[3] if len(sys.argv) > 3 else None
if anchor:  # window = from the launch number int(frac) of the anchor kernel to its last launch
    a = [r for r in rows if anchor in r[2]]
    cut, t1 = a[int(frac)][0], a[-1][1]
rows = [r for r in rows if cut <= r[0] <= t1]
busy_end = rows[0][0]
idle = 0
by = collections.Counter(); cnt = collections.Counter()
prev = rows[0][2]
for s, e, n in rows:
    if s > busy_end:
        g = s - busy_end
        idle += g
        key = prev.split("(")[0][-40:] + " -> " + n.split("(")[0][-40:]
        by[key] += g; cnt[key] += 1
    if e > busy_end:
        busy_end = e; prev = n
span = rows[-1][1] - rows[0][0]
print("span ms %.1f idle ms %.1f (%.1f%%)" % (span / 1e6, idle / 1e6, 100 * idle / span))
for k, v in by.most_common(25):
    print("%8.2f ms  n=%4d  avg %6.1f us  %s" % (v / 1e6, cnt[k], v / cnt[k] / 1e3, k))
