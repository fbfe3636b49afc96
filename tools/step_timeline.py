"""Timeline of the last bench step in a rocprofv3 kernel trace: idle gaps and who runs when.
usage: step_timeline.py <rocprof dir> [bin_us]"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
binw = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 500e3
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-36:]) for r in csv.DictReader(open(f)))
fins = [r for r in rows if "k_ransac_finish" in r[2]]
# a step ends with its last k_ransac_finish (two RANSAC calls per step when split): use chamfer as the step marker
ch = [r for r in rows if "k_chamfer_mfma" in r[2]]
t0, t1 = ch[-2][1], ch[-1][1]
seg = [r for r in rows if t0 <= r[0] < t1]
print("step span %.2f ms, %d kernels" % ((t1 - t0) / 1e6, len(seg)))
cur = t0; idle = 0
for s, e, n in seg:
    if s > cur: idle += s - cur
    cur = max(cur, e)
print("idle %.2f ms" % (idle / 1e6))
nb = int((t1 - t0) / binw) + 1
for b in range(nb):
    lo, hi = t0 + b * binw, t0 + (b + 1) * binw
    by = collections.Counter(); busy = []
    for s, e, n in seg:
        o = min(e, hi) - max(s, lo)
        if o > 0:
            by[n] += o; busy.append((max(s, lo), min(e, hi)))
    busy.sort(); c = lo; u = 0
    for s, e in busy:
        if e > c: u += e - max(s, c); c = e
    top = ", ".join("%s %.0f%%" % (k.strip(), 100 * v / binw) for k, v in by.most_common(3))
    print("%6.2f ms  busy %3.0f%%  %s" % ((lo - t0) / 1e6, 100 * u / binw, top))
