"""Chronological list of the GPU idle gaps (all streams merged) inside the last bench step of a rocprofv3 kernel trace.
usage: step_gaps.py <rocprof dir> [min_gap_us] [steps_back]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
ming = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 30e3
back = int(sys.argv[3]) if len(sys.argv) > 3 else 1
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-44:]) for r in csv.DictReader(open(f)))
ch = [r for r in rows if "k_chamfer_mfma" in r[2]]
t0, t1 = ch[-1 - back][1], ch[-back][1]
seg = [r for r in rows if t0 <= r[0] < t1]
print("step span %.2f ms, %d launches" % ((t1 - t0) / 1e6, len(seg)))
cur = t0; prev = "(previous step's chamfer)"; idle = 0; small = 0; nsmall = 0
for s, e, n in seg:
    if s > cur:
        g = s - cur; idle += g
        if g >= ming: print("  +%7.3f ms  gap %7.1f us   %s -> %s" % ((cur - t0) / 1e6, g / 1e3, prev.strip(), n.strip()))
        else: small += g; nsmall += 1
    if e > cur: cur = e; prev = n
print("idle %.2f ms of which %d gaps below %.0f us: %.2f ms" % (idle / 1e6, nsmall, ming / 1e3, small / 1e6))
