"""List the launches of one kernel from a rocprofv3 --kernel-trace csv (duration, grid) in order."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
pat = sys.argv[2]
rows = [r for r in csv.DictReader(open(f)) if pat in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for i, r in enumerate(rows):
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print(i, "us %.1f" % d, "grid", r["Grid_Size_X"], r["Grid_Size_Y"], "wg", r["Workgroup_Size_X"])
