"""Print the top kernels of a rocprofv3 --stats kernel_stats.csv."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0] if not sys.argv[1].endswith(".csv") else sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
for r in list(csv.DictReader(open(f)))[:n]:
    print(f"{r['Name'][:72]:72s} calls={r['Calls']:>5s} total_ms={float(r['TotalDurationNs'])/1e6:8.2f} avg_us={float(r['AverageNs'])/1e3:9.1f} {r['Percentage']}%")
