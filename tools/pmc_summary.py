"""Summarise a rocprofv3 --pmc counter_collection CSV per kernel: sum and per-launch mean."""
import collections, csv, glob, sys
d = sys.argv[1]
f = glob.glob(d + "/*/*counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0][:48]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    disp[k].add(r["Dispatch_Id"])
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1].values())):
    if not k.startswith(("cs::", "void cs::", "_ZN2cs")):
        continue
    n = len(disp[k])
    print(f"{k:50s} launches={n:5d} " + " ".join(f"{c}={x:.4g} (per launch {x/n:.4g})" for c, x in v.items()))
