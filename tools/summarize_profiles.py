"""Numbers DESIGN.md section 7 / profiles/README.md quote, from one measurement set: python tools/summarize_profiles.py <dir> <tag>"""
import csv, json, sys
o, tag = sys.argv[1].rstrip("/") + "/", sys.argv[2]
d = json.load(open(o + tag + "_default_line.json"))
r = d["roofline"]
print("DEFAULT: value %.1f (%s, %.2f ms/step)  sequential %.1f (%.2f ms)" % (d["value"], d["value_pass"], d["ms_per_step"],
      d["sequential"]["value"], d["sequential"]["ms_per_step"]))
print("  roofline %s frac %.4f executed %.4f avg %.4f ms achieved %.1f TF; solo frac %.4f avg %.4f ms; traffic %.1f MB mfma_busy %.3f valu %.3f" % (
      r["kernel"], r["frac"], r["frac_executed"], r["avg_launch_ms"], r["achieved"], r["solo"]["frac"], r["solo"]["avg_launch_ms"],
      (r.get("traffic") or 0) / 1e6, r.get("mfma_busy", 0), r.get("valu_active", 0)))
print("  per step event ms:", {k: round(v / d["steps"], 2) for k, v in d["kernel_ms"].items() if v > 0}, "conv frac", d["roofline_by_kernel"]["conv"].get("frac_of_matrix_peak"))
for k, v in d["workloads"].items():
    rr = v["roofline"]
    print("  leg %s: value %.1f (%s, %.2f ms)  sequential %.1f  %s frac %.4f avg %.4f ms traffic %.1f MB" % (k, v["value"], v["value_pass"], v["ms_per_step"],
          v["sequential"]["value"], rr["kernel"], rr["frac"], rr["avg_launch_ms"], (rr.get("traffic") or 0) / 1e6))
c = d["cpu_baseline"]
print("  cpu %.4f q/s on %d cores; %s" % (c["value"], c["cores"], c["sample"][-150:]))
for w in ("chair", "table", "stress"):
    l = json.load(open(o + "%s_%s_line.json" % (tag, w)))
    p = json.load(open(o + "%s_%s_profiled.json" % (tag, w))) if not __import__("os").path.exists(o + "%s_%s_line_profiled.json" % (tag, w)) else json.load(open(o + "%s_%s_line_profiled.json" % (tag, w)))
    rr = l["roofline"]
    name = {"chair": "k_ransac_prefilterILi1", "table": "k_ransac_prefilterILi1", "stress": "k_conv_dma"}[w]
    rows = [x for x in csv.DictReader(open(o + "%s_%s_kernel_stats.csv" % (tag, w))) if name in x["Name"]]
    calls = sum(int(x["Calls"]) for x in rows); tot = sum(float(x["TotalDurationNs"]) for x in rows)
    print("%s: value %.1f (%s, %.2f ms) sequential %.1f (%.2f ms) frac %.4f solo %s avg live %.4f ms; profiled run live %.4f ms vs rocprofv3 %.4f ms over %d launches; mfma %.3f traffic %.1f MB" % (
          w, l["value"], l["value_pass"], l["ms_per_step"], l["sequential"]["value"], l["sequential"]["ms_per_step"], rr["frac"],
          ("%.4f" % rr["solo"]["frac"]) if "solo" in rr else "-", rr["avg_launch_ms"], p["roofline"]["avg_launch_ms"], tot / calls / 1e6, calls,
          rr.get("mfma_busy", 0), (rr.get("traffic") or 0) / 1e6))
print(open(o + tag + "_chair_launch_census.txt").readline().strip())
print("csrc_sha", json.load(open(o + "pmc_chair.json"))["csrc_sha"])
