"""Distil the three rocprofv3 --pmc passes of tools/pmc_collect.sh into <outdir>/pmc.json (copied to
profiles/pmc_<workload>.json, which bench.py reads into `roofline.traffic` / `mfma_busy` / ...) and a text table of
every library kernel.  Corrections as MI355X_MICROARCH.md prescribes: FETCH_SIZE / WRITE_SIZE are KiB; on gfx950
FETCH_SIZE tallies 16-B/lane streaming reads at half their bytes, so kernels that stream through LDS-DMA or
dwordx4 loads (listed in WIDE) get FETCH doubled; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles,
SQ_VALU_MFMA_BUSY_CYCLES cycles; utilisation = busy / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)."""
import collections, csv, glob, json, sys

out = sys.argv[1]
workload = "chair"
if "--workload" in sys.argv:
    workload = sys.argv[sys.argv.index("--workload") + 1]
WIDE = ("k_ransac_prefilter", "k_knn_f16", "k_topk_f16", "k_conv_mfma", "k_conv_lacc")
DOMINANT = {"chair": "k_ransac_prefilter", "table": "k_ransac_prefilter", "stress": "k_conv_mfma"}[workload]


def load(tag):
    f = glob.glob(f"{out}/{tag}/*/*counter_collection.csv")
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    if not f:
        return agg, disp
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
    return agg, disp


fa, fd = load("FETCH_SIZE")
wa, wd = load("WRITE_SIZE")
sa, sd = load("SQ_VALU_MFMA_BUSY_CYCLES")
rows = []
for k in sorted(set(fa) | set(sa)):
    if "cs::" not in k and "_ZN2cs" not in k:
        continue
    short = k.split("(")[0].replace("void ", "")[:44]
    n = max(len(fd.get(k, ())), 1)
    fetch = fa[k]["FETCH_SIZE"] / n * 1024.0
    write = wa[k]["WRITE_SIZE"] / max(len(wd.get(k, ())), 1) * 1024.0 if k in wa else 0.0
    wide = any(wn in k for wn in WIDE)
    hbm = (2.0 if wide else 1.0) * fetch + write
    s = sa.get(k, {})
    ns = max(len(sd.get(k, ())), 1)
    gui = s.get("GRBM_GUI_ACTIVE", 0.0)
    simd_cycles = 1024.0 * gui / 8.0
    mfma = s.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / simd_cycles if simd_cycles else 0.0
    valu = 4.0 * s.get("SQ_ACTIVE_INST_VALU", 0.0) / simd_cycles if simd_cycles else 0.0
    wave = s.get("SQ_WAVE_CYCLES", 0.0)
    wait_any = s.get("SQ_WAIT_ANY", 0.0) / wave if wave else 0.0
    wait_inst = s.get("SQ_WAIT_INST_ANY", 0.0) / wave if wave else 0.0
    rows.append(dict(kernel=short, full=k, launches=n, hbm_bytes_per_launch=hbm, fetch_bytes_raw=fetch, write_bytes=write,
                     fetch_doubled=wide, mfma_busy=mfma, valu_active=valu, wait_any=wait_any, wait_inst=wait_inst,
                     kcycles_per_launch=gui / ns / 8.0 / 1e3))
rows.sort(key=lambda r: -r["hbm_bytes_per_launch"] * r["launches"])
with open(f"{out}/pmc_table.txt", "w") as f:
    f.write("# tools/pmc_collect.sh: rocprofv3 --kernel-trace --pmc {FETCH_SIZE | WRITE_SIZE | SQ set} -- python3 bench.py "
            + " ".join(sys.argv[2:]) + " --steps 3 --warmup 1 --no-cpu-baseline --no-overlap-probe --no-solo-probe\n")
    f.write("# hbm = (2x for 16-B/lane streaming kernels) FETCH_SIZE + WRITE_SIZE per launch; mfma_busy / valu_active = share of SIMD cycles;\n"
            "# wait_any / wait_inst = share of wave cycles parked (waitcnt, barrier) / stalled at issue\n")
    f.write(f"{'kernel':46s} {'launches':>8s} {'hbm MB/launch':>14s} {'mfma_busy':>9s} {'valu':>6s} {'wait_any':>8s} {'wait_inst':>9s} {'kcycles':>8s}\n")
    for r in rows:
        f.write(f"{r['kernel']:46s} {r['launches']:8d} {r['hbm_bytes_per_launch'] / 1e6:14.2f} {r['mfma_busy']:9.3f} "
                f"{r['valu_active']:6.3f} {r['wait_any']:8.3f} {r['wait_inst']:9.3f} {r['kcycles_per_launch']:8.0f}\n")
dom = [r for r in rows if DOMINANT in r["full"]]
if dom:
    # several template instances of one family: weight by launches
    n = sum(r["launches"] for r in dom)
    agg = {k: sum(r[k] * r["launches"] for r in dom) / n for k in
           ("hbm_bytes_per_launch", "mfma_busy", "valu_active", "wait_any", "wait_inst")}
    js = {"kernel": DOMINANT, "launches": n, **agg, "fetch_doubled": dom[0]["fetch_doubled"],
          "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_* (three separate passes, tools/pmc_collect.sh) on "
                    "`python bench.py " + " ".join(sys.argv[2:]) + " --steps 3 --warmup 1 --no-cpu-baseline "
                    "--no-overlap-probe --no-solo-probe`"}
    json.dump(js, open(f"{out}/pmc.json", "w"), indent=1)
    print(json.dumps(js))
print(open(f"{out}/pmc_table.txt").read())
