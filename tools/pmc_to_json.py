"""Distil the three rocprofv3 --pmc passes of tools/pmc_collect.sh into <outdir>/pmc.json (copied to
profiles/pmc_<workload>.json, which bench.py reads into `roofline.traffic` / `mfma_busy` and `roofline_by_kernel`)
and a text table of every library kernel.  Corrections as MI355X_MICROARCH.md prescribes: FETCH_SIZE / WRITE_SIZE are
KiB; on gfx950 FETCH_SIZE tallies 16-B/lane streaming reads at half their bytes, so kernels that stream through
LDS-DMA or dwordx4 loads (listed in WIDE) get FETCH doubled; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are
quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles; utilisation = busy / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs).
`valu_active` (round 5): the share of SIMD cycles the VECTOR ALU is busy = 2 cycles per wave64 instruction
(the SIMD-32 datapath, MI355X_MICROARCH.md cycle table) x SQ_ACTIVE_INST_VALU (one quad-cycle of issue per instruction =
the instruction count) / SIMD cycles.  Rounds 1-4 charged the 4 ISSUE cycles of an instruction instead: the waves of a SIMD
overlap their issue slots, so that figure exceeded 1 (1.08 on the table workload) and could not be added to `mfma_busy`.

The file carries `csrc_sha` (hash of corsair_amd/csrc sources at collection time): bench.py attaches the counters
only while the kernels are unchanged, otherwise it prints traffic: null and counters: "stale" (ADVICE r2)."""
import collections, csv, glob, hashlib, json, os, sys

out = sys.argv[1]
workload = "chair"
if "--workload" in sys.argv:
    workload = sys.argv[sys.argv.index("--workload") + 1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WIDE = ("k_ransac_prefilter", "k_knn_f16", "k_topk_f16", "k_conv_mfma", "k_conv_dma", "k_chamfer_f16")
DOMINANT = {"chair": "k_ransac_prefilter", "table": "k_ransac_prefilter", "stress": "k_conv_dma"}[workload]
# the FIRST-stage prefilter instantiation (<1, true> by default); the second stage launches <2, false> on a few survivors
DOM_MATCH = {"k_ransac_prefilter": ("k_ransac_prefilterILi1", "k_ransac_prefilter<1")}.get(DOMINANT, (DOMINANT,))
# kernel family of bench.py's `kernel_ms` -> (kernels that belong to it, kernels of which ONE launch = one library call)
FAMILY = {
    "conv": (("k_conv_dma", "k_conv_mfma", "k_conv_stem", "k_conv_generic"), ("k_conv_dma", "k_conv_mfma", "k_conv_stem", "k_conv_generic")),
    # one "kmap" profile scope per kernel map (its build kernel) + one per batch for the common tiling-order pass
    # (round 5: one "kmap" scope for the coordinate pyramid, one for the level kernels, one for the tiling order of a batch)
    "kmap": (("k_level_maps", "k_build_nbr", "k_row_keys", "k_order_finish", "k_pyr_", "k_insert", "k_emit_strided", "k_flag_first",
              "k_segments", "k_fill_table"),
             ("k_pyr_insert", "k_order_finish")),
    "knn": (("k_knn_f16", "k_knn_rescore_f16", "k_knf_pack", "k_knn_feat"), ("k_knf_pack_queries",)),
    "chamfer": (("k_chamfer",), ("k_chamfer_finish",)),
    "topk": (("k_topk", "k_tkf", "k_dist_matrix", "k_row_topk"), ("k_topk_finish",)),
    "ransac_pre": (("k_ransac_prefilterILi1", "k_ransac_prefilter<1"), ("k_ransac_prefilterILi1", "k_ransac_prefilter<1")),
    "ransac_hyp": (("k_ransac_hyp<", "k_ransac_hypILi"), ("k_ransac_hyp<", "k_ransac_hypILi")),
    "ransac_eval": (("k_ransac_count", "k_ransac_err", "k_ransac_scan", "k_ransac_survivors", "k_ransac_stage2",
                     "k_ransac_prefilterILi2", "k_ransac_prefilter<2"), ("k_ransac_scan1",)),
    "symcut": (("k_symcut",), ("k_symcut_kmeans",)),
}


def csrc_sha():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "corsair_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            with open(os.path.join(d, f), "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


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
    valu = 2.0 * s.get("SQ_ACTIVE_INST_VALU", 0.0) / simd_cycles if simd_cycles else 0.0
    wave = s.get("SQ_WAVE_CYCLES", 0.0)
    wait_any = s.get("SQ_WAIT_ANY", 0.0) / wave if wave else 0.0
    wait_inst = s.get("SQ_WAIT_INST_ANY", 0.0) / wave if wave else 0.0
    rows.append(dict(kernel=short, full=k, launches=n, hbm_bytes_per_launch=hbm, fetch_bytes_raw=fetch, write_bytes=write,
                     fetch_doubled=wide, mfma_busy=mfma, valu_active=valu, wait_any=wait_any, wait_inst=wait_inst,
                     kcycles_per_launch=gui / ns / 8.0 / 1e3))
rows.sort(key=lambda r: -r["hbm_bytes_per_launch"] * r["launches"])
with open(f"{out}/pmc_table.txt", "w") as f:
    f.write("# tools/pmc_collect.sh: rocprofv3 --kernel-trace --pmc {FETCH_SIZE | WRITE_SIZE | SQ set} -- python3 bench.py "
            + " ".join(sys.argv[2:]) + " --steps 3 --warmup 1 --no-cpu-baseline --no-overlap-probe --no-solo-probe --no-extra-workloads\n")
    f.write("# hbm = (2x for 16-B/lane streaming kernels) FETCH_SIZE + WRITE_SIZE per launch; mfma_busy = matrix pipe, valu = vector ALU (2 cycles per wave64 instruction): shares of SIMD cycles;\n"
            "# wait_any / wait_inst = share of wave cycles parked (waitcnt, barrier) / stalled at issue\n")
    f.write(f"{'kernel':46s} {'launches':>8s} {'hbm MB/launch':>14s} {'mfma_busy':>9s} {'valu':>6s} {'wait_any':>8s} {'wait_inst':>9s} {'kcycles':>8s}\n")
    for r in rows:
        f.write(f"{r['kernel']:46s} {r['launches']:8d} {r['hbm_bytes_per_launch'] / 1e6:14.2f} {r['mfma_busy']:9.3f} "
                f"{r['valu_active']:6.3f} {r['wait_any']:8.3f} {r['wait_inst']:9.3f} {r['kcycles_per_launch']:8.0f}\n")


def weighted(sel, keys):
    n = sum(r["launches"] for r in sel)
    cyc = sum(r["kcycles_per_launch"] * r["launches"] for r in sel) or 1.0
    agg = {"hbm_bytes_per_launch": sum(r["hbm_bytes_per_launch"] * r["launches"] for r in sel) / max(n, 1)}
    for k in keys:   # busy fractions: weighted by the time the kernels ran
        agg[k] = sum(r[k] * r["kcycles_per_launch"] * r["launches"] for r in sel) / cyc
    return n, agg


dom = [r for r in rows if any(m in r["full"] for m in DOM_MATCH)]
families = {}
for fam, (members, anchors) in FAMILY.items():
    sel = [r for r in rows if any(m in r["full"] for m in members)]
    calls = sum(r["launches"] for r in rows if any(a in r["full"] for a in anchors))
    if not sel or not calls:
        continue
    total_bytes = sum(r["hbm_bytes_per_launch"] * r["launches"] for r in sel)
    _, agg = weighted(sel, ("mfma_busy", "valu_active", "wait_any"))
    families[fam] = {"hbm_bytes_per_call": total_bytes / calls, "calls_profiled": calls,
                     "mfma_busy": agg["mfma_busy"], "valu_active": agg["valu_active"], "wait_any": agg["wait_any"],
                     "kernels": sorted({r["kernel"].split("<")[0] for r in sel})}
if dom:
    n, agg = weighted(dom, ("mfma_busy", "valu_active", "wait_any", "wait_inst"))
    js = {"kernel": DOMINANT, "launches": n, **agg, "fetch_doubled": dom[0]["fetch_doubled"], "families": families,
          "csrc_sha": csrc_sha(),
          "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_* (three separate passes, tools/pmc_collect.sh) on "
                    "`python bench.py " + " ".join(sys.argv[2:]) + " --steps 3 --warmup 1 --no-cpu-baseline "
                    "--no-overlap-probe --no-solo-probe --no-extra-workloads`"}
    json.dump(js, open(f"{out}/pmc.json", "w"), indent=1)
    print(json.dumps(js))
print(open(f"{out}/pmc_table.txt").read())
