"""Aggregate the rocprofv3 --pmc passes of tools/conv_pmc.sh per convolution kernel instance."""
import collections, csv, glob, sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for f in glob.glob(f"{out}/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_conv" not in k:
            continue
        k = k.split("(")[0].replace("void cs::", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[(k, f)].add(r["Dispatch_Id"])
print(f"{'kernel':28s} {'n':>4s} {'mfma_busy':>9s} {'valu':>6s} {'wait_any':>8s} {'wait_inst':>9s} {'lds_busy':>8s} {'bank_cf':>8s} {'wait_lds':>8s} {'vmem_act':>8s} {'fetch MB':>9s} {'kcyc':>7s}")
for k, s in sorted(agg.items()):
    n = max(len(v) for (kk, f), v in disp.items() if kk == k)
    gui = s["GRBM_GUI_ACTIVE"] / 2.0   # collected in two passes
    simd = 1024.0 * gui / 8.0
    wave = s["SQ_WAVE_CYCLES"] or 1.0
    print(f"{k:28s} {n:4d} {s['SQ_VALU_MFMA_BUSY_CYCLES'] / simd:9.3f} {4 * s['SQ_ACTIVE_INST_VALU'] / simd:6.3f} "
          f"{s['SQ_WAIT_ANY'] / wave:8.3f} {s['SQ_WAIT_INST_ANY'] / wave:9.3f} {s['SQ_LDS_IDX_ACTIVE'] / (256 * gui / 8):8.3f} "
          f"{s['SQ_LDS_BANK_CONFLICT'] / max(s['SQ_LDS_IDX_ACTIVE'], 1):8.3f} {s['SQ_WAIT_INST_LDS'] / wave:8.3f} "
          f"{4 * s['SQ_ACTIVE_INST_VMEM'] / simd:8.3f} {s['FETCH_SIZE'] * 1024 / n / 1e6:9.1f} {gui / n / 8 / 1e3:7.0f}")
