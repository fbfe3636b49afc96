"""Analyse the k_ransac_prefilter block trace (CS_PF_TRACE=<it0> CS_PF_TRACE_FILE=...): per-wave start /
loop start / end (100 MHz wall clock) and hardware placement."""
import sys
import numpy as np
t = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 4, 4)  # block, wave, field
bid = np.arange(len(t))
keep = t[:, 0, 2] > 0
print("xcc == block id % 8 for", (((t[keep, 0, 3] >> np.uint64(20)) & np.uint64(0xf)).astype(np.int64) == (bid[keep] & 7)).mean() * 100, "% of blocks")
t = t[keep]
w0 = t[:, 0, :]
t0 = w0[:, 0].min()
start, end = (w0[:, i].astype(np.int64) - int(t0) for i in (0, 2))
wd = (t[:, :, 1] >> np.uint64(32)).astype(np.int64); wb = (t[:, :, 1] & np.uint64(0xffffffff)).astype(np.int64)
life_w = (t[:, :, 2].astype(np.int64) - t[:, :, 0].astype(np.int64))
print("per wave: lifetime us %.1f, waiting for own DMA %.1f us (%.0f%%), at barrier %.1f us (%.0f%%)" % (
    life_w.mean() / 100, wd.mean() / 100, 100 * wd.sum() / life_w.sum(), wb.mean() / 100, 100 * wb.sum() / life_w.sum()))
loop = start
print("blocks", len(t), "span us", (end.max()) / 100.0)
print("block lifetime us: mean %.1f min %.1f max %.1f" % ((end - start).mean() / 100, (end - start).min() / 100, (end - start).max() / 100))
print("prologue us: mean %.2f max %.2f" % ((loop - start).mean() / 100, (loop - start).max() / 100))
hw = w0[:, 3]
xcc = ((hw >> np.uint64(20)) & np.uint64(0xf)).astype(np.int64)
hwid = (hw & np.uint64(0xfffff)).astype(np.int64)
cyc = (t[:, :, 3] >> np.uint64(24)).astype(np.float64)
ghz = cyc / (life_w / 100.0) / 1e3
print("in-kernel clock (s_memtime / s_memrealtime): median %.3f GHz, p10 %.3f, p90 %.3f" % (
    np.median(ghz), np.percentile(ghz, 10), np.percentile(ghz, 90)))
cu = (hwid >> 8) & 0xf; sh = (hwid >> 12) & 1; se = (hwid >> 13) & 0x7; simd = (hwid >> 4) & 0x3
key = xcc * 10000 + se * 100 + sh * 50 + cu
print("distinct CUs", len(np.unique(key)), "xcc counts", np.bincount(xcc))
# concurrency over time
ev = np.concatenate([np.stack([start, np.ones_like(start)], 1), np.stack([end, -np.ones_like(end)], 1)])
ev = ev[np.argsort(ev[:, 0], kind="stable")]
conc = np.cumsum(ev[:, 1])
T = ev[:, 0]
dur = np.diff(T)
print("time-weighted mean concurrent blocks %.1f" % ((conc[:-1] * dur).sum() / dur.sum()))
for frac in (0.25, 0.5, 0.75, 0.9, 0.95, 1.0):
    print("  finished by %.0f%% of span: %d blocks" % (frac * 100, (end <= frac * end.max()).sum()))
# per-CU busy
life = {}
for k, s_, e_ in zip(key, start, end):
    life.setdefault(k, []).append((s_, e_))
print("blocks per CU: min %d max %d" % (min(len(v) for v in life.values()), max(len(v) for v in life.values())))
