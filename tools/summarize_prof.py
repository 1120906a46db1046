#!/usr/bin/env python3
"""Condenses a tools/profile.sh output directory (gpurun_out/prof_<tag>) into profiles/<tag>_*.csv:
the rocprofv3 --kernel-trace --stats table and per-kernel PMC averages (HBM bytes corrected as
MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE are in KiB; gfx950 FETCH_SIZE reports 1/2 of
wide coalesced reads, so fetch bytes are given raw and doubled)."""
import collections
import csv
import glob
import os
import shutil
import sys

src = sys.argv[1]
tag = sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
if stats:
    shutil.copy(stats[0], os.path.join(dst, "%s_kernel_stats.csv" % tag))
for f in ("bench_trace.json",):
    p = os.path.join(src, f)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, "%s_%s" % (tag, f)))
rows = collections.OrderedDict()
for d in ("pmc_mfma", "pmc_fetch", "pmc_write"):
    fs = glob.glob(os.path.join(src, d, "*", "*_counter_collection.csv"))
    if not fs:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(fs[0])):
        agg[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in agg.items():
        if "k_" not in k:
            continue
        r = rows.setdefault(k, {})
        for c, x in v.items():
            r[c] = sum(x) / len(x)
            r["launches_" + c] = len(x)
with open(os.path.join(dst, "%s_pmc_per_launch.csv" % tag), "w") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAVE_CYCLES(quad)", "SQ_BUSY_CU_CYCLES", "GRBM_GUI_ACTIVE(sum 8 XCD)",
                "mfma_util=MFMA_BUSY/(4*SQ_BUSY_CU_CYCLES)", "FETCH_SIZE_KiB", "hbm_read_bytes_raw", "hbm_read_bytes_x2(gfx950 wide loads)",
                "WRITE_SIZE_KiB", "hbm_write_bytes"])
    for k, r in rows.items():
        mf, wc = r.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), r.get("SQ_WAVE_CYCLES", 0.0)
        fs_, ws = r.get("FETCH_SIZE", 0.0), r.get("WRITE_SIZE", 0.0)
        w.writerow([k.replace("(anonymous namespace)::", "")[:70], "%.0f" % mf, "%.0f" % wc, "%.0f" % r.get("SQ_BUSY_CU_CYCLES", 0),
                    "%.0f" % r.get("GRBM_GUI_ACTIVE", 0), "%.3f" % (mf / (4 * r.get("SQ_BUSY_CU_CYCLES", 0)) if r.get("SQ_BUSY_CU_CYCLES", 0) else 0), "%.1f" % fs_, "%.0f" % (fs_ * 1024),
                    "%.0f" % (fs_ * 2048), "%.1f" % ws, "%.0f" % (ws * 1024)])
print("wrote profiles/%s_*" % tag)
