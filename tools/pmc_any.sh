#!/bin/bash
# Generic PMC passes over one python command (run through gpurun): tools/pmc_any.sh <tag> <kernel-name filter> <python script> [args...]
# Pass 1: SQ issue / wait / MFMA / LDS counters; pass 2: FETCH_SIZE; pass 3: WRITE_SIZE; pass 4: L2 hit / miss. Counter passes never
# carry --kernel-trace (gpurun refuses mixed runs). Prints per-kernel per-launch averages and writes gpurun_out/pmc_<tag>.csv
export TMPDIR=/tmp
TAG=$1; FILT=$2; shift 2
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/sq -- python3 "$@" > $OUT/sq.log 2>&1 || echo "sq pass failed"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 "$@" > $OUT/fetch.log 2>&1 || echo "fetch pass failed"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 "$@" > $OUT/write.log 2>&1 || echo "write pass failed"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/l2 -- python3 "$@" > $OUT/l2.log 2>&1 || echo "l2 pass failed"
python3 - <<PY
import csv, glob, collections
rows = collections.OrderedDict()
for d in ("sq", "fetch", "write", "l2"):
    for f in glob.glob("$OUT/%s/*/*_counter_collection.csv" % d):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            agg[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, v in agg.items():
            if "$FILT" and "$FILT" not in k: continue
            r = rows.setdefault(k.replace("(anonymous namespace)::", "")[:60], {})
            for c, x in v.items():
                r[c] = sum(x) / len(x); r["n"] = len(x)
cols = sorted({c for r in rows.values() for c in r})
with open("$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG.csv", "w") as f:
    w = csv.writer(f); w.writerow(["kernel"] + cols)
    for k, r in rows.items(): w.writerow([k] + ["%.0f" % r.get(c, 0) for c in cols])
for k, r in rows.items():
    print(k)
    for c in cols: print("   %-28s %16.0f" % (c, r.get(c, 0)))
    if r.get("SQ_BUSY_CU_CYCLES"): print("   mfma_util = MFMA_BUSY/(4*BUSY_CU)  %.3f" % (r.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (4 * r["SQ_BUSY_CU_CYCLES"])))
    if r.get("TCC_HIT_sum") is not None and r.get("TCC_HIT_sum", 0) + r.get("TCC_MISS_sum", 0) > 0: print("   l2 hit rate %.3f" % (r["TCC_HIT_sum"] / (r["TCC_HIT_sum"] + r["TCC_MISS_sum"])))
    if "FETCH_SIZE" in r: print("   HBM-side read  %.1f MB raw (x2 for wide loads: %.1f MB), write %.1f MB" % (r["FETCH_SIZE"] * 1024 / 1e6, r["FETCH_SIZE"] * 2048 / 1e6, r.get("WRITE_SIZE", 0) * 1024 / 1e6))
PY
