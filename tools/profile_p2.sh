#!/bin/bash
# Profiles the polisher (P2) chain on the GPU box (run through gpurun): kernel trace + stats of the builder -> bi-GRU chain
# and of the bi-GRU at a chip-filling batch, and an MFMA-busy PMC pass. Usage: tools/profile_p2.sh <tag>
set -e
TAG=${1:-r01}
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_p2_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/chain -- python3 tools/bench_polish.py 8 > $OUT/chain.log 2> $OUT/chain.err || echo "chain trace failed"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/gru -- python3 tools/bench_gru.py 8192 3 > $OUT/gru.log 2> $OUT/gru.err || echo "gru trace failed"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc -- python3 tools/bench_gru.py 8192 2 > $OUT/pmc.log 2> $OUT/pmc.err || echo "pmc failed"
python3 - <<PY
import csv, glob, collections, shutil, os
out = "$OUT"; tag = "$TAG"; dst = os.path.join("$GRAFT_REPO_ROOT", "gpurun_out", "p2_summary_" + tag)
os.makedirs(dst, exist_ok=True)
for name in ("chain", "gru"):
    fs = glob.glob(os.path.join(out, name, "*", "*_kernel_stats.csv"))
    if fs: shutil.copy(fs[0], os.path.join(dst, "%s_p2_%s_kernel_stats.csv" % (tag, name)))
    lg = os.path.join(out, name + ".log")
    if os.path.exists(lg): shutil.copy(lg, os.path.join(dst, "%s_p2_%s.log" % (tag, name)))
fs = glob.glob(os.path.join(out, "pmc", "*", "*_counter_collection.csv"))
if fs:
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(fs[0])):
        agg[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    with open(os.path.join(dst, "%s_p2_pmc.csv" % tag), "w") as f:
        f.write("kernel,SQ_VALU_MFMA_BUSY_CYCLES,SQ_BUSY_CU_CYCLES,GRBM_GUI_ACTIVE,mfma_util=MFMA_BUSY/(4*SQ_BUSY_CU_CYCLES)\n")
        for k, c in agg.items():
            m = {n: sum(x) / len(x) for n, x in c.items()}
            if m.get("SQ_BUSY_CU_CYCLES", 0) > 0:
                f.write('"%s",%.0f,%.0f,%.0f,%.3f\n' % (k, m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0), m["SQ_BUSY_CU_CYCLES"], m.get("GRBM_GUI_ACTIVE", 0), m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (4 * m["SQ_BUSY_CU_CYCLES"])))
PY
ls $GRAFT_REPO_ROOT/gpurun_out/p2_summary_$TAG
