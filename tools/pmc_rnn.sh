#!/bin/bash
# PMC pass over the RNN micro-benchmark for two decoder variants (run through gpurun)
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_rnn
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for v in -1 8; do
  export PV_DEC_STAGGER=$v
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/v$v -- python3 tools/bench_rnn.py 4096 5 > $OUT/v$v.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for v in ("-1","8"):
    f = glob.glob("$OUT/v%s/*/*_counter_collection.csv" % v)[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        agg[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        dur[row["Kernel_Name"]].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    for k, c in agg.items():
        if "lstm" in k:
            d = sum(dur[k]) / len(dur[k])
            m = {n: sum(x)/len(x) for n, x in c.items()}
            clk = m["GRBM_GUI_ACTIVE"] / 8 / d
            print(v, k[:50], "dur_us %.0f clk_GHz %.3f mfma_busy_frac %.3f wait_any %.3f active %.3f" % (d/1e3, clk, m["SQ_VALU_MFMA_BUSY_CYCLES"]/(4*m["SQ_BUSY_CU_CYCLES"]), m["SQ_WAIT_ANY"]/m["SQ_WAVE_CYCLES"], m["SQ_ACTIVE_INST_ANY"]/m["SQ_WAVE_CYCLES"]))
PY
