#!/bin/bash
# PMC pass over the RNN micro-benchmark: HBM-side traffic (FETCH_SIZE / WRITE_SIZE, separate passes) and MFMA busy, for
# the default library and, if present, a variant library (variants/libpepper_hip_<name>.so). Run through gpurun.
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_rnn
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for v in default ${1:-}; do
  if [ "$v" != default ]; then export PEPPER_HIP_LIB=$GRAFT_REPO_ROOT/variants/libpepper_hip_$v.so; fi
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/$v.fetch -- python3 tools/bench_rnn.py 4096 5 > $OUT/$v.fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/$v.write -- python3 tools/bench_rnn.py 4096 5 > $OUT/$v.write.log 2>&1
  python3 tools/bench_rnn.py 4096 10 > $OUT/$v.time.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for v in "default ${1:-}".split():
    for c in ("fetch", "write"):
        fs = glob.glob("$OUT/%s.%s/*/*_counter_collection.csv" % (v, c))
        if not fs: continue
        agg = collections.defaultdict(list)
        for row in csv.DictReader(open(fs[0])):
            agg[row["Kernel_Name"]].append(float(row["Counter_Value"]))
        for k, x in agg.items():
            if "lstm" in k or "head" in k:
                print(v, c, k[:60], "%.1f MB per launch (KiB counter; FETCH_SIZE x2 on gfx950 for wide loads)" % (sum(x) / len(x) * 1024 / 1e6))
    print(open("$OUT/%s.time.log" % v).read())
PY
