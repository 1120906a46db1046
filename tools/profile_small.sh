#!/bin/bash
# rocprofv3 kernel-trace stats of the small-batch forms (run through gpurun): one caller's P1 calls (k_lstm_split) and the
# P2 sliding loop at 64 / 1000 chunks (k_gru_us). Usage: tools/profile_small.sh <tag>
set -e
TAG=${1:-r02b}
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_small_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p1 -- python3 tools/bench_single.py > $OUT/p1_single.json 2> $OUT/p1_single.err || echo "p1 trace failed"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p2_64 -- python3 tools/bench_gru.py 64 5 > $OUT/p2_b64.log 2> $OUT/p2_b64.err || echo "p2 64 trace failed"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p2_1000 -- python3 tools/bench_gru.py 1000 5 > $OUT/p2_b1000.log 2> $OUT/p2_b1000.err || echo "p2 1000 trace failed"
find $OUT -name "*kernel_stats.csv" | head
