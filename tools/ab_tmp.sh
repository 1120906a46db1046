set -e
cd /root/repo
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r01e
mkdir -p $OUT
rm -rf $OUT/trace
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 160 --warmup 16 --no-cpu-baseline --no-p2 --no-bf16 > $OUT/bench_trace.json 2> $OUT/bench_trace.err
