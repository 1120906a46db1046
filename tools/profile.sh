#!/bin/bash
# Profiles bench.py on the GPU box (run through gpurun). Usage: tools_profile.sh <tag>
# 1) kernel trace + stats, 2) PMC passes for HBM traffic and MFMA busy (separate runs, no tracing mixed in).
set -e
TAG=${1:-r01}
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
# (the trace run keeps the configs[2] secondary, so that k_gemm_bf16x3 / k_rec_bf16 appear in the kernel stats next to the fp32 chain)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 128 --warmup 16 --no-cpu-baseline --no-p2 --no-h2d --no-filepath > $OUT/bench_trace.json 2> $OUT/bench_trace.err || echo "trace run failed"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -- python3 bench.py --steps 32 --warmup 16 --min-seconds 0.05 --no-cpu-baseline --no-p2 --no-bf16 --no-h2d --no-filepath > $OUT/bench_pmc1.json 2> $OUT/bench_pmc1.err || echo "pmc1 failed"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 32 --warmup 16 --min-seconds 0.05 --no-cpu-baseline --no-p2 --no-bf16 --no-h2d --no-filepath > $OUT/bench_pmc2.json 2> $OUT/bench_pmc2.err || echo "pmc2 failed"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 32 --warmup 16 --min-seconds 0.05 --no-cpu-baseline --no-p2 --no-bf16 --no-h2d --no-filepath > $OUT/bench_pmc3.json 2> $OUT/bench_pmc3.err || echo "pmc3 failed"
find $OUT -name "*.csv" | head -50
