set -e
cd /root/repo
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/full_gpu_tests.log 2>&1
PV_BENCH_OVERLAP=1 timeout -k 10 600 python bench.py --no-cpu-baseline --no-p2 --no-bf16 > gpurun_out/bench_ov1.json 2> gpurun_out/bench_ov1.err
PV_BENCH_OVERLAP=0 timeout -k 10 600 python bench.py --no-cpu-baseline --no-p2 --no-bf16 > gpurun_out/bench_ov0.json 2> gpurun_out/bench_ov0.err
