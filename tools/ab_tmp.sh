set -e
cd /root/repo
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_polish_gpu.py -x -q -m gpu > gpurun_out/polish_test.log 2>&1
timeout -k 10 300 python tools/bench_polish.py 8 > gpurun_out/polish_bench.log 2>&1
