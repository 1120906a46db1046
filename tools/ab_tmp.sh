set -e
cd /root/repo
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_rnn_gpu.py -x -q -m gpu -k p2 > gpurun_out/gru_test.log 2>&1
timeout -k 10 300 python tools/bench_gru.py 64 > gpurun_out/gru_bench.log 2>&1
timeout -k 10 300 python tools/bench_gru.py 1000 >> gpurun_out/gru_bench.log 2>&1
timeout -k 10 300 python tools/bench_gru.py 8192 2 >> gpurun_out/gru_bench.log 2>&1
