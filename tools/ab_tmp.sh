set -e
cd /root/repo
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_rnn_gpu.py tests/test_polish_gpu.py -x -q -m gpu -k "p2 or gru" > gpurun_out/gru_test.log 2>&1
{
for B in 64 1000 4096; do
  PV_GRU_ROWS=16 timeout -k 10 300 python tools/bench_gru.py $B 2
  PV_GRU_ROWS=32 timeout -k 10 300 python tools/bench_gru.py $B 2
done
PV_GRU_ROWS=16 timeout -k 10 300 python tools/bench_gru.py 8192 2
timeout -k 10 300 python tools/bench_gru.py 8192 2
} > gpurun_out/gru_bench.log 2>&1
