set -e
cd /root/repo
mkdir -p gpurun_out
{
echo "== tests (tail 16 default below 8192)"; timeout -k 10 900 python -m pytest tests/test_rnn_gpu.py tests/test_host_mirror_gpu.py tests/test_pipeline_gpu.py -x -q -m gpu 2>&1 | tail -3
echo "== tests tail 32"; PV_TAIL_ROWS=32 timeout -k 10 900 python -m pytest tests/test_rnn_gpu.py -x -q -m gpu -k "p1" 2>&1 | tail -3
for B in 512 4096; do
echo "== fp32 B=$B"; timeout -k 10 120 python tools/bench_rnn.py $B 10
done
echo "== bf16x3"; PV_BENCH_DTYPE=1 timeout -k 10 120 python tools/bench_rnn.py 4096 10
} > gpurun_out/ab_tail.log 2>&1
