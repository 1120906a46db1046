set -e
cd /root/repo
mkdir -p gpurun_out
{
echo "== tests"; timeout -k 10 900 python -m pytest tests/test_rnn_gpu.py tests/test_pipeline_gpu.py -x -q -m gpu 2>&1 | tail -3
echo "== fp32"; timeout -k 10 120 python tools/bench_rnn.py 4096 10
echo "== bf16x3"; PV_BENCH_DTYPE=1 timeout -k 10 120 python tools/bench_rnn.py 4096 10
} > gpurun_out/ab_h0.log 2>&1
