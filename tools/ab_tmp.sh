set -e
cd /root/repo
mkdir -p gpurun_out
{
echo "== tests"; timeout -k 10 900 python -m pytest tests/test_rnn_gpu.py -x -q -m gpu -k p1 2>&1 | tail -3
echo "== fp32"; timeout -k 10 120 python tools/bench_rnn.py 4096 10
echo "== fp32 3000"; timeout -k 10 120 python tools/bench_rnn.py 3000 10
} > gpurun_out/ab_head.log 2>&1
