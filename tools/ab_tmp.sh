set -e
cd /root/repo
mkdir -p gpurun_out
{
echo "== tests"; timeout -k 10 900 python -m pytest tests/test_summary_gpu.py tests/test_polish_gpu.py -x -q -m gpu 2>&1 | tail -3
echo "== builder"; timeout -k 10 300 python tools/bench_builder.py 8
echo "== stamps"; PEPPER_HIP_LIB=/root/repo/variants/libpepper_hip_pstamps.so timeout -k 10 300 python tools/pstamps.py
} > gpurun_out/ab_pileup2.log 2>&1
