set -e
cd /root/repo
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/full_gpu_tests.log 2>&1
