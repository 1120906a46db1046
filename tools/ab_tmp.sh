set -e
cd /root/repo
mkdir -p gpurun_out
timeout -k 10 900 python bench.py > gpurun_out/bench_r01f.json 2> gpurun_out/bench_r01f.err
