set -e
cd /root/repo
mkdir -p gpurun_out
timeout -k 10 900 python bench.py > gpurun_out/bench_r01d.json 2> gpurun_out/bench_r01d.err
bash tools/profile.sh r01d > gpurun_out/profile_r01d.log 2>&1
