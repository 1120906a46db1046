set -e
cd /root/repo
mkdir -p gpurun_out
PV_BENCH_BACKEND=gloo timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 32 --warmup 8 > gpurun_out/bench_n2.log 2> gpurun_out/bench_n2.err
