set -e
cd /root/repo
mkdir -p gpurun_out
bash tools/pmc_rnn.sh nt > gpurun_out/pmc_rnn.log 2>&1
