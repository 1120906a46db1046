# unit-split vs direction-split GRU form at 64 / 256 / 1000 chunks (tools/bench_gru.py); run through gpurun
cd $GRAFT_REPO_ROOT
for B in 64 256 1000; do
  timeout -k 10 120 python tools/bench_gru.py $B 3 2>/dev/null | tail -1
  PV_GRU_USPLIT=0 timeout -k 10 120 python tools/bench_gru.py $B 3 2>/dev/null | tail -1
done
