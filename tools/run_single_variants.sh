# A/B of diagnostic library builds (variants/libpepper_hip_<name>.so, see build.build_variant) on tools/bench_single.py; run through gpurun
set -e
cd $GRAFT_REPO_ROOT
for v in default; do
  if [ $v = default ]; then unset PEPPER_HIP_LIB; else export PEPPER_HIP_LIB=$GRAFT_REPO_ROOT/variants/libpepper_hip_$v.so; fi
  timeout -k 10 200 python tools/bench_single.py > gpurun_out/single_$v.json 2> gpurun_out/single_$v.err
  python - <<PY
import json
d = json.load(open("gpurun_out/single_$v.json"))
for B in ("B64", "B512", "B1024"):
    print("$v", B, d["split"][B]["wall_ms"], d["split"][B]["kernels_ms"])
PY
done
