#!/bin/bash
# Same-box A/B of the image builder: the working tree's kernels against the committed ones.
#   tools/ab_builder.sh [regions]        (run from the repo root, in this container: it calls gpurun itself)
# Boxes differ by up to 15 % on single kernels, more than most builder changes are worth, so the candidate is built as a second
# library (variants/libpepper_hip_cand.so), the committed sources are built in place, and both run alternately, three times each,
# inside ONE gpurun call. The working tree is restored afterwards (git stash).
set -e
N=${1:-16}
python -c "from pepper_thesis_amd import build; print(build.build_variant('cand', []))" | tail -1
git stash -q
trap 'git stash pop -q; python -c "import __graft_entry__ as g; g.build()" > /dev/null' EXIT   # and the in-tree library is the working tree's again
python -c "import __graft_entry__ as g; g.build()" > /dev/null
/usr/local/graft/bin/gpurun --timeout 900 -- "for i in 1 2 3; do timeout -k 10 120 python tools/bench_builder.py $N 2>&1 | grep 'back to back\|builder' | grep -o \"k_[a-z_]*': [0-9.]*\|events: [0-9.]*\" | tr '\n' ' '; echo ' HEAD'; PEPPER_HIP_LIB=\$PWD/variants/libpepper_hip_cand.so timeout -k 10 120 python tools/bench_builder.py $N 2>&1 | grep 'back to back\|builder' | grep -o \"k_[a-z_]*': [0-9.]*\|events: [0-9.]*\" | tr '\n' ' '; echo ' CAND'; done" 2>&1 | grep "HEAD\|CAND"
