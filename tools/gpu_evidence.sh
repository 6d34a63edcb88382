#!/bin/bash
# Runs ON the GPU box: the round's evidence set.  usage: bash tools/gpu_evidence.sh <part: a|b>
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
mkdir -p gpurun_out/r02_final
if [ "${1:-a}" = "a" ]; then
  DSIM_MARGINS_OUT=gpurun_out/r02_final/margins.json timeout -k 10 900 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider > gpurun_out/r02_final/pytest.log 2>&1
  echo "pytest rc=$? $(tail -1 gpurun_out/r02_final/pytest.log)"
  timeout -k 10 600 python bench.py > gpurun_out/r02_final/bench.json 2> gpurun_out/r02_final/bench.err
  echo "bench rc=$?"; head -c 700 gpurun_out/r02_final/bench.json; echo
  bash tools/profile_sq.sh r02_main && bash tools/profile_sq.sh r02_hexa --workload hexa && bash tools/profile_sq.sh r02_mixed --workload mixed
else
  bash tools/profile_sq.sh r02_mixed_tm --workload mixed_type_major && bash tools/profile_sq.sh r02_sub5 --substeps 5 && \
  bash tools/profile_sq.sh r02_config3 --workload config3 --substeps 2 && bash tools/profile_sq.sh r02_config4 --workload config4 && \
  bash tools/profile_sq.sh r02_two_call --workload two_call_loop && bash tools/profile_sq.sh r02_c5 --workload config5 && \
  bash tools/profile_sq.sh r02_c5_lowdensity --workload config5 --slab-m 1024
fi
