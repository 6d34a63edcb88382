#!/bin/bash
# Runs ON the GPU box: the round's evidence set (TAG = r06).  usage: bash tools/gpu_evidence.sh <part: a|b|c|d>
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r06_final
mkdir -p $OUT
case "${1:-a}" in
a)
  DSIM_MARGINS_OUT=$OUT/margins.json timeout -k 10 1000 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider > $OUT/pytest.log 2>&1
  echo "pytest rc=$? $(tail -1 $OUT/pytest.log)"
  timeout -k 10 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
  echo "bench rc=$?"; head -c 600 $OUT/bench.json; echo
  DSIM_BENCH_FORCE_DIST=1 timeout -k 10 600 python bench.py --no-also --no-cpu-baseline > $OUT/bench_rccl_world1.json 2> $OUT/bench_rccl_world1.err
  echo "bench rccl world 1 rc=$?"; tail -c 400 $OUT/bench_rccl_world1.err
  ./tools/membench 4194304 > $OUT/membench.txt 2>&1; ./tools/membench --json 4194304 >> $OUT/membench.txt 2>&1
  bash tools/profile_sq.sh r06_main
  ;;
b)
  bash tools/profile_sq.sh r06_two_call_hexa --workload two_call_loop --two-call-kind hexa && \
  bash tools/profile_sq.sh r06_two_call_mixed --workload two_call_loop --two-call-kind mixed && \
  bash tools/profile_sq.sh r06_two_call_config5 --workload two_call_loop --two-call-kind config5 && \
  bash tools/profile_sq.sh r06_two_call_quad --workload two_call_loop
  ;;
c)
  export DSIM_PROFILE_BENCH="--steps 100 --warmup 5 --settle-seconds 0.3"      # (vector-heavy: profiled settled, bench.py Fleet.timed)
  bash tools/profile_sq.sh r06_sub5 --substeps 5 && bash tools/profile_sq.sh r06_hexa_sub5 --workload hexa --substeps 5 && \
  bash tools/profile_sq.sh r06_dyn_sub5 --workload dyn --substeps 5
  unset DSIM_PROFILE_BENCH
  bash tools/profile_sq.sh r06_c5 --workload config5 && bash tools/profile_sq.sh r06_dyn --workload dyn
  ;;
f)
  # launch duration against time under load (tools/clock_probe.py): what the device's power management does to each workload
  for W in "config2x1024 1" "config2x1024 5" "hexa 1" "hexa 5" "mixed 5" "dyn 5"; do
    set -- $W
    timeout -k 10 120 python tools/clock_probe.py --workload $1 --substeps $2 --seconds 4 2>&1 | grep -v "t=\|amdgpu.ids" > $OUT/clock_$1_sub$2.txt
    head -n 4 $OUT/clock_$1_sub$2.txt | cut -c1-260
  done
  ;;
e)
  bash tools/profile_sq.sh r06_mixed --workload mixed && bash tools/profile_sq.sh r06_hexa --workload hexa
  python tools/c5_chain_probe.py 400 > $OUT/c5_chain_probe.txt 2>&1; tail -n 4 $OUT/c5_chain_probe.txt
  ;;
d)
  bash tools/kernel_coverage.sh
  ;;
esac
