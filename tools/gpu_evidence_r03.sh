#!/bin/bash
# Runs ON the GPU box: round 3's evidence set.  usage: bash tools/gpu_evidence_r03.sh <part: a|b|c>
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r03_final
mkdir -p $OUT
case "${1:-a}" in
a)
  DSIM_MARGINS_OUT=$OUT/margins.json timeout -k 10 900 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider > $OUT/pytest.log 2>&1
  echo "pytest rc=$? $(tail -1 $OUT/pytest.log)"
  timeout -k 10 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
  echo "bench rc=$?"; head -c 600 $OUT/bench.json; echo
  DSIM_BENCH_BACKEND=gloo timeout -k 10 600 python bench.py --gpus 2 > $OUT/bench_2rank_gloo.json 2> $OUT/bench_2rank_gloo.err
  echo "bench2 rc=$?"
  ./tools/membench 4194304 > $OUT/membench.txt 2>&1; ./tools/membench --json 4194304 >> $OUT/membench.txt 2>&1
  bash tools/profile_sq.sh r03_main && bash tools/profile_sq.sh r03_mixed --workload mixed
  ;;
b)
  bash tools/profile_sq.sh r03_two_call --workload two_call_loop && bash tools/profile_sq.sh r03_c5 --workload config5 && \
  bash tools/profile_sq.sh r03_c5_lowdensity --workload config5 --slab-m 1024 && \
  DSIM_STORAGE=caller bash tools/profile_sq.sh r03_mixed_caller_order --workload mixed
  ;;
c)
  # two gloo ranks of config 5 sharing the one GPU, kernel trace of both ranks (the rehearsal path: gloo stages the packed
  # buffers through the host, so its host gaps are NOT those of RCCL)
  DSIM_BENCH_BACKEND=gloo rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_c5_2rank -- python3 bench.py --gpus 2 --workload config5 --steps 100 --warmup 10 --no-cpu-baseline --no-also > $OUT/kt_c5_2rank.log 2>&1
  echo "kt 2rank rc=$?"; find $OUT/kt_c5_2rank -name "*kernel_stats.csv" | head
  tail -c 1500 $OUT/kt_c5_2rank.log
  ;;
d)
  # the whole GPU suite once more on the final tree, the remaining workloads' counter passes, and a long two-rank soak
  DSIM_MARGINS_OUT=$OUT/margins.json timeout -k 10 900 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider > $OUT/pytest.log 2>&1
  echo "pytest rc=$? $(tail -1 $OUT/pytest.log)"
  bash tools/profile_sq.sh r03_hexa --workload hexa && bash tools/profile_sq.sh r03_sub5 --substeps 5 && \
  bash tools/profile_sq.sh r03_config3 --workload config3 && bash tools/profile_sq.sh r03_config4 --workload config4
  DSIM_BENCH_BACKEND=gloo timeout -k 10 600 python bench.py --gpus 2 --workload config5 --steps 3000 --warmup 20 --no-cpu-baseline --no-also > $OUT/soak_2rank_gloo.json 2> $OUT/soak_2rank_gloo.err
  echo "soak rc=$?"; grep '^{' $OUT/soak_2rank_gloo.json | tail -1 | head -c 1500; echo
  ;;
esac
