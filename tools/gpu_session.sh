#!/bin/bash
# Runs ON the GPU box (via gpurun): one measurement session — parity tests, A/B of build variants and kernel forms,
# kernel-trace profiles of the config-5 chain.  usage: bash tools/gpu_session.sh <tag>
set -u
TAG=${1:-r02}
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
DSIM_MARGINS_OUT=$OUT/margins.json timeout -k 10 900 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest rc=$? $(tail -1 $OUT/pytest.log)"
one() {  # label, args...
  local label=$1; shift
  timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-also "$@" 2>/dev/null | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', round(d['value']/1e9,3), 'Gds/s', round(d['roofline']['launch_us'],1), 'us frac', round(d['roofline']['frac'],3))" | tee -a $OUT/ab.txt
}
for round in 1 2; do
  one "mixed v4(default: two waves per tile)" --workload mixed
  one "mixed v3 (three waves)" --workload mixed --mixed-v3
  one "mixed v3 soa layout" --workload mixed --layout soa
  one "mixed type-major" --workload mixed_type_major
  one "two-call loop" --workload two_call_loop
  one "config5 slab128" --workload config5
  one "config5 slab1024(r01 def)" --workload config5 --slab-m 1024
done
for S in 128 1024; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_c5_$S -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-also --workload config5 --slab-m $S > $OUT/kt_c5_$S.log 2>&1
  echo "kt config5 slab $S rc=$?"
  f=$(find $OUT/kt_c5_$S -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 $f | cut -c1-160
done
