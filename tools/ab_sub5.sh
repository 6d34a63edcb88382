#!/bin/bash
# Runs ON the GPU box: the five-sub-step workloads (the examples' setting) with two builds of the library, interleaved.
# usage: bash tools/ab_sub5.sh <libA.so> <libB.so> [rounds]
set -u
A=$1; B=$2; R=${3:-2}
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for r in $(seq $R); do
  for L in $A $B; do
    for W in "config2x1024" "hexa" "mixed"; do
      python bench.py --workload $W --substeps 5 --steps 100 --warmup 10 --no-also --no-cpu-baseline --lib $L 2>/dev/null \
        | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$L'.split('/')[-1], '$W', 'sub5 launch_us %.1f frac %.3f' % (d['roofline']['launch_us'], d['roofline']['frac']))"
    done
    python bench.py --workload config2x1024 --substeps 1 --steps 100 --warmup 10 --no-also --no-cpu-baseline --lib $L 2>/dev/null \
      | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$L'.split('/')[-1], 'headline sub1 launch_us %.1f frac %.3f' % (d['roofline']['launch_us'], d['roofline']['frac']))"
  done
done
