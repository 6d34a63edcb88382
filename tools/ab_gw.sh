#!/bin/bash
# Runs ON the GPU box: same-box interleaved A/B of two builds on the headline, the five-sub-step workload (settled) and the
# BASELINE-size fleets (launch-latency bound).  usage: bash tools/ab_gw.sh <libA.so> <libB.so> [rounds]
A=$1; B=$2; R=${3:-2}
cd ${GRAFT_REPO_ROOT:-$(pwd)}
one() { python bench.py --no-also --no-cpu-baseline --lib $1 "${@:3}" 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%-28s %-22s launch_us %8.2f  host_us %8.2f' % ('$1'.split('/')[-1], '$2', d['roofline']['launch_us'], d['ms_per_step']*1e3))"; }
for r in $(seq $R); do for L in $A $B; do
  one $L headline --steps 200 --warmup 20
  one $L sub5_settled --substeps 5 --steps 100 --warmup 10 --settle-seconds 0.3
  one $L config2_4096_sub5 --workload config2 --substeps 5 --steps 2000 --warmup 200
  one $L config4_65536_sub1 --workload config4 --steps 2000 --warmup 200
  one $L hexa_sub1 --workload hexa --steps 200 --warmup 20
done; done
