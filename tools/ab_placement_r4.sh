#!/bin/bash
# interleaved same-box A/B of placement by trial on / off for the two-call loop of one fleet kind: bash tools/ab_placement_r4.sh <kind> [pairs]
KIND=${1:-quad}; PAIRS=${2:-4}
for r in $(seq 1 $PAIRS); do for P in 1 0; do
  echo -n "$KIND placement=$P: "
  DSIM_PLACEMENT=$P timeout -k 10 120 python bench.py --workload two_call_loop --two-call-kind $KIND --steps 100 --warmup 20 --no-cpu-baseline --no-also 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['launch_us'],1),'us', [ (r['array'][:12], r.get('decided_by','')[:24], r.get('chosen_pass_us'), r.get('first_pass_us')) for r in d.get('placement',[]) if 'decided_by' in r])"
done; done
