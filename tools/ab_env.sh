#!/bin/bash
# dev helper (GPU box): interleaved A/B over "ENV=... ENV=..." settings, 3 rounds
# usage: bash tools/ab_env.sh "<bench args>" "VAR=a VAR2=b" "VAR=c" ...
ARGS=$1; shift
for round in 1 2 3; do
  for E in "$@"; do
    env $E python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-also $ARGS 2>/dev/null | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$E | $ARGS |', round(d['value']/1e9,2), 'Gds/s', round(d['roofline']['launch_us'],1), 'us frac', round(d['roofline']['frac'],3))"
  done
done
