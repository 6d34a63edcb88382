#!/bin/bash
# dev helper (runs on the GPU box): interleaved A/B of differently-tuned builds of the same ABI
# usage: bash tools/ab_bench.sh "<bench args>" lib1.so lib2.so ...
ARGS=$1; shift
for round in 1 2 3; do
  for L in "$@"; do
    python bench.py --lib $L --steps 100 --warmup 10 --no-cpu-baseline --no-also $ARGS 2>/dev/null | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L $ARGS', round(d['value']/1e9,2), 'Gds/s', round(d['roofline']['launch_us'],1), 'us frac', round(d['roofline']['frac'],3))"
  done
done
