# interleaved same-box A/B of several builds of the library: bash tools/ab_libs.sh "<a.so> <b.so> ..." <bench args...>
LIBS=$1; shift
for r in $(seq 1 ${REPS:-3}); do for L in $LIBS; do
  echo -n "$L: "; timeout -k 10 120 python bench.py --lib $L --steps 200 --warmup 20 --no-cpu-baseline --no-also "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,2),'us')"
done; done
