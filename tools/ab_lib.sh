# interleaved same-box A/B of two builds of the library: bash tools/ab_lib.sh <other.so> <bench args...>
OTHER=$1; shift
for r in 1 2 3 4; do for L in dronesim_amd/libdronesim_amd.so $OTHER; do
  echo -n "$L: "; timeout -k 10 120 python bench.py --lib $L --steps 200 --warmup 20 --no-cpu-baseline --no-also "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,2),'us')"
done; done
