for r in 1 2 3; do for L in dronesim_amd/libdronesim_amd.so build/libdsim_phys4.so build/libdsim_phys6.so; do
  echo -n "$L: "; timeout -k 10 120 python bench.py --lib $L --workload two_call_loop --steps 100 --warmup 20 --no-cpu-baseline --no-also 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,1),'us', [p['chosen_pass_us'] for p in d.get('placement',[])])"
done; done
