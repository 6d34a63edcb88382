# the reference-shaped loop from several fresh processes: how often does the placement find its place?
for r in 1 2 3 4 5 6 7 8 9 10; do
  timeout -k 10 120 python bench.py --workload two_call_loop --steps 100 --warmup 20 --no-cpu-baseline --no-also 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,1),'us', [(p['array'][:4], p.get('candidates'), p.get('first_pass_us'), p.get('chosen_pass_us'), p.get('decided_by','')[:22], p.get('state_block','')[:5]) for p in d.get('placement',[])])"
done
