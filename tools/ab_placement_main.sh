# A/B of the targets' placement on the headline line, fresh processes on one box; run on the GPU box
for r in 1 2 3 4 5 6; do for P in 1; do
  echo -n "placement $P headline: " ; DSIM_PLACEMENT=$P timeout -k 10 120 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-also 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['launch_us'],1),'us', round(d['roofline']['frac'],3), [(p['candidates'], p['first_pass_us'], p['chosen_pass_us'], p['decided_by'][:24]) for p in (d.get('placement') or [])])"
done; done
