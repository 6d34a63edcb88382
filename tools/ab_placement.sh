# A/B of the observation-row placement (dronesim_amd/placement.py) on the reference-shaped loop; run on the GPU box
mkdir -p gpurun_out/r03g
for r in 1 2 3 4; do for P in 1 0; do
  echo -n "placement $P two_call_loop: " ; DSIM_PLACEMENT=$P timeout -k 10 120 python bench.py --workload two_call_loop --steps 20 --warmup 5 --no-cpu-baseline --no-also 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e9,2),'Gds/s', round(d['ms_per_step']*1e3,1),'us', round(d['roofline']['frac'],3), json.dumps(d.get('placement')))"
done; done
