mkdir -p gpurun_out/r03g
for r in 1 2 3; do for L in tile64 tile256 tile1024 tile4096; do
  echo -n "$L " ; timeout -k 10 120 python bench.py --layout $L --steps 20 --warmup 5 --no-cpu-baseline --no-also 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e9,2),'Gds/s', round(d['roofline']['launch_us'],1),'us', round(d['roofline']['frac'],3), d['roofline']['kernel'][:60])"
done; done
