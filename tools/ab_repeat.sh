# the same bench line from several fresh processes on one box: how much of the spread is per process (placement)?
mkdir -p gpurun_out/r03g
for r in 1 2 3 4 5 6 7 8; do
  timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e9,2),'Gds/s', round(d['roofline']['launch_us'],1),'us', round(d['roofline']['frac'],3))"
done
