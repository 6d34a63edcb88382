#!/bin/bash
# Runs ON the GPU box (via gpurun): round-3 measurement session.  usage: bash tools/gpu_r03.sh <part>
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r03
mkdir -p $OUT
one() {  # label, args...
  local label=$1; shift
  timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-also "$@" 2>$OUT/last.err | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', round(d['value']/1e9,3), 'Gds/s', round(d['roofline']['launch_us'],1), 'us frac', round(d['roofline']['frac'],3), d.get('exchange',''))" | tee -a $OUT/ab.txt
}
kt() {  # tag, args...
  local tag=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$tag -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-also "$@" > $OUT/kt_$tag.log 2>&1
  echo "kt $tag rc=$?"
  f=$(find $OUT/kt_$tag -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if int(r["Calls"]) >= 5:
        print(f'  {r["Name"][:90]:90s} {r["Calls"]:>6s} avg {float(r["AverageNs"])/1e3:8.2f} us  min {float(r["MinNs"])/1e3:8.2f}')
PY
}
case "${1:-a}" in
a)
  for round in 1 2; do
    one "main" 
    one "two-call loop" --workload two_call_loop
    one "mixed (auto: type-major storage)" --workload mixed
    DSIM_STORAGE=caller one "mixed (storage=caller: k_step_mixed4)" --workload mixed
    one "mixed type-major (caller-sorted)" --workload mixed_type_major
    one "config5 slab128" --workload config5
    DSIM_STORAGE=caller one "config5 slab128 storage=caller" --workload config5
    one "config5 slab1024(r01 def)" --workload config5 --slab-m 1024
  done
  kt two_call --workload two_call_loop
  kt mixed --workload mixed
  kt c5 --workload config5
  ;;
c)
  for round in 1 2; do
    one "main 1023 replicas" --replicas 1023
    one "two-call 1024" --workload two_call_loop
    one "two-call 1023 replicas" --workload two_call_loop --replicas 1023
    one "two-call 1023 replicas soa" --workload two_call_loop --replicas 1023 --layout soa
    DSIM_STORAGE=caller one "c5 caller defer" --workload config5
    DSIM_STORAGE=caller DSIM_DEFER_FB=0 one "c5 caller inline-fb" --workload config5
    one "c5 auto defer" --workload config5
    DSIM_DEFER_FB=0 one "c5 auto inline-fb" --workload config5
  done
  ;;
d)
  for round in 1 2; do
    one "mixed auto separate launches" --workload mixed --runs separate
    one "mixed auto one launch" --workload mixed --runs one
    one "c5 auto one launch" --workload config5
    one "c5 auto separate" --workload config5 --runs separate
    DSIM_STORAGE=caller one "c5 caller" --workload config5
    one "c5 lowdensity one launch" --workload config5 --slab-m 1024
  done
  kt c5b --workload config5
  ;;
e)
  for round in 1 2 3; do
    one "c5 auto" --workload config5
    DSIM_STORAGE=caller one "c5 caller" --workload config5
    one "c5 lowdensity" --workload config5 --slab-m 1024
  done
  kt c5e --workload config5
  kt c5e_low --workload config5 --slab-m 1024
  ;;
f)
  for round in 1 2 3; do
    one "main tile64"
    one "main tile256" --layout tile256
    one "main tile1024" --layout tile1024
    one "main tile4096" --layout tile4096
    one "hexa tile64" --workload hexa
    one "hexa tile1024" --workload hexa --layout tile1024
    one "mixed tile1024" --workload mixed --layout tile1024
  done
  ;;
g)
  for round in 1 2 3; do
    one "c5 alone" --workload config5
    one "c5 mirror peer, split" --workload config5 --mirror-peer
    DSIM_DW_SPLIT=0 one "c5 mirror peer, one grid" --workload config5 --mirror-peer
  done
  kt c5_mirror --workload config5 --mirror-peer
  ;;
h)
  for round in 1 2 3; do
    one "c5 mirror peer, split, side stream high priority" --workload config5 --mirror-peer
    DSIM_HALO_PRIO=0 one "c5 mirror peer, split, normal priority" --workload config5 --mirror-peer
  done
  kt c5_mirror --workload config5 --mirror-peer
  ;;
i)
  for round in 1 2 3; do
    one "c5 mirror peer, device-scope events" --workload config5 --mirror-peer
    DSIM_TORCH_EVENTS=1 one "c5 mirror peer, torch events" --workload config5 --mirror-peer
  done
  kt c5_mirror --workload config5 --mirror-peer
  ;;
j)
  for round in 1 2 3; do
    one "c5 mirror: wire on a side stream, split query" --workload config5 --mirror-peer
    DSIM_HALO_ONE_STREAM=1 one "c5 mirror: ONE stream, split query" --workload config5 --mirror-peer
    DSIM_HALO_ONE_STREAM=1 DSIM_DW_SPLIT=0 one "c5 mirror: ONE stream, one-grid query" --workload config5 --mirror-peer
  done
  DSIM_HALO_ONE_STREAM=1 DSIM_DW_SPLIT=0 kt c5_mirror_one --workload config5 --mirror-peer
  ;;
b)
  DSIM_BENCH_BACKEND=gloo one "config5 2 gloo ranks, halo split" --workload config5 --gpus 2
  DSIM_BENCH_BACKEND=gloo DSIM_DW_SPLIT=0 one "config5 2 gloo ranks, halo one-grid" --workload config5 --gpus 2
  DSIM_BENCH_BACKEND=gloo DSIM_DW_EXCHANGE=allgather one "config5 2 gloo ranks, allgather" --workload config5 --gpus 2
  ;;
esac
