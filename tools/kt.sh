#!/bin/bash
# Runs ON the GPU box: kernel-trace stats of one bench.py workload, printed.  usage: bash tools/kt.sh <tag> <bench args...>
set -u
TAG=$1; shift
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-also $* > $OUT/kt.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/kt/*/*kernel_stats.csv")
for r in csv.DictReader(open(f[0])):
    if int(r["Calls"]) >= 5:
        print(f'{r["Name"][:64]:64s} {r["Calls"]:>6s} avg {float(r["AverageNs"])/1e3:8.2f} us  min {float(r["MinNs"])/1e3:8.2f}')
PY
