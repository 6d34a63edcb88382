#!/bin/bash
# Runs ON the GPU box: kernel-trace stats of one python script, printed.  usage: bash tools/kt_py.sh <tag> <script> [args...]
set -u
TAG=$1; shift
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 "$@" > $OUT/log.txt 2>&1
tail -3 $OUT/log.txt
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/kt/*/*kernel_stats.csv"))
for r in csv.DictReader(open(f[-1])):
    if int(r["Calls"]) >= 20:
        print(f'{r["Name"][:96]:96s} {r["Calls"]:>6s} avg {float(r["AverageNs"])/1e3:8.2f} us')
PY
