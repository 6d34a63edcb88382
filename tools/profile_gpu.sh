#!/bin/bash
# Runs ON the GPU box (via gpurun): kernel-trace stats + HBM counter passes for bench.py's default
# workload.  Output under gpurun_out/prof_<tag>/ ; summarise with tools/summarise_profile.py.
# usage: bash tools/profile_gpu.sh <tag> [extra bench args]
set -u
TAG=${1:-r01}; shift || true
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
B="$ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-also $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $B > $OUT/kt.log 2>&1
echo "kt rc=$?"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $B > $OUT/fetch.log 2>&1
echo "fetch rc=$?"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $B > $OUT/write.log 2>&1
echo "write rc=$?"
find $OUT -name "*.csv" | head -20
