#!/bin/bash
# Runs ON the GPU box: SQ wave-cycle counters (two --pmc passes of 8 SQ slots) and the HBM counter passes for one
# bench.py workload.  Output under gpurun_out/<tag>/ ; summarise with tools/summarise_sq.py.
# usage: bash tools/profile_sq.sh <tag> <bench args...>
set -u
TAG=$1; shift
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
# DSIM_PROFILE_BENCH: the timing arguments of the profiled command (default: the driver's own; the vector-heavy workloads are
# profiled settled, "--steps 100 --warmup 5 --settle-seconds 0.3": bench.py Fleet.timed)
B="bench.py ${DSIM_PROFILE_BENCH:---steps 20 --warmup 5} --no-cpu-baseline --no-also $*"
# what was profiled, and how it was timed: "settled" (the launches right behind --settle-seconds of the same load: the sustained
# clock) or "from_idle" (the driver's contract: W warm-up steps, then K timed) - read by tools/summarise_sq.py into the summary
case "$B" in *--settle-seconds*) PROTO=settled ;; *) PROTO=from_idle ;; esac
printf '{"profiled_command": "python3 %s", "protocol": "%s"}\n' "$B" "$PROTO" > $OUT/meta.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $B > $OUT/kt.log 2>&1; echo "kt rc=$?"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VALU --output-format csv -d $OUT/sq1 -- python3 $B > $OUT/sq1.log 2>&1; echo "sq1 rc=$?"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_BUSY_CYCLES --output-format csv -d $OUT/sq2 -- python3 $B > $OUT/sq2.log 2>&1; echo "sq2 rc=$?"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $B > $OUT/fetch.log 2>&1; echo "fetch rc=$?"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $B > $OUT/write.log 2>&1; echo "write rc=$?"
