#!/bin/bash
# Runs ON the GPU box: which kernel instances does `pytest -m gpu` launch?  -> gpurun_out/cov/kernels.txt
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/cov
mkdir -p $OUT
cd $ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 -m pytest tests -m gpu -q --no-header -p no:cacheprovider -x > $OUT/pytest.log 2>&1
tail -1 $OUT/pytest.log
python3 - "$OUT" <<'PY'
import csv, glob, sys
names = set()
for f in glob.glob(sys.argv[1] + "/kt/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        names.add(r["Name"])
open(sys.argv[1] + "/kernels.txt", "w").write("\n".join(sorted(names)) + "\n")
print(len(names), "distinct kernels launched")
PY
rm -rf $OUT/kt          # (the trace of the whole suite is beyond what gpurun copies back; the list is what is kept)
