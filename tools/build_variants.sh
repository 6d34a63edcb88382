#!/bin/bash
# dev helper (build container): differently-tuned builds of the same ABI for A/B runs on the GPU box
#   bash tools/build_variants.sh name "extra hipcc flags" [name2 "flags2" ...]   ->  build/libdsim_<name>.so
# then on the box:  python bench.py --lib build/libdsim_<name>.so ...
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $ROOT/build
while [ $# -ge 2 ]; do
  # (the flags of the default build, from __graft_entry__.HIPCC_FLAGS, so that a variant differs in what it is asked to)
  FLAGS=$(cd $ROOT && python3 -c "import __graft_entry__ as g; print(' '.join(g.HIPCC_FLAGS))")
  /opt/rocm/bin/hipcc $FLAGS $2 -o $ROOT/build/libdsim_$1.so $ROOT/dronesim_amd/csrc/dsim_api.hip
  echo "built build/libdsim_$1.so ($2)"
  shift 2
done
