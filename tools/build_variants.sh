#!/bin/bash
# dev helper (build container): differently-tuned builds of the same ABI for A/B runs on the GPU box
#   bash tools/build_variants.sh name "extra hipcc flags" [name2 "flags2" ...]   ->  build/libdsim_<name>.so
# then on the box:  python bench.py --lib build/libdsim_<name>.so ...
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $ROOT/build
while [ $# -ge 2 ]; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared $2 -o $ROOT/build/libdsim_$1.so $ROOT/dronesim_amd/csrc/dsim_api.hip
  echo "built build/libdsim_$1.so ($2)"
  shift 2
done
