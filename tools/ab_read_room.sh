#!/bin/bash
# Runs ON the GPU box: the headline from fresh processes with the targets behind the state block in its own allocation (the
# default for large fleets, fleet.FleetState) and with them in a block of their own (DSIM_NO_READ_ROOM=1), interleaved.
# usage: bash tools/ab_read_room.sh [processes per arm]
N=${1:-6}
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for i in $(seq $N); do for ARM in 0 1; do
  DSIM_NO_READ_ROOM=$ARM python bench.py --steps 100 --warmup 10 --no-also --no-cpu-baseline 2>/dev/null \
    | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('targets %s: launch_us %.1f frac %.3f' % ('in a block of their own ' if $ARM else 'behind the state block', d['roofline']['launch_us'], d['roofline']['frac']))"
done; done
