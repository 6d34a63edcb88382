#!/usr/bin/env python3
"""Dev probe (GPU box): does the targets' walk (25 candidates of 168 MB allocated and freed at the first step_fused) in front
of the rows' walk change what the rows' walk finds?  One fresh process per call: python tools/targets_walk_probe.py [skip]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from dronesim_amd.envs import CtrlAviary  # noqa: E402

skip = len(sys.argv) > 1 and sys.argv[1] == "skip"
if skip:
    def no_walk(self, targets, control_timestep):
        targets._placed = True
    CtrlAviary._place_targets = no_walk
fl = bench.Fleet(4096, 1024, 0, 1, "tile64", 1)
fl.make_two_call_loop()
w, d, reg = fl.timed(100, 20, min_s=0.05)
log = [(r["array"][:14], r.get("decided_by", "")[:30], r.get("chosen_pass_us"), r.get("first_pass_us")) for r in fl.env.ctx.placement_log]
print("skip targets walk" if skip else "with targets walk", round(d / (100 * reg) * 1e6, 1), "us", log, flush=True)
