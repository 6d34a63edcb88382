#!/usr/bin/env python3
"""Dev probe (GPU box): the Python side of one fused step of a BASELINE-size fleet (4 096 quads, 5 sub-steps): the whole
step_fused call, the bare ctypes call with a prepared argument block, and the pieces in between."""
import ctypes
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from dronesim_amd import _native as nat  # noqa: E402

fl = bench.Fleet(4096, 1, 0, 5, "tile64", 1)
env, tg = fl.env, fl.tgt
for _ in range(200):
    env.step_fused(tg)
torch.cuda.synchronize()


def per_call(fn, n=20000):
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    return dt / n * 1e6


N = 3000
print("step_fused (host enqueue + device, drained):", round(per_call(lambda: env.step_fused(tg), N), 2), "us")
plan = env._fused_plan
_, args, sview, tview, ref = plan[:5]
lib, h, n = env.ctx.lib, env.ctx.handle, env.NUM_DRONES
sp = env.ctx.stream_ptr()
print("bare lib.dsim_step with a prepared block:   ", round(per_call(lambda: lib.dsim_step(h, sp, n, sview, tview, ref), N), 2), "us")
print("ctx.stream_ptr():                            ", round(per_call(env.ctx.stream_ptr), 3), "us")
print("_targets_ptrs(targets):                      ", round(per_call(lambda: env._targets_ptrs(tg)), 3), "us")
raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
if raw is not None:
    print("torch._C._cuda_getCurrentRawStream(0):       ", round(per_call(lambda: raw(0)), 3), "us")
print("bench.Fleet.step():                          ", round(per_call(fl.step, N), 2), "us")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(3000):
    fl.step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(8)
