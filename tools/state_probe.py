#!/usr/bin/env python3
"""Does the fused step's time depend on WHICH allocation the state block (and the targets right behind it) lies in?  One
4 194 304-drone fleet; ten further (state, targets) pairs allocated one after the other and all held; 20 fused launches timed
on each (contents copied over; the flight itself is irrelevant here).  usage: python tools/state_probe.py"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from dronesim_amd import _native as nat  # noqa: E402

fl = bench.Fleet(4096, 1024, 0, 1, "tile64", 1)
env, tg = fl.env, fl.tgt
for _ in range(30):
    env.step_fused(tg)
args = env.step_args()
lib, h, n = env.ctx.lib, env.ctx.handle, env.NUM_DRONES
sview, tview, ref = env.state.view(), tg.view(), ctypes.byref(args)
S0, T0 = env.state.data, tg.data
keep = []


def timed(S, T):
    sview.base, tview.base = S.data_ptr(), T.data_ptr()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for it in range(24):
        if it == 4:
            e0.record()
        nat.check(lib.dsim_step(h, env.ctx.stream_ptr(), n, sview, tview, ref))
    e1.record()
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) * 1e3 / 20, 1)


print("as allocated:", timed(S0, T0), flush=True)
for k in range(10):
    S, T = torch.empty_like(S0), torch.empty_like(T0)
    S.copy_(S0); T.copy_(T0)
    keep += [S, T]
    print(f"pair {k}: state+targets {timed(S, T)} us   this state with the ORIGINAL targets {timed(S, T0)} us   original state with these targets {timed(S0, T)} us", flush=True)
