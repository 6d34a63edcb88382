#!/usr/bin/env python3
"""How much does the Env.step launch depend on WHERE its observation rows lie?  One 4 194 304-drone env; a series of
candidate arrays for the rows (2 GiB of ballast between them); for each the time of the real launch (k_physics_fast)
writing its rows there, alone and inside env.step() (which also transposes the action).  (A plain two-stream copy
probe over the same arrays was flat, 91-96 us, where the launch went from 148 to 204 us: it predicts nothing and was
dropped; profiles/r03_placement_probe.txt keeps that run.)
usage: python tools/placement_probe.py [--out FILE]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dronesim_amd.envs import CtrlAviary  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--candidates", type=int, default=12)
    ap.add_argument("--ballast-gib", type=float, default=2.0)
    a = ap.parse_args()
    n = 4096 * 1024
    side = 64
    ij = np.arange(n) % 4096
    xyz = np.stack([(ij % side) * 1.0, (ij // side) * 1.0, np.full(n, 0.5)], 1)
    env = CtrlAviary(["robobee"], n, initial_xyzs=xyz, aggregate_phy_steps=1, noise_seed=1, dict_io=False, layout="tile64",
                     placement=False)
    cmd = torch.full((n, 4), 0.4, device=env.ctx.device)
    rows, keep = [], []
    for k in range(a.candidates):
        obs = torch.zeros((n, 20), dtype=torch.float32, device=env.ctx.device)
        keep.append(obs)
        env._obs_buf = obs
        for _ in range(3):
            env.step(cmd)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            env.step(cmd)
        e1.record()
        torch.cuda.synchronize()
        # the physics launch alone, straight through the C-ABI (no action transpose in front of it)
        import ctypes
        from dronesim_amd import _native as nat
        args = env.step_args()
        args.action = env._action_buf.data_ptr()
        args.obs_out, args.obs_width = obs.data_ptr(), 20
        ref, view = ctypes.byref(args), env.state.view()
        f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for it in range(23):
            if it == 3:
                f0.record()
            nat.check(env.ctx.lib.dsim_physics(env.ctx.handle, env.ctx.stream_ptr(), n, view, env._last_action.data_ptr(), ref))
        f1.record()
        torch.cuda.synchronize()
        rows.append({"candidate": k, "va": hex(obs.data_ptr()), "physics_launch_us": round(f0.elapsed_time(f1) * 1e3 / 20, 1), "gib_allocated_before": round(k * (a.ballast_gib + 0.3125), 2),
                     "env_step_us": round(e0.elapsed_time(e1) * 1e3 / 20, 1)})
        print(rows[-1], flush=True)
        if a.ballast_gib > 0:
            keep.append(torch.empty((int(a.ballast_gib * (1 << 30)),), dtype=torch.uint8, device=env.ctx.device))
    if a.out:
        with open(a.out, "w") as fh:
            json.dump(rows, fh, indent=1)


if __name__ == "__main__":
    main()
