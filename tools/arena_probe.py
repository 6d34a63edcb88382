#!/usr/bin/env python3
"""Env.step launch time (k_physics_fast, 4 194 304 drones) with the observation rows (a) in their own allocation (the
default), (b) in ONE allocation with the state block, right behind it, (c) in one allocation, in front of it.
usage: python tools/arena_probe.py VARIANT"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dronesim_amd import _native as nat  # noqa: E402
from dronesim_amd.envs import CtrlAviary  # noqa: E402


def main():
    variant = sys.argv[1]
    n = 4096 * 1024
    ij = np.arange(n) % 4096
    xyz = np.stack([(ij % 64) * 1.0, (ij // 64) * 1.0, np.full(n, 0.5)], 1)
    env = CtrlAviary(["robobee"], n, initial_xyzs=xyz, aggregate_phy_steps=1, noise_seed=1, dict_io=False, layout="tile64",
                     placement=False)
    dev = env.ctx.device
    ns, no = env.state.data.numel(), n * 20
    if variant == "separate":
        obs = torch.zeros((n, 20), dtype=torch.float32, device=dev)
    else:
        arena = torch.zeros((ns + no + (1 << 20),), dtype=torch.float32, device=dev)
        if variant == "behind":
            st, obs = arena[:ns], arena[ns: ns + no].view(n, 20)
        else:
            obs, st = arena[:no].view(n, 20), arena[no: no + ns]
        st.view(env.state.data.shape).copy_(env.state.data)
        env.state.data = st.view(env.state.data.shape)
    env.reset()
    env._obs_buf = obs
    cmd = torch.full((n, 4), 0.4, device=dev)
    env.step(cmd)
    args = env.step_args()
    args.action = env._action_buf.data_ptr()
    args.obs_out, args.obs_width = obs.data_ptr(), 20
    ref, view = ctypes.byref(args), env.state.view()
    out = []
    for rep in range(3):
        f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for it in range(23):
            if it == 3:
                f0.record()
            nat.check(env.ctx.lib.dsim_physics(env.ctx.handle, env.ctx.stream_ptr(), n, view, env._last_action.data_ptr(), ref))
        f1.record()
        torch.cuda.synchronize()
        out.append(round(f0.elapsed_time(f1) * 1e3 / 20, 1))
    print(variant, out, "state", hex(env.state.data.data_ptr()), "obs", hex(obs.data_ptr()), flush=True)


if __name__ == "__main__":
    main()
