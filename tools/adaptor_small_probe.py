#!/usr/bin/env python3
"""Dev probe (GPU box): Env.step of VelocityAviary at BASELINE's literal size (4 096 quads x 5 sub-steps), host-paced:
microseconds per step with the action a fresh tensor every other step.  usage: python tools/adaptor_small_probe.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dronesim_amd.envs import VelocityAviary
n = 4096
ij = np.arange(n)
xyz = np.stack([(ij % 64) * 1.0, (ij // 64) * 1.0, np.full(n, 0.5)], 1)
env = VelocityAviary(["robobee"], n, initial_xyzs=xyz, aggregate_phy_steps=5, noise_seed=1, dict_io=False)
act = torch.tensor([1.0, 0.0, 0.2, 0.5], device=env.ctx.device).repeat(n, 1)
for _ in range(200): env.step(act)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3000): env.step(act.clone() if _ % 2 else act)
torch.cuda.synchronize(); print("adaptor env step, 4096 drones x 5 sub-steps:", round((time.perf_counter() - t0) / 3000 * 1e6, 2), "us")
