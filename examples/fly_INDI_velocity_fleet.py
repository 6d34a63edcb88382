#!/usr/bin/env python3
"""The loop of the reference's examples/fly_INDI_velocity.py (:168-195) on a fleet: VelocityAviary, every drone
commanded along the unit direction (0.2, 0.2, 0.2) at 2 % of its maximum speed.

    python examples/fly_INDI_velocity_fleet.py --num_drones 4096 --duration_sec 3
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from dronesim_amd.envs import VelocityAviary  # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--drone", default="robobee")
    ap.add_argument("--num_drones", type=int, default=4096)
    ap.add_argument("--duration_sec", type=float, default=3.0)
    A = ap.parse_args(argv)
    n, AGGR, FREQ = A.num_drones, 5, 240
    side = int(np.ceil(np.sqrt(n)))
    xyz = np.stack([(np.arange(n) % side) * 2.0, (np.arange(n) // side) * 2.0, np.full(n, 1.0)], 1)
    env = VelocityAviary([A.drone], n, initial_xyzs=xyz, aggregate_phy_steps=AGGR, freq=FREQ, dict_io=False)
    action = torch.tensor([0.2, 0.2, 0.2, 0.02]).repeat(n, 1)             # fly_INDI_velocity.py:186-192
    steps = int(A.duration_sec * FREQ / AGGR)
    START = time.time()
    for _ in range(steps):
        obs, reward, done, info = env.step(action)
    vel = obs[:, 10:13].cpu().numpy()
    el = time.time() - START
    want = env.SPEED_LIMIT[0] * 0.02 * np.ones(3) / np.sqrt(3.0)
    print(f"{n} drones x {steps} env steps in {el:.2f} s wall; mean velocity {vel.mean(0).round(3)} m/s, "
          f"commanded {want.round(3)} m/s")
    env.close()
    return vel, want


if __name__ == "__main__":
    main()
