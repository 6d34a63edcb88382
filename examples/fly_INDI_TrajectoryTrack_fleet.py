#!/usr/bin/env python3
"""The loop of the reference's examples/fly_INDI_TrajectoryTrack.py (:127-260) on a fleet.

    python examples/fly_INDI_TrajectoryTrack_fleet.py --num_drones 65536 --duration_sec 5

Gates (-3,0,2) -> (0.5,1,5) -> (3,0,2), minimum-snap polynomials sampled ON THE DEVICE (the reference pre-samples
a 1200-row table on the host with trajGen; the polynomial coefficients used here are that generator's output,
kept as a fixture), 240 Hz physics, 2 physics steps per control step, every drone offset on a 1 m grid and
started at its own phase of the lap.  Needs tests/golden/traj_track_waypoints.npz (coefficients).
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from dronesim_amd.envs import CtrlAviary  # noqa: E402
from dronesim_amd.fleet import TrajectoryTargets, WaypointTargets  # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--num_drones", type=int, default=4096)
    ap.add_argument("--duration_sec", type=float, default=5.0)
    ap.add_argument("--targets", default="table", choices=["table", "sampler"],
                    help="table: the pre-sampled 1200-row waypoint table indexed per drone (the example's own form); "
                         "sampler: get_des_state evaluated per drone on the device")
    A = ap.parse_args(argv)
    g = np.load(os.path.join(ROOT, "tests", "golden", "traj_track_waypoints.npz"))
    n, AGGR, FREQ = A.num_drones, 2, 240                                   # fly_INDI_TrajectoryTrack.py:108,162-164
    side = int(np.ceil(np.sqrt(n)))
    off = np.stack([np.arange(n) % side, np.arange(n) // side, np.zeros(n)], 1).astype(np.float64)
    env = CtrlAviary(["robobee"], n, initial_xyzs=g["gates"][0][None, :] + off, aggregate_phy_steps=AGGR, freq=FREQ,
                     dict_io=False)
    n_wp = g["target_pos"].shape[0]
    if A.targets == "table":
        wp0 = (np.arange(n) * n_wp // 6) % n_wp                             # :187-189
        tgt = WaypointTargets(env.ctx, n, g["target_pos"], g["target_vel"], g["target_acc"], g["target_yaw"],
                              wp_counters=wp0, offsets=off)
    else:
        tgt = TrajectoryTargets(env.ctx, n, g["coeffs"], g["TS"], t0=np.zeros(n), offsets=off)
    dt_ctrl = AGGR / FREQ
    steps = int(A.duration_sec * FREQ / AGGR)
    START = time.time()
    for k in range(steps):
        if A.targets == "sampler":
            tgt.sample(dt_ctrl)
        env.step_fused(tgt, control_timestep=dt_ctrl, action=np.full((n, 4), 0.4, dtype=np.float32) if k == 0 else None)
    pos = env.state.pos.T.cpu().numpy()
    el = time.time() - START
    z = pos[:, 2]
    print(f"{n} drones x {steps} env steps in {el:.2f} s wall ({n * steps / el:.3e} drone-steps/s incl. host loop); "
          f"altitude range [{z.min():.2f}, {z.max():.2f}] m")
    env.close()
    return pos - off


if __name__ == "__main__":
    main()
