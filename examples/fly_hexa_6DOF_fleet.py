#!/usr/bin/env python3
"""The loop of the reference's examples/fly_hexa_6DOF_simple.py (:150-240) on a fleet: morphing hexas with the
6-DOF INDI law + WLS allocation hold their altitude and stay level while chasing a lateral circle (the
controller's target attitude is level: lateral force comes from the tilted rotors).  With the shipped URDF the
lateral loop is loose — its control-effectiveness rows (hexa_6DOF.urdf:30-36) are sized for the 0.2 kg main
body, the composite vehicle weighs 0.86 kg — so the drones orbit their moving target within a few decimetres
rather than sitting on it; altitude and attitude are tight.

    python examples/fly_hexa_6DOF_fleet.py --num_drones 4096 --duration_sec 5
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from dronesim_amd.envs import CtrlAviary  # noqa: E402
from dronesim_amd.fleet import WaypointTargets  # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--num_drones", type=int, default=4096)
    ap.add_argument("--duration_sec", type=float, default=5.0)
    ap.add_argument("--control_freq_hz", type=int, default=48)
    A = ap.parse_args(argv)
    n, FREQ = A.num_drones, 240
    AGGR = FREQ // A.control_freq_hz
    side = int(np.ceil(np.sqrt(n)))
    off = np.stack([(np.arange(n) % side) * 2.0, (np.arange(n) // side) * 2.0, np.zeros(n)], 1)
    init = np.array([0.0, 0.0, 0.6])                                      # fly_hexa_6DOF_simple.py:152
    R, PERIOD = 0.3, 15
    NUM_WP = A.control_freq_hz * PERIOD                                    # :157-169
    i = np.arange(NUM_WP)
    tp = np.stack([R * np.cos(i / NUM_WP * 4 * np.pi + np.pi / 2) + init[0],
                   R * np.sin(i / NUM_WP * 4 * np.pi + np.pi / 2) - R + init[1], np.full(NUM_WP, init[2])], 1)
    env = CtrlAviary(["hexa_6DOF"], n, initial_xyzs=init + off, aggregate_phy_steps=AGGR, freq=FREQ, dict_io=False)
    wp0 = (np.arange(n) * NUM_WP // 6) % NUM_WP                            # :170-172
    tgt = WaypointTargets(env.ctx, n, tp, np.zeros_like(tp), np.zeros_like(tp), np.zeros(NUM_WP), wp_counters=wp0,
                          offsets=off)
    steps = int(A.duration_sec * A.control_freq_hz)
    START = time.time()
    for k in range(steps):
        env.step_fused(tgt, control_timestep=AGGR / FREQ)
    el = time.time() - START
    rigid = env.state.rigid_aos()
    tilt = 2 * np.arcsin(np.clip(np.linalg.norm(rigid[:, 3:5], axis=1), 0, 1))
    wp = tgt.counters.cpu().numpy()[:n]
    d = rigid[:, 0:3] - off - tp[(wp - 1) % NUM_WP]
    err_xy, err_z = np.linalg.norm(d[:, :2], axis=1), np.abs(d[:, 2])
    fb = env.ctx.query(0)                                                 # QUERY_WLS_FALLBACKS
    print(f"{n} hexas x {steps} env steps in {el:.2f} s wall; lateral error median {np.median(err_xy):.3f} m, "
          f"altitude error max {err_z.max():.3f} m, max tilt {np.degrees(tilt.max()):.2f} deg, WLS fallbacks {fb}")
    env.close()
    return err_xy, err_z, tilt, fb


if __name__ == "__main__":
    main()
