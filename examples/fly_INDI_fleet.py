#!/usr/bin/env python3
"""The loop of the reference's examples/fly_INDI.py (:147-245) on a fleet, with this package.

    python examples/fly_INDI_fleet.py --num_drones 4096 --duration_sec 5

Same defaults as the reference script (robobee, 240 Hz physics, 48 Hz control, aggregate physics
steps, start (0,1,0.5) -> hover at (0,0,0.5) while the yaw target ramps, initial action 0.4); each
drone gets its own xy offset so that a fleet flies the figure side by side.  No GUI / plotting.
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from dronesim_amd.envs import CtrlAviary  # noqa: E402
from dronesim_amd.fleet import Targets  # noqa: E402

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--drone", default="robobee")
    ap.add_argument("--num_drones", type=int, default=4096)
    ap.add_argument("--simulation_freq_hz", type=int, default=240)
    ap.add_argument("--control_freq_hz", type=int, default=48)
    ap.add_argument("--duration_sec", type=int, default=15)
    ap.add_argument("--fused", type=int, default=1, help="1: one launch per loop iteration; 0: env.step + controller")
    A = ap.parse_args()

    n = A.num_drones
    side = int(np.ceil(np.sqrt(n)))
    off = np.stack([(np.arange(n) % side) * 2.0, (np.arange(n) // side) * 2.0, np.zeros(n)], 1)
    INIT_XYZS = np.array([0.0, 1.0, 0.5]) + off                       # fly_INDI.py:147
    AGGR = int(A.simulation_freq_hz / A.control_freq_hz)              # :139-141
    env = CtrlAviary([A.drone], n, initial_xyzs=INIT_XYZS, initial_rpys=np.zeros((n, 3)), freq=A.simulation_freq_hz,
                     aggregate_phy_steps=AGGR, dict_io=False)
    CTRL_EVERY_N_STEPS = int(np.floor(env.SIM_FREQ / A.control_freq_hz))   # :213
    tgt = Targets(env.ctx, n)
    target_pos = (np.array([0.0, 0.0, 0.5]) + off).astype(np.float32)      # :235 (+ the drone's own offset)
    START = time.time()
    k = 0
    if A.fused:
        import torch
        from dronesim_amd.fleet import frozen
        # the hover targets never change: on the device once, and promised unchanged (fleet.frozen), so that every later
        # set() copies nothing; only the yaw ramp — one constant per step — is filled
        pos_dev = frozen(torch.from_numpy(np.ascontiguousarray(target_pos.T)).to(env.ctx.device))
        for i in range(0, int(A.duration_sec * env.SIM_FREQ), AGGR):
            tgt.set(pos=pos_dev, yaw=0.4 + k / 200.0)                       # TARGET_RPYS[wp], :165-167
            env.step_fused(tgt, control_timestep=CTRL_EVERY_N_STEPS * env.TIMESTEP,
                           action=np.full((n, 4), 0.4, dtype=np.float32) if k == 0 else None)   # :214
            k += 1
    else:
        from dronesim_amd.control import INDIControl
        ctrl = INDIControl(A.drone, env=env)
        action = np.full((n, 4), 0.4, dtype=np.float32)
        for i in range(0, int(A.duration_sec * env.SIM_FREQ), AGGR):
            obs, reward, done, info = env.step(action)                                      # :223
            action, _, _ = ctrl.computeControlFromState(CTRL_EVERY_N_STEPS * env.TIMESTEP, None,   # :229-239
                                                        target_pos=target_pos, target_rpy=np.array([0, 0, 0.4 + k / 200.0]))
            k += 1
    pos = env.state.pos.T.cpu().numpy()
    el = time.time() - START
    err = np.linalg.norm(pos - target_pos, axis=1)
    print(f"{n} drones x {k} env steps in {el:.2f} s wall ({n * k / el:.3e} drone-steps/s incl. host loop); "
          f"distance to hover target: median {np.median(err):.3f} m, max {err.max():.3f} m")
    env.close()
