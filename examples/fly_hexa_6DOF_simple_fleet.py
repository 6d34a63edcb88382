#!/usr/bin/env python3
"""The loop of the reference's examples/fly_hexa_6DOF_simple.py (:150-240) on a fleet of its own airframe,
hexa_6DOF_simple.urdf: morphing-hexa physics flown by the QUAD controller class on six actuators
(dronesim/control/INDIControl.py with actuator_nr = 6, output_nr = 4: hexa_6DOF_simple.urdf:28-33) through the
reference-shaped surfaces — obs = env.step(action); action = ctrl.computeControlFromState(obs) (:214-221).  The
vehicle tilts to move sideways, like a quad; the example's initial action is 0.1 on all six rotors (:206-208).

    python examples/fly_hexa_6DOF_simple_fleet.py --num_drones 4096 --duration_sec 5
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from dronesim_amd.control import INDIControl  # noqa: E402
from dronesim_amd.envs import CtrlAviary  # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--num_drones", type=int, default=4096)
    ap.add_argument("--duration_sec", type=float, default=5.0)
    ap.add_argument("--control_freq_hz", type=int, default=48)
    A = ap.parse_args(argv)
    import torch
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
    env = CtrlAviary(["hexa_6DOF_simple"], n, initial_xyzs=init + off, aggregate_phy_steps=AGGR, freq=FREQ, dict_io=False,
                     ground_plane=False)
    ctrl = INDIControl("hexa_6DOF_simple", env=env)                         # :202
    dev = env.ctx.device
    table = torch.from_numpy(tp.astype(np.float32)).to(dev)
    offs = torch.from_numpy(off.astype(np.float32)).to(dev)
    wp = torch.from_numpy((np.arange(n) * NUM_WP // 6) % NUM_WP).to(dev)   # :170-172
    action = torch.full((n, 6), 0.1, device=dev)                            # :206-208
    steps = int(A.duration_sec * A.control_freq_hz)
    START = time.time()
    for k in range(steps):
        obs, reward, done, info = env.step(action)                         # :214
        target = table[wp] + offs                                          # :222-225 (+ the drone's own offset)
        target[:, 2] = init[2]
        action, _, _ = ctrl.computeControlFromState(AGGR / FREQ, None, target_pos=target, target_rpy=np.zeros(3))   # :219-233
        wp = torch.where(wp < NUM_WP - 1, wp + 1, torch.zeros_like(wp))    # :236-240
    el = time.time() - START
    rigid = env.state.rigid_aos()
    tilt = 2 * np.arcsin(np.clip(np.linalg.norm(rigid[:, 3:5], axis=1), 0, 1))
    d = rigid[:, 0:3] - off - tp[(wp.cpu().numpy() - 1) % NUM_WP]
    err_xy, err_z = np.linalg.norm(d[:, :2], axis=1), np.abs(d[:, 2])
    print(f"{n} hexa_6DOF_simple x {steps} env steps in {el:.2f} s wall; lateral error median {np.median(err_xy):.3f} m, "
          f"altitude error max {err_z.max():.3f} m, max tilt {np.degrees(tilt.max()):.2f} deg")
    env.close()
    return err_xy, err_z, tilt


if __name__ == "__main__":
    main()
