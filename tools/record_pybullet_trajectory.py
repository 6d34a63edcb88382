#!/usr/bin/env python3
"""P4 closure kit — records the fixture that would pin the integrator (DESIGN.md "Oracle").

The reference delegates rigid-body integration to PyBullet (`p.stepSimulation`, dronesim/envs/BaseAviary.py:542-543;
gravity / time step :673-675, URDF_USE_INERTIA_FROM_FILE :689, read-back :718-732).  PyBullet is absent from the build
container and from the GPU box, so this repo's restatement of Bullet's floating-base step (oracle/dsim_oracle.c:
orc_bullet_step, dronesim_amd/csrc/dsim_device.h: bullet_step) is "parity unpinned".  This script closes that gap the
day it is run ON A MACHINE THAT HAS `pybullet`, `gym` and the reference checkout: it drives the reference's own
CtrlAviary + INDIControl through the default flight of examples/fly_INDI.py (1 robobee from (0, 1, 0.5) to (0, 0, 0.5),
yaw target 0.4 + k/200, initial action 0.4, 240 Hz physics, 48 Hz control, AGGR_PHY_STEPS = 5) with the rotor noise
switched off, and stores every Env.step's action, state vector and the controller memory:

    python tools/record_pybullet_trajectory.py --reference /path/to/dronesim --out tests/golden/pybullet_fly_INDI.npz

240 physics steps = 48 control steps by default (--steps); the flight touches the ground plane around t = 0.35 s, which
the fixture therefore covers (contact is not modelled here: tests compare up to `first_contact_step`).  Nothing is
stubbed, vendored or re-implemented: without a real PyBullet the script refuses to run.

tests/test_p4_closure.py consumes the file and is skipped while it does not exist.
"""
import argparse
import os
import sys

import numpy as np


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference", help="checkout of enac-drones/dronesim")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                  "tests", "golden", "pybullet_fly_INDI.npz"))
    ap.add_argument("--steps", type=int, default=48, help="control steps (x5 physics steps each)")
    ap.add_argument("--drone", default="robobee")
    a = ap.parse_args(argv)
    try:
        import pybullet as p
    except ImportError:
        sys.exit("record_pybullet_trajectory: `pybullet` is not installed here; run this where it is (nothing is faked)")
    sys.path.insert(0, a.reference)
    from dronesim.control.INDIControl import INDIControl
    from dronesim.envs.BaseAviary import Physics
    from dronesim.envs.CtrlAviary import CtrlAviary

    sim_freq, ctrl_freq = 240, 48
    aggr = int(sim_freq / ctrl_freq)                                     # fly_INDI.py:139-141
    init_xyz, init_rpy = np.array([[0.0, 1.0, 0.5]]), np.zeros((1, 3))  # fly_INDI.py:147-148
    env = CtrlAviary(drone_model=[a.drone], num_drones=1, initial_xyzs=init_xyz, initial_rpys=init_rpy,
                     physics=Physics.PYB, neighbourhood_radius=10, freq=sim_freq, aggregate_phy_steps=aggr,
                     gui=False, record=False, obstacles=False, user_debug_gui=False)
    # the reference adds unseeded Gaussian rotor noise every sub-step (BaseAviary.py:1518-1525): zero it for a
    # reproducible fixture by making the draws deterministic zeros for the duration of the recording
    real_normal = np.random.normal
    np.random.normal = lambda loc=0.0, scale=1.0, size=None: np.zeros(size) if size is not None else 0.0
    ctrl = INDIControl(drone_model=a.drone)
    dt_ctrl = aggr / sim_freq                                            # fly_INDI.py:231
    action = {"0": np.array([0.4, 0.4, 0.4, 0.4])}                       # fly_INDI.py:214
    rec = {k: [] for k in ("action", "state", "cmd", "last_vel", "last_rates", "last_thrust", "target_yaw", "contacts")}
    try:
        for k in range(a.steps):
            obs, _, _, _ = env.step(action)                             # fly_INDI.py:223
            yaw = 0.4 + k / 200.0                                        # fly_INDI.py:165-167, 235-237
            rec["action"].append(np.asarray(action["0"], dtype=np.float64).copy())
            rec["state"].append(np.asarray(obs["0"]["state"], dtype=np.float64).copy())
            rec["contacts"].append(len(p.getContactPoints(bodyA=env.DRONE_IDS[0], physicsClientId=env.CLIENT)))
            cmd, _, _ = ctrl.computeControlFromState(control_timestep=dt_ctrl, state=obs["0"]["state"],
                                                     target_pos=np.array([0.0, 0.0, 0.5]),
                                                     target_rpy=np.array([0.0, 0.0, yaw]))
            action = {"0": cmd}
            rec["cmd"].append(np.asarray(cmd, dtype=np.float64).copy())
            rec["last_vel"].append(np.asarray(ctrl.last_vel, dtype=np.float64).copy())
            rec["last_rates"].append(np.asarray(ctrl.last_rates, dtype=np.float64).copy())
            rec["last_thrust"].append(float(ctrl.last_thrust))
            rec["target_yaw"].append(yaw)
    finally:
        np.random.normal = real_normal
        env.close()
    contacts = np.asarray(rec["contacts"])
    first = int(np.argmax(contacts > 0)) if (contacts > 0).any() else a.steps
    out = {k: np.asarray(v) for k, v in rec.items()}
    out.update(init_xyz=init_xyz, init_rpy=init_rpy, sim_freq=sim_freq, ctrl_freq=ctrl_freq, aggr=aggr,
               first_contact_step=first, drone=a.drone,
               pybullet_api_version=np.int64(p.getAPIVersion()), numpy_version=np.__version__)
    np.savez(a.out, **out)
    print(f"wrote {a.out}: {a.steps} control steps, first ground contact at step {first}")


if __name__ == "__main__":
    main()
