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

360 physics steps = 72 control steps by default (--steps); the flight touches the ground plane around t = 0.35 s and
leaves it about half a second later, which the fixture therefore covers (`contacts`, `first_contact_step`: the in-flight
part pins P4, the touchdown interval is what DSIM_OPT_PLANE is compared with).  Nothing is stubbed, vendored or
re-implemented: without a real PyBullet the script refuses to run.

Round 3 additions, written into the same file:
  * `helpers_*` — the three closed-form PyBullet helpers the controllers use (C8: INDIControl.py:225, 301, 388, 428) on a
    seeded zoo of attitudes (whole sphere, both signs of w, both gimbal branches, non-unit quaternions):
    p.getEulerFromQuaternion, p.getQuaternionFromEuler, p.getMatrixFromQuaternion — inputs and outputs;
  * `hexa_*` — one hover-and-step flight of hexa_6DOF with the reference's own INDIControl_6DOF (PyBullet flies it as an
    ARTICULATED body: six revolute arm joints held by default motors, hexa_6DOF.urdf:382-434; this repo flies the rigid
    composite): states, actions, commands — what quantifies the locked-joint approximation.

Round 5 additions, written into the same file:
  * `dyn_*` — one flight on Physics.DYN (BaseAviary._dynamics, BaseAviary.py:1767-1828) with the REAL engine as the pose store
    (p.resetBasePositionAndOrientation / resetBaseVelocity / getBase...): the branch is dead code in the fork for plumbing
    reasons (it reads self.KF, self.M, self.J, self.J_INV, self.L, self.GRAVITY, self.DRONE_MODEL and indexes the action as an
    array, :527), so those attributes are supplied from env.drones[0] and _preprocessAction hands an RPM array through — exactly
    what tests/golden/make_goldens.py:capture_dynamics does with a stand-in store.  Pins the stand-in (does Bullet return the
    quaternion it was given? the placeholder angular velocity?) and C8 inside that loop;
  * `noisy_*` — the fly_INDI.py flight again with the rotor noise ON and every draw logged (np.random.normal wrapped: a seeded
    generator, the normals stored per sub-step in the order the reference draws them, BaseAviary.py:1518-1521): states and
    normals, so that the oracle and the kernels can be fed the very same draws (noise replay) — pins the noise ENTRY POINTS
    of the force map inside the engine (which link, which frame), sub-step by sub-step.

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
    ap.add_argument("--steps", type=int, default=72, help="control steps (x5 physics steps each)")
    ap.add_argument("--hexa-steps", type=int, default=96, help="control steps of the hexa_6DOF hover-and-step flight (0: skip)")
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
    out.update(record_helpers(p))
    if a.hexa_steps > 0:
        out.update(record_hexa(p, a.hexa_steps, sim_freq, ctrl_freq))
    out.update(record_dyn(p, a.drone, sim_freq))
    out.update(record_noisy(p, a.drone, sim_freq, ctrl_freq))
    np.savez(a.out, **out)
    print(f"wrote {a.out}: {a.steps} control steps, first ground contact at step {first}")


def record_helpers(p, seed=20260):
    """C8: the three helpers on a seeded attitude zoo (inputs stored with the outputs, so the consumer needs no generator)."""
    rng = np.random.default_rng(seed)
    q = rng.normal(size=(2000, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)                          # whole sphere, w of both signs
    rpy = np.stack([rng.uniform(-np.pi, np.pi, 400), np.zeros(400), rng.uniform(-np.pi, np.pi, 400)], 1)
    sarg = np.concatenate([np.full(100, 1.0), 1.0 - rng.uniform(0, 8e-6, 100), 0.99999 - rng.uniform(3e-6, 5e-5, 200)])
    rpy[:, 1] = np.arcsin(sarg * np.where(rng.uniform(size=400) < 0.5, -1.0, 1.0))       # both gimbal branches and their edge
    qg = np.array([p.getQuaternionFromEuler(r.tolist()) for r in rpy])
    qn = rng.normal(size=(400, 4))
    qn = qn / np.linalg.norm(qn, axis=1, keepdims=True) * rng.uniform(0.5, 1.5, (400, 1))    # non-unit
    quats = np.concatenate([q, qg, qn])
    eulers_in = np.concatenate([rpy, rng.uniform(-np.pi, np.pi, (1000, 3))])
    return {"helpers_quat_in": quats,
            "helpers_euler_from_quat": np.array([p.getEulerFromQuaternion(x.tolist()) for x in quats]),
            "helpers_matrix_from_quat": np.array([p.getMatrixFromQuaternion(x.tolist()) for x in quats]),
            "helpers_euler_in": eulers_in,
            "helpers_quat_from_euler": np.array([p.getQuaternionFromEuler(x.tolist()) for x in eulers_in])}


def record_hexa(p, steps, sim_freq, ctrl_freq):
    """hexa_6DOF, hover at its start for a third of the flight, then a 0.5 m step in x and 0.3 m in z
    (examples/fly_hexa_6DOF.py drives the same classes), noise off."""
    from dronesim.control.INDIControl_6DOF import INDIControl as INDIControl6
    from dronesim.envs.BaseAviary import Physics
    from dronesim.envs.CtrlAviary import CtrlAviary
    aggr = int(sim_freq / ctrl_freq)
    init_xyz, init_rpy = np.array([[0.0, 0.0, 1.0]]), np.zeros((1, 3))
    env = CtrlAviary(drone_model=["hexa_6DOF"], num_drones=1, initial_xyzs=init_xyz, initial_rpys=init_rpy,
                     physics=Physics.PYB, neighbourhood_radius=10, freq=sim_freq, aggregate_phy_steps=aggr,
                     gui=False, record=False, obstacles=False, user_debug_gui=False)
    real_normal = np.random.normal
    np.random.normal = lambda loc=0.0, scale=1.0, size=None: np.zeros(size) if size is not None else 0.0
    ctrl = INDIControl6(drone_model="hexa_6DOF")
    action = {"0": np.full(6, 0.5)}                                       # INDIControl_6DOF.reset: cmd = 0.5
    rec = {k: [] for k in ("hexa_action", "hexa_state", "hexa_cmd", "hexa_target_pos", "hexa_joint_angles")}
    n_joints = p.getNumJoints(env.DRONE_IDS[0], physicsClientId=env.CLIENT)
    try:
        for k in range(steps):
            obs, _, _, _ = env.step(action)
            tpos = init_xyz[0] + (np.array([0.5, 0.0, 0.3]) if k >= steps // 3 else 0.0)
            rec["hexa_action"].append(np.asarray(action["0"], dtype=np.float64).copy())
            rec["hexa_state"].append(np.asarray(obs["0"]["state"], dtype=np.float64).copy())
            rec["hexa_joint_angles"].append([p.getJointState(env.DRONE_IDS[0], j, physicsClientId=env.CLIENT)[0]
                                             for j in range(n_joints)])
            out = ctrl.computeControlFromState(control_timestep=aggr / sim_freq, state=obs["0"]["state"], target_pos=tpos)
            action = {"0": out[0]}
            rec["hexa_cmd"].append(np.asarray(out[0], dtype=np.float64).copy())
            rec["hexa_target_pos"].append(tpos.copy())
    finally:
        np.random.normal = real_normal
        env.close()
    res = {k: np.asarray(v) for k, v in rec.items()}
    res.update(hexa_init_xyz=init_xyz, hexa_init_rpy=init_rpy, hexa_aggr=aggr)
    return res


def record_dyn(p, drone, sim_freq, steps=48, aggr=5):
    """Physics.DYN with the real engine as the pose store (module docstring).  48 Env.steps of 5 sub-steps: smooth
    differential thrust about hover, as tests/golden/make_goldens.py:capture_dynamics flies it."""
    from dronesim.envs.BaseAviary import DroneModel, Physics
    from dronesim.envs.CtrlAviary import CtrlAviary
    init_xyz, init_rpy = np.array([[0.0, 1.0, 1.5]]), np.array([[0.1, -0.05, 0.4]])
    env = CtrlAviary(drone_model=[drone], num_drones=1, initial_xyzs=init_xyz, initial_rpys=init_rpy, physics=Physics.DYN,
                     neighbourhood_radius=10, freq=sim_freq, aggregate_phy_steps=aggr, gui=False, record=False, obstacles=False,
                     user_debug_gui=False)
    d = env.drones[0]
    # what the fork left commented out (BaseAviary.py:200-235): the attributes _dynamics reads on self
    env.KF, env.KM, env.M, env.J, env.J_INV, env.L = d.KF, d.KM, d.M, d.J, d.J_INV, d.L
    env.GRAVITY, env.DRONE_MODEL = env.G * d.M, DroneModel.CF2X                      # :226; the examples' "default: CF2X"
    env._preprocessAction = lambda action: action                                     # upstream CtrlAviary: the RPM array as it is
    hover = np.sqrt(d.M * env.G / (4 * d.KF)) / 20000.0
    rng = np.random.default_rng(41)
    t = np.arange(steps)[:, None] * (aggr / sim_freq)
    pwm = np.clip(hover * (1.0 + 0.02 * np.sin(2 * np.pi * rng.uniform(0.5, 3.0, (1, 4)) * t + rng.uniform(0, 2 * np.pi, (1, 4))))
                  + rng.normal(0, 0.002, (steps, 4)), 0.0, 1.0)
    init = np.concatenate([env.pos[0], env.quat[0], env.rpy[0], env.vel[0], env.ang_v[0]]).copy()
    rec = []
    try:
        for k in range(steps):
            env.step((20000.0 * pwm[k])[None, :])
            rec.append(np.concatenate([env.pos[0], env.quat[0], env.rpy[0], env.vel[0], env.ang_v[0], env.rpy_rates[0]]).copy())
    finally:
        env.close()
    return {"dyn_init": init, "dyn_pwm": pwm, "dyn_states": np.asarray(rec), "dyn_aggr": aggr, "dyn_arm": d.L}


def record_noisy(p, drone, sim_freq, ctrl_freq, steps=48, seed=20265):
    """The fly_INDI.py flight with the rotor noise ON and logged: np.random.normal is wrapped by a seeded generator that
    records every draw; per physics sub-step the reference draws f_noise[4] ~ N(0, .01) then m_noise[4] ~ N(0, .001)
    (BaseAviary.py:1518-1521).  Started at z = 2 m so that the ground plane stays out of it."""
    from dronesim.control.INDIControl import INDIControl
    from dronesim.envs.BaseAviary import Physics
    from dronesim.envs.CtrlAviary import CtrlAviary
    aggr = int(sim_freq / ctrl_freq)
    init_xyz, init_rpy = np.array([[0.0, 1.0, 2.0]]), np.zeros((1, 3))
    env = CtrlAviary(drone_model=[drone], num_drones=1, initial_xyzs=init_xyz, initial_rpys=init_rpy, physics=Physics.PYB,
                     neighbourhood_radius=10, freq=sim_freq, aggregate_phy_steps=aggr, gui=False, record=False, obstacles=False,
                     user_debug_gui=False)
    gen = np.random.default_rng(seed)
    draws = []
    real_normal = np.random.normal

    def logged(loc=0.0, scale=1.0, size=None):
        v = gen.normal(loc, scale, size)
        draws.append((float(scale), np.atleast_1d(np.asarray(v, dtype=np.float64)).copy()))
        return v
    np.random.normal = logged
    ctrl = INDIControl(drone_model=drone)
    action = {"0": np.array([0.4, 0.4, 0.4, 0.4])}
    rec = {k: [] for k in ("noisy_action", "noisy_state", "noisy_cmd")}
    try:
        for k in range(steps):
            obs, _, _, _ = env.step(action)
            rec["noisy_action"].append(np.asarray(action["0"], dtype=np.float64).copy())
            rec["noisy_state"].append(np.asarray(obs["0"]["state"], dtype=np.float64).copy())
            cmd, _, _ = ctrl.computeControlFromState(control_timestep=aggr / sim_freq, state=obs["0"]["state"],
                                                     target_pos=np.array([0.0, 0.0, 2.0]), target_rpy=np.array([0.0, 0.0, 0.4]))
            action = {"0": cmd}
            rec["noisy_cmd"].append(np.asarray(cmd, dtype=np.float64).copy())
    finally:
        np.random.normal = real_normal
        env.close()
    # the draws in order: (f_noise[4], m_noise[4]) per sub-step — anything else would mean the reference draws differently
    assert len(draws) == 2 * steps * aggr and all(d[1].shape == (4,) for d in draws), "unexpected draw pattern"
    assert all(abs(draws[2 * i][0] - 0.01) < 1e-12 and abs(draws[2 * i + 1][0] - 0.001) < 1e-12 for i in range(steps * aggr))
    f = np.array([draws[2 * i][1] for i in range(steps * aggr)]).reshape(steps, aggr, 4)
    m = np.array([draws[2 * i + 1][1] for i in range(steps * aggr)]).reshape(steps, aggr, 4)
    res = {k: np.asarray(v) for k, v in rec.items()}
    res.update(noisy_f_noise=f, noisy_m_noise=m, noisy_init_xyz=init_xyz, noisy_init_rpy=init_rpy, noisy_aggr=aggr, noisy_seed=seed)
    return res


if __name__ == "__main__":
    main()
