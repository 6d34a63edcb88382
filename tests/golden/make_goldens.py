#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING the reference.

TEST INFRASTRUCTURE.  Runs only in the build container (needs /root/reference);
the GPU box and the product never import this.  Nothing from the reference is
copied: this script imports the reference package from where it lies, feeds it
seeded inputs and stores inputs + outputs as .npz (fp64).

What is the real reference and what is a stand-in
-------------------------------------------------
* dronesim.utils.math, dronesim.control.wls_alloc, dronesim.utils.trajGen and
  dronesim.utils.trajutils import UNMODIFIED -> their goldens are clean.
* dronesim.control.INDIControl / INDIControl_6DOF import `pybullet`, `gym`,
  `pybullet_data` at module top (INDIControl.py:11,18; BaseAviary.py:9,14-15).
  None is installed here (ordinary ModuleNotFoundError, nothing was refused).
  The controllers use exactly three closed-form pybullet helpers
  (INDIControl.py:225,301,388,428).  We inject an in-memory module providing
  those three functions (restated below from the published Bullet 3.x
  formulas: ZYX Euler with the |sarg|>=0.99999 gimbal clamp, half-angle
  quaternion product + normalise, and btMatrix3x3::setRotation) and empty
  `gym` / `pybullet_data` shells.  The controller code that runs is the
  reference's own, unmodified.  CAVEAT: the goldens therefore inherit the
  stand-in's fidelity to Bullet's Euler convention, which nothing in the
  reference pins.

Usage:  python tests/golden/make_goldens.py   (writes tests/golden/*.npz)
"""
import math
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


# ----------------------------------------------------------------------------
# stand-ins for the three pybullet math helpers + empty gym shell
# ----------------------------------------------------------------------------
def _getEulerFromQuaternion(q):
    x, y, z, w = (float(q[0]), float(q[1]), float(q[2]), float(q[3]))
    sqx, sqy, sqz, squ = x * x, y * y, z * z, w * w
    sarg = -2.0 * (x * z - w * y)
    if sarg <= -0.99999:
        return (0.0, -0.5 * math.pi, 2.0 * math.atan2(x, -y))
    if sarg >= 0.99999:
        return (0.0, 0.5 * math.pi, 2.0 * math.atan2(-x, y))
    return (
        math.atan2(2.0 * (y * z + w * x), squ - sqx - sqy + sqz),
        math.asin(sarg),
        math.atan2(2.0 * (x * y + w * z), squ + sqx - sqy - sqz),
    )


def _getQuaternionFromEuler(e):
    phi, the, psi = float(e[0]) / 2.0, float(e[1]) / 2.0, float(e[2]) / 2.0
    q = [
        math.sin(phi) * math.cos(the) * math.cos(psi) - math.cos(phi) * math.sin(the) * math.sin(psi),
        math.cos(phi) * math.sin(the) * math.cos(psi) + math.sin(phi) * math.cos(the) * math.sin(psi),
        math.cos(phi) * math.cos(the) * math.sin(psi) - math.sin(phi) * math.sin(the) * math.cos(psi),
        math.cos(phi) * math.cos(the) * math.cos(psi) + math.sin(phi) * math.sin(the) * math.sin(psi),
    ]
    n = math.sqrt(sum(c * c for c in q))
    return tuple(c / n for c in q)


def _getMatrixFromQuaternion(q):
    x, y, z, w = (float(q[0]), float(q[1]), float(q[2]), float(q[3]))
    d = x * x + y * y + z * z + w * w
    s = 2.0 / d
    xs, ys, zs = x * s, y * s, z * s
    wx, wy, wz = w * xs, w * ys, w * zs
    xx, xy, xz = x * xs, x * ys, x * zs
    yy, yz, zz = y * ys, y * zs, z * zs
    return (
        1.0 - (yy + zz), xy - wz, xz + wy,
        xy + wz, 1.0 - (xx + zz), yz - wx,
        xz - wy, yz + wx, 1.0 - (xx + yy),
    )


def install_standins():
    pb = types.ModuleType("pybullet")
    pb.getEulerFromQuaternion = _getEulerFromQuaternion
    pb.getQuaternionFromEuler = _getQuaternionFromEuler
    pb.getMatrixFromQuaternion = _getMatrixFromQuaternion
    sys.modules["pybullet"] = pb
    sys.modules["pybullet_data"] = types.ModuleType("pybullet_data")
    gym = types.ModuleType("gym")

    class Env:  # BaseAviary subclasses gym.Env; never instantiated here
        pass

    gym.Env = Env
    gym.spaces = types.ModuleType("gym.spaces")
    sys.modules["gym"] = gym
    sys.modules["gym.spaces"] = gym.spaces
    if REF not in sys.path:
        sys.path.insert(0, REF)


# ----------------------------------------------------------------------------
# seeded case generators
# ----------------------------------------------------------------------------
def quat_from_euler(r, p, y):
    return np.array(_getQuaternionFromEuler((r, p, y)))


def make_cases(rng, n, n_act, reset_thrust, reset_cmd):
    """Random + hand-picked controller inputs (one row per case)."""
    c = {}
    rpy = np.stack(
        [rng.uniform(-1.2, 1.2, n), rng.uniform(-1.2, 1.2, n), rng.uniform(-math.pi, math.pi, n)], 1
    )
    # hand-picked attitude edge cases
    rpy[0] = (0.0, 0.0, 0.0)
    rpy[1] = (0.0, 0.0, math.pi - 1e-3)       # yaw at +pi, target across the wrap
    rpy[2] = (0.0, 0.0, -math.pi + 1e-3)
    rpy[3] = (0.3, 1.5699, 0.4)               # inside the gimbal clamp (sarg>=0.99999)
    rpy[4] = (0.3, -1.5699, -2.0)
    rpy[5] = (1.5, 0.2, 1.0)                  # large roll: det(G)=T^2 cos(phi) small
    rpy[6] = (-1.55, -0.4, -3.0)
    rpy[7] = (2.8, 0.1, 0.5)                  # upside-down-ish
    c["quat"] = np.stack([quat_from_euler(*r) for r in rpy])
    # a few non-unit and negative-w quaternions (pybullet helpers do not renormalise)
    c["quat"][8] *= 1.0 + 1e-4
    c["quat"][9] *= -1.0
    c["pos"] = rng.uniform(-5, 5, (n, 3))
    c["vel"] = rng.uniform(-3, 3, (n, 3))
    c["ang_vel"] = rng.uniform(-2, 2, (n, 3))
    c["target_pos"] = c["pos"] + rng.uniform(-1.5, 1.5, (n, 3))
    c["target_pos"][10] = c["pos"][10] + (30.0, -40.0, 25.0)   # saturates accel_e clip
    c["target_pos"][11] = c["pos"][11] - (30.0, -40.0, 25.0)
    c["target_vel"] = rng.uniform(-1, 1, (n, 3))
    c["target_acc"] = rng.uniform(-1, 1, (n, 3))
    c["target_rpy"] = np.stack(
        [rng.uniform(-0.5, 0.5, n), rng.uniform(-0.5, 0.5, n), rng.uniform(-2 * math.pi, 2 * math.pi, n)], 1
    )
    c["target_rpy"][1, 2] = -math.pi + 0.2
    c["target_rpy"][2, 2] = math.pi - 0.2
    c["target_rpy_rates"] = rng.uniform(-1, 1, (n, 3))
    c["dt"] = rng.choice([1 / 240, 2 / 240, 5 / 240], n)
    # controller memory
    c["last_vel"] = c["vel"] + rng.uniform(-0.05, 0.05, (n, 3))
    c["last_rates"] = rng.uniform(-2, 2, (n, 3))
    c["last_thrust"] = rng.uniform(-0.5, 1.5, n)
    c["cmd"] = rng.uniform(0.0, 1.0, (n, n_act))
    c["cmd"][12] = 0.0                           # PWM clip at the floor
    c["cmd"][13] = 1.0                           # and at the ceiling
    # second half: gentle near-hover cases so that cmd_out is NOT saturated and
    # cmd_out - cmd pins the unclipped increment du
    h = n // 2
    g = slice(h, n)
    m = n - h
    grpy = np.stack([rng.uniform(-0.15, 0.15, m), rng.uniform(-0.15, 0.15, m),
                     rng.uniform(-math.pi, math.pi, m)], 1)
    c["quat"][g] = np.stack([quat_from_euler(*r) for r in grpy])
    c["vel"][g] = rng.uniform(-0.3, 0.3, (m, 3))
    c["last_vel"][g] = c["vel"][g] + rng.uniform(-1e-3, 1e-3, (m, 3))
    c["ang_vel"][g] = rng.uniform(-0.05, 0.05, (m, 3))
    for j in range(h, n):   # last_rates ~ current body rates
        Rm = np.array(_getMatrixFromQuaternion(c["quat"][j])).reshape(3, 3)
        c["last_rates"][j] = Rm.T @ c["ang_vel"][j] + rng.uniform(-1e-3, 1e-3, 3)
    c["target_pos"][g] = c["pos"][g] + rng.uniform(-0.05, 0.05, (m, 3))
    c["target_vel"][g] = c["vel"][g] + rng.uniform(-0.02, 0.02, (m, 3))
    c["target_acc"][g] = rng.uniform(-0.02, 0.02, (m, 3))
    c["target_rpy"][g, 2] = grpy[:, 2] + rng.uniform(-0.02, 0.02, m)
    c["last_thrust"][g] = rng.uniform(0.2, 0.6, m)
    c["cmd"][g] = rng.uniform(0.35, 0.65, (m, n_act))
    # first-call-after-reset cases
    for k in (14, 15):
        c["last_vel"][k] = 0
        c["last_rates"][k] = 0
        c["last_thrust"][k] = reset_thrust
        c["cmd"][k] = reset_cmd
    return c


def run_quad(ctrl_cls, model, cases, six_dof):
    """One computeControl per case on the reference controller; also the
    sub-function outputs (thrust, target_euler) of _INDIPositionControl."""
    n = cases["pos"].shape[0]
    ctrl = ctrl_cls(drone_model=model)
    n_act = ctrl.indi_actuator_nr
    out = {
        "cmd_out": np.zeros((n, n_act)), "pos_e": np.zeros((n, 3)), "yaw_e": np.zeros(n),
        "last_vel_out": np.zeros((n, 3)), "last_rates_out": np.zeros((n, 3)),
        "last_thrust_out": np.zeros(n),
        "pc_thrust": np.zeros(n), "pc_target_euler": np.zeros((n, 3)),
    }
    for i in range(n):
        def load_state():
            ctrl.reset()
            ctrl.last_vel = cases["last_vel"][i].copy()
            ctrl.last_rates = cases["last_rates"][i].copy()
            ctrl.last_thrust = float(cases["last_thrust"][i])
            ctrl.cmd = cases["cmd"][i].copy()

        # sub-function C2 alone
        load_state()
        r = ctrl._INDIPositionControl(
            float(cases["dt"][i]), cases["pos"][i].copy(), cases["quat"][i].copy(),
            cases["vel"][i].copy(), cases["target_pos"][i].copy(), cases["target_rpy"][i].copy(),
            cases["target_vel"][i].copy(), cases["target_acc"][i].copy(),
        )
        out["pc_thrust"][i] = r[0]
        out["pc_target_euler"][i] = r[1]
        # full call C1
        load_state()
        kw = dict(
            control_timestep=float(cases["dt"][i]), cur_pos=cases["pos"][i].copy(),
            cur_quat=cases["quat"][i].copy(), cur_vel=cases["vel"][i].copy(),
            cur_ang_vel=cases["ang_vel"][i].copy(), target_pos=cases["target_pos"][i].copy(),
            target_vel=cases["target_vel"][i].copy(), target_acc=cases["target_acc"][i].copy(),
            target_rpy=cases["target_rpy"][i].copy(),
            target_rpy_rates=cases["target_rpy_rates"][i].copy(),
        )
        cmd, pos_e, yaw_e = ctrl.computeControl(**kw)
        out["cmd_out"][i] = cmd
        out["pos_e"][i] = pos_e
        out["yaw_e"][i] = yaw_e
        out["last_vel_out"][i] = ctrl.last_vel
        out["last_rates_out"][i] = ctrl.last_rates
        out["last_thrust_out"][i] = ctrl.last_thrust
    return out


def run_sequence(ctrl_cls, model, rng, n_seq, n_steps, dt):
    """Multi-call sequences: controller memory evolution over a smooth synthetic
    state track (the physics half is absent here, so the state is a seeded
    random walk; what is pinned is the controller's recursion)."""
    ctrl = ctrl_cls(drone_model=model)
    n_act = ctrl.indi_actuator_nr
    S = {k: np.zeros((n_seq, n_steps, d)) for k, d in
         [("pos", 3), ("quat", 4), ("vel", 3), ("ang_vel", 3), ("target_pos", 3),
          ("target_vel", 3), ("target_acc", 3), ("target_rpy", 3)]}
    O = {"cmd_out": np.zeros((n_seq, n_steps, n_act)), "pos_e": np.zeros((n_seq, n_steps, 3)),
         "yaw_e": np.zeros((n_seq, n_steps)), "last_thrust_out": np.zeros((n_seq, n_steps))}
    for s in range(n_seq):
        ctrl.reset()
        pos = rng.uniform(-2, 2, 3)
        vel = rng.uniform(-0.5, 0.5, 3)
        rpy = np.array([rng.uniform(-0.3, 0.3), rng.uniform(-0.3, 0.3), rng.uniform(-3, 3)])
        om = rng.uniform(-0.5, 0.5, 3)
        tgt = pos + rng.uniform(-1, 1, 3)
        for k in range(n_steps):
            vel = vel + rng.normal(0, 0.05, 3)
            pos = pos + vel * dt
            om = 0.9 * om + rng.normal(0, 0.1, 3)
            rpy = rpy + om * dt
            q = quat_from_euler(*rpy)
            trpy = np.array([0.0, 0.0, 0.4 + k / 200.0])
            tv = rng.uniform(-0.2, 0.2, 3)
            ta = rng.uniform(-0.2, 0.2, 3)
            S["pos"][s, k], S["quat"][s, k], S["vel"][s, k], S["ang_vel"][s, k] = pos, q, vel, om
            S["target_pos"][s, k], S["target_vel"][s, k] = tgt, tv
            S["target_acc"][s, k], S["target_rpy"][s, k] = ta, trpy
            cmd, pos_e, yaw_e = ctrl.computeControl(
                control_timestep=dt, cur_pos=pos.copy(), cur_quat=q.copy(), cur_vel=vel.copy(),
                cur_ang_vel=om.copy(), target_pos=tgt.copy(), target_vel=tv.copy(),
                target_acc=ta.copy(), target_rpy=trpy.copy())
            O["cmd_out"][s, k] = cmd
            O["pos_e"][s, k] = pos_e
            O["yaw_e"][s, k] = yaw_e
            O["last_thrust_out"][s, k] = ctrl.last_thrust
    S.update(O)
    S["dt"] = np.array(dt)
    return S


def main():
    install_standins()
    from dronesim.control.INDIControl import INDIControl
    from dronesim.control.INDIControl_6DOF import INDIControl as INDIControl_6DOF  # same class name in both modules
    from dronesim.control.wls_alloc import wls_alloc
    from dronesim.utils import math as rmath
    from dronesim.utils.trajGen import trajGenerator

    # ---- (v) utils/math.py helpers --------------------------------------
    rng = np.random.default_rng(20240807)
    n = 64
    q1 = rng.normal(size=(n, 4)); q1 /= np.linalg.norm(q1, axis=1, keepdims=True)
    q2 = rng.normal(size=(n, 4)); q2 /= np.linalg.norm(q2, axis=1, keepdims=True)
    inv_comp = np.stack([rmath.quat_inv_comp(a, b) for a, b in zip(q1, q2)])
    comp = np.stack([rmath.quat_comp(a, b) for a, b in zip(q1, q2)])
    wrapped = np.stack([rmath.quat_wrap_shortest(a.copy()) for a in inv_comp])
    ang = np.concatenate([rng.uniform(-12, 12, n - 4), [math.pi, -math.pi, 3 * math.pi, -5 * math.pi]])
    nang = np.array([rmath.norm_ang(float(a)) for a in ang])
    eul = np.stack([_getEulerFromQuaternion(q) for q in q1])
    rot = np.stack([_getMatrixFromQuaternion(q) for q in q1])
    np.savez(os.path.join(OUT, "math_helpers.npz"), q1=q1, q2=q2, quat_inv_comp=inv_comp,
             quat_comp=comp, quat_wrap_shortest=wrapped, ang=ang, norm_ang=nang,
             euler_standin=eul, matrix_standin=rot)

    # ---- (iv) wls_alloc ---------------------------------------------------
    umin = np.zeros(6); umax = np.full(6, 9600.0)
    uc = np.array([4614, 4210, 4210, 4614, 4210, 4210.0])
    A = np.array([[0.0, -0.015, 0.015, 0.0, -0.015, 0.015],
                  [0.015, -0.010, -0.010, 0.015, -0.010, -0.010],
                  [0.103, 0.103, 0.103, -0.103, -0.103, -0.103],
                  [-0.0009] * 6])
    v = np.array([240, -240.5658, 600.0, 1.8532])
    Wv = np.array([100, 100, 1, 10.0])
    du, it = wls_alloc(v, umin - uc, umax - uc, A, None, None, Wv, None, (umin - uc).copy())
    wl = {"main_v": v, "main_umin": umin - uc, "main_umax": umax - uc, "main_B": A, "main_Wv": Wv,
          "main_up": umin - uc, "main_du": du, "main_iter": np.array(it)}
    # seeded hexa-shaped cases (6x6 B = G1/0.05 of hexa_6DOF, Wv/Wu as INDIControl_6DOF.py:607-628)
    G1 = np.array([[-7.5, -15.0, -7.5, 7.5, 15.0, 7.5], [-13.0, 0.0, 13.0, 13.0, 0.0, -13.0],
                   [-5.0, 5.0, -5.0, 5.0, -5.0, 5.0], [-2.0, 4.0, -2.0, -2.0, 4.0, -2.0],
                   [-3.0, 0.0, 3.0, -3.0, 0.0, 3.0], [1.5] * 6])
    B6 = G1 / 0.05
    Wv6 = np.array([1000, 1000, 0.1, 10, 10, 100.0]); Wu6 = np.ones(6)
    m = 96
    vs = rng.normal(0, 1, (m, 6)) * np.array([20, 20, 5, 3, 3, 6.0])
    cmds = rng.uniform(0, 1, (m, 6))
    scale = np.where(np.arange(m) % 3 == 2, 1e4, 1.0)   # x1e4 bounds force >1 iteration
    vs[np.arange(m) % 3 == 2] *= 3e4
    dus = np.full((m, 6), np.nan); its = np.zeros(m, dtype=np.int64); ok = np.zeros(m, dtype=bool)
    for i in range(m):
        lo, hi = (0.0 - cmds[i]) * scale[i], (1.0 - cmds[i]) * scale[i]
        try:
            r, it_ = wls_alloc(vs[i], lo, hi, B6, None, None, Wv6, Wu6, None)
        except Exception:          # reference raises on its own alpha bug -> recorded as failure
            r, it_ = None, -1
        its[i] = it_
        if r is not None:
            dus[i] = r; ok[i] = True
    wl.update(hexa_B=B6, hexa_Wv=Wv6, hexa_Wu=Wu6, hexa_v=vs, hexa_cmd=cmds, hexa_scale=scale,
              hexa_du=dus, hexa_iter=its, hexa_ok=ok)
    np.savez(os.path.join(OUT, "wls_alloc.npz"), **wl)

    # ---- (i)-(iii) controllers ---------------------------------------------
    for model, cls, six, rt, rc, seed in [
        ("robobee", INDIControl, False, 0.0, 0.0, 11),
        ("tello", INDIControl, False, 0.0, 0.0, 12),
        ("hexa_6DOF", INDIControl_6DOF, True, 0.3, 0.5, 13),
    ]:
        rng = np.random.default_rng(seed)
        n_act = 6 if six else 4
        cases = make_cases(rng, 192, n_act, rt, rc)
        out = run_quad(cls, model, cases, six)
        cases.update(out)
        np.savez(os.path.join(OUT, f"indi_single_{model}.npz"), **cases)
        seq = run_sequence(cls, model, rng, 4, 60, 5 / 240)
        np.savez(os.path.join(OUT, f"indi_sequence_{model}.npz"), **seq)
        # parsed controller constants, so the build's own parameter table is pinned too
        c = cls(drone_model=model)
        np.savez(os.path.join(OUT, f"ctrl_params_{model}.npz"),
                 G1=c.G1, kp=c.guidance_indi_pos_gain, kd=c.guidance_indi_speed_gain,
                 att=np.array([c.indi_gains.att.p, c.indi_gains.att.q, c.indi_gains.att.r]),
                 rate=np.array([c.indi_gains.rate.p, c.indi_gains.rate.q, c.indi_gains.rate.r]),
                 pwm2rpm_scale=np.array(c.PWM2RPM_SCALE), pwm2rpm_const=np.array(c.PWM2RPM_CONST),
                 min_pwm=np.array(c.MIN_PWM), max_pwm=np.array(c.MAX_PWM), m=c.m,
                 kf=c.KF, km=c.KM, pinv_G1=np.linalg.pinv(c.G1 / 0.05))

    # ---- config-3 waypoint tables (fly_INDI_TrajectoryTrack.py:127-186) -----
    gates = np.vstack((np.array([[-3.0, 0, 2]]), np.array([0.5, 1, 5]), np.array([3, 0, 2])))
    traj = trajGenerator(gates, max_vel=0.7, gamma=1e6)
    ts = np.arange(0, traj.TS[-1], 1 / 96)
    P, V, Ac, Y = [], [], [], []
    for ti in ts:
        st = traj.get_des_state(ti)
        P.append(st.pos); V.append(st.vel); Ac.append(st.acc); Y.append(st.yaw)
    np.savez(os.path.join(OUT, "traj_track_waypoints.npz"), TS=traj.TS, coeffs=traj.coeffs,
             t=ts, target_pos=np.array(P), target_vel=np.array(V), target_acc=np.array(Ac),
             target_yaw=np.array(Y), gates=gates)
    print("goldens written to", OUT)


if __name__ == "__main__":
    main()
