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


# ----------------------------------------------------------------------------
# Env-side pieces of the path: the reference's OWN force-map / aero / action-adaptor
# code run against recording stand-ins for the Bullet calls it makes
# ----------------------------------------------------------------------------
def capture_env_side():
    """BaseAviary._physics (P2/P3), _drag (P6), _groundEffect (P7), _downwash (P8),
    CtrlAviary/VelocityAviary/RPYTAviary._preprocessAction (P1, adaptors) are plain methods that
    read a few attributes of `self` and talk to Bullet only through p.applyExternalForce /
    p.applyExternalTorque / p.getLinkStates.  They are called here UNBOUND on a namespace object
    carrying those attributes, with recorders in place of the three Bullet calls: what is stored
    is every (link, vector, frame flag) the reference hands to the engine.  The Drone records come
    from the reference's own URDF parser (BaseAviary._parseURDFParameters)."""
    import pybullet as p                                  # the in-memory stand-in module
    from types import SimpleNamespace as NS
    from dronesim.envs.BaseAviary import BaseAviary, Drone
    from dronesim.envs.CtrlAviary import CtrlAviary
    from dronesim.envs.VelocityAviary import VelocityAviary
    from dronesim.envs.RPYTAviary import RPYTAviary
    from dronesim.control.INDIControl import INDIControl

    calls = []
    p.LINK_FRAME, p.WORLD_FRAME = 1, 2                   # Bullet's enum values
    p.applyExternalForce = lambda body, link, forceObj, posObj, flags, physicsClientId=0: calls.append(
        (0, link, [float(x) for x in forceObj], [float(x) for x in posObj], flags))
    p.applyExternalTorque = lambda body, link, torqueObj, flags, physicsClientId=0: calls.append(
        (1, link, [float(x) for x in torqueObj], [0.0, 0.0, 0.0], flags))

    def drone_of(model):
        return Drone(*BaseAviary._parseURDFParameters(None, model + ".urdf"))

    out = {}
    # ---- P0: every scalar the reference's URDF parser yields ------------------
    for model in ("robobee", "tello", "hexa_6DOF"):
        d = drone_of(model)
        for k in ("M", "L", "KF", "KM", "MAX_SPEED_KMH", "GND_EFF_COEFF", "PROP_RADIUS", "DW_COEFF_1",
                  "DW_COEFF_2", "DW_COEFF_3", "INDI_ACTUATOR_NR", "INDI_OUTPUT_NR"):
            out[f"{model}_{k}"] = np.array(getattr(d, k))
        for k in ("J", "DRAG_COEFF", "PWM2RPM_SCALE", "PWM2RPM_CONST", "G1", "MIN_PWM", "MAX_PWM"):
            out[f"{model}_{k}"] = np.array(getattr(d, k), dtype=np.float64)
        out[f"{model}_TYPE"] = np.array(d.TYPE)

    # ---- P2 / P3: rotor force map incl. the noise draws -----------------------
    rng = np.random.default_rng(31)
    for model in ("robobee", "tello", "hexa_6DOF"):
        d = drone_of(model)
        # _physics dispatches to self._quad_copter_physics / self._morphing_hexa_physics: bind the reference's own
        Fake = type("Fake", (), {"_quad_copter_physics": BaseAviary._quad_copter_physics,
                                 "_morphing_hexa_physics": BaseAviary._morphing_hexa_physics})
        fake = Fake()
        fake.drones, fake.DRONE_IDS, fake.CLIENT = [d], [7], 0
        n_act, K = d.INDI_ACTUATOR_NR, 24
        cmds = rng.uniform(np.array(d.MIN_PWM), np.array(d.MAX_PWM), (K, n_act))
        cmds[0] = 0.4
        cmds[1] = np.array(d.MAX_PWM)
        cmds[2] = np.array(d.MIN_PWM)
        seeds = 5000 + np.arange(K)
        rec_kind, rec_link, rec_vec, rec_pos, rec_flag, fn, mn = [], [], [], [], [], [], []
        for i in range(K):
            np.random.seed(int(seeds[i]))                 # the reference draws from the global numpy RNG
            calls.clear()
            BaseAviary._physics(fake, cmds[i].copy(), 0)
            rec_kind.append([c[0] for c in calls]); rec_link.append([c[1] for c in calls])
            rec_vec.append([c[2] for c in calls]); rec_pos.append([c[3] for c in calls])
            rec_flag.append([c[4] for c in calls])
            np.random.seed(int(seeds[i]))                 # the same stream again, as plain inputs for the oracle:
            fn.append(np.random.normal(0, 0.01, n_act))   # f_noise first, m_noise second (BaseAviary.py:1518-1521, 1429-1430)
            mn.append(np.random.normal(0, 0.001, n_act))
        out.update({f"{model}_fm_cmd": cmds, f"{model}_fm_seed": seeds, f"{model}_fm_kind": np.array(rec_kind),
                    f"{model}_fm_link": np.array(rec_link), f"{model}_fm_vec": np.array(rec_vec),
                    f"{model}_fm_pos": np.array(rec_pos), f"{model}_fm_flag": np.array(rec_flag),
                    f"{model}_fm_f_noise": np.array(fn), f"{model}_fm_m_noise": np.array(mn)})

    # ---- P6 / P7 / P8: aero add-ons (dead code in the fork: they read self.KF etc., which the fork
    #      moved into self.drones[i]; the attributes are supplied from the parsed Drone) ----------
    d = drone_of("robobee")
    K = 32
    rpy = np.stack([rng.uniform(-1.9, 1.9, K), rng.uniform(-1.4, 1.4, K), rng.uniform(-3.1, 3.1, K)], 1)
    rpy[0] = 0.0
    quat = np.stack([quat_from_euler(*e) for e in rpy])
    # what getEulerFromQuaternion hands back for that quaternion (what BaseAviary stores in self.rpy)
    rpy_store = np.stack([_getEulerFromQuaternion(q) for q in quat])
    vel = rng.uniform(-3, 3, (K, 3))
    rpm = rng.uniform(2000, 19000, (K, 4))
    pos = np.stack([rng.uniform(-2, 2, K), rng.uniform(-2, 2, K), rng.uniform(0.02, 1.0, K)], 1)
    h_clip = 0.05                                          # GND_EFF_H_CLIP: its definition is commented out (BaseAviary.py:235)
    prop_h = rng.uniform(0.01, 0.6, (K, 4))                 # rotor heights getLinkStates would report (inputs)
    drag_vec, gnd_vec, gnd_n = [], [], []
    for i in range(K):
        fake = NS(DRONE_IDS=[7], CLIENT=0, quat=quat[i:i + 1], vel=vel[i:i + 1], rpy=rpy_store[i:i + 1],
                  DRAG_COEFF=d.DRAG_COEFF, KF=d.KF, GND_EFF_COEFF=d.GND_EFF_COEFF, PROP_RADIUS=d.PROP_RADIUS,
                  GND_EFF_H_CLIP=h_clip)
        calls.clear()
        BaseAviary._drag(fake, rpm[i].copy(), 0)
        assert len(calls) == 1 and calls[0][1] == 4 and calls[0][4] == p.LINK_FRAME
        drag_vec.append(calls[0][2])
        # Bullet returns ragged per-link tuples (pos3, orn4, ...), which numpy >= 1.24 refuses to pack the way
        # the reference's np.array(...) expects; the stand-in pads every field to 4 so that the reference's
        # `link_states[j, 0][2]` reads the same number: the rotor height, an INPUT of this case
        p.getLinkStates = lambda body, linkIndices, computeLinkVelocity, computeForwardKinematics, physicsClientId, _i=i: [
            [(0.0, 0.0, float(prop_h[_i, j]) if j < 4 else float(pos[_i, 2]), 0.0)] + [(0.0, 0.0, 0.0, 1.0)] * 7
            for j in linkIndices]
        calls.clear()
        try:
            BaseAviary._groundEffect(fake, rpm[i].copy(), 0)
            g = np.zeros(4)
            for c in calls:
                assert c[0] == 0 and c[4] == p.LINK_FRAME and c[2][0] == 0 and c[2][1] == 0
                g[c[1]] = c[2][2]
            gnd_vec.append(g); gnd_n.append(len(calls))
        except Exception as e:                              # numpy's ragged-array refusal on newer versions
            raise RuntimeError(f"_groundEffect did not run: {e}")
    out.update(aero_quat=quat, aero_rpy=rpy_store, aero_vel=vel, aero_rpm=rpm, aero_pos=pos, aero_prop_h=prop_h,
               aero_h_clip=np.array(h_clip), aero_drag=np.array(drag_vec), aero_gnd=np.array(gnd_vec),
               aero_gnd_calls=np.array(gnd_n))
    # downwash: a small world, force on every drone
    M = 48
    wpos = np.stack([rng.uniform(0, 14, M), rng.uniform(0, 14, M), rng.uniform(0.5, 6, M)], 1)
    wpos[1] = wpos[0] + [0.0, 0.0, 0.5]                      # directly above
    wpos[3] = wpos[2] + [9.99, 0.0, 1.0]                     # just inside / just outside the 10 m cut
    wpos[5] = wpos[4] + [10.01, 0.0, 1.0]
    fake = NS(DRONE_IDS=list(range(M)), CLIENT=0, NUM_DRONES=M, pos=wpos, DW_COEFF_1=d.DW_COEFF_1,
              DW_COEFF_2=d.DW_COEFF_2, DW_COEFF_3=d.DW_COEFF_3, PROP_RADIUS=d.PROP_RADIUS)
    dw = np.zeros(M)
    for i in range(M):
        calls.clear()
        BaseAviary._downwash(fake, i)
        for c in calls:
            assert c[0] == 0 and c[1] == 4 and c[4] == p.LINK_FRAME and c[2][0] == 0 and c[2][1] == 0
            dw[i] += c[2][2]
    out.update(dw_pos=wpos, dw_fz=dw)

    # ---- P1 + the two alternate action adaptors -------------------------------
    d = drone_of("robobee")
    a = rng.uniform(-0.5, 1.5, (16, 4))
    clipped = CtrlAviary._preprocessAction(NS(drones=[d]), {"0": a[0]})
    out["clip_in"] = a
    out["clip_out"] = np.stack([CtrlAviary._preprocessAction(NS(drones=[d]), {"0": v})["0"] for v in a])
    K = 48
    for model in ("robobee", "tello"):
        d = drone_of(model)
        ctrl = INDIControl(drone_model=model)
        rpy = np.stack([rng.uniform(-0.6, 0.6, K), rng.uniform(-0.6, 0.6, K), rng.uniform(-3.1, 3.1, K)], 1)
        st = np.zeros((K, 20))
        st[:, 0:3] = rng.uniform(-2, 2, (K, 3))
        st[:, 3:7] = np.stack([quat_from_euler(*e) for e in rpy])
        st[:, 7:10] = np.stack([_getEulerFromQuaternion(q) for q in st[:, 3:7]])
        st[:, 10:13] = rng.uniform(-1, 1, (K, 3))
        st[:, 13:16] = rng.uniform(-1, 1, (K, 3))
        mem = {"last_vel": st[:, 10:13] + rng.normal(0, 0.02, (K, 3)), "last_rates": rng.uniform(-1, 1, (K, 3)),
               "last_thrust": rng.uniform(-1, 1, K), "cmd": rng.uniform(0.2, 0.8, (K, 4))}
        act_v = np.concatenate([rng.normal(0, 1, (K, 3)), rng.uniform(-1, 1, (K, 1))], 1)
        act_v[0, :3] = 0.0                                   # zero direction: v_unit_vector = 0 branch
        act_r = np.concatenate([rng.uniform(-2, 2, (K, 3)), rng.uniform(-1, 1, (K, 1))], 1)
        res = {}
        for name, cls, acts in (("vel", VelocityAviary, act_v), ("rpyt", RPYTAviary, act_r)):
            o_cmd, o_lv, o_lr, o_lt = [], [], [], []
            for i in range(K):
                ctrl.reset()
                ctrl.last_vel, ctrl.last_rates = mem["last_vel"][i].copy(), mem["last_rates"][i].copy()
                ctrl.last_thrust, ctrl.cmd = float(mem["last_thrust"][i]), mem["cmd"][i].copy()
                fake = NS(ctrl=[ctrl], AGGR_PHY_STEPS=5, TIMESTEP=1 / 240, SPEED_LIMIT=[d.MAX_SPEED_KMH * (1000 / 3600)],
                          _getDroneStateVector=lambda k, _i=i: st[_i].copy())
                r = cls._preprocessAction(fake, {"0": acts[i].copy()})["0"]
                o_cmd.append(np.array(r)); o_lv.append(ctrl.last_vel.copy()); o_lr.append(np.array(ctrl.last_rates).copy())
                o_lt.append(ctrl.last_thrust)
            res[name] = (np.array(o_cmd), np.array(o_lv), np.array(o_lr), np.array(o_lt))
        out.update({f"{model}_ad_state": st, f"{model}_ad_last_vel": mem["last_vel"], f"{model}_ad_last_rates": mem["last_rates"],
                    f"{model}_ad_last_thrust": mem["last_thrust"], f"{model}_ad_cmd": mem["cmd"],
                    f"{model}_ad_vel_action": act_v, f"{model}_ad_rpyt_action": act_r})
        for name in ("vel", "rpyt"):
            out.update({f"{model}_ad_{name}_cmd_out": res[name][0], f"{model}_ad_{name}_last_vel_out": res[name][1],
                        f"{model}_ad_{name}_last_rates_out": res[name][2], f"{model}_ad_{name}_last_thrust_out": res[name][3]})
    # ---- the fourth shipped airframe, hexa_6DOF_simple (the same links and joints as hexa_6DOF, a four-output control
    #      block): parser scalars and the force map, from their own generator so that everything above stays bit for bit ----
    rng = np.random.default_rng(32)
    model = "hexa_6DOF_simple"
    d = drone_of(model)
    for k in ("M", "L", "KF", "KM", "MAX_SPEED_KMH", "GND_EFF_COEFF", "PROP_RADIUS", "DW_COEFF_1",
              "DW_COEFF_2", "DW_COEFF_3", "INDI_ACTUATOR_NR", "INDI_OUTPUT_NR"):
        out[f"{model}_{k}"] = np.array(getattr(d, k))
    for k in ("J", "DRAG_COEFF", "PWM2RPM_SCALE", "PWM2RPM_CONST", "G1", "MIN_PWM", "MAX_PWM"):
        out[f"{model}_{k}"] = np.array(getattr(d, k), dtype=np.float64)
    out[f"{model}_TYPE"] = np.array(d.TYPE)
    Fake = type("Fake", (), {"_quad_copter_physics": BaseAviary._quad_copter_physics,
                             "_morphing_hexa_physics": BaseAviary._morphing_hexa_physics})
    fake = Fake()
    fake.drones, fake.DRONE_IDS, fake.CLIENT = [d], [7], 0
    n_act, K = d.INDI_ACTUATOR_NR, 24
    cmds = rng.uniform(np.array(d.MIN_PWM), np.array(d.MAX_PWM), (K, n_act))
    cmds[0] = 0.1                                             # the example's initial action (fly_hexa_6DOF_simple.py:206-208)
    cmds[1] = np.array(d.MAX_PWM)
    cmds[2] = np.array(d.MIN_PWM)
    seeds = 7000 + np.arange(K)
    rec_kind, rec_link, rec_vec, rec_pos, rec_flag, fn, mn = [], [], [], [], [], [], []
    for i in range(K):
        np.random.seed(int(seeds[i]))
        calls.clear()
        BaseAviary._physics(fake, cmds[i].copy(), 0)
        rec_kind.append([c[0] for c in calls]); rec_link.append([c[1] for c in calls])
        rec_vec.append([c[2] for c in calls]); rec_pos.append([c[3] for c in calls])
        rec_flag.append([c[4] for c in calls])
        np.random.seed(int(seeds[i]))
        fn.append(np.random.normal(0, 0.01, n_act))
        mn.append(np.random.normal(0, 0.001, n_act))
    out.update({f"{model}_fm_cmd": cmds, f"{model}_fm_seed": seeds, f"{model}_fm_kind": np.array(rec_kind),
                f"{model}_fm_link": np.array(rec_link), f"{model}_fm_vec": np.array(rec_vec),
                f"{model}_fm_pos": np.array(rec_pos), f"{model}_fm_flag": np.array(rec_flag),
                f"{model}_fm_f_noise": np.array(fn), f"{model}_fm_m_noise": np.array(mn)})
    np.savez(os.path.join(OUT, "env_side.npz"), **out)
    print("env-side goldens written")


def capture_roll_sweep():
    """INDIControl.computeControl with the roll angle swept through +-90 deg, where det(G) = T^2 cos(roll) of the
    position law's 3x3 system (INDIControl.py:314-339) goes through zero and np.linalg.pinv changes regime."""
    from dronesim.control.INDIControl import INDIControl
    ctrl = INDIControl(drone_model="robobee")
    deltas = np.array([1e-1, 3e-2, 1e-2, 1e-3, 1e-4, 1e-6, 1e-9, 0.0])
    rolls = np.concatenate([s * (math.pi / 2 - d * t) for s in (1.0, -1.0) for t in (1.0, -1.0) for d in [deltas]])
    rng = np.random.default_rng(77)
    n = len(rolls)
    pitch, yaw = rng.uniform(-0.4, 0.4, n), rng.uniform(-3, 3, n)
    c = {"roll": rolls, "quat": np.stack([quat_from_euler(r, p_, y) for r, p_, y in zip(rolls, pitch, yaw)]),
         "pos": rng.uniform(-1, 1, (n, 3)), "vel": rng.uniform(-0.5, 0.5, (n, 3)), "ang_vel": rng.uniform(-0.5, 0.5, (n, 3)),
         "target_pos": rng.uniform(-1, 1, (n, 3)), "target_yaw": rng.uniform(-3, 3, n),
         "last_vel": rng.uniform(-0.5, 0.5, (n, 3)), "last_rates": rng.uniform(-0.5, 0.5, (n, 3)),
         "last_thrust": rng.uniform(-0.5, 0.5, n), "cmd": rng.uniform(0.3, 0.7, (n, 4)), "dt": np.array(5 / 240)}
    out = {"cmd_out": np.zeros((n, 4)), "last_thrust_out": np.zeros(n), "last_rates_out": np.zeros((n, 3))}
    for i in range(n):
        ctrl.reset()
        ctrl.last_vel, ctrl.last_rates = c["last_vel"][i].copy(), c["last_rates"][i].copy()
        ctrl.last_thrust, ctrl.cmd = float(c["last_thrust"][i]), c["cmd"][i].copy()
        cmd, _, _ = ctrl.computeControl(control_timestep=float(c["dt"]), cur_pos=c["pos"][i].copy(), cur_quat=c["quat"][i].copy(),
                                        cur_vel=c["vel"][i].copy(), cur_ang_vel=c["ang_vel"][i].copy(),
                                        target_pos=c["target_pos"][i].copy(),
                                        target_rpy=np.array([0.0, 0.0, c["target_yaw"][i]]))
        out["cmd_out"][i], out["last_thrust_out"][i], out["last_rates_out"][i] = cmd, ctrl.last_thrust, ctrl.last_rates
    c.update(out)
    np.savez(os.path.join(OUT, "indi_roll_sweep.npz"), **c)
    print("roll-sweep goldens written")


def capture_dynamics():
    """Physics.DYN: BaseAviary._dynamics (BaseAviary.py:1767-1828) and the step() loop around it (:510-547), the
    reference's OWN explicit rigid-body integrator — the one whose arithmetic is Python in the reference tree, so the one
    rigid-body mode that CAN be pinned here.  The branch is dead code in the fork for plumbing reasons: it reads self.KF,
    self.KM, self.M, self.J, self.J_INV, self.L, self.GRAVITY and self.DRONE_MODEL, which the fork moved into
    self.drones[i] or left commented out (:200-235), and it indexes the action as an array (:527) where CtrlAviary now
    returns a dict.  Those attributes are supplied from the reference's own URDF parser; GRAVITY = G * M as the
    commented line :226 defines it.  Under DYN the engine is a POSE STORE: the reference writes the new pose and
    velocity with p.resetBasePositionAndOrientation / p.resetBaseVelocity (:1814-1826) and reads them back with
    p.getBasePositionAndOrientation / p.getBaseVelocity (:726-732); p.stepSimulation is skipped (:541-543).  The
    stand-in for those four calls stores and returns what it was given.

    (A) single calls of _dynamics on seeded states — incl. quat / rpy pairs that do NOT belong together, which shows
        that the rotation comes from self.quat and the angle sum from self.rpy — for both mixers;
    (B) flights through the reference's own BaseAviary.step() loop (unbound, on an object carrying the attributes
        above; _preprocessAction hands the RPM array through, as upstream's CtrlAviary does), AGGR_PHY_STEPS 5 and 1:
        pins the refresh of self.rpy from the stored quaternion between sub-steps (:513-520, 729)."""
    import pybullet as p
    from dronesim.envs.BaseAviary import BaseAviary, Drone, DroneModel, Physics

    store = {}
    p.resetBasePositionAndOrientation = lambda body, pos, quat, physicsClientId=0: store.__setitem__(
        ("pose", body), (tuple(float(x) for x in pos), tuple(float(x) for x in quat)))
    p.resetBaseVelocity = lambda body, lin, ang, physicsClientId=0: store.__setitem__(
        ("vel", body), (tuple(float(x) for x in lin), tuple(float(x) for x in ang)))
    p.getBasePositionAndOrientation = lambda body, physicsClientId=0: store[("pose", body)]
    p.getBaseVelocity = lambda body, physicsClientId=0: store[("vel", body)]

    def drone_of(model):
        return Drone(*BaseAviary._parseURDFParameters(None, model + ".urdf"))

    def attrs(d, dm, dt):
        return dict(KF=d.KF, KM=d.KM, M=d.M, J=d.J, J_INV=d.J_INV, L=d.L, GRAVITY=9.8 * d.M, DRONE_MODEL=dm, TIMESTEP=dt,
                    CLIENT=0)

    out = {}
    rng = np.random.default_rng(41)
    # ---- (A) single calls ----------------------------------------------------------------------------------------
    K = 64
    for model, dm, tag in (("robobee", DroneModel.CF2X, "robobee_x"), ("tello", DroneModel.CF2X, "tello_x"),
                           ("robobee", DroneModel.CF2P, "robobee_plus"), ("tello", DroneModel.HB, "tello_hb")):
        d = drone_of(model)
        dt = 1 / 240
        rpy = np.stack([rng.uniform(-1.3, 1.3, K), rng.uniform(-1.3, 1.3, K), rng.uniform(-math.pi, math.pi, K)], 1)
        rpy[0] = 0.0
        rpy[1] = (0.2, 1.5699, -0.7)                          # inside the gimbal clamp of getEulerFromQuaternion
        rpy[2] = (0.0, 0.0, math.pi - 1e-4)                   # the angle sum crosses +pi
        quat = np.stack([quat_from_euler(*e) for e in rpy])
        rpy_in = np.stack([_getEulerFromQuaternion(q) for q in quat])   # what BaseAviary stores in self.rpy (:729)
        rpy_in[3] = rpy_in[3] + (0.3, -0.2, 0.5)              # a pair that does not belong together
        pos = rng.uniform(-5, 5, (K, 3))
        vel = rng.uniform(-3, 3, (K, 3))
        rates = rng.uniform(-4, 4, (K, 3))
        rates[0] = 0.0
        pwm = rng.uniform(0.0, 1.0, (K, 4))
        pwm[0] = math.sqrt(d.M * 9.8 / (4 * d.KF)) / 20000.0   # hover
        rpm = 20000.0 * pwm
        o = {k: [] for k in ("pos", "quat", "vel", "ang_v", "rates")}
        for i in range(K):
            fake = type("Fake", (), {})()
            fake.__dict__.update(attrs(d, dm, dt), DRONE_IDS=[5], pos=pos[i:i + 1].copy(), quat=quat[i:i + 1].copy(),
                                 rpy=rpy_in[i:i + 1].copy(), vel=vel[i:i + 1].copy(), rpy_rates=rates[i:i + 1].copy())
            store.clear()
            BaseAviary._dynamics(fake, rpm[i].copy(), 0)
            o["pos"].append(store[("pose", 5)][0]); o["quat"].append(store[("pose", 5)][1])
            o["vel"].append(store[("vel", 5)][0]); o["ang_v"].append(store[("vel", 5)][1])
            o["rates"].append(fake.rpy_rates[0].copy())
        out.update({f"{tag}_dt": np.array(dt), f"{tag}_pos": pos, f"{tag}_quat": quat, f"{tag}_rpy": rpy_in, f"{tag}_vel": vel,
                    f"{tag}_rates": rates, f"{tag}_pwm": pwm, f"{tag}_rpm": rpm,
                    f"{tag}_mixer_plus": np.array(dm != DroneModel.CF2X)})
        out.update({f"{tag}_{k}_out": np.array(v) for k, v in o.items()})

    # ---- (B) flights through BaseAviary.step() -----------------------------------------------------------------------
    d = drone_of("robobee")
    hover = math.sqrt(d.M * 9.8 / (4 * d.KF)) / 20000.0
    for tag, aggr, steps in (("flight5", 5, 48), ("flight1", 1, 120)):
        N = 3
        Fake = type("Fake", (), {"_dynamics": BaseAviary._dynamics,
                                 "_updateAndStoreKinematicInformation": BaseAviary._updateAndStoreKinematicInformation,
                                 "_preprocessAction": lambda self, a: a,       # upstream CtrlAviary: the RPM array as it is
                                 "_computeObs": lambda self: None, "_computeReward": lambda self: -1,
                                 "_computeDone": lambda self: False, "_computeInfo": lambda self: {}})
        fake = Fake()
        fake.__dict__.update(attrs(d, DroneModel.CF2X, 1 / 240), DRONE_IDS=[11, 12, 13], NUM_DRONES=N, RECORD=False, GUI=False,
                             USER_DEBUG=False, USE_GUI_RPM=False, AGGR_PHY_STEPS=aggr, PHYSICS=Physics.DYN, step_counter=0,
                             last_clipped_action=None)
        pos0 = np.array([[0.0, 1.0, 0.5], [1.0, 0.0, 1.5], [-2.0, 0.5, 3.0]])
        rpy0 = np.array([[0.0, 0.0, 0.0], [0.1, -0.05, 0.4], [-0.3, 0.2, 3.0]])
        store.clear()
        for k in range(N):        # what _housekeeping leaves: loadURDF at INIT_XYZS / INIT_RPYS, zero velocities (:640-714)
            store[("pose", 11 + k)] = (tuple(pos0[k]), _getQuaternionFromEuler(rpy0[k]))
            store[("vel", 11 + k)] = ((0.0, 0.0, 0.0), (0.0, 0.0, 0.0))
        fake.pos, fake.quat, fake.rpy = np.zeros((N, 3)), np.zeros((N, 4)), np.zeros((N, 3))
        fake.vel, fake.ang_v, fake.rpy_rates = np.zeros((N, 3)), np.zeros((N, 3)), np.zeros((N, 3))
        BaseAviary._updateAndStoreKinematicInformation(fake)                   # reset() ends with it (:421)
        init = np.concatenate([fake.pos, fake.quat, fake.rpy, fake.vel, fake.ang_v], 1).copy()
        # smooth commands about hover: differential thrust that rolls / pitches / yaws the vehicles by a few tenths of a rad
        t = np.arange(steps)[:, None, None] * (aggr / 240.0)
        ph = rng.uniform(0, 2 * math.pi, (1, N, 4))
        fr = rng.uniform(0.5, 3.0, (1, N, 4))
        pwm = np.clip(hover * (1.0 + 0.02 * np.sin(2 * math.pi * fr * t + ph)) + rng.normal(0, 0.002, (steps, N, 4)), 0.0, 1.0)
        rec = []
        for k in range(steps):
            BaseAviary.step(fake, 20000.0 * pwm[k])
            rec.append(np.concatenate([fake.pos, fake.quat, fake.rpy, fake.vel, fake.ang_v, fake.rpy_rates], 1).copy())
        assert fake.step_counter == steps * aggr
        out.update({f"{tag}_aggr": np.array(aggr), f"{tag}_init": init, f"{tag}_pwm": pwm, f"{tag}_states": np.array(rec)})
    np.savez(os.path.join(OUT, "dynamics.npz"), **out)
    print("dynamics goldens written")


def main():
    install_standins()
    if len(sys.argv) > 1 and sys.argv[1] == "dyn":        # only the Physics.DYN file
        capture_dynamics()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "env":        # only the env-side file
        capture_env_side()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "roll":       # only the roll-sweep file
        capture_roll_sweep()
        return
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
        # the fourth shipped airframe: the QUAD controller class on six actuators (examples/fly_hexa_6DOF_simple.py:18, 202)
        ("hexa_6DOF_simple", INDIControl, False, 0.0, 0.0, 14),
    ]:
        rng = np.random.default_rng(seed)
        n_act = 6 if (six or model == "hexa_6DOF_simple") else 4
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
    capture_env_side()
    capture_roll_sweep()
    capture_dynamics()
    print("goldens written to", OUT)


if __name__ == "__main__":
    main()
