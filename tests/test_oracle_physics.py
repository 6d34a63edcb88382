"""Analytic known-answer tests that pin the oracle's rigid-body half.

The reference delegates integration to PyBullet (absent, unpinned) and has no tests, so the
physics half of the oracle is "parity unpinned" against the reference itself; what pins it
are these closed-form checks of the restated Bullet step semantics.  CPU only.
"""
import math

import numpy as np
import pytest

from dronesim_amd import params
from oracle import oracle as orc

DT = 1.0 / 240.0


def _O(model="robobee"):
    t = params.builtin_type(model)
    return t, orc.Oracle([t])


def _rest(n=1, z=0.5):
    r = np.zeros((n, 13)); r[:, 2] = z; r[:, 6] = 1.0
    return r


def test_free_fall_one_step_hand_calculation():
    """(1) zero thrust: v1 = -g dt (no damping at v=0), z1 = z0 + v1 dt (semi-implicit);
    second step includes the damping c(1+|v|)v with c = 0.04f."""
    t, O = _O()
    r, m = _rest(), O.reset_mem(1)          # cmd = 0 -> zero thrust
    O.physics(r, m, 1, DT)
    v1 = -9.8 * DT
    assert r[0, 9] == v1 and r[0, 2] == 0.5 + v1 * DT       # exact: same fp64 operations
    O.physics(r, m, 1, DT)
    c = float(np.float32(0.04))
    v2 = v1 + (-9.8 - c * (1 + abs(v1)) * v1) * DT
    np.testing.assert_allclose(r[0, 9], v2, rtol=1e-15)
    np.testing.assert_allclose(r[0, 2], 0.5 + v1 * DT + v2 * DT, rtol=1e-15)
    np.testing.assert_array_equal(r[0, 3:7], [0, 0, 0, 1])  # no rotation appears


def test_terminal_velocity_of_damped_fall():
    """(1b) long fall converges to g = c(1+v)v  =>  v = (-1 + sqrt(1+4g/c))/2."""
    t, O = _O()
    r, m = _rest(z=1e5), O.reset_mem(1)
    O.physics(r, m, 240 * 60, DT)
    c = float(np.float32(0.04))
    vt = (-1 + math.sqrt(1 + 4 * 9.8 / c)) / 2
    np.testing.assert_allclose(-r[0, 9], vt, rtol=1e-9)


def test_hover_equilibrium():
    """(2) sum F = m g  =>  zero acceleration (robobee rpm ~ 9585, PWM ~ 0.479)."""
    t, O = _O()
    assert abs(t.hover_pwm - 0.4793) < 1e-3
    r, m = _rest(), O.reset_mem(1)
    m[:, 7:11] = t.hover_pwm
    O.physics(r, m, 240, DT)
    assert np.abs(r[0, 7:13]).max() < 1e-12 and abs(r[0, 2] - 0.5) < 1e-12


def test_single_rotor_first_step():
    """(3) one rotor on: dw = J^-1 (r x F + yaw torque) dt from rest, dv = (F/m - g) dt."""
    t, O = _O()
    r, m = _rest(), O.reset_mem(1)
    m[0, 7] = 0.5                       # rotor 0 at (+.11,+.11,0), spin sign -1
    O.physics(r, m, 1, DT)
    rpm = 20000 * 0.5
    F, tq = t.kf * rpm ** 2, t.km * rpm ** 2
    tau = np.cross(t.rotor_pos[0], [0, 0, F]) + np.array([0, 0, -tq])
    np.testing.assert_allclose(r[0, 10:13], tau / np.array(t.inertia) * DT, rtol=1e-14)
    np.testing.assert_allclose(r[0, 9], (F / t.mass - 9.8) * DT, rtol=1e-14)


def test_yaw_torque_sign_convention():
    """(5) z torque = -t0 + t1 - t2 + t3 (BaseAviary.py:1527): speeding rotors 1,3 yaws positive."""
    t, O = _O("tello")                  # symmetric arms: no roll/pitch torque from equal pairs
    for hi, sign in (((1, 3), +1), ((0, 2), -1)):
        r, m = _rest(), O.reset_mem(1)
        m[0, 7:11] = 0.4
        for j in hi:
            m[0, 7 + j] = 0.6
        O.physics(r, m, 1, DT)
        assert np.sign(r[0, 12]) == sign and abs(r[0, 10]) < 1e-12 and abs(r[0, 11]) < 1e-12


def test_torque_free_spin_about_principal_axis():
    """(4) spin about z with hover thrust: w decays only by damping, attitude advances by the exact
    axis-angle of each step: yaw(k) = sum_k w_k dt."""
    t, O = _O()
    r, m = _rest(), O.reset_mem(1)
    m[:, 7:11] = t.hover_pwm
    w = 3.0
    r[0, 12] = w
    c = float(np.float32(0.04))
    yaw = 0.0
    for _ in range(100):
        O.physics(r, m, 1, DT)
        w = w - c * (1 + abs(w)) * w * DT
        yaw += w * DT
        np.testing.assert_allclose(r[0, 12], w, rtol=1e-13)
        np.testing.assert_allclose(orc.euler_from_quat(r[0, 3:7])[2], yaw, rtol=1e-12)
    assert abs(r[0, 10]) < 1e-15 and abs(r[0, 11]) < 1e-15


def test_gyroscopic_term_conserves_momentum_direction():
    """Torque-free tumbling of an asymmetric-spin state, damping removed: |L| (world) is conserved
    to first order by the body-frame gyroscopic term w x Jw."""
    t = params.builtin_type("robobee")
    t.lin_damping = t.ang_damping = 0.0
    O = orc.Oracle([t])
    r, m = _rest(), O.reset_mem(1)
    m[:, 7:11] = t.hover_pwm
    r[0, 10:13] = [2.0, 0.5, 3.0]
    J = np.array(t.inertia)

    def L(r):
        R = orc.matrix_from_quat(r[0, 3:7])
        return R @ (J * (R.T @ r[0, 10:13]))

    L0 = L(r)
    O.physics(r, m, 480, DT / 8)
    assert np.linalg.norm(L(r) - L0) / np.linalg.norm(L0) < 2e-2      # explicit Euler drift only
    t2 = params.builtin_type("robobee"); t2.lin_damping = t2.ang_damping = 0.0
    t2.inertia = (1e-3, 1e-3, 1e-3)                                   # sphere: no gyroscopic torque at all
    O2 = orc.Oracle([t2])
    r2 = _rest(); r2[0, 10:13] = [2.0, 0.5, 3.0]
    O2.physics(r2, m, 100, DT)
    np.testing.assert_allclose(r2[0, 10:13], [2.0, 0.5, 3.0], rtol=1e-12)


def test_semi_implicit_ordering():
    """(6) position uses the UPDATED velocity: x1 = x0 + (v0 + a dt) dt."""
    t, O = _O()
    r, m = _rest(), O.reset_mem(1)
    m[:, 7:11] = t.hover_pwm
    r[0, 7] = 1.0
    c = float(np.float32(0.04))
    v1 = 1.0 - c * 2.0 * 1.0 * DT
    O.physics(r, m, 1, DT)
    np.testing.assert_allclose(r[0, 7], v1, rtol=1e-15)
    np.testing.assert_allclose(r[0, 0], v1 * DT, rtol=1e-15)


def test_quaternion_stays_unit_and_angle_clamp():
    """(7) unit norm after 1e5 steps; angular motion per step clamped to pi/4; velocity clamp 100."""
    t, O = _O()
    r, m = _rest(z=1e6), O.reset_mem(1)
    m[0, 7:11] = [0.55, 0.45, 0.5, 0.48]
    r[0, 10:13] = [0.3, -0.2, 0.5]
    O.physics(r, m, 100000, DT)
    assert abs(np.linalg.norm(r[0, 3:7]) - 1) < 1e-14
    assert np.isfinite(r).all() and np.abs(r[0, 7:13]).max() <= 100.0
    # |w| dt > pi/4: Bullet clamps the ANGLE used in sin/cos to pi/4 but scales the unclamped
    # angular velocity by sin(pi/8)/fAngle_clamped, then normalises (btTransformUtil quirk, kept):
    # rotation = 2 atan2(|w| sin(pi/8)/fA, cos(pi/8)), fA = (pi/4)/dt
    t2 = params.builtin_type("robobee"); t2.lin_damping = t2.ang_damping = 0.0
    O2 = orc.Oracle([t2])
    r2 = _rest(); r2[0, 12] = 90.0                 # 90 rad/s * 1/50 s = 1.8 rad > pi/4
    m2 = O2.reset_mem(1); m2[:, 7:11] = t2.hover_pwm
    O2.physics(r2, m2, 1, 1 / 50)
    fA = (math.pi / 4) / (1 / 50)
    expect = 2 * math.atan2(90.0 * math.sin(math.pi / 8) / fA, math.cos(math.pi / 8))
    np.testing.assert_allclose(orc.euler_from_quat(r2[0, 3:7])[2], expect, rtol=1e-12)
    assert expect < 1.8                            # less than the unclamped 1.8 rad
    r3 = _rest(); r3[0, 10] = 250.0
    O2.physics(r3, m2, 1, DT)
    assert r3[0, 10] == 100.0                      # maxCoordinateVelocity clamp


def test_noise_enters_as_the_reference_applies_it():
    """BaseAviary.py:1518-1543: x/y force noise reuses entries 0,1 for all four rotors (4 fn0, 4 fn1);
    moment noise 0,1 go to the base x/y torque; z torque uses all four noisy rotor torques."""
    t, O = _O("tello")
    r, m = _rest(), O.reset_mem(1)
    m[:, 7:11] = t.hover_pwm
    nz = np.zeros((1, 1, 12))
    nz[0, 0, 0:4] = [0.01, -0.02, 0.003, 0.004]       # f_noise
    nz[0, 0, 6:10] = [1e-4, -2e-4, 3e-4, 5e-4]        # m_noise
    O.physics(r, m, 1, DT, noise=nz)
    np.testing.assert_allclose(r[0, 7], 4 * 0.01 / t.mass * DT, rtol=1e-12)
    np.testing.assert_allclose(r[0, 8], 4 * -0.02 / t.mass * DT, rtol=1e-12)
    np.testing.assert_allclose(r[0, 9], (0.01 - 0.02 + 0.003 + 0.004) / t.mass * DT, rtol=1e-9, atol=1e-16)
    a = 0.0475
    fz = np.array([0.01, -0.02, 0.003, 0.004])
    tx = 1e-4 + sum(p[1] * f for p, f in zip(t.rotor_pos, fz))
    ty = -2e-4 - sum(p[0] * f for p, f in zip(t.rotor_pos, fz))
    tz = (-1e-4 - 2e-4 - 3e-4 + 5e-4) + sum(p[0] * -0.02 - p[1] * 0.01 for p in t.rotor_pos)
    np.testing.assert_allclose(r[0, 10:13], np.array([tx, ty, tz]) / np.array(t.inertia) * DT, rtol=1e-9, atol=1e-15)


def test_state_vector_layout():
    """P5 (BaseAviary.py:780-790): [pos3 quat4 rpy3 vel3 ang_v3 last_action]."""
    import ctypes
    t, O = _O()
    rigid = np.arange(13, dtype=np.float64) / 10
    rigid[3:7] = orc.quat_from_euler([0.1, -0.2, 0.3])
    out = np.zeros(20)
    P = t.to_c()
    D = ctypes.POINTER(ctypes.c_double)
    la = np.array([0.1, 0.2, 0.3, 0.4])
    orc.lib().orc_state_vector(ctypes.byref(P), rigid.ctypes.data_as(D), la.ctypes.data_as(D), out.ctypes.data_as(D))
    np.testing.assert_array_equal(out[0:7], rigid[0:7])
    np.testing.assert_allclose(out[7:10], [0.1, -0.2, 0.3], atol=1e-15)
    np.testing.assert_array_equal(out[10:16], rigid[7:13])
    np.testing.assert_array_equal(out[16:20], la)


def test_wrench_mapping_agrees_with_the_reference_dynamics_formula():
    """Cross-check against the reference's OWN explicit model (BaseAviary._dynamics, :1767-1828, dead
    code, CF2X mixer): from level rest one step of the Bullet-style restatement (rotor wrench P2 + step
    P4) and one step of _dynamics must agree exactly (no damping at v = 0, no gyro at w = 0) — signs,
    lever arms, yaw-torque convention and semi-implicit order; with damping removed they stay together
    over a short manoeuvre up to the rpy-rate vs body-rate difference, which shrinks with the step."""
    t = params.builtin_type("tello")          # symmetric X layout: rotor levers (0.0475, 0.0475)
    t.arm = 0.0475 * np.sqrt(2.0)             # the mixer's L / sqrt(2) on the PYB force map's own lever (the URDF's arm is 0.0635)
    t.lin_damping = t.ang_damping = 0.0
    O = orc.Oracle([t])
    cmd = np.array([0.52, 0.47, 0.50, 0.49])

    def run(dt, steps):
        r = _rest(); m = O.reset_mem(1); m[0, 7:11] = cmd
        pos, rpy, vel, rr = np.array([0, 0, 0.5]), np.zeros(3), np.zeros(3), np.zeros(3)
        rpm = 20000.0 * cmd
        for _ in range(steps):
            O.physics(r, m, 1, dt)
            # (the rpy SUM is carried here, not re-read from the quaternion: this test compares two integrators)
            pos, _, vel, rr, rpy = orc.dynamics(t, dt, rpm, pos, orc.quat_from_euler(rpy), rpy, vel, rr)
        return r[0], pos, rpy, vel, rr

    r, pos, rpy, vel, rr = run(DT, 1)
    np.testing.assert_allclose(r[0:3], pos, rtol=0, atol=1e-15)
    np.testing.assert_allclose(r[7:10], vel, rtol=0, atol=1e-15)
    np.testing.assert_allclose(r[10:13], rr, rtol=1e-13)              # world w == body rates == rpy rates at level
    np.testing.assert_allclose(orc.euler_from_quat(r[3:7]), rpy, rtol=0, atol=1e-7)   # exp-map vs rpy sum: O(angle^2)
    errs = []
    for div in (1, 2, 4):
        r, pos, rpy, vel, rr = run(DT / div, 24 * div)                # 0.1 s manoeuvre
        errs.append(np.abs(r[0:3] - pos).max() + np.abs(orc.euler_from_quat(r[3:7]) - rpy).max())
    assert errs[0] < 2e-3 and errs[2] < errs[0]                       # small, and not growing with refinement


def test_step_is_a_consistent_discretisation_of_the_damped_newton_euler_equations():
    """Independent formulation: integrate m v' = R F_b + m g - m c (1+|v|) v,  J w_b' = tau_b - w_b x J w_b -
    c (1+|w|) J w_b,  q' = 1/2 q (0, w_b) with scipy's adaptive Runge-Kutta (rtol 1e-11) under a constant body
    wrench that excites translation, all three rotation axes and the gyroscopic coupling; the restated
    semi-implicit step must converge to it at first order in dt (error halves when dt halves)."""
    from scipy.integrate import solve_ivp
    t = params.builtin_type("robobee")
    O = orc.Oracle([t])
    m_, J, c, g = t.mass, np.asarray(t.inertia, dtype=float), float(np.float32(0.04)), t.gravity
    cmd = np.array([0.500, 0.494, 0.497, 0.503])
    P = t.to_c()
    import ctypes
    D = ctypes.POINTER(ctypes.c_double)
    F, tau, rpm = np.zeros(3), np.zeros(3), np.zeros(4)
    f = orc.lib().orc_quad_wrench
    f.argtypes = [ctypes.POINTER(type(P)), D, D, D, D, D, D]
    f(ctypes.byref(P), cmd.ctypes.data_as(D), None, None, F.ctypes.data_as(D), tau.ctypes.data_as(D), rpm.ctypes.data_as(D))
    w0 = np.array([0.8, -0.5, 1.2])                       # body rates at t = 0 (level start: world = body)

    def rhs(_, y):
        pos, q, v, wb = y[0:3], y[3:7], y[7:10], y[10:13]
        q = q / np.linalg.norm(q)
        R = orc.matrix_from_quat(q)
        a = R @ F / m_ + np.array([0, 0, -g]) - c * (1 + np.linalg.norm(v)) * v
        dw = (tau - np.cross(wb, J * wb) - c * (1 + np.linalg.norm(wb)) * J * wb) / J
        x, y_, z, w = q
        dq = 0.5 * np.array([w * wb[0] + y_ * wb[2] - z * wb[1], w * wb[1] + z * wb[0] - x * wb[2],
                             w * wb[2] + x * wb[1] - y_ * wb[0], -x * wb[0] - y_ * wb[1] - z * wb[2]])
        return np.concatenate([v, dq, a, dw])

    T_end = 0.5
    y0 = np.concatenate([[0, 0, 1.0], [0, 0, 0, 1.0], [0.3, -0.2, 0.1], w0])
    ref = solve_ivp(rhs, (0, T_end), y0, rtol=1e-11, atol=1e-13).y[:, -1]
    R_end = orc.matrix_from_quat(ref[3:7] / np.linalg.norm(ref[3:7]))
    errs = []
    for div in (1, 2, 4):
        dt = (1.0 / 240.0) / div
        r = np.zeros((1, 13)); r[0, 0:3] = y0[0:3]; r[0, 6] = 1.0; r[0, 7:10] = y0[7:10]; r[0, 10:13] = w0
        mem = O.reset_mem(1); mem[0, 7:11] = cmd
        O.physics(r, mem, int(round(T_end / dt)), dt)
        e_pos = np.abs(r[0, 0:3] - ref[0:3]).max()
        e_vel = np.abs(r[0, 7:10] - ref[7:10]).max()
        e_w = np.abs(r[0, 10:13] - R_end @ ref[10:13]).max()             # the state holds WORLD-frame rates
        q = r[0, 3:7]
        e_q = min(np.abs(q - ref[3:7] / np.linalg.norm(ref[3:7])).max(), np.abs(q + ref[3:7] / np.linalg.norm(ref[3:7])).max())
        errs.append(max(e_pos, e_vel, e_w, e_q))
    assert errs[0] < 5e-2 and 1.7 < errs[0] / errs[1] < 2.3 and 1.7 < errs[1] / errs[2] < 2.3, errs


def test_threefry_known_answers_and_noise_moments():
    """The rotor-noise generator (product-defined: the reference's draws are unseeded): Threefry4x32 against the
    published known-answer vectors of Random123 (13 and 20 rounds — same round function, rotation constants and key
    schedule as the 12-round form the product uses), and the N(0,1) moments of the Box-Muller normals built on it."""
    import ctypes
    L = orc.lib()
    U4 = ctypes.c_uint32 * 4
    f = L.orc_threefry4x32
    f.argtypes = [U4, U4, ctypes.c_int]
    f.restype = None
    kat = [
        (13, [0] * 4, [0] * 4, [0x531c7e4f, 0x39491ee5, 0x2c855a92, 0x3d6abf9a]),
        (20, [0] * 4, [0] * 4, [0x9c6ca96a, 0xe17eae66, 0xfc10ecd4, 0x5256a7d8]),
        (20, [0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0, 0x082efa98, 0xec4e6c89],
         [0x59cd1dbb, 0xb8879579, 0x86b5d00c, 0xac8b6d84]),
    ]
    for rounds, ctr, key, want in kat:
        x = U4(*ctr)
        f(x, U4(*key), rounds)
        assert list(x) == want, (rounds, [hex(v) for v in x])
    O = orc.Oracle([params.builtin_type("robobee")])
    z = np.array([O.noise_normals(0x1234ABCD5, i, s, 4) for i in range(4000) for s in range(6)])   # 192 000 normals
    # 8 + 8-bit Box-Muller pairs on the cell centres of the 256 x 256 lattice: variance exactly 1 by construction, kurtosis 2.977, |n| <= 3.535
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1.0) < 0.01
    assert abs((z ** 3).mean()) < 0.03 and abs((z ** 4).mean() - 2.9767) < 0.06        # skewness, kurtosis of the 256 x 256 grid
    assert np.abs(z).max() <= math.sqrt(2 * 1.0013550008475642 * math.log(512.0)) + 1e-9 and np.abs(z).min() > 0.0
    c = np.corrcoef(z.T)                                                               # the 8 normals of a sub-step
    assert np.abs(c - np.eye(8)).max() < 0.035          # 24 000 sub-steps: sampling sigma 0.0065, 28 pairs
    lag = np.corrcoef(z[:-1, 0], z[1:, 0])[0, 1]                                       # consecutive counters: the two halves of one
    assert abs(lag) < 0.035                                                            # block, then the next block
    # an even sub-step and the odd one behind it share a block and draw from different words of it
    a_, b_ = O.noise_normals(9, 5, 10, 4), O.noise_normals(9, 5, 11, 4)
    assert len(set(np.round(np.concatenate([a_, b_]), 12))) == 16
    # the exact population moments of the grid: every (radius byte, angle byte) pair is equally likely
    k = (np.arange(256) + 0.5) / 256.0
    r = np.sqrt(-2.0 * 1.0013550008475642 * np.log(k))
    th = 2 * np.pi * (np.arange(256) + 0.5) / 256.0
    pop = (r[:, None] * np.cos(th)[None, :]).ravel()
    assert abs(pop.var() - 1.0) < 1e-12 and abs(pop.mean()) < 1e-12 and abs((pop ** 4).mean() - 2.9766571967915207) < 1e-9 and np.abs(pop).min() > 7e-4
    # a six-actuator sub-step takes the six normals of its body wrench from its half of a block (rows 6 .. 11: zero)
    h = O.noise_normals(77, 3, 9, 6)
    assert h.shape == (12,) and len(set(np.round(h[:6], 12))) == 6 and np.all(h[6:] == 0.0)


# ---------------------------------------------------------------------------
# DSIM_OPT_PLANE: the ground plane of the reference's world (BaseAviary.py:680) as a product-defined contact model
# (oracle/dsim_oracle.c: orc_plane_contact) — analytic behaviour checks; there is nothing of Bullet to pin it against
# ---------------------------------------------------------------------------
PLANE = 1 << 10


def test_plane_rest_drop_and_liftoff():
    t, O = _O()
    h = t.collision_below
    # resting on the plane with no thrust: stays put
    r, m = _rest(z=h), O.reset_mem(1)
    O.physics(r, m, 240, DT, options=PLANE)
    assert abs(r[0, 2] - h) < 5e-4 and np.abs(r[0, 7:13]).max() < 2e-3 and abs(r[0, 6] - 1) < 1e-6   # (24 Gauss-Seidel sweeps: sub-mm, mrad/s residuals)
    # dropped from half a metre: comes to rest on the plane, never sinks into it by more than a millimetre or so
    r = _rest(z=0.5)
    zmin = 1.0
    for _ in range(360):
        O.physics(r, m, 1, DT, options=PLANE)
        zmin = min(zmin, r[0, 2])
    assert zmin > h - 2e-3 and abs(r[0, 2] - h) < 1e-3 and np.abs(r[0, 7:10]).max() < 5e-3
    # without the option it falls through
    r2 = _rest(z=0.5)
    O.physics(r2, m, 360, DT)
    assert r2[0, 2] < -1.0
    # thrust above weight lifts it off again (contact only pushes)
    m[0, 7:11] = 0.6
    O.physics(r, m, 120, DT, options=PLANE)
    assert r[0, 2] > h + 0.3 and r[0, 9] > 0.5


def test_plane_friction_stops_a_slide_over_the_coulomb_distance():
    t, O = _O()
    h, mu, g = t.collision_below, t.contact_friction, t.gravity
    r, m = _rest(z=h), O.reset_mem(1)
    v0 = 1.0
    r[0, 7] = v0
    x_stop = None
    for k in range(480):
        O.physics(r, m, 1, DT, options=PLANE)
        if x_stop is None and abs(r[0, 7]) < 1e-3:
            x_stop = r[0, 0]
    assert x_stop is not None
    d = v0 * v0 / (2 * mu * g)                                 # 0.102 m; Bullet's velocity damping shortens it a little
    assert 0.85 * d < x_stop < 1.02 * d, (x_stop, d)
    assert abs(r[0, 2] - h) < 1e-3 and np.abs(r[0, 3:6]).max() < 2e-2     # stays on the plane, does not tumble
    # a frictionless plane lets it glide (only the 4 % / s air damping acts)
    import dataclasses
    t0 = dataclasses.replace(t, contact_friction=0.0)
    O0 = orc.Oracle([t0])
    r = _rest(z=h); r[0, 7] = v0
    O0.physics(r, m, 240, DT, options=PLANE)
    assert r[0, 7] > 0.9 * v0


def test_plane_edge_landing_levels_the_vehicle():
    """Dropped tilted by 0.5 rad: the low rim touches first, the contact torque rotates it flat, it ends level at rest."""
    t, O = _O()
    r, m = _rest(z=0.4), O.reset_mem(1)
    r[0, 3:7] = orc.quat_from_euler([0.5, 0.0, 0.3])
    first_touch_z = None
    for k in range(600):
        O.physics(r, m, 1, DT, options=PLANE)
        if first_touch_z is None and abs(r[0, 10:13]).max() > 1e-3:
            first_touch_z = r[0, 2]
    roll, pitch, _ = orc.euler_from_quat(r[0, 3:7])
    # first contact when the low rim reaches the plane: z = h cos(tilt) + r sin(tilt), well above the level rest height
    assert first_touch_z is not None and first_touch_z > t.collision_below + 0.03
    assert abs(roll) < 0.02 and abs(pitch) < 0.02 and abs(r[0, 2] - t.collision_below) < 2e-3
    assert np.abs(r[0, 7:13]).max() < 2e-2


def test_plane_config1_default_flight_touches_down_and_takes_off():
    """examples/fly_INDI.py defaults (start at z = 0.5, initial action 0.4 < hover, controller thrust from 0): with
    the plane on the drone sinks onto it, sits there while the INDI thrust state winds up, lifts off and reaches the
    target — instead of passing through z = 0 to -0.14 m as the plane-less model does."""
    t, O = _O()
    dtc = 5 * DT
    zs = {}
    for opt in (0, PLANE):
        r = _rest(z=0.5); r[0, 1] = 1.0
        m = O.reset_mem(1)
        z = []
        for k in range(720):
            tgt = np.array([[0, 0, 0.5, 0, 0, 0, 0, 0, 0, 0.4 + k / 200.0]])
            a6 = None
            if k == 0:
                a6 = np.zeros((1, 6)); a6[0, :4] = 0.4
            assert O.step(r, m, tgt, 5, DT, dtc, action=a6, options=opt) == 0
            z.append(r[0, 2])
        zs[opt] = np.array(z)
        assert np.abs(r[0, 0:3] - [0, 0, 0.5]).max() < 0.03                 # both end hovering on the target
    assert zs[0].min() < -0.1                                               # no plane: through the floor
    assert t.collision_below - 2e-3 < zs[PLANE].min() < t.collision_below + 0.01    # plane: rests on it
    on_ground = (zs[PLANE] < t.collision_below + 5e-3).sum() * dtc
    assert 0.03 < on_ground < 1.5                                           # for a few control periods, then lifts off


@pytest.mark.parametrize("model", ["robobee", "tello", "hexa_6DOF"])
def test_plane_every_airframe_rests_tips_back_and_lifts_off(model):
    """Each shipped airframe: at rest on its collision cylinder with idle rotors it stays put; put down tilted by
    0.3 rad with a spin it settles level on the plane; above hover thrust it leaves the ground (contact only pushes)."""
    t, O = _O(model)
    h = t.rest_height                                              # (hexa: the base link's COM is 11 mm above the composite's)
    na = t.n_act
    r, m = _rest(z=h), O.reset_mem(1)
    m[0, 7:7 + na] = 0.0
    O.physics(r, m, 240, DT, options=PLANE)
    assert abs(r[0, 2] - h) < 5e-4 and np.abs(r[0, 7:13]).max() < 5e-3
    r = _rest(z=h + t.collision_radius * math.sin(0.3) + 0.01)
    r[0, 3:7] = orc.quat_from_euler([0.3, 0.0, -1.0])
    r[0, 12] = 2.0                                                 # spinning about the vertical: friction stops it
    for _ in range(720):
        O.physics(r, m, 1, DT, options=PLANE)
    roll, pitch, _ = orc.euler_from_quat(r[0, 3:7])
    assert abs(roll) < 0.02 and abs(pitch) < 0.02 and abs(r[0, 2] - h) < 2e-3 and np.abs(r[0, 7:13]).max() < 2e-2
    m[0, 7:7 + na] = min(1.0, 1.3 * t.hover_pwm)
    O.physics(r, m, 120, DT, options=PLANE)
    assert r[0, 2] > h + 0.1 and r[0, 9] > 0.2


def test_hexa_state_is_the_base_link_the_composite_com_is_what_flies():
    """p.getBasePositionAndOrientation reports the BASE link's centre of mass (BaseAviary.py:726-732); the morphing hexa
    is integrated as the rigid composite of its links, whose centre of mass is 11 mm lower (dsim_type_params.base_offset).
    Torque-free, force-free, undamped: the composite COM reconstructed from the state moves in a straight line while the
    reported point circles it, and the reported velocity is v_com + w x (R d)."""
    import dataclasses
    t0 = params.builtin_type("hexa_6DOF")
    d = np.asarray(t0.base_offset)
    assert abs(d[2] - 0.011) < 1e-3 and np.abs(d[:2]).max() < 1e-4
    t = dataclasses.replace(t0, gravity=0.0, lin_damping=0.0, ang_damping=0.0, kf=0.0, km=0.0)
    O = orc.Oracle([t])
    r, m = _rest(z=1.0), O.reset_mem(1)
    w0 = np.array([3.0, 0.0, 0.0])                                   # about a principal axis: stays constant
    vcom = np.array([0.1, -0.2, 0.05])
    r[0, 10:13] = w0
    r[0, 7:10] = vcom + np.cross(w0, d)                              # level: R = I
    com0 = r[0, 0:3] - d
    for k in range(1, 241):
        O.physics(r, m, 1, DT)
        R = np.array(orc.matrix_from_quat(r[0, 3:7])).reshape(3, 3)
        np.testing.assert_allclose(r[0, 0:3] - R @ d, com0 + vcom * k * DT, rtol=0, atol=1e-12)
        np.testing.assert_allclose(r[0, 7:10], vcom + np.cross(r[0, 10:13], R @ d), rtol=0, atol=1e-12)
    np.testing.assert_allclose(r[0, 10:13], w0, rtol=0, atol=1e-12)
    assert abs(np.linalg.norm(r[0, 0:3] - (com0 + vcom * 240 * DT)) - np.linalg.norm(d)) < 1e-12
    # a quad's reported point IS its centre of mass
    assert tuple(params.builtin_type("robobee").base_offset) == (0.0, 0.0, 0.0)
