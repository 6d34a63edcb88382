"""Shared helpers for the parity tests (test infrastructure).

Two error metrics live here.

``rel_err`` — |gpu - oracle| / (|oracle| + 1): the round-1 metric.  For a position near +-50 m it
accepts 5 mm per step, about the size of one step's displacement, so on its own it says little
about the position update.  It is kept for quantities of order one and as a drift measure for
closed-loop trajectories.

``increment_ratio`` / ``assert_step_parity`` — the per-step bar applied to the state INCREMENT
of one step, which is what one launch computes:

    |d_gpu - d_oracle| <= 1e-4 |d_oracle| + k ulp32(M),      d = state_after - state_before,

``M`` = the largest magnitude that enters the fp32 update of that field (the field itself before and
after the step and the largest term added to it: ``step_terms``), ``k`` = a small count of fp32
roundings per physics sub-step (``K_ULP``).  The second term is the floor no fp32 kernel can go
below — the oracle computes the same update in fp64 from the same fp32 inputs — and it is per drone
and per field, never a blanket multiple of the bar: at x = 50 m it is k x 3.8e-6 m, not 5e-3 m.
"""
import math

import numpy as np

from oracle import oracle as orc

REL_TOL = 1e-4          # BASELINE.json north_star: per-step state within 1e-4 relative error
K_ULP = 2.0             # fp32 roundings allowed per physics sub-step / control evaluation, in ulps of the largest term
K_ULP_NOISE = 3.0       # ... with the rotor noise on: every rotor force and moment is then a sum of two terms and the lateral force and
                        # moment components exist at all — half again as many roundings in the wrench (assert_step_parity(noise=True))

# natural scale of each quantity: |gpu - oracle| / (|oracle| + scale) (drift metric, see above)
RIGID_SCALE = np.array([1.0] * 3 + [1.0] * 4 + [1.0] * 3 + [1.0] * 3)      # m, -, m/s, rad/s
MEM_SCALE = np.array([1.0] * 3 + [1.0] * 3 + [1.0] + [1.0] * 6)


def rel_err(got, ref, scale):
    return np.abs(got - ref) / (np.abs(ref) + scale[: ref.shape[1]])


def f32(a):
    """Round to fp32-representable values so GPU (fp32) and oracle (fp64) see identical inputs."""
    return np.asarray(a, dtype=np.float32).astype(np.float64)


def ulp32(x):
    """Spacing of fp32 numbers at |x| (element-wise)."""
    a = np.abs(np.asarray(x, dtype=np.float64)).astype(np.float32)
    return np.spacing(np.maximum(a, np.float32(1e-30))).astype(np.float64)


def step_terms(types, type_id, rigid, mem, tgt, dt_phys, dt_ctrl, substeps, control=True, action=None):
    """Largest magnitude entering the fp32 update of every state field during one fused step, per drone:
    (terms_rigid [n,13], terms_mem [n,13]).  Each entry is a sum of |coefficient| x |operand| over the
    operands of that field's update as the reference writes it (BaseAviary.py:1487-1543 wrench,
    p.stepSimulation, INDIControl.py:278-296, 433-459 / INDIControl_6DOF.py:399-413, 560-628):
      vel      (g + sum_i F_i / m + c (1 + |v|) |v|) dt      gravity and thrust accelerations cancel at hover
      ang vel  (sum_i |r_i| F_i / J_min + gyro |w|^2 + c (1 + |w|) |w|) dt + |w|      the rotor moments cancel pairwise; |w|: R w_b
      quat     1
      thrust   |v| / dt_ctrl + kd (kp |pos_e| + |v*| + |v|) + |a*|     the finite-difference acceleration
      cmd_j    sum_i |alloc_ji| term(nu_i),   nu_{0..2}: krate (katt + |w|) + |w| / dt_ctrl,
               nu_3 (quad): thrust term;  nu_{3..5} (hexa): the acceleration term
    """
    n = rigid.shape[0]
    tr = np.zeros((n, 13))
    tm = np.zeros((n, 13))
    tid = np.zeros(n, dtype=np.int64) if type_id is None else np.asarray(type_id).astype(np.int64)
    if tgt.shape[0] == 1 and n != 1:
        tgt = np.broadcast_to(tgt, (n, tgt.shape[1]))
    v = np.abs(rigid[:, 7:10]).max(1)
    w = np.linalg.norm(rigid[:, 10:13], axis=1)
    for k, t in enumerate(types):
        s = np.flatnonzero(tid == k)
        if s.size == 0:
            continue
        na = t.n_act
        cmd = np.clip((mem[s, 7:7 + na] if action is None else np.asarray(action)[s, :na]),   # the action the physics applies
                      np.asarray(t.pwm_min)[:na], np.asarray(t.pwm_max)[:na])
        rpm = np.asarray(t.pwm2rpm_scale)[:na] * cmd + np.asarray(t.pwm2rpm_const)[:na]
        F = t.kf * rpm ** 2                                               # [m, na]
        arm = np.linalg.norm(np.asarray(t.rotor_pos)[:na], axis=1)
        acc = (t.gravity + F.sum(1) / t.mass)
        alpha = ((F * arm).sum(1) + t.km * (rpm ** 2).sum(1)) / min(t.inertia)
        # Bullet's damping m v (c + c |v|), J w (c + c |w|) and the gyroscopic term w x J w (row P4): negligible in gentle
        # flight, the largest terms of the update near the velocity clamps (tests/util.py:random_fleet, envelope=)
        J = np.asarray(t.inertia, dtype=np.float64)
        gyro = max(abs(J[2] - J[1]) / J[0], abs(J[0] - J[2]) / J[1], abs(J[1] - J[0]) / J[2])
        vn = np.linalg.norm(rigid[s, 7:10], axis=1)
        acc = acc + t.lin_damping * (1.0 + vn) * vn
        alpha = alpha + gyro * w[s] ** 2 + t.ang_damping * (1.0 + w[s]) * w[s]
        tr[s, 0:3] = (v[s] + acc * dt_phys * substeps)[:, None] * dt_phys
        tr[s, 3:7] = 1.0
        # (a vehicle flown as a composite about another point than the reported one — the morphing hexa, params.base_offset —
        # reports v_com + w x (R d): operands of magnitude |w| |d|)
        tr[s, 7:10] = (acc * dt_phys + w[s] * float(np.linalg.norm(getattr(t, 'base_offset', (0.0, 0.0, 0.0)))))[:, None]
        # (every kernel turns w between the world and the body frame — the body-frame loop carries R^T w and stores R w_b — so every
        # world coordinate of w is a sum of three products of magnitude up to |w|: at 130 rad/s the rounding of R alone is worth
        # ulp32(130) in a coordinate that happens to be small)
        tr[s, 10:13] = (alpha * dt_phys + w[s])[:, None]
        if not control:
            continue
        v_new = v[s] + acc * dt_phys * substeps                           # bound on |v| after the physics
        w_new = w[s] + alpha * dt_phys * substeps
        pos_e = np.abs(tgt[s, 0:3] - rigid[s, 0:3]).max(1) + v_new * dt_phys * substeps
        # the finite-difference acceleration (v - last_vel) / dt: after a physics step v carries its own fp32 rounding,
        # amplified by 1 / dt; in a control-only call on given inputs the subtraction of two fp32 numbers is exact
        dv = v_new if dt_phys > 0 else np.abs(rigid[s, 7:10] - mem[s, 0:3]).max(1)
        a_term = dv / dt_ctrl + t.kd_pos * (t.kp_pos * pos_e + np.abs(tgt[s, 3:6]).max(1) + v_new) + np.abs(tgt[s, 6:9]).max(1)
        a_term = np.minimum(a_term, 6.0 + dv / dt_ctrl)                   # the clip at +-6 bounds what survives
        rate_term = (np.max(t.rate_gain) * (np.max(t.att_gain) + w_new) + w_new / dt_ctrl)
        tm[s, 0:3] = tr[s, 7:10]
        tm[s, 3:6] = (2.0 * w_new)[:, None] + tr[s, 10:13]
        tm[s, 6] = a_term
        A = np.abs(np.asarray(t.alloc, dtype=np.float64))                 # [n_act][n_out]
        if t.kind != 1:                                                   # the quad law (on four or six actuators)
            nu = np.stack([rate_term] * 3 + [a_term + np.abs(mem[s, 6])], 1)
            tm[s, 7:7 + na] = nu @ A[:na, :4].T
        else:
            A2 = np.abs(np.asarray(t.wls_first_iteration()[1], dtype=np.float64))
            nu = np.stack([rate_term] * 3 + [a_term] * 3, 1)
            tm[s, 7:13] = nu @ A[:6, :6].T + (A2 @ np.ones(6))[None, :]
    return tr, tm


NOISE_MAX_SIGMA = 4.8546       # |n| <= sqrt(2 corr ln 131072) sigma: the fine Box-Muller lattice of the rotor noise (dsim_device.h:box_muller16; the coarse one ends at 3.535)


_ROTOR_NOISE_MAPS = {}


def hexa_noise_maps(t):
    """Six-actuator types draw the noise of the BODY WRENCH (dsim_device.h:noise_normals): W = L z, z six unit normals, L the Cholesky
    factor of M diag(sigma^2) M^T, M the 6 x 12 map from the per-rotor force / moment noise of BaseAviary.py:1429-1457 to the wrench.
    Returns (L [6,6] as the device holds it, P [12,6]): P z = per-rotor noise values (f[6] in N, m[6] in N m) whose wrench under the
    ORACLE's per-rotor map is L z — what the tests hand to the (unchanged) oracle.  L follows dsim_api.hip:to_dev step by step: M
    from the fp32 images of the rotor axes and of r x a, the factor rounded to fp32 over 0.01."""
    key = (np.asarray(t.rotor_pos, dtype=np.float64).tobytes(), np.asarray(t.rotor_axis, dtype=np.float64).tobytes(),
           np.asarray(t.rotor_spin, dtype=np.float64).tobytes())
    if key in _ROTOR_NOISE_MAPS:
        return _ROTOR_NOISE_MAPS[key]
    r = np.asarray(t.rotor_pos, dtype=np.float64)[:6]
    a = np.asarray(t.rotor_axis, dtype=np.float64)[:6]
    sp = np.asarray(t.rotor_spin, dtype=np.float64)[:6]
    sig = np.array([0.01] * 6 + [0.001] * 6)
    a32, rxa32 = f32(a), f32(np.cross(r, a))
    M32 = np.zeros((6, 12))
    M32[0:3, 0:6], M32[3:6, 0:6], M32[3:6, 6:12] = a32.T, rxa32.T, (sp[:, None] * a32).T
    L = np.linalg.cholesky((M32 * sig) @ (M32 * sig).T)
    L = f32(L / 0.01) * 0.01
    M = np.zeros((6, 12))                                       # the oracle's map: fp64 geometry, actual forces and moments
    M[0:3, 0:6], M[3:6, 0:6], M[3:6, 6:12] = a.T, np.cross(r, a).T, (sp[:, None] * a).T
    S = np.diag(sig ** 2)
    P = S @ M.T @ np.linalg.solve(M @ S @ M.T, L)
    _ROTOR_NOISE_MAPS[key] = (L, P)
    return L, P


def rotor_noise(t, u):
    """(f_noise[n_act], m_noise[n_act]), in N and N m, of one sub-step from the unit normals u = Oracle.noise_normals(...) of the
    launch: quads draw them per rotor (BaseAviary.py:1518-1521); six-actuator types draw the six normals of the body wrench, and the
    per-rotor values returned here add up to exactly that wrench under the oracle's map (hexa_noise_maps)."""
    na = t.n_act
    if na == 4:
        return u[0:4] * 0.01, u[4:8] * 0.001
    n = hexa_noise_maps(t)[1] @ np.asarray(u[0:6], dtype=np.float64)
    return n[0:6], n[6:12]


def noise_terms(types, type_id, n, dt_phys, substeps):
    """([n,13], [n,13]) what the rotor noise adds to the magnitudes of step_terms (BaseAviary.py:1518-1525: f_noise ~ N(0, .01)
    per rotor — and on the lateral axes of every rotor link —, m_noise ~ N(0, .001)): at zero command the noise IS the
    largest term of the velocity and rate updates, and an increment that happens to be small (a draw near zero) must not
    shrink the bar below the rounding of the terms that made it."""
    tr, tm = np.zeros((n, 13)), np.zeros((n, 13))
    tid = np.zeros(n, dtype=np.int64) if type_id is None else np.asarray(type_id).astype(np.int64)
    for k, t in enumerate(types):
        s = tid == k
        fmax, mmax = NOISE_MAX_SIGMA * 0.01, NOISE_MAX_SIGMA * 0.001
        arm = np.linalg.norm(np.asarray(t.rotor_pos)[: t.n_act], axis=1)
        tr[s, 7:10] = t.n_act * fmax / t.mass * dt_phys
        tr[s, 10:13] = (t.n_act * mmax + 2.0 * (arm * fmax).sum()) / min(t.inertia) * dt_phys
        if t.n_act == 6:            # the wrench W = L z, |z_j| <= NOISE_MAX_SIGMA: row sums of |L|
            w = np.abs(hexa_noise_maps(t)[0]).sum(1) * NOISE_MAX_SIGMA
            tr[s, 7:10] = w[0:3].max() / t.mass * dt_phys
            tr[s, 10:13] = w[3:6].max() / min(t.inertia) * dt_phys
        tr[s, 0:3] = tr[s, 7:10] * dt_phys * substeps
        tm[s, 0:3], tm[s, 3:6] = tr[s, 7:10], tr[s, 10:13]
    return tr, tm


def tilt_gain(types, type_id, rigid):
    """[n,13] factor on the ulp term of the controller-memory fields: the quad law's pitch increment is
    w.a / (T cos^2(roll)) (INDIControl.py:314-339: det G = T^2 cos(roll)), so the fp32 rounding of cos(roll) reaches
    the attitude set-point, and through it the cmd fields, amplified by 1 / cos^2(roll).  1 for the thrust row and
    for the 6-DOF law (its target attitude is forced to zero, INDIControl_6DOF.py:495)."""
    n = rigid.shape[0]
    g = np.ones((n, 13))
    tid = np.zeros(n, dtype=np.int64) if type_id is None else np.asarray(type_id).astype(np.int64)
    q = rigid[:, 3:7]
    ra, rb = 2 * (q[:, 1] * q[:, 2] + q[:, 3] * q[:, 0]), q[:, 3] ** 2 - q[:, 0] ** 2 - q[:, 1] ** 2 + q[:, 2] ** 2
    c2 = rb ** 2 / np.maximum(ra ** 2 + rb ** 2, 1e-300)
    for k, t in enumerate(types):
        if t.kind != 1:                        # the quad law, on four or six actuators
            s = tid == k
            g[s, 7:7 + t.n_act] = (1.0 / np.maximum(c2[s], 1e-8))[:, None]
    return g


TINY = 1e-3     # m, m/s, rad/s, PWM: below k ulp32(1e-3) = k x 1.2e-10 a difference is numerical zero (a drone at rest with
                # zero command has rates of 1e-20 on both sides, from cancelling denormal-sized terms)


def increment_ratio(got, ref, prev, terms, k, part=None):
    """|d_got - d_ref| / (REL_TOL (|d_ref| + part) + k ulp32(max(|prev|, |ref|, terms, TINY))): <= 1 passes.
    part: the magnitude of a PART of the increment that is itself only known to REL_TOL — the contribution of an input
    that comes from another fp32 evaluation, e.g. the neighbour-downwash force (hundreds of exp() terms summed in fp32,
    tests/util.py:assert_downwash): where thrust and gravity cancel, |d_ref| says nothing about its size."""
    d_ref = ref - prev
    M = np.maximum(np.maximum(np.maximum(np.abs(prev), np.abs(ref)), terms), TINY)
    return np.abs(got - ref) / (REL_TOL * (np.abs(d_ref) + (0.0 if part is None else part)) + k * ulp32(M))


WORST = {}     # test label -> worst ratio seen (printed by conftest at the end of the session)
WHERE = {}     # test label -> where that worst ratio sits: block, field, the numbers (so that a thin margin can be traced
               # to the term that produces it; written to DSIM_MARGINS_OUT beside the ratios)
RIGID_NAMES = ["x", "y", "z", "qx", "qy", "qz", "qw", "vx", "vy", "vz", "wx", "wy", "wz"]
MEM_NAMES = ["last_vx", "last_vy", "last_vz", "last_p", "last_q", "last_r", "last_thrust", "cmd0", "cmd1", "cmd2", "cmd3", "cmd4", "cmd5"]


PLANE_SWEEPS = 24       # DSIM_PLANE_ITERS / ORC_PLANE_ITERS


def plane_terms(types, type_id, rigid, dt_ctrl, dt_phys=1.0 / 240.0):
    """DSIM_OPT_PLANE: what the contact solve adds to the magnitudes of step_terms, per drone ([n,13], [n,13]); to be
    used with k = K_ULP x sub-steps x (1 + PLANE_SWEEPS) — one more rounding of the contact terms per sweep.
      - a contact point moves at u = v + w x r, |u| <= |v| + |w| R (R: COM to the rim of the collision cylinder); an
        impulse that changes u by du changes v by at most du and w by at most du / (2 rho) (rho = sqrt(J_min / m): the
        maximum of (l / J) / (1 / m + l^2 / J) over the lever arm l);
      - the target velocity of a point is its gap over dt, gap = z + r_z: operands |z| / dt and R / dt, entering once
        per sub-step, not once per sweep (hence the division: k ulp(M) then charges them K_ULP ulps per sub-step);
      - the controller differentiates the new velocity and rates."""
    n = rigid.shape[0]
    tr = np.zeros((n, 13))
    tm = np.zeros((n, 13))
    tid = np.zeros(n, dtype=np.int64) if type_id is None else np.asarray(type_id).astype(np.int64)
    for k, t in enumerate(types):
        s = np.flatnonzero(tid == k)
        R = math.hypot(t.collision_radius, t.collision_below)
        u = np.linalg.norm(rigid[s, 7:10], axis=1) + np.linalg.norm(rigid[s, 10:13], axis=1) * R
        u = u + (np.abs(rigid[s, 2]) + R) / dt_phys / (1 + PLANE_SWEEPS)
        rho = math.sqrt(min(t.inertia) / t.mass)
        tr[s, 7:10] = u[:, None]
        tr[s, 10:13] = (u / (2 * rho))[:, None]
        tr[s, 0:3] = tr[s, 7:10] * dt_phys
        tm[s, 0:3] = tr[s, 7:10]
        tm[s, 3:6] = tr[s, 10:13]
        tm[s, 6] = u / dt_ctrl
        A = np.abs(np.asarray(t.alloc, dtype=np.float64))
        na = t.n_act
        rate = np.max(t.rate_gain) * u / (2 * rho) + u / (2 * rho) / dt_ctrl
        n_out = A.shape[1]
        nu = np.stack([rate] * 3 + [u / dt_ctrl] * (n_out - 3), 1)
        tm[s, 7:7 + na] = nu @ A[:na, :n_out].T
    return tr, tm


def assert_step_parity(label, types, type_id, prev_rigid, prev_mem, tgt, got_rigid, got_mem, ref_rigid, ref_mem,
                       dt_phys, dt_ctrl, substeps, control=True, k=None, action=None, extra_terms=None, part_rigid=None, noise=False):
    """One step of the HIP path against one step of the oracle from the same (fp32-representable) state,
    judged on the increments (module docstring).  got_mem / ref_mem may be None (physics only); action [n, n_act] =
    the explicit action of this step where it is not the stored cmd."""
    k = (K_ULP_NOISE if noise else K_ULP) * max(1, substeps) if k is None else k
    tr, tm = step_terms(types, type_id, prev_rigid, prev_mem, tgt, dt_phys, dt_ctrl, max(1, substeps), control, action)
    if extra_terms is not None:
        tr, tm = tr + extra_terms[0], tm + extra_terms[1]
    rr = increment_ratio(got_rigid, ref_rigid, prev_rigid, tr, k, part_rigid)
    worst = float(rr.max())
    i, f = (int(x) for x in np.unravel_index(rr.argmax(), rr.shape))
    where = ("rigid", i, f)
    detail = dict(field=RIGID_NAMES[f], drone=i, err=float(got_rigid[i, f] - ref_rigid[i, f]), increment=float(ref_rigid[i, f] - prev_rigid[i, f]),
                  value=float(ref_rigid[i, f]), largest_term=float(tr[i, f]), k_ulp=float(k))
    if got_mem is not None:
        kk = k * tilt_gain(types, type_id, ref_rigid)
        # (the controller memory copies the new velocity into last_vel: what is only known to REL_TOL in the velocity — the
        # part of its increment that came from the neighbour-downwash force — is only known to REL_TOL there too)
        part_mem = None
        if part_rigid is not None:
            # ... and the law differentiates it: (v - last_vel) / dt_ctrl is the measured acceleration (INDIControl.py:285-291,
            # INDIControl_6DOF.py:399-413), which goes into the thrust state and, through the allocation, into the commands
            part_mem = np.zeros_like(ref_mem)
            part_mem[:, 0:3] = part_rigid[:, 7:10]
            pa = part_rigid[:, 7:10].sum(1) / dt_ctrl
            part_mem[:, 6] = pa
            tid_ = np.zeros(ref_mem.shape[0], dtype=np.int64) if type_id is None else np.asarray(type_id).astype(np.int64)
            for k_, t_ in enumerate(types):
                s_ = tid_ == k_
                A_ = np.abs(np.asarray(t_.alloc, dtype=np.float64))
                col = A_[: t_.n_act, 3] if t_.kind != 1 else A_[:6, 3:6].sum(1)
                part_mem[np.ix_(s_, 7 + np.arange(t_.n_act))] = pa[s_, None] * col[None, :]
        rm = increment_ratio(got_mem, ref_mem, prev_mem, tm, kk, part_mem)
        if rm.max() > worst:
            worst = float(rm.max())
            i, f = (int(x) for x in np.unravel_index(rm.argmax(), rm.shape))
            where = ("mem", i, f)
            detail = dict(field=MEM_NAMES[f], drone=i, err=float(got_mem[i, f] - ref_mem[i, f]), increment=float(ref_mem[i, f] - prev_mem[i, f]),
                          value=float(ref_mem[i, f]), largest_term=float(tm[i, f]), k_ulp=float(np.broadcast_to(kk, rm.shape)[i, f]))
    if worst >= WORST.get(label, 0.0):
        detail["ulp32_of_M"] = float(ulp32(max(abs(detail["value"]), abs(detail["value"] - detail["increment"]), detail["largest_term"], TINY)))
        WHERE[label] = detail
    WORST[label] = max(WORST.get(label, 0.0), worst)
    assert worst <= 1.0, (label, worst, where)
    return worst


def control_bound(types, type_id, rigid, prev_mem, tgt, ref_mem, dt_ctrl, k=K_ULP):
    """Per-case tolerance [n,13] of one computeControl call on the controller memory (last_vel3 last_rates3
    last_thrust cmd6): REL_TOL |d_ref| + k ulp32(M), M from step_terms with no physics in front."""
    _, tm = step_terms(types, type_id, rigid, prev_mem, tgt, 0.0, dt_ctrl, 1, True)
    M = np.maximum(np.maximum(np.maximum(np.abs(prev_mem), np.abs(ref_mem)), tm), TINY)
    return REL_TOL * np.abs(ref_mem - prev_mem) + k * tilt_gain(types, type_id, rigid) * ulp32(M)


def assert_control_parity(label, types, type_id, rigid, prev_mem, tgt, got_mem, ref_mem, dt_ctrl, k=K_ULP, slack=None):
    """computeControl on the HIP path against the oracle from the same fp32-representable inputs, judged on the
    increments of the controller memory.  slack [n,13] (optional) is added per case — used against the reference's
    fp64-input goldens, where it is |oracle(fp32-rounded inputs) - golden|, the measured effect of input rounding."""
    tol = control_bound(types, type_id, rigid, prev_mem, tgt, ref_mem, dt_ctrl, k)
    if slack is not None:
        tol = tol + slack
    ratio = np.abs(got_mem - ref_mem) / tol
    worst = float(ratio.max())
    WORST[label] = max(WORST.get(label, 0.0), worst)
    assert worst <= 1.0, (label, worst, tuple(int(x) for x in np.unravel_index(ratio.argmax(), ratio.shape)))
    return worst


ENVELOPES = ("omega_clamp", "vel_clamp", "pi4", "tiny_omega", "non_unit", "tumbling", "wreck")


def random_fleet(rng, n, n_act=4, tilt=0.5, speed=2.0, rate=1.5, spread=50.0, envelope=None):
    """Seeded in-flight fleet state: rigid [n,13], mem [n,13], targets [n,10] (fp32-representable).

    envelope: None = gentle flight (the defaults above), or one of ENVELOPES — the corners of Bullet's floating-base
    step (row P4, BaseAviary.py:542-543) that gentle flight never reaches:
      "omega_clamp"  angular-velocity coordinates at 90-130 rad/s, mixed with lanes that have ONE such coordinate and
                     lanes just under 100 rad/s (applyDeltaVeeMultiDof's clamp of every world coordinate to +-100; the
                     body-frame loop's world-frame detour triggers on |w_b| >= 100)
      "vel_clamp"    velocity coordinates at 95-105 m/s (the same clamp on the linear coordinates; damping pulls the
                     ones under 100 away from it, thrust pushes some onto it)
      "pi4"          |w| up to 170 rad/s in random directions: with dt_phys = 1/100 or 1/60 s the rotation per sub-step
                     exceeds pi/4 and Bullet's angular-motion clamp engages (unreachable at 240 Hz)
      "tiny_omega"   |w| < 1e-3 rad/s, some lanes exactly zero (Bullet's Taylor branch of the exponential map)
      "non_unit"     quaternions of length 0.5-1.5 (the helpers do not normalise; Bullet's step does)
      "tumbling"     attitudes over the whole sphere (tilt to pi), both signs of w
      "wreck"        all of it at once: a tumbling lane with clamped rates and velocities and a non-unit quaternion"""
    assert envelope is None or envelope in ENVELOPES, envelope
    rpy = np.stack([rng.uniform(-tilt, tilt, n), rng.uniform(-tilt, tilt, n), rng.uniform(-math.pi, math.pi, n)], 1)
    quat = np.stack([orc.quat_from_euler(r) for r in rpy]) if n <= 20000 else _quat_from_euler_np(rpy)
    pos = np.concatenate([rng.uniform(-spread, spread, (n, 2)), rng.uniform(0.5, 20.0, (n, 1))], 1)
    vel = rng.uniform(-speed, speed, (n, 3))
    om = rng.uniform(-rate, rate, (n, 3))
    sign = lambda shape: np.where(rng.uniform(size=shape) < 0.5, -1.0, 1.0)
    if envelope in ("tumbling", "wreck"):
        quat = rng.normal(size=(n, 4))
        quat /= np.linalg.norm(quat, axis=1, keepdims=True)
    if envelope in ("non_unit", "wreck"):
        quat = quat * rng.uniform(0.5, 1.5, (n, 1))
    if envelope in ("omega_clamp", "wreck"):
        big = sign((n, 3)) * rng.uniform(90.0, 130.0, (n, 3))
        kind = rng.integers(0, 4, n)                      # 0, 1: every coordinate; 2: one coordinate; 3: |w| just under 100
        one = np.zeros((n, 3)); one[np.arange(n), rng.integers(0, 3, n)] = 1.0
        om = np.where((kind <= 1)[:, None], big, np.where((kind == 2)[:, None], one * big + (1 - one) * om, om))
        d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
        om = np.where((kind == 3)[:, None], d * rng.uniform(97.0, 101.0, (n, 1)), om)
    if envelope in ("vel_clamp", "wreck"):
        vel = sign((n, 3)) * rng.uniform(95.0, 105.0, (n, 3))
    if envelope == "pi4":
        d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
        om = d * rng.uniform(40.0, 170.0, (n, 1))          # pi / (4 dt) = 78.5 rad/s at 100 Hz, 47.1 at 60 Hz
    if envelope == "tiny_omega":
        om = rng.uniform(-1.0, 1.0, (n, 3)) * (1e-3 / math.sqrt(3.0)) * rng.uniform(0.0, 1.0, (n, 1))
        om[rng.uniform(size=n) < 0.1] = 0.0
    rigid = f32(np.concatenate([pos, quat, vel, om], 1))
    mem = np.zeros((n, 13))
    mem[:, 0:3] = rigid[:, 7:10] + rng.uniform(-0.02, 0.02, (n, 3))
    mem[:, 3:6] = rng.uniform(-rate, rate, (n, 3))
    mem[:, 6] = rng.uniform(0.0, 1.0, n)
    mem[:, 7:7 + n_act] = rng.uniform(0.3, 0.7, (n, n_act))
    mem = f32(mem)
    tgt = np.concatenate([rigid[:, 0:3] + rng.uniform(-1, 1, (n, 3)), rng.uniform(-0.5, 0.5, (n, 3)),
                          rng.uniform(-0.5, 0.5, (n, 3)), rng.uniform(-4, 4, (n, 1))], 1)
    return rigid, mem, f32(tgt)


def _quat_from_euler_np(rpy):
    h = rpy / 2.0
    s, c = np.sin(h), np.cos(h)
    q = np.stack([
        s[:, 0] * c[:, 1] * c[:, 2] - c[:, 0] * s[:, 1] * s[:, 2],
        c[:, 0] * s[:, 1] * c[:, 2] + s[:, 0] * c[:, 1] * s[:, 2],
        c[:, 0] * c[:, 1] * s[:, 2] - s[:, 0] * s[:, 1] * c[:, 2],
        c[:, 0] * c[:, 1] * c[:, 2] + s[:, 0] * s[:, 1] * s[:, 2]], 1)
    return q / np.linalg.norm(q, axis=1, keepdims=True)


def attitude_zoo(rng, n_random=2000):
    """Quaternions (xyzw, fp32-representable) that exercise every branch of p.getEulerFromQuaternion
    (BaseAviary.py:729): random attitudes over the whole sphere, both signs of w, both gimbal branches
    (|sarg| >= 0.99999, exactly +-90 deg pitch and just inside the clamp), attitudes just OUTSIDE the clamp
    (|sarg| a few 1e-6 under 0.99999: the ill-conditioned end of asin), and non-unit quaternions (scaled
    0.5 .. 1.5: the helper does not normalise).  Cases within fp32 rounding of the branch threshold are
    excluded by construction — there Bullet's own function is discontinuous (pitch jumps by 4.5e-3 rad)."""
    out = []
    q = rng.normal(size=(n_random, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    out.append(q)                                                     # w of both signs
    rpy = np.stack([rng.uniform(-math.pi, math.pi, 400), np.zeros(400), rng.uniform(-math.pi, math.pi, 400)], 1)
    s_in = np.concatenate([np.full(100, 1.0), 1.0 - rng.uniform(0, 8e-6, 100)])            # inside the clamp
    s_out = 0.99999 - rng.uniform(3e-6, 5e-5, 200)                                         # just outside
    sarg = np.concatenate([s_in, s_out]) * np.where(rng.uniform(size=400) < 0.5, -1.0, 1.0)
    rpy[:, 1] = np.arcsin(sarg)
    out.append(_quat_from_euler_np(rpy))
    q = rng.normal(size=(400, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    out.append(q * rng.uniform(0.5, 1.5, (400, 1)))                   # non-unit
    q = f32(np.concatenate(out))
    sarg = -2.0 * (q[:, 0] * q[:, 2] - q[:, 3] * q[:, 1])
    keep = np.abs(np.abs(sarg) - 0.99999) > 1e-6                      # off the discontinuity
    return q[keep]


# ---------------------------------------------------------------------------------------------------------------------
# neighbour downwash (formula P8): tolerance per receiver
# ---------------------------------------------------------------------------------------------------------------------
FLT_MIN = 1.17549435e-38     # smallest normal fp32: v_exp_f32 flushes what lies below it to zero


def p8_terms(types, type_id, recv_pos, world_pos, cutoff=10.0, chunk=256):
    """Per receiver of formula P8 (BaseAviary.py:1736-1763; oracle/dsim_oracle.c:orc_downwash), in fp64 numpy:
    (n_pairs, largest |term|, straddle, flush) — the number of contributing pairs, the largest single term, the summed
    magnitude of the terms of pairs that sit within 2e-5 m of the cut-off circle or of dz = 0, where the fp32 and the
    fp64 evaluation of `dz > 0 and dxy < 10` may legitimately disagree, and sum(alpha) x FLT_MIN: a term is alpha x
    exp(..), and an exponential below the smallest normal fp32 is flushed to zero by the hardware while fp64 keeps it
    (forces of 1e-30 N: "no force" either way)."""
    recv_pos, world_pos = np.asarray(recv_pos, np.float64), np.asarray(world_pos, np.float64)
    n = recv_pos.shape[0]
    tid = np.zeros(n, dtype=np.int64) if type_id is None else np.asarray(type_id).astype(np.int64)
    d0 = np.array([t.dw_coeff[0] for t in types])[tid]; d1 = np.array([t.dw_coeff[1] for t in types])[tid]
    d2 = np.array([t.dw_coeff[2] for t in types])[tid]; pr = np.array([t.prop_radius for t in types])[tid]
    n_pairs, t_max, straddle, flush = np.zeros(n, dtype=np.int64), np.zeros(n), np.zeros(n), np.zeros(n)
    chunk = max(1, min(chunk, (1 << 23) // max(1, world_pos.shape[0])))          # <= 8 M pairs (64 MB per temporary) at a time
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        dz = world_pos[None, :, 2] - recv_pos[s:e, None, 2]
        dxy = np.hypot(world_pos[None, :, 0] - recv_pos[s:e, None, 0], world_pos[None, :, 1] - recv_pos[s:e, None, 1])
        near = (dz > -2e-5) & (dxy < cutoff + 2e-5)
        dzs = np.where(dz > 1e-12, dz, 1.0)
        alpha = d0[s:e, None] * (pr[s:e, None] / (4 * dzs)) ** 2
        term = alpha * np.exp(-0.5 * (dxy / (d1[s:e, None] * dzs + d2[s:e, None])) ** 2)
        hit = (dz > 0) & (dxy < cutoff)
        flush[s:e] = np.where(hit, alpha, 0.0).sum(1) * FLT_MIN
        edge = near & ((np.abs(dxy - cutoff) < 2e-5) | (np.abs(dz) < 2e-5))
        n_pairs[s:e] = hit.sum(1)
        t_max[s:e] = np.where(hit, term, 0.0).max(1)
        straddle[s:e] = np.where(edge & (dz > 1e-12), term, 0.0).sum(1)
    return n_pairs, t_max, straddle, flush


def assert_downwash(label, got, ref, types, type_id, recv_pos, world_pos):
    """|got - ref| <= 1e-4 |ref| + n_pairs ulp32(largest term) (+ what straddles the cut-off, + the flushed exponentials),
    per receiver.  The terms of one receiver all have the same sign, so |ref| is the sum of their magnitudes: the bar
    is relative to what was summed, with no absolute floor that would hide the weak far-field terms."""
    n_pairs, t_max, straddle, flush = p8_terms(types, type_id, recv_pos, world_pos)
    tol = REL_TOL * np.abs(ref) + n_pairs * ulp32(np.maximum(t_max, 1e-300)) * (t_max > 0) + straddle + flush + 1e-300
    ratio = np.abs(np.asarray(got, np.float64) - ref) / tol
    worst = float(ratio.max())
    WORST[label] = max(WORST.get(label, 0.0), worst)
    assert worst <= 1.0, (label, worst, int(ratio.argmax()), float(got[ratio.argmax()]), float(ref[ratio.argmax()]))
    return worst
