"""Oracle (oracle/dsim_oracle.c, fp64) against the golden vectors produced by the
reference's own controller code (tests/golden/make_goldens.py).  CPU only."""
import os
import sys

import numpy as np
import pytest

from dronesim_amd import params
from oracle import oracle as orc

TOL = 1e-9  # fp64 restatement vs fp64 reference; the SVD pinv differs from LAPACK's at ~1e-13


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_math_helpers(golden_dir):
    g = _load(golden_dir, "math_helpers.npz")
    for a, b, ic, co, wr in zip(g["q1"], g["q2"], g["quat_inv_comp"], g["quat_comp"], g["quat_wrap_shortest"]):
        np.testing.assert_array_equal(orc.quat_inv_comp(a, b), ic)   # same arithmetic, bit-exact
        np.testing.assert_array_equal(orc.quat_comp(a, b), co)
        np.testing.assert_array_equal(orc.quat_wrap_shortest(ic), wr)
    for a, na in zip(g["ang"], g["norm_ang"]):
        assert orc.norm_ang(a) == na
    for q, e, R in zip(g["q1"], g["euler_standin"], g["matrix_standin"]):
        np.testing.assert_allclose(orc.euler_from_quat(q), e, rtol=0, atol=1e-15)
        np.testing.assert_allclose(orc.matrix_from_quat(q).ravel(), R, rtol=0, atol=1e-15)


def test_pinv_matches_numpy():
    rng = np.random.default_rng(0)
    for shape in [(3, 3), (4, 4), (6, 6), (12, 6), (12, 3), (4, 6)]:
        A = rng.normal(size=shape)
        np.testing.assert_allclose(orc.pinv(A), np.linalg.pinv(A), rtol=1e-10, atol=1e-12)
    # rank-deficient: numpy zeroes singular values below rcond*smax
    A = np.array([[1.0, 2, 3], [2, 4, 6], [1, 0, 1]])
    np.testing.assert_allclose(orc.pinv(A), np.linalg.pinv(A), rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("model", ["robobee", "tello", "hexa_6DOF_simple"])
def test_ctrl_params_match_reference(golden_dir, model):
    g = _load(golden_dir, f"ctrl_params_{model}.npz")
    t = params.builtin_type(model)
    np.testing.assert_array_equal(t.G1, g["G1"])
    assert t.kp_pos == g["kp"] and t.kd_pos == g["kd"]
    np.testing.assert_array_equal(t.att_gain, g["att"])
    np.testing.assert_array_equal(t.rate_gain, g["rate"])
    np.testing.assert_array_equal(t.pwm2rpm_scale, g["pwm2rpm_scale"])
    np.testing.assert_array_equal(t.pwm_min, g["min_pwm"])
    np.testing.assert_array_equal(t.pwm_max, g["max_pwm"])
    assert t.ctrl_mass == g["m"] and t.kf == g["kf"] and t.km == g["km"]
    np.testing.assert_allclose(t.alloc, g["pinv_G1"], rtol=0, atol=0)
    np.testing.assert_allclose(orc.pinv(t.G1 / 0.05), g["pinv_G1"], rtol=1e-12, atol=1e-15)


def _mem_from_case(g, i, n_act):
    mem = np.zeros(13)
    mem[0:3], mem[3:6], mem[6] = g["last_vel"][i], g["last_rates"][i], g["last_thrust"][i]
    mem[7:7 + n_act] = g["cmd"][i]
    return mem


@pytest.mark.parametrize("model", ["robobee", "tello", "hexa_6DOF_simple"])
def test_indi_single_call(golden_dir, model):
    """The quad law of INDIControl.py — on four actuators, and on the six of hexa_6DOF_simple (examples/
    fly_hexa_6DOF_simple.py:18: G1 is 4 x 6, pinv(G1 / 0.05) 6 x 4, all six commands incremented and clipped)."""
    g = _load(golden_dir, f"indi_single_{model}.npz")
    t = params.builtin_type(model)
    O = orc.Oracle([t])
    n = g["pos"].shape[0]
    na = t.n_act
    assert g["cmd_out"].shape[1] == na
    for i in range(n):
        # C2 alone
        mem = _mem_from_case(g, i, na)
        thrust, te, pe = orc.indi_position(t, float(g["dt"][i]), g["pos"][i], g["quat"][i], g["vel"][i],
                                           g["target_pos"][i], g["target_rpy"][i], g["target_vel"][i],
                                           g["target_acc"][i], mem)
        scale = 1.0 + abs(g["pc_thrust"][i])
        assert abs(thrust - g["pc_thrust"][i]) <= TOL * scale, (i, thrust, g["pc_thrust"][i])
        np.testing.assert_allclose(te, g["pc_target_euler"][i], rtol=0, atol=TOL * (1 + np.abs(te).max()))
        np.testing.assert_array_equal(pe, g["pos_e"][i])
    # C1 batch
    rigid = np.concatenate([g["pos"], g["quat"], g["vel"], g["ang_vel"]], 1)
    mem = np.stack([_mem_from_case(g, i, na) for i in range(n)])
    tgt = np.concatenate([g["target_pos"], g["target_vel"], g["target_acc"], g["target_rpy"][:, 2:3]], 1)
    for dt in np.unique(g["dt"]):
        sel = np.where(g["dt"] == dt)[0]
        r, m, tg = rigid[sel].copy(), mem[sel].copy(), tgt[sel].copy()
        rc, pos_e, yaw_e = O.control(r, m, tg, float(dt))
        assert rc == 0
        np.testing.assert_allclose(m[:, 7:7 + na], g["cmd_out"][sel], rtol=0, atol=1e-8)
        np.testing.assert_array_equal(pos_e, g["pos_e"][sel])
        np.testing.assert_allclose(yaw_e, g["yaw_e"][sel], rtol=0, atol=1e-12)
        np.testing.assert_array_equal(m[:, 0:3], g["last_vel_out"][sel])
        np.testing.assert_allclose(m[:, 3:6], g["last_rates_out"][sel], rtol=0, atol=1e-14)
        np.testing.assert_allclose(m[:, 6], g["last_thrust_out"][sel], rtol=0,
                                   atol=TOL * (1 + np.abs(g["last_thrust_out"][sel]).max()))


@pytest.mark.parametrize("model", ["robobee", "tello", "hexa_6DOF_simple"])
def test_indi_sequence(golden_dir, model):
    g = _load(golden_dir, f"indi_sequence_{model}.npz")
    t = params.builtin_type(model)
    O = orc.Oracle([t])
    S, K = g["pos"].shape[:2]
    mem = O.reset_mem(S)
    for k in range(K):
        rigid = np.concatenate([g["pos"][:, k], g["quat"][:, k], g["vel"][:, k], g["ang_vel"][:, k]], 1)
        tgt = np.concatenate([g["target_pos"][:, k], g["target_vel"][:, k], g["target_acc"][:, k],
                              g["target_rpy"][:, k, 2:3]], 1)
        rc, pos_e, yaw_e = O.control(rigid.copy(), mem, tgt, float(g["dt"]))
        assert rc == 0
        np.testing.assert_allclose(mem[:, 7:7 + t.n_act], g["cmd_out"][:, k], rtol=0, atol=1e-8, err_msg=f"step {k}")
        np.testing.assert_allclose(yaw_e, g["yaw_e"][:, k], rtol=0, atol=1e-12)
        np.testing.assert_allclose(mem[:, 6], g["last_thrust_out"][:, k], rtol=0, atol=1e-8)


def test_wls_alloc_main_case(golden_dir):
    """The reference's only known-answer check (wls_alloc.py:381-408, Matlab lsqlin)."""
    g = _load(golden_dir, "wls_alloc.npz")
    u, it, rc = orc.wls_alloc(g["main_v"], g["main_umin"], g["main_umax"], g["main_B"], Wv=g["main_Wv"],
                              Wu=None, up=g["main_up"])
    assert rc == 0 and it == int(g["main_iter"]) == 6
    np.testing.assert_allclose(u, g["main_du"], rtol=1e-9, atol=1e-6)
    matlab = np.array([-4614.0, 426.064612091305, 5390.0, -4614.0, -4210.0, 5390.0])
    np.testing.assert_allclose(u, matlab, rtol=1e-8, atol=1e-5)


def test_wls_alloc_hexa_cases(golden_dir):
    g = _load(golden_dir, "wls_alloc.npz")
    multi = 0
    for i in range(g["hexa_v"].shape[0]):
        lo = (0.0 - g["hexa_cmd"][i]) * g["hexa_scale"][i]
        hi = (1.0 - g["hexa_cmd"][i]) * g["hexa_scale"][i]
        u, it, rc = orc.wls_alloc(g["hexa_v"][i], lo, hi, g["hexa_B"], Wv=g["hexa_Wv"], Wu=g["hexa_Wu"])
        assert (rc == 0) == bool(g["hexa_ok"][i]), i
        if rc == 0:
            assert it == g["hexa_iter"][i], (i, it, g["hexa_iter"][i])
            multi += it > 1
            np.testing.assert_allclose(u, g["hexa_du"][i], rtol=1e-7, atol=1e-7 * g["hexa_scale"][i])
    assert multi >= 5  # the fixture really exercises the active-set loop


def test_indi_6dof_single_call(golden_dir):
    g = _load(golden_dir, "indi_single_hexa_6DOF.npz")
    gp = _load(golden_dir, "ctrl_params_hexa_6DOF.npz")
    # controller constants only (physics constants of the hexa are not needed here)
    t = params.DroneType(
        name="hexa_6DOF", kind=params.KIND_HEXA6DOF, n_act=6, mass=0.86, inertia=(1, 1, 1),
        kf=float(gp["kf"]), km=float(gp["km"]), pwm2rpm_scale=gp["pwm2rpm_scale"],
        pwm2rpm_const=gp["pwm2rpm_const"], pwm_min=gp["min_pwm"], pwm_max=gp["max_pwm"],
        rotor_pos=[(0, 0, 0)] * 6, rotor_axis=[(0, 0, 1)] * 6, rotor_spin=[1] * 6, G1=gp["G1"],
        kp_pos=float(gp["kp"]), kd_pos=float(gp["kd"]), att_gain=gp["att"], rate_gain=gp["rate"],
        reset_thrust=0.3, reset_cmd=0.5)
    O = orc.Oracle([t])
    n = g["pos"].shape[0]
    rigid = np.concatenate([g["pos"], g["quat"], g["vel"], g["ang_vel"]], 1)
    mem = np.stack([_mem_from_case(g, i, 6) for i in range(n)])
    tgt = np.concatenate([g["target_pos"], g["target_vel"], g["target_acc"], g["target_rpy"][:, 2:3]], 1)
    for dt in np.unique(g["dt"]):
        sel = np.where(g["dt"] == dt)[0]
        r, m, tg = rigid[sel].copy(), mem[sel].copy(), tgt[sel].copy()
        rc, pos_e, yaw_e = O.control(r, m, tg, float(dt))
        assert rc == 0
        np.testing.assert_allclose(m[:, 7:13], g["cmd_out"][sel], rtol=0, atol=1e-7)
        np.testing.assert_allclose(yaw_e, g["yaw_e"][sel], rtol=0, atol=1e-12)
        np.testing.assert_allclose(m[:, 6], g["last_thrust_out"][sel], rtol=0, atol=1e-8)


def test_trajectory_sampler_matches_reference_table(golden_dir):
    """orc_traj_sample (trajGen.get_des_state + stateful get_yaw) against the 1200-row table the
    reference's own trajGenerator produced for the fly_INDI_TrajectoryTrack gates."""
    g = _load(golden_dir, "traj_track_waypoints.npz")
    ys = np.zeros(3)
    for k, t in enumerate(g["t"]):
        o = orc.traj_sample(g["coeffs"], g["TS"], t, ys)
        np.testing.assert_allclose(o[0:3], g["target_pos"][k], rtol=0, atol=1e-9)
        np.testing.assert_allclose(o[3:6], g["target_vel"][k], rtol=0, atol=1e-9)
        np.testing.assert_allclose(o[6:9], g["target_acc"][k], rtol=0, atol=1e-9)
        assert abs(o[9] - g["target_yaw"][k]) < 1e-5, k   # yaw integrates acos() of nearly parallel headings (ill-conditioned)


def test_golden_fixtures_regenerate_bit_identically(tmp_path):
    """The committed fixtures are exactly what tests/golden/make_goldens.py produces from the reference tree
    (runs only where /root/reference is mounted: the build container, not the GPU box)."""
    import glob
    import importlib.util
    import os
    if not os.path.isdir("/root/reference/dronesim"):
        pytest.skip("reference tree not present")
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("make_goldens", os.path.join(here, "make_goldens.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    mg.OUT = str(tmp_path)
    saved = {k: sys.modules.get(k) for k in ("pybullet", "pybullet_data", "gym", "gym.spaces")}
    argv = sys.argv
    import warnings
    try:
        sys.argv = ["make_goldens.py"]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")             # the reference's own numpy deprecations
            mg.main()
    finally:
        sys.argv = argv
        for k, v in saved.items():                      # the generator's stand-in modules must not leak into other tests
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    files = sorted(glob.glob(os.path.join(here, "*.npz")))
    assert len(files) == 18
    for f in files:
        a, b = np.load(f, allow_pickle=True), np.load(os.path.join(str(tmp_path), os.path.basename(f)), allow_pickle=True)
        assert set(a.files) == set(b.files), f
        for k in a.files:
            assert np.array_equal(a[k], b[k], equal_nan=a[k].dtype.kind == "f"), (f, k)


def test_euler_and_matrix_conventions_match_scipy():
    """Row C8: the three PyBullet math helpers are restated (nothing of Bullet is available here).  Their
    convention — quaternion xyzw, roll-pitch-yaw = rotations about the FIXED x, y, z axes in that order
    (URDF / Bullet 'rpy') — is cross-checked against an independent implementation, scipy's Rotation."""
    from scipy.spatial.transform import Rotation as R
    rng = np.random.default_rng(7)
    rpy = np.stack([rng.uniform(-np.pi, np.pi, 500), rng.uniform(-1.5, 1.5, 500), rng.uniform(-np.pi, np.pi, 500)], 1)
    for e in rpy:
        q = orc.quat_from_euler(e)
        qs = R.from_euler("xyz", e).as_quat()                       # extrinsic xyz, scalar-last
        assert min(np.abs(q - qs).max(), np.abs(q + qs).max()) < 1e-14
        np.testing.assert_allclose(orc.matrix_from_quat(q), R.from_quat(q).as_matrix(), rtol=0, atol=1e-14)
        back = orc.euler_from_quat(q)
        np.testing.assert_allclose(back, R.from_quat(q).as_euler("xyz"), rtol=0, atol=1e-9)
        np.testing.assert_allclose(back, e, rtol=0, atol=1e-9)


def test_roll_sweep_through_the_pinv_singularity_vs_reference(golden_dir):
    """det G = T^2 cos(roll): the reference's np.linalg.pinv (INDIControl.py:336) is swept through roll = +-90 deg
    (tests/golden/indi_roll_sweep.npz, |roll| - 90 deg from 1e-1 rad down to 0).  The oracle's SVD pinv follows the
    reference to rounding down to 1e-4 rad and stays with it (ill-conditioned, but the same regime) below."""
    g = _load(golden_dir, "indi_roll_sweep.npz")
    O = orc.Oracle([params.builtin_type("robobee")])
    n = len(g["roll"])
    rigid = np.concatenate([g["pos"], g["quat"], g["vel"], g["ang_vel"]], 1)
    mem = np.zeros((n, 13))
    mem[:, 0:3], mem[:, 3:6], mem[:, 6], mem[:, 7:11] = g["last_vel"], g["last_rates"], g["last_thrust"], g["cmd"]
    tgt = np.concatenate([g["target_pos"], np.zeros((n, 6)), g["target_yaw"][:, None]], 1)
    rc, _, _ = O.control(rigid, mem, tgt, float(g["dt"]))
    assert rc == 0 and np.isfinite(mem).all()
    d = np.abs(np.abs(g["roll"]) - np.pi / 2)
    err = np.abs(mem[:, 7:11] - g["cmd_out"]).max(1)
    assert err[d >= 1e-4].max() < 1e-10 and err[d < 1e-4].max() < 5e-3
    np.testing.assert_allclose(mem[:, 6], g["last_thrust_out"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(mem[:, 3:6], g["last_rates_out"], rtol=0, atol=1e-13)
