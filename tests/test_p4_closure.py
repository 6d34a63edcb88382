"""P4 closure kit (SURVEY.md 8c / VERDICT r01 item 9): the day tests/golden/pybullet_fly_INDI.npz exists — one noise-free
flight of examples/fly_INDI.py recorded from the reference's own PyBullet run by tools/record_pybullet_trajectory.py —
these tests pin the integrator restatement (oracle: orc_bullet_step; device: bullet_step) and the three Euler /
quaternion helpers (C8) against the engine.  PyBullet cannot be installed here, nothing is stubbed in its place, so the
tests are skipped while the fixture is absent."""
import os

import numpy as np
import pytest

from dronesim_amd import params
from oracle import oracle as orc

FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pybullet_fly_INDI.npz")
needs_fixture = pytest.mark.skipif(not os.path.exists(FIXTURE),
                                   reason="tests/golden/pybullet_fly_INDI.npz absent: record it with "
                                          "tools/record_pybullet_trajectory.py on a machine that has pybullet")
DT = 1.0 / 240.0


def _load():
    g = np.load(FIXTURE, allow_pickle=False)
    K = int(min(g["first_contact_step"], g["state"].shape[0]))      # the in-flight part: pinned exactly; the contact part below
    return g, K


def _rigid(state_row):
    return np.concatenate([state_row[0:7], state_row[10:16]])


@needs_fixture
def test_oracle_env_step_matches_pybullet_step_by_step():
    """Every Env.step of the recording: PyBullet's state before + the recorded action -> the oracle's P1-P4 -> PyBullet's
    state after, at 1e-9 (fp64 vs fp64: the restated step IS Bullet's step or it is not), rpy column included (C8)."""
    g, K = _load()
    t = params.builtin_type(str(g["drone"]))
    O = orc.Oracle([t])
    prev = np.concatenate([g["init_xyz"][0], orc.quat_from_euler(g["init_rpy"][0]), np.zeros(6)])
    assert K >= 8, "the recording reaches the ground too early to pin anything"
    for k in range(K):
        r = prev[None, :].copy()
        a6 = np.zeros((1, 6)); a6[0, :4] = g["action"][k]
        O.physics(r, O.reset_mem(1), int(g["aggr"]), DT, action=a6)
        want = g["state"][k]
        np.testing.assert_allclose(r[0], _rigid(want), rtol=0, atol=1e-9, err_msg=f"step {k}")
        np.testing.assert_allclose(orc.euler_from_quat(r[0, 3:7]), want[7:10], rtol=0, atol=1e-9, err_msg=f"rpy, step {k}")
        prev = _rigid(want)


@needs_fixture
def test_oracle_closed_loop_reproduces_the_recorded_commands():
    """The whole loop (env.step -> computeControlFromState) from the initial state alone reproduces the recorded
    commands and controller memory: closes C8's use inside the controller too."""
    g, K = _load()
    t = params.builtin_type(str(g["drone"]))
    O = orc.Oracle([t])
    rigid = np.concatenate([g["init_xyz"][0], orc.quat_from_euler(g["init_rpy"][0]), np.zeros(6)])[None, :]
    mem = O.reset_mem(1)
    dtc = float(g["aggr"]) * DT
    for k in range(K):
        a6 = np.zeros((1, 6)); a6[0, :4] = 0.4 if k == 0 else mem[0, 7:11]
        O.physics(rigid, mem, int(g["aggr"]), DT, action=a6)
        tgt = np.array([[0, 0, 0.5, 0, 0, 0, 0, 0, 0, g["target_yaw"][k]]])
        assert O.control(rigid, mem, tgt, dtc)[0] == 0
        np.testing.assert_allclose(mem[0, 7:11], g["cmd"][k], rtol=0, atol=1e-8, err_msg=f"cmd, step {k}")
        np.testing.assert_allclose(mem[0, 6], g["last_thrust"][k], rtol=0, atol=1e-8)


@needs_fixture
@pytest.mark.gpu
def test_device_env_step_matches_pybullet_step_by_step():
    """The HIP path against the engine itself: every recorded Env.step from PyBullet's own previous state, judged on
    the increments at the per-step bar (tests/util.py)."""
    import torch
    from dronesim_amd.envs import CtrlAviary
    from tests.util import assert_step_parity, f32
    g, K = _load()
    t = params.builtin_type(str(g["drone"]))
    env = CtrlAviary([str(g["drone"])], 1, initial_xyzs=g["init_xyz"], initial_rpys=g["init_rpy"],
                     aggregate_phy_steps=int(g["aggr"]), noise_seed=0, dict_io=False)
    prev = env.state.rigid_aos()
    for k in range(K):
        env.state.set_fields(0, torch.from_numpy(np.ascontiguousarray(prev.T)))
        r0 = env.state.rigid_aos()
        act = f32(g["action"][k][None, :])
        env.step(torch.from_numpy(act.astype(np.float32)))
        want = f32(_rigid(g["state"][k])[None, :])
        assert_step_parity("pybullet_fixture", [t], None, r0, env.state.mem_aos(), np.zeros((1, 10)),
                           env.state.rigid_aos(), None, want, None, DT, DT, int(g["aggr"]), control=False, action=act)
        prev = want
    env.close()


@needs_fixture
def test_plane_contact_model_against_the_recorded_touchdown():
    """The part of the recording after the first ground contact, against DSIM_OPT_PLANE (a product-defined contact
    model, not Bullet's: DESIGN.md section 7).  What it is expected to share with the engine, stated loosely on
    purpose — the day the fixture exists these numbers say how far the model is from Bullet's manifold and solver:
    the vehicle comes to rest ON the plane at its collision shape's height, stays there while the thrust state winds up,
    and leaves the ground within 0.1 s of the recording."""
    g, _ = _load()
    first = int(g["first_contact_step"])
    n_rec = g["state"].shape[0]
    if first >= n_rec - 4:
        pytest.skip("the recording ends before the vehicle has settled on the plane")
    t = params.builtin_type(str(g["drone"]))
    O = orc.Oracle([t])
    PLANE = 1 << 10
    dtc = float(g["aggr"]) * DT
    zs_rec = g["state"][:, 2]
    # engine: lowest height reached and the control step at which it leaves the ground again
    on = g["contacts"] > 0
    rest_rec = float(zs_rec[on].min())
    assert abs(rest_rec - t.rest_height) < 5e-3, (rest_rec, t.rest_height)         # Bullet rests it on the same cylinder
    leave_rec = int(np.flatnonzero(on)[-1]) + 1 if on.any() else first
    # this model: the oracle's closed loop from the initial state, plane on
    rigid = np.concatenate([g["init_xyz"][0], orc.quat_from_euler(g["init_rpy"][0]), np.zeros(6)])[None, :]
    mem = O.reset_mem(1)
    zs, touching = [], []
    for k in range(n_rec):
        a6 = np.zeros((1, 6)); a6[0, :4] = 0.4 if k == 0 else mem[0, 7:11]
        O.physics(rigid, mem, int(g["aggr"]), DT, action=a6, options=PLANE)
        tgt = np.array([[0, 0, 0.5, 0, 0, 0, 0, 0, 0, g["target_yaw"][k]]])
        assert O.control(rigid, mem, tgt, dtc)[0] == 0
        # (getContactPoints lists points inside Bullet's contact breaking threshold, 0.02 m, not only touching ones)
        zs.append(rigid[0, 2]); touching.append(rigid[0, 2] < t.rest_height + 0.02)
    zs, touching = np.array(zs), np.array(touching)
    assert abs(zs.min() - rest_rec) < 5e-3
    if touching.any() and leave_rec < n_rec:
        leave = int(np.flatnonzero(touching)[-1]) + 1
        assert abs(leave - leave_rec) * dtc < 0.1, (leave, leave_rec)


def _has(key):
    return os.path.exists(FIXTURE) and key in np.load(FIXTURE, allow_pickle=False).files


@pytest.mark.skipif(not _has("helpers_quat_in"), reason="the fixture holds no C8 helper table (record it with round 3's recorder)")
def test_c8_helpers_match_pybullet_on_the_attitude_zoo():
    """C8 pinned directly: the oracle's restatement of the three PyBullet helpers (the goldens of the control half were
    generated with a stand-in for them, tests/golden/make_goldens.py:42-81) against the engine's own functions on the
    recorded zoo — whole sphere, both signs of w, both gimbal branches and their edge, non-unit quaternions."""
    g = np.load(FIXTURE, allow_pickle=False)
    q = g["helpers_quat_in"]
    e = np.array([orc.euler_from_quat(x) for x in q])
    sarg = -2.0 * (q[:, 0] * q[:, 2] - q[:, 3] * q[:, 1])
    off_edge = np.abs(np.abs(sarg) - 0.99999) > 1e-9                  # on the branch threshold Bullet itself is discontinuous
    np.testing.assert_allclose(e[off_edge], g["helpers_euler_from_quat"][off_edge], rtol=0, atol=1e-12)
    m = np.array([orc.matrix_from_quat(x).reshape(9) for x in q])
    np.testing.assert_allclose(m, g["helpers_matrix_from_quat"], rtol=0, atol=1e-12)
    qe = np.array([orc.quat_from_euler(x) for x in g["helpers_euler_in"]])
    np.testing.assert_allclose(qe, g["helpers_quat_from_euler"], rtol=0, atol=1e-12)


@pytest.mark.skipif(not _has("hexa_state"), reason="the fixture holds no hexa_6DOF flight (record it with round 3's recorder)")
def test_hexa_locked_joint_approximation_is_quantified():
    """PyBullet flies hexa_6DOF as an articulated body (six revolute arm joints held by default motors,
    hexa_6DOF.urdf:382-434); this repo flies the rigid composite with the joints locked.  Against the
    recorded hover-and-step flight: how far the arm joints actually move in the engine, and how far one Env.step of the
    rigid model is from the engine's, step by step from the engine's own previous state — the numbers are printed; the
    bounds below are the ones under which the approximation was accepted (1 mrad of joint motion, 1e-3 relative per step)."""
    g = np.load(FIXTURE, allow_pickle=False)
    t = params.builtin_type("hexa_6DOF")
    O = orc.Oracle([t])
    aggr = int(g["hexa_aggr"])
    joint_motion = float(np.abs(g["hexa_joint_angles"] - g["hexa_joint_angles"][0]).max())
    prev = np.concatenate([g["hexa_init_xyz"][0], orc.quat_from_euler(g["hexa_init_rpy"][0]), np.zeros(6)])
    worst = 0.0
    for k in range(g["hexa_state"].shape[0]):
        r = prev[None, :].copy()
        a6 = g["hexa_action"][k][None, :].copy()
        O.physics(r, O.reset_mem(1), aggr, DT, action=a6)
        want = _rigid(g["hexa_state"][k])
        inc = np.abs(want - prev)
        worst = max(worst, float((np.abs(r[0] - want) / (inc + 1e-6)).max()))
        prev = want
    print(f"hexa_6DOF: arm joints move {joint_motion:.3e} rad in the engine; rigid-composite step differs by {worst:.3e} of the increment")
    assert joint_motion < 1e-3 and worst < 1e-3


@pytest.mark.skipif(not _has("dyn_states"), reason="the fixture holds no Physics.DYN flight (record it with round 5's recorder)")
def test_dyn_flight_with_the_real_engine_as_the_pose_store():
    """Row D1 against the engine: the reference's _dynamics / step() loop with the REAL PyBullet storing and returning the
    pose (tests/golden/dynamics.npz was recorded with a stand-in store): the oracle's Env.step on Physics.DYN, step by step
    from the recording's own previous state — the quaternion Bullet hands back, rpy = its getEulerFromQuaternion, the
    placeholder angular velocity, rpy_rates."""
    g = np.load(FIXTURE, allow_pickle=False)
    t = params.builtin_type(str(g["drone"]))
    assert t.arm == float(g["dyn_arm"])
    O = orc.Oracle([t])
    aggr = int(g["dyn_aggr"])
    init, pwm, states = g["dyn_init"], g["dyn_pwm"], g["dyn_states"]
    prev = np.concatenate([init[0:7], init[10:16]])
    rates = np.zeros((1, 3))
    for k in range(pwm.shape[0]):
        r = prev[None, :].copy()
        a6 = np.zeros((1, 6)); a6[0, :4] = pwm[k]
        assert O.dyn_physics(r, rates, O.reset_mem(1), aggr, DT, action=a6) == 0
        want = states[k]
        np.testing.assert_allclose(r[0, 0:7], want[0:7], rtol=0, atol=1e-9, err_msg=f"pose, step {k}")
        np.testing.assert_allclose(orc.euler_from_quat(r[0, 3:7]), want[7:10], rtol=0, atol=1e-9, err_msg=f"rpy, step {k}")
        np.testing.assert_allclose(r[0, 7:10], want[10:13], rtol=0, atol=1e-9)
        np.testing.assert_array_equal(want[13:16], -1.0)                        # what getBaseVelocity reports under DYN
        np.testing.assert_allclose(rates[0], want[16:19], rtol=0, atol=1e-9)
        prev = np.concatenate([want[0:7], want[10:16]])
        rates = want[16:19][None, :].copy()


@pytest.mark.skipif(not _has("noisy_state"), reason="the fixture holds no noisy flight (record it with round 5's recorder)")
def test_noisy_flight_replays_through_the_oracle_with_the_recorded_draws():
    """The fly_INDI.py flight with the rotor noise ON: the reference's own draws (f_noise[4], m_noise[4] per sub-step,
    BaseAviary.py:1518-1521), fed to the oracle as a noise replay — every Env.step from the engine's own previous state must
    land on the engine's next state: pins WHERE the normals enter the force map (rotor link z forces, the shared lateral
    components of draws 0 and 1, the base torque) through the engine, not only through the recorded apply* calls."""
    g = np.load(FIXTURE, allow_pickle=False)
    t = params.builtin_type(str(g["drone"]))
    O = orc.Oracle([t])
    aggr = int(g["noisy_aggr"])
    prev = np.concatenate([g["noisy_init_xyz"][0], orc.quat_from_euler(g["noisy_init_rpy"][0]), np.zeros(6)])
    for k in range(g["noisy_state"].shape[0]):
        r = prev[None, :].copy()
        a6 = np.zeros((1, 6)); a6[0, :4] = g["noisy_action"][k]
        nz = np.zeros((1, aggr, 12))
        nz[0, :, 0:4], nz[0, :, 6:10] = g["noisy_f_noise"][k], g["noisy_m_noise"][k]
        O.physics(r, O.reset_mem(1), aggr, DT, action=a6, noise=nz)
        want = _rigid(g["noisy_state"][k])
        np.testing.assert_allclose(r[0], want, rtol=0, atol=1e-9, err_msg=f"step {k}")
        prev = want
