"""Oracle vs what the reference's OWN env-side code hands to the physics engine (tests/golden/env_side.npz,
captured by tests/golden/make_goldens.py::capture_env_side: BaseAviary._physics / _drag / _groundEffect /
_downwash and the three _preprocessAction variants run unbound against recording stand-ins for
p.applyExternalForce / applyExternalTorque / getLinkStates).  This pins rows P0-P3, P6-P8 and the action
adaptors of SURVEY.md 8a on reference-run data; only the integrator itself (P4, inside Bullet) stays
unpinned.  CPU only."""
import ctypes
import os

import numpy as np
import pytest

from dronesim_amd import params
from oracle import oracle as orc

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "env_side.npz"))
D = ctypes.POINTER(ctypes.c_double)
LINK_FRAME = 1


def _d(a):
    return a.ctypes.data_as(D)


@pytest.mark.parametrize("model", ["robobee", "tello", "hexa_6DOF", "hexa_6DOF_simple"])
def test_type_table_matches_reference_urdf_parser(model):
    """P0: the build's own URDF reader vs BaseAviary._parseURDFParameters."""
    t = params.builtin_type(model)
    g = lambda k: G[f"{model}_{k}"]
    assert t.n_act == int(g("INDI_ACTUATOR_NR"))
    assert t.kf == float(g("KF")) and t.km == float(g("KM"))
    np.testing.assert_array_equal(np.asarray(t.G1)[: int(g("INDI_OUTPUT_NR")), : t.n_act], g("G1"))
    np.testing.assert_array_equal(np.asarray(t.pwm2rpm_scale)[: t.n_act], g("PWM2RPM_SCALE"))
    np.testing.assert_array_equal(np.asarray(t.pwm2rpm_const)[: t.n_act], g("PWM2RPM_CONST"))
    np.testing.assert_array_equal(np.asarray(t.pwm_min)[: t.n_act], g("MIN_PWM"))
    np.testing.assert_array_equal(np.asarray(t.pwm_max)[: t.n_act], g("MAX_PWM"))
    np.testing.assert_array_equal(np.asarray(t.drag_coeff), g("DRAG_COEFF"))
    assert t.gnd_eff_coeff == float(g("GND_EFF_COEFF")) and t.prop_radius == float(g("PROP_RADIUS"))
    np.testing.assert_array_equal(np.asarray(t.dw_coeff), [g("DW_COEFF_1"), g("DW_COEFF_2"), g("DW_COEFF_3")])
    assert t.max_speed_kmh == float(g("MAX_SPEED_KMH"))
    if not model.startswith("hexa_6DOF"):
        # quads: one massive link.  (hexa: the reference's parser reads the FIRST link only, 0.2 kg, a value the
        # path never uses; Bullet sums all links of the URDF, 0.86 kg, which is what the type table carries)
        assert t.mass == float(g("M"))
        np.testing.assert_array_equal(np.asarray(t.inertia), np.diag(g("J")))
    else:
        assert float(g("M")) == 0.2 and abs(t.mass - 0.86) < 1e-12


@pytest.mark.parametrize("model", ["robobee", "tello"])
def test_quad_force_map_vs_reference_calls(model):
    """P2: four LINK_FRAME forces [fn0, fn1, F_i] on rotor links 0..3 (at the links' inertial origins) and one
    LINK_FRAME torque [mn0, mn1, tau_z] on the base (-1) — summed into a body wrench and compared with the
    oracle's orc_quad_wrench fed with the same numpy normals."""
    t = params.builtin_type(model)
    P = t.to_c()
    f = orc.lib().orc_quad_wrench
    f.argtypes = [ctypes.POINTER(type(P)), D, D, D, D, D, D]
    r = np.asarray(t.rotor_pos)[:4]
    for i in range(G[f"{model}_fm_cmd"].shape[0]):
        kind, link, vec = G[f"{model}_fm_kind"][i], G[f"{model}_fm_link"][i], G[f"{model}_fm_vec"][i]
        assert list(kind) == [0, 0, 0, 0, 1] and list(link) == [0, 1, 2, 3, -1]
        assert (G[f"{model}_fm_flag"][i] == LINK_FRAME).all() and not G[f"{model}_fm_pos"][i].any()
        F_ref = vec[:4].sum(0)
        tau_ref = np.cross(r, vec[:4]).sum(0) + vec[4]
        F, tau, rpm = np.zeros(3), np.zeros(3), np.zeros(4)
        cmd, fn, mn = (G[f"{model}_fm_{k}"][i].copy() for k in ("cmd", "f_noise", "m_noise"))
        f(ctypes.byref(P), _d(cmd), _d(fn), _d(mn), _d(F), _d(tau), _d(rpm))
        np.testing.assert_allclose(F, F_ref, rtol=1e-13, atol=1e-15)
        np.testing.assert_allclose(tau, tau_ref, rtol=1e-12, atol=1e-16)


@pytest.mark.parametrize("m", ["hexa_6DOF", "hexa_6DOF_simple"])
def test_hexa_force_map_vs_reference_calls(m):
    """P3: per prop j one LINK_FRAME force [0,0,F_j] and one LINK_FRAME torque [0,0,±tau_j] on link 2j+1;
    the link frames (tilted props) come from the URDF joints.  Both morphing-hexa URDFs (the same links and joints)."""
    t = params.builtin_type(m)
    P = t.to_c()
    f = orc.lib().orc_hexa_wrench
    f.argtypes = [ctypes.POINTER(type(P)), D, D, D, D, D, D]
    r, ax = np.asarray(t.rotor_pos)[:6], np.asarray(t.rotor_axis)[:6]
    for i in range(G[f"{m}_fm_cmd"].shape[0]):
        kind, link, vec = G[f"{m}_fm_kind"][i], G[f"{m}_fm_link"][i], G[f"{m}_fm_vec"][i]
        assert list(kind) == [0, 1] * 6 and list(link) == [1, 1, 3, 3, 5, 5, 7, 7, 9, 9, 11, 11]
        assert (G[f"{m}_fm_flag"][i] == LINK_FRAME).all() and not vec[:, :2].any()
        Fz, Tz = vec[0::2, 2], vec[1::2, 2]
        F_ref = (ax * Fz[:, None]).sum(0)
        tau_ref = np.cross(r, ax * Fz[:, None]).sum(0) + (ax * Tz[:, None]).sum(0)
        F, tau, rpm = np.zeros(3), np.zeros(3), np.zeros(6)
        cmd, fn, mn = (G[f"{m}_fm_{k}"][i].copy() for k in ("cmd", "f_noise", "m_noise"))
        f(ctypes.byref(P), _d(cmd), _d(fn), _d(mn), _d(F), _d(tau), _d(rpm))
        np.testing.assert_allclose(F, F_ref, rtol=1e-13, atol=1e-15)
        np.testing.assert_allclose(tau, tau_ref, rtol=1e-12, atol=1e-16)


def test_drag_and_ground_effect_vs_reference_calls():
    """P6: the vector _drag applies (link 4, LINK_FRAME); P7: the per-rotor z forces of _groundEffect for the
    rotor heights getLinkStates reported, incl. the |roll|,|pitch| < pi/2 gate (no call at all beyond it)."""
    t = params.builtin_type("robobee")
    t.gnd_eff_h_clip = float(G["aero_h_clip"])
    P = t.to_c()
    fd = orc.lib().orc_drag
    fd.argtypes = [ctypes.POINTER(type(P)), D, D, D, D]
    fg = orc.lib().orc_ground_effect
    fg.argtypes = [ctypes.POINTER(type(P)), D, D, D, D]
    gated = 0
    for i in range(G["aero_quat"].shape[0]):
        q, v, rpm = G["aero_quat"][i].copy(), G["aero_vel"][i].copy(), G["aero_rpm"][i].copy()
        out = np.zeros(3)
        fd(ctypes.byref(P), _d(q), _d(v), _d(rpm), _d(out))
        np.testing.assert_allclose(out, G["aero_drag"][i], rtol=1e-13, atol=1e-18)
        R = orc.matrix_from_quat(q)
        want = G["aero_gnd"][i]
        assert G["aero_gnd_calls"][i] in (0, 4)
        gated += G["aero_gnd_calls"][i] == 0
        for j in range(4):
            # the oracle derives rotor j's height from the pose: place the base so that it equals the recorded one
            pos = np.array([0.0, 0.0, G["aero_prop_h"][i, j] - R[2] @ np.asarray(t.rotor_pos)[j]])
            dF = np.zeros(6)
            fg(ctypes.byref(P), _d(pos), _d(q), _d(rpm), _d(dF))
            np.testing.assert_allclose(dF[j], want[j], rtol=1e-12, atol=1e-18)
    assert 0 < gated < G["aero_quat"].shape[0]          # both sides of the attitude gate are covered


def test_downwash_vs_reference_calls():
    """P8: sum of the z forces _downwash applies to each drone of a 48-drone world."""
    t = params.builtin_type("robobee")
    O = orc.Oracle([t])
    pos = G["dw_pos"]
    rigid = np.zeros((pos.shape[0], 13)); rigid[:, 0:3] = pos; rigid[:, 6] = 1.0
    fz = O.downwash(rigid, pos)
    np.testing.assert_allclose(fz, G["dw_fz"], rtol=1e-12, atol=1e-300)
    assert (G["dw_fz"] != 0).sum() > 10


def test_pwm_clip_vs_reference():
    """P1: CtrlAviary._preprocessAction."""
    t = params.builtin_type("robobee")
    P = t.to_c()
    f = orc.lib().orc_preprocess_action
    f.argtypes = [ctypes.POINTER(type(P)), D, D]
    for a, want in zip(G["clip_in"], G["clip_out"]):
        out = np.zeros(6)
        f(ctypes.byref(P), _d(np.concatenate([a, [0, 0]])), _d(out))
        np.testing.assert_array_equal(out[:4], want)


@pytest.mark.parametrize("model", ["robobee", "tello"])
@pytest.mark.parametrize("mode", ["vel", "rpyt"])
def test_action_adaptors_vs_reference(model, mode):
    """VelocityAviary / RPYTAviary._preprocessAction: command and controller memory after the call."""
    t = params.builtin_type(model)
    O = orc.Oracle([t])
    g = lambda k: G[f"{model}_ad_{k}"]
    st = g("state")
    n = st.shape[0]
    rigid = np.concatenate([st[:, 0:7], st[:, 10:16]], 1).copy()
    mem = np.zeros((n, 13))
    mem[:, 0:3], mem[:, 3:6], mem[:, 6], mem[:, 7:11] = g("last_vel"), g("last_rates"), g("last_thrust"), g("cmd")
    assert O.adaptor_step(0 if mode == "vel" else 1, rigid, mem, g(f"{mode}_action"), 5, 1 / 240, 5 / 240) == 0
    np.testing.assert_allclose(mem[:, 7:11], g(f"{mode}_cmd_out"), rtol=0, atol=1e-11)
    np.testing.assert_allclose(mem[:, 3:6], g(f"{mode}_last_rates_out"), rtol=0, atol=1e-12)
    np.testing.assert_allclose(mem[:, 6], g(f"{mode}_last_thrust_out"), rtol=0, atol=1e-11)
    if mode == "vel":
        np.testing.assert_allclose(mem[:, 0:3], g("vel_last_vel_out"), rtol=0, atol=1e-13)
