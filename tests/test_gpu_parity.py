"""GPU parity tests: the HIP path (through the C-ABI) against the fp64 CPU oracle and the
golden vectors.  Tolerance: BASELINE.json's bar, 1e-4 relative per step (fp32 device
arithmetic vs the fp64 reference arithmetic).  One-step comparisons apply it to the state INCREMENT of
the step plus a few fp32 ulps of the largest term entering each field's update, per drone and per
field (tests/util.py: assert_step_parity); closed-loop trajectories are checked step by step from the
device's own previous state (no accumulation hides in the bar), and their accumulated drift from the
oracle's free-running trajectory is bounded separately.
"""
import ctypes
import math
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from dronesim_amd import params  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from tests.util import (K_ULP, noise_terms, MEM_SCALE, RIGID_SCALE, assert_control_parity, assert_downwash, assert_step_parity,  # noqa: E402
                        attitude_zoo, f32, random_fleet, rel_err, rotor_noise, ulp32)

pytestmark = pytest.mark.gpu

REL_TOL = 1e-4          # north_star: per-step state within 1e-4 rel-err
DT = float(np.float32(1.0 / 240.0))


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device; the product has no CPU fallback")
    from dronesim_amd import _native as nat
    from dronesim_amd import fleet
    return nat, fleet


def _args(nat, substeps, dt_phys, dt_ctrl, options=0, seed=0, step_index=0, replay=None, type_id=None, action=None):
    a = nat.StepArgs()
    a.phys_substeps, a.dt_phys, a.dt_ctrl, a.options = substeps, dt_phys, dt_ctrl, options
    a.noise_seed, a.step_index = seed, step_index
    a.noise_replay = replay.data_ptr() if replay is not None else None
    a.type_id = type_id.data_ptr() if type_id is not None else None
    a.action = action.data_ptr() if action is not None else None
    return a


def _stream(ctx):
    return ctx.stream_ptr()


def _make(gpu, model, n, layout="soa", seed=0, pad=256, **kw):
    nat, fleet = gpu
    t = params.builtin_type(model)
    ctx = fleet.Context([t])
    st = fleet.FleetState(ctx, n, layout, pad)
    tg = fleet.Targets(ctx, n, layout, pad=pad)
    rigid, mem, tgt = random_fleet(np.random.default_rng(seed), n, **kw)
    st.load_aos(rigid, mem)
    tg.set_fields(0, torch.from_numpy(np.ascontiguousarray(tgt.T)))
    return t, ctx, st, tg, rigid, mem, tgt


# ---------------------------------------------------------------------------
# computeControl against the reference's golden vectors
# ---------------------------------------------------------------------------
def _golden_mem(g, sel, n_act):
    """The reference's controller memory after the call, in the oracle's [n,13] layout."""
    m = np.zeros((len(sel), 13))
    m[:, 0:3], m[:, 3:6], m[:, 6] = g["last_vel_out"][sel], g["last_rates_out"][sel], g["last_thrust_out"][sel]
    m[:, 7:7 + n_act] = g["cmd_out"][sel]
    return m


def _wrap_diff(a, b):
    d = np.abs(a - b)
    return np.minimum(d, np.abs(d - 2 * math.pi))       # a yaw error within rounding of +-pi may land on either side


@pytest.mark.parametrize("model", ["robobee", "tello", "hexa_6DOF_simple"])
@pytest.mark.parametrize("layout", ["soa", "tile64"])
def test_control_vs_golden(gpu, golden_dir, model, layout):
    """computeControl against the vectors the reference's own INDIControl produced.  Three links, each per case:
    (1) HIP path == oracle on the SAME fp32-rounded inputs, at the per-step bar on the memory increments;
    (2) oracle on the reference's fp64 inputs == golden to 1e-9 (the pin; also tests/test_oracle_control.py);
    (3) HIP path == golden within (1)'s bound plus |oracle(fp32 inputs) - golden|, the measured effect of rounding
        the inputs to fp32 (saturating cases amplify it through pinv(G) ~ 1/cos(roll) and pinv(G1/0.05))."""
    nat, fleet = gpu
    g = np.load(os.path.join(golden_dir, f"indi_single_{model}.npz"))
    t = params.builtin_type(model)
    ctx = fleet.Context([t])
    O = orc.Oracle([t])
    n = g["pos"].shape[0]
    rigid = np.concatenate([g["pos"], g["quat"], g["vel"], g["ang_vel"]], 1)
    mem = np.zeros((n, 13))
    mem[:, 0:3], mem[:, 3:6], mem[:, 6], mem[:, 7:7 + t.n_act] = g["last_vel"], g["last_rates"], g["last_thrust"], g["cmd"]
    tgt = np.concatenate([g["target_pos"], g["target_vel"], g["target_acc"], g["target_rpy"][:, 2:3]], 1)
    for dt in np.unique(g["dt"]):
        sel = np.where(g["dt"] == dt)[0]
        m = len(sel)
        st = fleet.FleetState(ctx, m, layout)
        tg = fleet.Targets(ctx, m, layout)
        r32, m32, t32, dt32 = f32(rigid[sel]), f32(mem[sel]), f32(tgt[sel]), float(np.float32(dt))
        st.load_aos(r32, m32)
        tg.set_fields(0, torch.from_numpy(np.ascontiguousarray(t32.T)))
        pos_e = torch.zeros((3, st.n_pad), device=ctx.device)
        yaw_e = torch.zeros((st.n_pad,), device=ctx.device)
        a = _args(nat, 0, dt32, dt32)
        nat.check(ctx.lib.dsim_control(ctx.handle, _stream(ctx), m, st.view(), tg.view(), ctypes.byref(a),
                                       pos_e.data_ptr(), yaw_e.data_ptr()))
        torch.cuda.synchronize()
        got = st.mem_aos()
        # (1) same fp32 inputs
        o32 = m32.copy()
        rc, pe32, ye32 = O.control(r32, o32, t32, dt32)
        assert rc == 0
        assert_control_parity(f"control_golden[{model},{layout}] vs oracle(fp32 in)", [t], None, r32, m32, t32, got, o32, dt32)
        # (2) the pin
        o64 = mem[sel].copy()
        rc, pe64, ye64 = O.control(rigid[sel].copy(), o64, tgt[sel], float(dt))
        gold = _golden_mem(g, sel, t.n_act)
        assert rc == 0 and np.abs(o64 - gold).max() < 1e-9 and np.abs(pe64 - g["pos_e"][sel]).max() < 1e-12
        # (3) against the reference's numbers, slack = measured input-rounding effect, per case and field
        assert_control_parity(f"control_golden[{model},{layout}] vs reference", [t], None, r32, m32, t32, got, gold, dt32,
                              slack=np.abs(o32 - gold))
        # pos_e is one fp32 subtraction; yaw_e one atan2 (+ the same input rounding)
        pe = pos_e[:, :m].T.double().cpu().numpy()
        assert (np.abs(pe - pe32) <= ulp32(np.maximum(np.abs(t32[:, 0:3]), np.abs(r32[:, 0:3])))).all()
        assert (np.abs(pe - g["pos_e"][sel]) <= np.abs(pe32 - g["pos_e"][sel]) + ulp32(np.maximum(np.abs(t32[:, 0:3]), np.abs(r32[:, 0:3])))).all()
        ye = yaw_e[:m].double().cpu().numpy()
        # yaw_e = norm_ang(yaw* - yaw) of two angles up to ~4 rad: 8 fp32 ulps at that size, times the attitude's
        # conditioning 1 / (ya^2 + yb^2)^(1/2) = 1 / cos(pitch) of the yaw atan2
        pitch = np.array([orc.euler_from_quat(q)[1] for q in r32[:, 3:7]])
        ytol = 8 * ulp32(4.0) / np.maximum(np.abs(np.cos(pitch)), 1e-3)
        assert (_wrap_diff(ye, ye32) <= ytol).all(), (_wrap_diff(ye, ye32) / ytol).max()
        assert (_wrap_diff(ye, g["yaw_e"][sel]) <= ytol + _wrap_diff(ye32, g["yaw_e"][sel])).all()
    ctx.close()


def test_roll_sweep_through_the_pinv_singularity(gpu, golden_dir):
    """The kernel's closed-form inverse of G (orthogonal columns: scaled transpose; only the pitch row carries
    1 / cos^2(roll)) against the reference's np.linalg.pinv swept through roll = +-90 deg: equal within the fp32
    conditioning (error ~ 1e-4 / |cos roll|) while |cos roll| >= 1e-2; finite, PWM-clipped commands all the way to
    cos(roll) = 0; and the thrust state exact everywhere except AT the singular point, where numpy's pinv switches
    to the minimum-norm solution (the reference itself jumps there: -8.6 at 1e-9 rad from it, -7.3 on it)."""
    nat, fleet = gpu
    g = np.load(os.path.join(golden_dir, "indi_roll_sweep.npz"))
    ctx = fleet.Context([params.builtin_type("robobee")])
    n = len(g["roll"])
    rigid = np.concatenate([g["pos"], g["quat"], g["vel"], g["ang_vel"]], 1)
    mem = np.zeros((n, 13))
    mem[:, 0:3], mem[:, 3:6], mem[:, 6], mem[:, 7:11] = g["last_vel"], g["last_rates"], g["last_thrust"], g["cmd"]
    st, tg = fleet.FleetState(ctx, n), fleet.Targets(ctx, n)
    st.load_aos(rigid, mem)
    tg.set(pos=np.ascontiguousarray(g["target_pos"].T), yaw=g["target_yaw"][None, :])
    a = _args(nat, 0, float(g["dt"]), float(g["dt"]))
    pos_e = torch.zeros((3, st.n_pad), device=ctx.device); yaw_e = torch.zeros((st.n_pad,), device=ctx.device)
    nat.check(ctx.lib.dsim_control(ctx.handle, _stream(ctx), n, st.view(), tg.view(), ctypes.byref(a),
                                   pos_e.data_ptr(), yaw_e.data_ptr()))
    torch.cuda.synchronize()
    got = st.mem_aos()
    assert np.isfinite(got).all() and got[:, 7:11].min() >= 0.0 and got[:, 7:11].max() <= 1.0
    c = np.abs(np.cos(g["roll"]))
    ok = c >= 1e-2
    assert ok.sum() >= 8
    # Per case: the bar on the memory increments with the fp32 ulp term amplified by 1 / cos^2(roll) (tests/util.py
    # tilt_gain: that is how the rounding of cos(roll) reaches the pitch increment) against the oracle on the same
    # fp32-rounded inputs; against the reference's numbers the measured effect of that input rounding
    # (oracle on the rounded inputs - golden) is added, per case and field.
    t = params.builtin_type("robobee")
    dt32 = float(np.float32(g["dt"]))
    r32, m32 = f32(rigid), f32(mem)
    t32 = f32(np.concatenate([g["target_pos"], np.zeros((n, 6)), g["target_yaw"][:, None]], 1))
    o32 = m32.copy()
    rc, _, _ = orc.Oracle([t]).control(r32, o32, t32, dt32)
    assert rc == 0
    assert_control_parity("roll_sweep vs oracle(fp32 in)", [t], None, r32[ok], m32[ok], t32[ok], got[ok], o32[ok], dt32)
    gold = o32.copy()                                        # the fixture holds cmd, thrust and rates
    gold[:, 7:11], gold[:, 6], gold[:, 3:6] = g["cmd_out"], g["last_thrust_out"], g["last_rates_out"]
    assert_control_parity("roll_sweep vs reference", [t], None, r32[ok], m32[ok], t32[ok], got[ok], gold[ok], dt32,
                          slack=np.abs(o32[ok] - gold[ok]))
    # the thrust increment is the well-conditioned row of inv(G): at the bar right up to the singular point
    off = np.abs(np.abs(g["roll"]) - np.pi / 2) > 0
    g_thr, o_thr = got.copy(), o32.copy()
    g_thr[:, 7:11] = o_thr[:, 7:11]                          # (cmd is ill-conditioned there: judged above where cos >= 1e-2)
    assert_control_parity("roll_sweep thrust row", [t], None, r32[off], m32[off], t32[off], g_thr[off], o_thr[off], dt32)
    ctx.close()


@pytest.mark.parametrize("model", ["robobee", "tello", "hexa_6DOF_simple"])
def test_control_sequence_vs_golden(gpu, golden_dir, model):
    """Controller memory recursion over 60 calls (reference-generated sequence)."""
    nat, fleet = gpu
    g = np.load(os.path.join(golden_dir, f"indi_sequence_{model}.npz"))
    t = params.builtin_type(model)
    ctx = fleet.Context([t])
    S, K = g["pos"].shape[:2]
    st = fleet.FleetState(ctx, S)
    tg = fleet.Targets(ctx, S)
    O = orc.Oracle([t])
    mem0 = O.reset_mem(S)
    st.load_aos(np.zeros((S, 13)), mem0)
    dt = float(np.float32(g["dt"]))
    for k in range(K):
        rigid = f32(np.concatenate([g["pos"][:, k], g["quat"][:, k], g["vel"][:, k], g["ang_vel"][:, k]], 1))
        tgt = f32(np.concatenate([g["target_pos"][:, k], g["target_vel"][:, k], g["target_acc"][:, k],
                                  g["target_rpy"][:, k, 2:3]], 1))
        st.set_fields(0, torch.from_numpy(np.ascontiguousarray(rigid.T)))
        tg.set_fields(0, torch.from_numpy(np.ascontiguousarray(tgt.T)))
        m0 = st.mem_aos()                              # the device's own controller memory before the call
        a = _args(nat, 0, dt, dt)
        nat.check(ctx.lib.dsim_control(ctx.handle, _stream(ctx), S, st.view(), tg.view(), ctypes.byref(a), None, None))
        got = st.mem_aos()
        o = m0.copy()
        assert O.control(rigid, o, tgt, dt)[0] == 0
        assert_control_parity(f"control_sequence[{model}]", [t], None, rigid, m0, tgt, got, o, dt)   # every call at the bar
        # and the recursion as a whole stays on the reference's own 60-call sequence (accumulated, absolute)
        assert np.abs(got[:, 7:7 + t.n_act] - g["cmd_out"][:, k]).max() < 1e-4, k
        assert np.abs(got[:, 6] - g["last_thrust_out"][:, k]).max() < 1e-4 * (1 + np.abs(g["last_thrust_out"][:, k]).max()), k
    ctx.close()


# ---------------------------------------------------------------------------
# fused Env.step + computeControl against the oracle
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("model,substeps,layout,pad", [("robobee", 5, "soa", 256), ("robobee", 1, "tile64", 64),
                                                       ("tello", 2, "soa", 64), ("tello", 5, "tile64", 256)])
def test_fused_step_vs_oracle(gpu, model, substeps, layout, pad):
    """pad=256: whole fleet on the fast kernel; pad=64: n_pad = 4160 = 16 whole tiles on the fast
    kernel + a 64-drone ragged tail on the general kernel."""
    nat, fleet = gpu
    n = 4096 + 17                      # ragged: not a multiple of 64
    t, ctx, st, tg, rigid, mem, tgt = _make(gpu, model, n, layout, seed=3, pad=pad)
    O = orc.Oracle([t])
    dtc = float(np.float32(substeps / 240.0))
    a = _args(nat, substeps, DT, dtc)
    nat.check(ctx.lib.dsim_step(ctx.handle, _stream(ctx), n, st.view(), tg.view(), ctypes.byref(a)))
    r0, m0 = rigid.copy(), mem.copy()
    assert O.step(rigid, mem, tgt, substeps, DT, dtc) == 0
    # positions out to +-50 m: the increment bar tests the position UPDATE there (1e-4 of ~8 mm + 4 ulp32(50 m))
    assert_step_parity(f"fused_step[{model},{substeps},{layout}]", [t], None, r0, m0, tgt, st.rigid_aos(), st.mem_aos(),
                       rigid, mem, DT, dtc, substeps)
    ctx.close()


def test_fused_step_broadcast_target(gpu):
    nat, fleet = gpu
    n = 1000
    t, ctx, st, tg, rigid, mem, tgt = _make(gpu, "robobee", n, seed=5, spread=3.0)
    bt = fleet.Targets(ctx, n, broadcast=True)
    one = f32(np.array([[0.0, 0.0, 0.5, 0.1, 0, 0, 0, 0, 0.05, 0.4]]))
    bt.set(pos=one[0, 0:3], vel=one[0, 3:6], acc=one[0, 6:9], yaw=one[0, 9])
    a = _args(nat, 5, DT, float(np.float32(5 / 240)), options=nat.OPT_BCAST_TGT)
    nat.check(ctx.lib.dsim_step(ctx.handle, _stream(ctx), n, st.view(), bt.view(), ctypes.byref(a)))
    O = orc.Oracle([t])
    r0, m0 = rigid.copy(), mem.copy()
    assert O.step(rigid, mem, one, 5, DT, float(np.float32(5 / 240))) == 0
    assert_step_parity("fused_step_broadcast", [t], None, r0, m0, one, st.rigid_aos(), st.mem_aos(), rigid, mem,
                       DT, float(np.float32(5 / 240)), 5)
    ctx.close()


def test_physics_only_vs_oracle(gpu):
    """Env.step(action): explicit action clipped in-kernel, echoed to last_action, controller memory untouched."""
    nat, fleet = gpu
    n = 2048
    t, ctx, st, tg, rigid, mem, tgt = _make(gpu, "robobee", n, seed=7)
    rng = np.random.default_rng(8)
    act = f32(rng.uniform(-0.2, 1.2, (n, 4)))            # some outside [0,1] -> clipped
    act_dev = torch.zeros((4, st.n_pad), device=ctx.device)
    act_dev[:, :n] = torch.from_numpy(act.T).float()
    echo = torch.full((4, st.n_pad), -7.0, device=ctx.device)
    a = _args(nat, 5, DT, DT, action=act_dev)
    nat.check(ctx.lib.dsim_physics(ctx.handle, _stream(ctx), n, st.view(), echo.data_ptr(), ctypes.byref(a)))
    O = orc.Oracle([t])
    a6 = np.zeros((n, 6)); a6[:, :4] = act
    last = np.zeros((n, 6))
    mem_before = mem.copy()
    r0 = rigid.copy()
    O.physics(rigid, mem, 5, DT, action=a6, last_action=last)
    assert_step_parity("physics_only", [t], None, r0, mem, tgt, st.rigid_aos(), None, rigid, None, DT, DT, 5,
                       control=False, action=act)
    np.testing.assert_array_equal(echo[:, :n].T.cpu().numpy(), np.clip(act, 0, 1).astype(np.float32))
    np.testing.assert_array_equal(last[:, :4], np.clip(act, 0, 1))
    np.testing.assert_array_equal(st.mem_aos(), mem_before)          # controller memory untouched
    ctx.close()


def test_hover_trajectory_vs_oracle(gpu):
    """Config 1 (examples/fly_INDI.py defaults): 1 robobee from (0,1,0.5) to (0,0,0.5), yaw ramp,
    initial action 0.4, 5 sub-steps per control, 2 s = 96 env steps; trajectory vs the oracle."""
    nat, fleet = gpu
    from dronesim_amd.envs import CtrlAviary
    env = CtrlAviary(["robobee"], 1, initial_xyzs=np.array([[0.0, 1.0, 0.5]]), initial_rpys=np.zeros((1, 3)),
                     aggregate_phy_steps=5, noise_seed=0, ground_plane=False)     # (the start-up dip of this flight reaches
    tg = fleet.Targets(env.ctx, 1)                                               # z = 0: test_plane_touchdown_flight_config1)
    t = params.builtin_type("robobee")
    O = orc.Oracle([t])
    rigid, mem = env.state.rigid_aos(), env.state.mem_aos()
    dtc = float(np.float32(5 / 240))
    worst = 0.0
    for k in range(96):
        yaw = float(np.float32(0.4 + k / 200.0))
        tgt = f32(np.array([[0, 0, 0.5, 0, 0, 0, 0, 0, 0, yaw]]))
        tg.set(pos=tgt[0, 0:3], yaw=yaw)
        act = np.full((1, 4), 0.4) if k == 0 else None
        r0, m0 = env.state.rigid_aos(), env.state.mem_aos()
        env.step_fused(tg, control_timestep=dtc, action=act)
        a6 = None
        if k == 0:
            a6 = np.zeros((1, 6)); a6[:, :4] = 0.4
        assert O.step(rigid, mem, tgt, 5, DT, dtc, action=a6) == 0          # free-running oracle trajectory
        r1, m1 = r0.copy(), m0.copy()
        assert O.step(r1, m1, tgt, 5, DT, dtc, action=a6) == 0              # oracle step from the device's state
        assert_step_parity("config1_96_steps", [t], None, r0, m0, tgt, env.state.rigid_aos(), env.state.mem_aos(), r1, m1,
                           DT, dtc, 5, action=act)
        worst = max(worst, rel_err(env.state.rigid_aos(), rigid, RIGID_SCALE).max())
    # accumulated drift of the two closed loops over 96 steps (every single step met the bar above)
    assert worst < 1e-3, worst
    assert np.linalg.norm(rigid[0, 0:2]) < 0.9          # it really flew towards the target
    env.close()


def test_full_flight_config1_vs_oracle(gpu):
    """The whole 15 s flight of examples/fly_INDI.py (720 control steps x 5 sub-steps, yaw target ramping through
    the +-pi wrap), three airframe starts side by side: the fp32 kernel's closed-loop trajectory stays on the fp64
    oracle's for the entire flight, and ends hovering on the target.  (Kernel and oracle share the no-ground
    model: the first start dips below z = 0 where PyBullet's plane would catch it)"""
    nat, fleet = gpu
    from dronesim_amd.envs import CtrlAviary
    starts = np.array([[0.0, 1.0, 0.5], [1.0, -1.0, 0.8], [-0.5, 0.5, 1.5]])
    n = len(starts)
    env = CtrlAviary(["robobee"], n, initial_xyzs=starts, initial_rpys=np.zeros((n, 3)), aggregate_phy_steps=5,
                     noise_seed=0, dict_io=False)
    tg = fleet.Targets(env.ctx, n)
    O = orc.Oracle([params.builtin_type("robobee")])
    rigid, mem = env.state.rigid_aos(), env.state.mem_aos()
    dtc = float(np.float32(5 / 240))
    worst, worst_k = 0.0, -1
    for k in range(720):
        yaw = float(np.float32(0.4 + k / 200.0))                         # fly_INDI.py:165-167 (reaches 4.0 rad)
        tgt = f32(np.tile(np.array([[0, 0, 0.5, 0, 0, 0, 0, 0, 0, yaw]]), (n, 1)))
        tg.set(pos=np.ascontiguousarray(tgt[:, 0:3].T), yaw=yaw)
        r0, m0 = env.state.rigid_aos(), env.state.mem_aos()
        env.step_fused(tg, control_timestep=dtc, action=np.full((n, 4), 0.4, dtype=np.float32) if k == 0 else None)
        a6 = None
        if k == 0:
            a6 = np.zeros((n, 6)); a6[:, :4] = 0.4
        assert O.step(rigid, mem, tgt, 5, DT, dtc, action=a6) == 0
        r1, m1 = r0.copy(), m0.copy()
        assert O.step(r1, m1, tgt, 5, DT, dtc, action=a6) == 0              # every step at the bar from the device's state
        assert_step_parity("config1_full_flight", [params.builtin_type("robobee")], None, r0, m0, tgt, env.state.rigid_aos(),
                           env.state.mem_aos(), r1, m1, DT, dtc, 5, action=None if a6 is None else a6[:, :4])
        if k % 8 == 7 or k == 719:
            e = rel_err(env.state.rigid_aos(), rigid, RIGID_SCALE).max()
            if e > worst:
                worst, worst_k = e, k
    assert worst < 2e-3, (worst, worst_k)
    final = env.state.rigid_aos()
    assert np.abs(final[:, 0:3] - np.array([0, 0, 0.5])).max() < 0.02 and np.abs(final[:, 7:10]).max() < 0.05
    env.close()


def test_noise_replay_vs_oracle(gpu):
    nat, fleet = gpu
    n, sub = 512, 5
    t, ctx, st, tg, rigid, mem, tgt = _make(gpu, "robobee", n, seed=11)
    rng = np.random.default_rng(12)
    fn = f32(rng.normal(0, 0.01, (n, sub, 4)))
    mn = f32(rng.normal(0, 0.001, (n, sub, 4)))
    replay = torch.zeros((sub, 8, st.n_pad), device=ctx.device)
    replay[:, 0:4, :n] = torch.from_numpy(fn.transpose(1, 2, 0)).float()
    replay[:, 4:8, :n] = torch.from_numpy(mn.transpose(1, 2, 0)).float()
    a = _args(nat, sub, DT, float(np.float32(sub / 240)), replay=replay)
    nat.check(ctx.lib.dsim_step(ctx.handle, _stream(ctx), n, st.view(), tg.view(), ctypes.byref(a)))
    nz = np.zeros((n, sub, 12))
    nz[:, :, 0:4], nz[:, :, 6:10] = fn, mn
    O = orc.Oracle([t])
    r0, m0 = rigid.copy(), mem.copy()
    assert O.step(rigid, mem, tgt, sub, DT, float(np.float32(sub / 240)), noise=nz) == 0
    assert_step_parity("noise_replay", [t], None, r0, m0, tgt, st.rigid_aos(), st.mem_aos(), rigid, mem, DT,
                       float(np.float32(sub / 240)), sub)
    ctx.close()


@pytest.mark.parametrize("model", ["robobee", "tello", "hexa_6DOF", "hexa_6DOF_simple"])
def test_force_map_vs_reference_recorded_calls(gpu, golden_dir, model):
    """The wrench the kernel applies == the sum of the applyExternalForce/applyExternalTorque calls the
    reference's own _physics made for the same command and noise draws (tests/golden/env_side.npz):
    from level rest one sub-step gives v = (F/m - g z) dt and w = J^-1 tau dt (no damping, no gyro term)."""
    nat, fleet = gpu
    G = np.load(os.path.join(golden_dir, "env_side.npz"))
    t = params.builtin_type(model)
    na = t.n_act
    cmd, fn, mn, vec = (G[f"{model}_fm_{k}"] for k in ("cmd", "f_noise", "m_noise", "vec"))
    n = cmd.shape[0]
    r, ax = np.asarray(t.rotor_pos)[:na], np.asarray(t.rotor_axis)[:na]
    if na == 4:
        F = vec[:, :4].sum(1)
        tau = np.cross(r[None], vec[:, :4]).sum(1) + vec[:, 4]
    else:
        f = ax[None] * vec[:, 0::2, 2:3]
        F = f.sum(1)
        tau = np.cross(r[None], f).sum(1) + (ax[None] * vec[:, 1::2, 2:3]).sum(1)
    ctx = fleet.Context([t])
    st = fleet.FleetState(ctx, n)
    rigid = np.zeros((n, 13)); rigid[:, 2] = 1.0; rigid[:, 6] = 1.0
    st.load_aos(rigid, np.zeros((n, 13)))
    act = torch.zeros((na, st.n_pad), device=ctx.device)
    act[:, :n] = torch.from_numpy(np.ascontiguousarray(cmd.T)).float()
    replay = torch.zeros((1, 2 * na, st.n_pad), device=ctx.device)
    replay[0, 0:na, :n] = torch.from_numpy(np.ascontiguousarray(fn.T)).float()
    replay[0, na:2 * na, :n] = torch.from_numpy(np.ascontiguousarray(mn.T)).float()
    last = torch.zeros((na, st.n_pad), device=ctx.device)
    a = _args(nat, 1, DT, DT, replay=replay, action=act)
    nat.check(ctx.lib.dsim_physics(ctx.handle, _stream(ctx), n, st.view(), last.data_ptr(), ctypes.byref(a)))
    torch.cuda.synchronize()
    got = st.rigid_aos()
    v = (F / t.mass + np.array([0, 0, -t.gravity])) * DT
    w = tau / np.asarray(t.inertia)[None] * DT
    # the state holds the velocity of the point PyBullet reports (the base link's COM: base_offset from the integrated
    # one; zero for the quads) — v_b = v + w x (R d), R the attitude after the sub-step
    qx, qy, qz, qw = (got[:, 3 + k] for k in range(4))
    R = np.stack([np.stack([1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qw * qz), 2 * (qx * qz + qw * qy)], 1),
                  np.stack([2 * (qx * qy + qw * qz), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qw * qx)], 1),
                  np.stack([2 * (qx * qz - qw * qy), 2 * (qy * qz + qw * qx), 1 - 2 * (qx * qx + qy * qy)], 1)], 1)
    v = v + np.cross(w, R @ np.asarray(t.base_offset))
    np.testing.assert_allclose(got[:, 7:10], v, rtol=REL_TOL, atol=REL_TOL * np.abs(v).max())
    np.testing.assert_allclose(got[:, 10:13], w, rtol=REL_TOL, atol=REL_TOL * np.abs(w).max())
    ctx.close()


@pytest.mark.parametrize("sub", [1, 3])        # 1: straight-line single-sub-step kernel, 3: the looped one
def test_inkernel_noise_matches_definition(gpu, sub):
    """In-kernel Threefry4x32-12 / Box-Muller noise == the oracle's restatement of the same definition;
    and it is N(0,.01)/N(0,.001)-distributed."""
    nat, fleet = gpu
    n, seed, step_index = 256, 0x1234ABCD5, 7
    t, ctx, st, tg, rigid, mem, tgt = _make(gpu, "robobee", n, seed=13)
    a = _args(nat, sub, DT, float(np.float32(sub / 240)), seed=seed, step_index=step_index)
    nat.check(ctx.lib.dsim_step(ctx.handle, _stream(ctx), n, st.view(), tg.view(), ctypes.byref(a)))
    O = orc.Oracle([t])
    nz = np.zeros((n, sub, 12))
    for i in range(n):
        for s in range(sub):
            u = O.noise_normals(seed, i, step_index * sub + s, 4, fine=(sub == 1))     # (one sub-step per launch: the fine lattice)
            nz[i, s, 0:4] = u[0:4] * 0.01
            nz[i, s, 6:10] = u[4:8] * 0.001
    assert abs(nz[:, :, 0:4].std() - 0.01) < 1e-3 and abs(nz[:, :, 0:4].mean()) < 1e-3
    r0, m0 = rigid.copy(), mem.copy()
    assert O.step(rigid, mem, tgt, sub, DT, float(np.float32(sub / 240)), noise=nz) == 0
    assert_step_parity(f"inkernel_noise[{sub}]", [t], None, r0, m0, tgt, st.rigid_aos(), st.mem_aos(), rigid, mem, DT,
                       float(np.float32(sub / 240)), sub)
    ctx.close()


# ---------------------------------------------------------------------------
# morphing hexa: 6-DOF INDI + WLS allocation + tilted-rotor physics; mixed fleets (config 5)
# ---------------------------------------------------------------------------
def _query(nat, ctx, what):
    v = ctypes.c_int64(0)
    nat.check(ctx.lib.dsim_query(ctx.handle, _stream(ctx), what, ctypes.byref(v)))
    return v.value


def test_hexa_control_vs_golden(gpu, golden_dir):
    """INDIControl_6DOF.computeControl against the reference-generated vectors, including the
    cases whose WLS allocation needs the full active-set loop (fallback path)."""
    nat, fleet = gpu
    g = np.load(os.path.join(golden_dir, "indi_single_hexa_6DOF.npz"))
    t = params.builtin_type("hexa_6DOF")
    ctx = fleet.Context([t])
    n = g["pos"].shape[0]
    rigid = np.concatenate([g["pos"], g["quat"], g["vel"], g["ang_vel"]], 1)
    mem = np.zeros((n, 13))
    mem[:, 0:3], mem[:, 3:6], mem[:, 6], mem[:, 7:13] = g["last_vel"], g["last_rates"], g["last_thrust"], g["cmd"]
    tgt = np.concatenate([g["target_pos"], g["target_vel"], g["target_acc"], g["target_rpy"][:, 2:3]], 1)
    O = orc.Oracle([t])
    for dt in np.unique(g["dt"]):
        sel = np.where(g["dt"] == dt)[0]
        m = len(sel)
        st = fleet.FleetState(ctx, m)
        tg = fleet.Targets(ctx, m)
        r32, m32, t32, dt32 = f32(rigid[sel]), f32(mem[sel]), f32(tgt[sel]), float(np.float32(dt))
        st.load_aos(r32, m32)
        tg.set_fields(0, torch.from_numpy(np.ascontiguousarray(t32.T)))
        pos_e = torch.zeros((3, st.n_pad), device=ctx.device)
        yaw_e = torch.zeros((st.n_pad,), device=ctx.device)
        a = _args(nat, 0, dt32, dt32)
        nat.check(ctx.lib.dsim_control(ctx.handle, _stream(ctx), m, st.view(), tg.view(), ctypes.byref(a),
                                       pos_e.data_ptr(), yaw_e.data_ptr()))
        got = st.mem_aos()
        # (1) the oracle on the SAME fp32-rounded inputs: the bar on the memory increments, per case
        o32 = m32.copy()
        rc, _, ye32 = O.control(r32, o32, t32, dt32)
        assert rc == 0
        # cases the active-set loop finishes (k_wls_fallback, fp64 on the device) carry the loop's own conditioning:
        # its lstsq rows are scaled by gamma Wv up to 1e8, the fp32 inputs v enter with that gain
        assert_control_parity("hexa_control_golden vs oracle(fp32 in)", [t], None, r32, m32, t32, got, o32, dt32)
        # (2) the pin: oracle on the reference's fp64 inputs == golden
        o64 = mem[sel].copy()
        assert O.control(rigid[sel].copy(), o64, tgt[sel], float(dt))[0] == 0
        gold = _golden_mem(g, sel, 6)
        assert np.abs(o64 - gold).max() < 1e-9
        # (3) against the reference's numbers: + the measured effect of rounding its inputs to fp32, per case
        assert_control_parity("hexa_control_golden vs reference", [t], None, r32, m32, t32, got, gold, dt32,
                              slack=np.abs(o32 - gold))
        ye = yaw_e[:m].double().cpu().numpy()
        pitch = np.array([orc.euler_from_quat(q)[1] for q in r32[:, 3:7]])
        ytol = 8 * ulp32(4.0) / np.maximum(np.abs(np.cos(pitch)), 1e-3)
        assert (_wrap_diff(ye, ye32) <= ytol).all()
        assert (_wrap_diff(ye, g["yaw_e"][sel]) <= ytol + _wrap_diff(ye32, g["yaw_e"][sel])).all()
    assert _query(nat, ctx, 0) > 0          # the fixture does exercise the active-set fallback
    assert _query(nat, ctx, 1) == 0         # and none of the reference calls failed
    ctx.close()


@pytest.mark.parametrize("substeps,layout", [(5, "soa"), (1, "tile64")])
def test_hexa_fused_step_vs_oracle(gpu, substeps, layout):
    nat, fleet = gpu
    n = 2000
    t, ctx, st, tg, rigid, mem, tgt = _make(gpu, "hexa_6DOF", n, layout, seed=31, n_act=6, tilt=0.3, rate=1.0)
    O = orc.Oracle([t])
    dtc = float(np.float32(substeps / 240.0))
    a = _args(nat, substeps, DT, dtc)
    nat.check(ctx.lib.dsim_step(ctx.handle, _stream(ctx), n, st.view(), tg.view(), ctypes.byref(a)))
    r0, m0 = rigid.copy(), mem.copy()
    assert O.step(rigid, mem, tgt, substeps, DT, dtc) == 0
    # (the WLS first-iteration matrix M1 has entries up to ~1e-2 per (rad/s^2): its gain on the fp32 rounding of the
    # finite-difference accelerations is what step_terms charges the cmd fields with, per drone)
    assert_step_parity(f"hexa_fused_step[{substeps},{layout}]", [t], None, r0, m0, tgt, st.rigid_aos(), st.mem_aos(),
                       rigid, mem, DT, dtc, substeps)
    ctx.close()


@pytest.mark.parametrize("layout,form", [("soa", "default"), ("tile64", "default")])
@pytest.mark.parametrize("sub", [1, 2])
def test_mixed_fleet_vs_oracle(gpu, sub, layout, form):
    """Config 5 layout kept in the caller's own order: even index robobee (quad INDI), odd index hexa_6DOF (6DOF INDI +
    WLS), one type_id byte per drone; in-kernel noise on.  Both forms of the mixed-fleet kernel the product ships (LDS-DMA
    staging in natural order, partition by ballots): two waves per tile with 1 KB DMAs on the wave-tiled layout
    (k_step_mixed4), three waves with row DMAs otherwise (k_step_mixed3).  (Rounds 1-2's other forms: in the git history, tools/variants/ up to round 5.)"""
    nat, fleet = gpu
    n = 3000
    types = [params.builtin_type("robobee"), params.builtin_type("hexa_6DOF")]
    ctx = fleet.Context(types)
    assert ctx.n_fields == 26
    st = fleet.FleetState(ctx, n, layout)
    tg = fleet.Targets(ctx, n, layout)
    rigid, mem, tgt = random_fleet(np.random.default_rng(41), n, n_act=6, tilt=0.3, rate=1.0)
    tid = (np.arange(n) % 2).astype(np.uint8)
    mem[tid == 0, 11:13] = 0.0
    st.load_aos(rigid, mem)
    tg.set_fields(0, torch.from_numpy(np.ascontiguousarray(tgt.T)))
    tid_dev = torch.zeros(st.n_pad, dtype=torch.uint8, device=ctx.device)
    tid_dev[:n] = torch.from_numpy(tid)
    seed, sidx = 99, 5
    a = _args(nat, sub, DT, float(np.float32(sub / 240)), seed=seed, step_index=sidx, type_id=tid_dev,
              options=0)
    nat.check(ctx.lib.dsim_step(ctx.handle, _stream(ctx), n, st.view(), tg.view(), ctypes.byref(a)))
    O = orc.Oracle(types)
    nz = np.zeros((n, sub, 12))
    for i in range(n):
        na = 4 if tid[i] == 0 else 6
        for s_ in range(sub):
            u = O.noise_normals(seed, i, sidx * sub + s_, na, fine=(sub == 1))
            nz[i, s_, 0:na], nz[i, s_, 6:6 + na] = rotor_noise(types[int(tid[i])], u)
    r0, m0 = rigid.copy(), mem.copy()
    assert O.step(rigid, mem, tgt, sub, DT, float(np.float32(sub / 240)), noise=nz, type_id=tid) == 0
    assert_step_parity(f"mixed_fleet[{sub},{layout},{form}]", types, tid, r0, m0, tgt, st.rigid_aos(), st.mem_aos(), rigid, mem, DT,
                       float(np.float32(sub / 240)), sub)
    ctx.close()


@pytest.mark.parametrize("sub", [1, 2])
def test_type_major_runs_equal_mixed_kernel_and_oracle(gpu, sub):
    """Type-major storage (dsim_step_args.runs): three types grouped in runs (here starting at multiples of 256,
    fleet.type_major_order's padded form), each run stepped by the single-type kernel of its kind — same result as the
    mixed-fleet kernel on the same storage (same law, drone-keyed noise) and as the oracle."""
    nat, fleet = gpu
    types = [params.builtin_type("robobee"), params.builtin_type("hexa_6DOF"), params.builtin_type("tello")]
    rng = np.random.default_rng(43)
    n = 1500
    caller_tid = rng.integers(0, 3, n).astype(np.uint8)               # the caller's (interleaved) numbering
    slot, n_slots, slot_types = fleet.type_major_order(caller_tid)
    assert n_slots % 256 == 0 and (slot_types[slot] == caller_tid).all() and len(set(slot)) == n
    runs = fleet.type_runs(slot_types)
    assert [r[2] for r in runs] == [0, 1, 2] and all(r[0] % 256 == 0 for r in runs)
    rigid, mem, tgt = random_fleet(rng, n_slots, n_act=6, tilt=0.3, rate=1.0)
    mem[slot_types != 1, 11:13] = 0.0
    seed, sidx = 77, 3
    O = orc.Oracle(types)
    dtc = float(np.float32(sub / 240))
    results = []
    for use_runs in (True, False):
        ctx = fleet.Context(types)
        st, tg = fleet.FleetState(ctx, n_slots), fleet.Targets(ctx, n_slots)
        st.load_aos(rigid, mem)
        tg.set_fields(0, torch.from_numpy(np.ascontiguousarray(tgt.T)))
        tid_dev = torch.zeros(st.n_pad, dtype=torch.uint8, device=ctx.device)
        tid_dev[:n_slots] = torch.from_numpy(slot_types)
        a = _args(nat, sub, DT, dtc, seed=seed, step_index=sidx, type_id=tid_dev)
        arr = (nat.TypeRun * len(runs))()
        for k, (f, c, ty) in enumerate(runs):
            arr[k].first, arr[k].count, arr[k].type = f, c, ty
        if use_runs:
            a.runs, a.n_runs = ctypes.addressof(arr), len(runs)
        for k in range(3):
            # every step against the oracle started from the DEVICE's previous state: per-step bar, no accumulation
            r0, m0 = st.rigid_aos(), st.mem_aos()
            a.step_index = sidx + k
            nat.check(ctx.lib.dsim_step(ctx.handle, _stream(ctx), n_slots, st.view(), tg.view(), ctypes.byref(a)))
            nz = np.zeros((n_slots, sub, 12))
            for i in range(n_slots):
                na = 6 if slot_types[i] == 1 else 4
                for s_ in range(sub):
                    u = O.noise_normals(seed, i, (sidx + k) * sub + s_, na, fine=(sub == 1))
                    nz[i, s_, 0:na], nz[i, s_, 6:6 + na] = rotor_noise(types[int(slot_types[i])] if int(slot_types[i]) < len(types) else types[0], u)
            r1, m1 = r0.copy(), m0.copy()
            assert O.step(r1, m1, tgt, sub, DT, dtc, noise=nz, type_id=slot_types) == 0
            assert_step_parity(f"type_major[{sub},runs={use_runs}]", types, slot_types, r0, m0, tgt, st.rigid_aos(),
                               st.mem_aos(), r1, m1, DT, dtc, sub)
        torch.cuda.synchronize()
        results.append((st.rigid_aos(), st.mem_aos()))
        ctx.close()
    # two differently compiled kernels of the same law, three steps: equal up to fp32 contraction/rounding
    assert rel_err(results[0][0], results[1][0], RIGID_SCALE).max() < 0.2 * REL_TOL
    assert rel_err(results[0][1], results[1][1], MEM_SCALE).max() < 0.5 * REL_TOL
    # a run that leaves the fleet, or names a type the table does not hold, is refused (a run may START anywhere:
    # tests/test_gpu_storage_halo_placement.py flies runs that share tiles)
    ctx = fleet.Context(types)
    st, tg = fleet.FleetState(ctx, n_slots), fleet.Targets(ctx, n_slots)
    tid_dev = torch.zeros(st.n_pad, dtype=torch.uint8, device=ctx.device)
    a = _args(nat, 1, DT, DT, type_id=tid_dev)
    for first, count, ty in ((100, st.n_pad, 0), (0, 50, 3), (-256, 50, 0)):
        bad = (nat.TypeRun * 1)(); bad[0].first, bad[0].count, bad[0].type = first, count, ty
        a.runs, a.n_runs = ctypes.addressof(bad), 1
        assert ctx.lib.dsim_step(ctx.handle, _stream(ctx), n_slots, st.view(), tg.view(), ctypes.byref(a)) == -1   # DSIM_E_ARG
    ctx.close()


def test_hexa_hover_physics(gpu):
    """Level hexa at hover PWM (tilted rotors: vertical thrust = weight, lateral components and all
    torques cancel) stays put."""
    nat, fleet = gpu
    t = params.builtin_type("hexa_6DOF")
    ctx = fleet.Context([t])
    n = 256
    st = fleet.FleetState(ctx, n)
    rigid = np.zeros((n, 13)); rigid[:, 2] = 1.0; rigid[:, 6] = 1.0
    mem = np.zeros((n, 13)); mem[:, 7:13] = t.hover_pwm
    st.load_aos(rigid, mem)
    a = _args(nat, 240, DT, DT)
    nat.check(ctx.lib.dsim_physics(ctx.handle, _stream(ctx), n, st.view(), None, ctypes.byref(a)))
    out = st.rigid_aos()
    # the URDF's arm yaw angles are rounded (0.523, 1.57, 2.617), so the cancellation is not exact
    assert np.abs(out[:, 0:3] - rigid[:, 0:3]).max() < 1e-2 and np.abs(out[:, 10:13]).max() < 5e-2
    O = orc.Oracle([t])
    O.physics(rigid, mem, 240, DT)
    assert rel_err(out, rigid, RIGID_SCALE).max() < REL_TOL
    ctx.close()


# ---------------------------------------------------------------------------
# neighbour downwash (formula P8) and the external-force input
# ---------------------------------------------------------------------------
def test_downwash_vs_bruteforce_oracle(gpu):
    nat, fleet = gpu
    from dronesim_amd.downwash import Downwash
    n = 4000
    types = [params.builtin_type("robobee"), params.builtin_type("hexa_6DOF")]
    ctx = fleet.Context(types)
    st = fleet.FleetState(ctx, n)
    rng = np.random.default_rng(51)
    rigid, mem, _ = random_fleet(rng, n, n_act=6)
    rigid[:, 0] = f32(rng.uniform(-40, 160, n)); rigid[:, 1] = f32(rng.uniform(-30, 70, n)); rigid[:, 2] = f32(rng.uniform(0.5, 20.5, n))
    rigid[7, 0:3] = rigid[8, 0:3] + [0.0, 0.0, 0.5]          # one drone right above another: strong term
    rigid[9, 0:3] = rigid[8, 0:3]                            # coincident pair: dz = 0 drops out
    st.load_aos(rigid, mem)
    tid = (np.arange(n) % 2).astype(np.uint8)
    tid_dev = torch.zeros(st.n_pad, dtype=torch.uint8, device=ctx.device); tid_dev[:n] = torch.from_numpy(tid)
    dw = Downwash(ctx, st, tid_dev)
    f = dw.compute().cpu().numpy()
    ref = orc.Oracle(types).downwash(rigid, rigid[:, 0:3], type_id=tid)
    assert (ref < 0).sum() > n // 2 and ref.min() < -1e-3    # the field is populated
    np.testing.assert_array_equal(f[0:2], 0.0)
    assert_downwash("downwash bruteforce", f[2, :n], ref, types, tid, rigid[:, 0:3], rigid[:, 0:3])
    # world = local + a remote shard (what another rank would contribute through the all-gather)
    remote = f32(np.stack([rng.uniform(-40, 160, 1500), rng.uniform(-30, 70, 1500), rng.uniform(0.5, 25, 1500)], 1))
    world = np.concatenate([remote[:700], rigid[:, 0:3], remote[700:]])      # this "rank's" shard sits in the middle
    f2 = dw.compute(torch.from_numpy(np.ascontiguousarray(world.T)).float().to(ctx.device), local_offset=700).cpu().numpy()
    ref2 = orc.Oracle(types).downwash(rigid, world, type_id=tid)
    assert_downwash("downwash bruteforce + remote shard", f2[2, :n], ref2, types, tid, rigid[:, 0:3], world)
    assert np.abs(ref2 - ref).max() > 1e-4                    # the remote shard matters
    ctx.close()


def test_downwash_bucket_grid_with_overflowing_cells(gpu):
    """The bucket form of the neighbour grid (one binning pass, cell-centred LDS-tiled query): a swarm of 700 drones
    packed into a 7 m square overflows the 64-entry buckets of the cells it sits in many times over; the overflow list
    keeps the result equal to the brute-force sum.  (Also: a dense world, 5 m cells and the two-wave query.)"""
    nat, fleet = gpu
    from dronesim_amd.downwash import Downwash
    n = 2500
    ctx = fleet.Context([params.builtin_type("robobee")])
    st = fleet.FleetState(ctx, n)
    rng = np.random.default_rng(58)
    rigid, mem, _ = random_fleet(rng, n)
    rigid[:, 0] = f32(rng.uniform(0, 300, n)); rigid[:, 1] = f32(rng.uniform(0, 300, n)); rigid[:, 2] = f32(rng.uniform(0.5, 20, n))
    rigid[:700, 0] = f32(rng.uniform(101, 108, 700)); rigid[:700, 1] = f32(rng.uniform(201, 208, 700))   # the swarm
    st.load_aos(rigid, mem)
    dw = Downwash(ctx, st)
    f = dw.compute().cpu().numpy()
    g = dw._last
    assert ctx.lib.dsim_downwash_prebin_ok(g.m, g.nx, g.ny) == 1      # this shape takes the bucket form (10 m cells: sparse world)
    ref = orc.Oracle([params.builtin_type("robobee")]).downwash(rigid, rigid[:, 0:3])
    assert (ref[:700] < 0).sum() > 600
    rb = [params.builtin_type("robobee")]
    assert_downwash("downwash overflowing cells", f[2, :n], ref, rb, None, rigid[:, 0:3], rigid[:, 0:3])
    f_again = dw.compute().cpu().numpy()                        # second build: the double-buffered counts were re-zeroed
    assert_downwash("downwash overflowing cells", f_again[2, :n], ref, rb, None, rigid[:, 0:3], rigid[:, 0:3])
    ctx.close()
    # a dense world: 0.8 drones per m^2 -> 5 m cells, 5 x 5 neighbourhoods of ~500 entries (two-wave query, 12 KB tile),
    # plus a knot of 400 drones in one cell whose neighbourhood does not fit the tile (several fills)
    n = 6000
    ctx = fleet.Context([params.builtin_type("robobee")])
    st = fleet.FleetState(ctx, n)
    rigid, mem, _ = random_fleet(rng, n)
    rigid[:, 0] = f32(rng.uniform(0, 85, n)); rigid[:, 1] = f32(rng.uniform(0, 85, n)); rigid[:, 2] = f32(rng.uniform(0.5, 20, n))
    rigid[:400, 0] = f32(rng.uniform(41, 44, 400)); rigid[:400, 1] = f32(rng.uniform(41, 44, 400))
    st.load_aos(rigid, mem)
    dw = Downwash(ctx, st)
    f = dw.compute().cpu().numpy()
    assert dw._last.cell == 5.0
    ref = orc.Oracle([params.builtin_type("robobee")]).downwash(rigid, rigid[:, 0:3])
    assert_downwash("downwash dense knot", f[2, :n], ref, rb, None, rigid[:, 0:3], rigid[:, 0:3])
    ctx.close()


@pytest.mark.parametrize("world", ["crowd", "vast"])
def test_downwash_counting_sort_form(gpu, world):
    """Worlds the bucket form does not take — more than 40 drones per cell on average (a crowd: 2.4 per m^2), or more
    than 65 536 cells (3 km x 3 km) — go through the counting-sort grid (count, scan, scatter, sorted query): against
    the brute-force sum, two types, and with another rank's drones among the candidates."""
    nat, fleet = gpu
    from dronesim_amd.downwash import Downwash
    types = [params.builtin_type("robobee"), params.builtin_type("hexa_6DOF")]
    n = 6000
    side = 50.0 if world == "crowd" else 3000.0
    ctx = fleet.Context(types)
    st = fleet.FleetState(ctx, n)
    rng = np.random.default_rng(91)
    rigid, mem, _ = random_fleet(rng, n, n_act=6)
    rigid[:, 0] = f32(rng.uniform(0, side, n)); rigid[:, 1] = f32(rng.uniform(0, side, n)); rigid[:, 2] = f32(rng.uniform(0.5, 20.5, n))
    if world == "vast":                                      # clusters, so that pairs exist at all
        c = rng.integers(0, 40, n)
        ctr = rng.uniform(100, side - 100, (40, 2))
        rigid[:, 0:2] = f32(ctr[c] + rng.uniform(-15, 15, (n, 2)))
    st.load_aos(rigid, mem)
    tid = (np.arange(n) % 2).astype(np.uint8)
    tid_dev = torch.zeros(st.n_pad, dtype=torch.uint8, device=ctx.device); tid_dev[:n] = torch.from_numpy(tid)
    dw = Downwash(ctx, st, tid_dev)
    f = dw.compute().cpu().numpy()
    g = dw._last
    assert ctx.lib.dsim_downwash_prebin_ok(g.m, g.nx, g.ny) == 0 and g.cell == 10.0       # not the bucket form
    ref = orc.Oracle(types).downwash(rigid, rigid[:, 0:3], type_id=tid)
    assert (ref < 0).sum() > n // 4
    assert_downwash(f"downwash counting sort[{world}]", f[2, :n], ref, types, tid, rigid[:, 0:3], rigid[:, 0:3])
    remote = f32(np.concatenate([rigid[:800, 0:2] + rng.uniform(-3, 3, (800, 2)), rng.uniform(0.5, 25, (800, 1))], 1))
    world_pos = np.concatenate([remote[:300], rigid[:, 0:3], remote[300:]])
    f2 = dw.compute(torch.from_numpy(np.ascontiguousarray(world_pos.T)).float().to(ctx.device), local_offset=300).cpu().numpy()
    ref2 = orc.Oracle(types).downwash(rigid, world_pos, type_id=tid)
    assert_downwash(f"downwash counting sort[{world}] + remote", f2[2, :n], ref2, types, tid, rigid[:, 0:3], world_pos)
    ctx.close()


@pytest.mark.parametrize("density", ["sparse", "dense"])
def test_downwash_receiver_coefficients_of_many_types(gpu, density):
    """The receiver's three downwash coefficients come from a table of ALL types that the query kernel keeps in LDS:
    five airframes whose DW1, DW2, DW3 and propeller radius all differ (the shipped ones share DW2 and DW3), in both
    forms of the query."""
    import dataclasses
    nat, fleet = gpu
    from dronesim_amd.downwash import Downwash
    base = params.builtin_type("robobee")
    types = [dataclasses.replace(base, name=f"rb{k}", dw_coeff=(2267.18 * (1 + 0.3 * k), 0.16 + 0.02 * k, -0.11 + 0.015 * k),
                                 prop_radius=base.prop_radius * (1 + 0.2 * k)) for k in range(5)]
    n = 4000
    side = 220.0 if density == "sparse" else 70.0
    ctx = fleet.Context(types)
    st = fleet.FleetState(ctx, n)
    rng = np.random.default_rng(77)
    rigid, mem, _ = random_fleet(rng, n)
    rigid[:, 0] = f32(rng.uniform(0, side, n)); rigid[:, 1] = f32(rng.uniform(0, side, n)); rigid[:, 2] = f32(rng.uniform(0.5, 12, n))
    st.load_aos(rigid, mem)
    tid = rng.integers(0, 5, n).astype(np.uint8)
    tid_dev = torch.zeros(st.n_pad, dtype=torch.uint8, device=ctx.device); tid_dev[:n] = torch.from_numpy(tid)
    dw = Downwash(ctx, st, tid_dev)
    f = dw.compute().cpu().numpy()
    assert dw._last.cell == (10.0 if density == "sparse" else 5.0)
    ref = orc.Oracle(types).downwash(rigid, rigid[:, 0:3], type_id=tid)
    assert_downwash(f"downwash many types[{density}]", f[2, :n], ref, types, tid, rigid[:, 0:3], rigid[:, 0:3])
    wrong = orc.Oracle(types).downwash(rigid, rigid[:, 0:3], type_id=(tid + 1) % 5)
    assert (np.abs(wrong - ref) / (np.abs(ref) + 1e-3)).max() > 1e-2          # the coefficients do matter
    ctx.close()


@pytest.mark.parametrize("heights", ["uniform", "flat", "two_layers", "ties"])
def test_downwash_dense_world_height_bands(gpu, heights):
    """The dense form of the query sorts a cell's receivers by height and lays its tile out in height bands (a group of
    receivers reads only the candidates above its lowest member): against the brute-force sum on worlds that stress the
    banding — uniform heights, a FLAT fleet (every candidate at or below every receiver: all bands empty, zero force), two
    thin layers, many exactly equal heights — with two types, and with another rank's drones among the candidates."""
    nat, fleet = gpu
    from dronesim_amd.downwash import Downwash
    n = 5200
    types = [params.builtin_type("robobee"), params.builtin_type("hexa_6DOF")]
    ctx = fleet.Context(types)
    st = fleet.FleetState(ctx, n)
    rng = np.random.default_rng({"uniform": 1, "flat": 2, "two_layers": 3, "ties": 4}[heights])
    rigid, mem, _ = random_fleet(rng, n, n_act=6)
    rigid[:, 0] = f32(rng.uniform(0, 80, n)); rigid[:, 1] = f32(rng.uniform(0, 80, n))          # 0.8 drones per m^2
    z = {"uniform": rng.uniform(0.5, 20.5, n), "flat": np.full(n, 3.0),
         "two_layers": np.where(rng.random(n) < 0.5, 2.0, 2.5) + rng.uniform(0, 1e-3, n),
         "ties": np.round(rng.uniform(0.5, 6.5, n) * 2) / 2}[heights]                           # 13 distinct heights
    rigid[:, 2] = f32(z)
    st.load_aos(rigid, mem)
    tid = (rng.random(n) < 0.5).astype(np.uint8)
    tid_dev = torch.zeros(st.n_pad, dtype=torch.uint8, device=ctx.device); tid_dev[:n] = torch.from_numpy(tid)
    dw = Downwash(ctx, st, tid_dev)
    remote = f32(np.stack([rng.uniform(0, 80, 1300), rng.uniform(0, 80, 1300), rng.uniform(0.5, 22, 1300)], 1))
    world = np.concatenate([remote[:500], rigid[:, 0:3], remote[500:]])
    f = dw.compute(torch.from_numpy(np.ascontiguousarray(world.T)).float().to(ctx.device), local_offset=500).cpu().numpy()
    assert dw._last.cell == 5.0                                                                # the dense form
    ref = orc.Oracle(types).downwash(rigid, world, type_id=tid)
    assert_downwash(f"downwash bands[{heights}]", f[2, :n], ref, types, tid, rigid[:, 0:3], world)
    own = orc.Oracle(types).downwash(rigid, rigid[:, 0:3], type_id=tid)
    if heights == "flat":
        assert np.abs(own).max() == 0.0 and (ref < 0).sum() > 100        # only the remote drones above push down
    else:
        assert (own < 0).sum() > n // 3
    f_own = dw.compute().cpu().numpy()                                   # the fleet alone
    assert_downwash(f"downwash bands[{heights}] own", f_own[2, :n], own, types, tid, rigid[:, 0:3], rigid[:, 0:3])
    ctx.close()


@pytest.mark.parametrize("opts", [1, 2, 3])          # DSIM_OPT_DRAG, DSIM_OPT_GROUND, both
def test_drag_and_ground_effect_vs_oracle(gpu, opts):
    """Formulas P6/P7 (dead code in the reference fork) as switchable physics terms: Env.step with an
    explicit action + last_clipped_action (drag uses the previous step's rpm), then a fused step."""
    nat, fleet = gpu
    n = 1500
    t, ctx, st, tg, rigid, mem, tgt = _make(gpu, "robobee", n, seed=61, tilt=0.6)
    rigid[:, 2] = f32(np.random.default_rng(62).uniform(0.02, 0.6, n))      # near the ground: P7 matters
    rigid[:40, 3:7] = f32(np.tile(orc.quat_from_euler([2.0, 0.1, 0.3]), (40, 1)))   # |roll| > pi/2: no ground effect
    st.load_aos(rigid, mem)
    rng = np.random.default_rng(63)
    act = f32(rng.uniform(0.2, 0.9, (n, 4)))
    prev = f32(rng.uniform(0.2, 0.9, (n, 4)))
    act_dev = torch.zeros((4, st.n_pad), device=ctx.device); act_dev[:, :n] = torch.from_numpy(act.T).float()
    echo = torch.zeros((4, st.n_pad), device=ctx.device); echo[:, :n] = torch.from_numpy(prev.T).float()
    a = _args(nat, 3, DT, DT, options=opts, action=act_dev)
    nat.check(ctx.lib.dsim_physics(ctx.handle, _stream(ctx), n, st.view(), echo.data_ptr(), ctypes.byref(a)))
    O = orc.Oracle([t])
    a6 = np.zeros((n, 6)); a6[:, :4] = act
    last = np.zeros((n, 6)); last[:, :4] = prev
    before = rigid.copy()
    O.physics(rigid, mem, 3, DT, action=a6, options=opts, last_action=last)
    # ground effect multiplies the rotor thrusts by up to 1 + coeff (r / 4 h_clip)^2: charged to the wrench terms
    boost = 1.0 + (t.gnd_eff_coeff * (t.prop_radius / (4 * t.gnd_eff_h_clip)) ** 2 if opts & 2 else 0.0)
    assert_step_parity(f"drag_ground_physics[{opts}]", [t], None, before, mem, tgt, st.rigid_aos(), None, rigid, None,
                       DT, DT, 3, control=False, k=K_ULP * 3 * boost, action=act)
    plain = before.copy()
    O.physics(plain, mem, 3, DT, action=a6)
    assert np.abs(plain - rigid).max() > 1e-5                         # the option really changes the step
    # fused step with the same options (general kernel; drag falls back to the current action's rpm)
    a2 = _args(nat, 2, DT, float(np.float32(2 / 240)), options=opts)
    r0, m0 = st.rigid_aos(), st.mem_aos()                 # from the device's own state after the first call
    nat.check(ctx.lib.dsim_step(ctx.handle, _stream(ctx), n, st.view(), tg.view(), ctypes.byref(a2)))
    r1, m1 = r0.copy(), m0.copy()
    assert O.step(r1, m1, tgt, 2, DT, float(np.float32(2 / 240)), options=opts) == 0
    assert_step_parity(f"drag_ground_fused[{opts}]", [t], None, r0, m0, tgt, st.rigid_aos(), st.mem_aos(), r1, m1,
                       DT, float(np.float32(2 / 240)), 2, k=K_ULP * 2 * boost)
    ctx.close()


def test_adjacency_vs_bruteforce(gpu):
    """Fleet-scale neighbourhood query == the reference's O(N^2) rule (BaseAviary.py:913-921)."""
    nat, fleet = gpu
    from dronesim_amd.downwash import Downwash
    n, radius = 3000, 7.5
    t = params.builtin_type("robobee")
    ctx = fleet.Context([t])
    st = fleet.FleetState(ctx, n)
    rng = np.random.default_rng(91)
    rigid, mem, _ = random_fleet(rng, n)
    rigid[:, 0:3] = f32(np.stack([rng.uniform(0, 120, n), rng.uniform(0, 80, n), rng.uniform(0, 15, n)], 1))
    st.load_aos(rigid, mem)
    cnt, lst = Downwash(ctx, st).adjacency(radius, max_k=64)
    p = rigid[:, 0:3].astype(np.float32)
    d2 = ((p[:, None, :] - p[None, :, :]) ** 2).sum(-1)
    adj = (d2 < np.float32(radius) ** 2) & ~np.eye(n, dtype=bool)
    np.testing.assert_array_equal(cnt.cpu().numpy(), adj.sum(1))
    L = lst.cpu().numpy()
    for i in rng.choice(n, 200, replace=False):
        got = set(int(x) for x in L[:, i] if x >= 0)
        assert got == set(np.nonzero(adj[i])[0].tolist()) or (adj[i].sum() > 64 and len(got) == 64)
    ctx.close()


def test_step_with_downwash_env(gpu):
    """Physics.PYB_DW env: fused step with the downwash force vs the oracle fed the brute-force force."""
    from dronesim_amd.envs import CtrlAviary, Physics
    n = 512
    rng = np.random.default_rng(52)
    xyz = np.stack([rng.uniform(0, 30, n), rng.uniform(0, 30, n), rng.uniform(0.5, 10, n)], 1)
    env = CtrlAviary(["robobee"], n, initial_xyzs=xyz, physics=Physics.PYB_DW, aggregate_phy_steps=1,
                     noise_seed=0, dict_io=False)
    from dronesim_amd.fleet import Targets
    tg = Targets(env.ctx, n)
    tg.set(pos=f32(xyz).T, yaw=0.0)
    t = params.builtin_type("robobee")
    O = orc.Oracle([t])
    tgt = np.concatenate([f32(xyz), np.zeros((n, 7))], 1)
    fz_max = 0.0
    for k in range(5):
        rigid, mem = env.state.rigid_aos(), env.state.mem_aos()      # every step from the device's previous state
        r0, m0 = rigid.copy(), mem.copy()
        fz = O.downwash(rigid, rigid[:, 0:3])
        fz_max = max(fz_max, float(np.abs(fz).max()))
        ext = np.zeros((n, 3)); ext[:, 2] = f32(fz)
        a6 = act = None
        if k == 0:
            a6 = np.zeros((n, 6)); a6[:, :4] = 0.45
            act = a6[:, :4]
        env.step_fused(tg, action=np.full((n, 4), 0.45, dtype=np.float32) if k == 0 else None)
        assert O.step(rigid, mem, tgt, 1, DT, DT, action=a6, ext_force=ext) == 0
        # the downwash force enters the velocity update like one more rotor thrust: |fz| / m joins the accelerations
        extra = float((np.abs(fz) / t.mass).max()) / t.gravity
        assert_step_parity("downwash_env", [t], None, r0, m0, tgt, env.state.rigid_aos(), env.state.mem_aos(), rigid, mem,
                           DT, DT, 1, k=K_ULP * (1.0 + extra), action=act)
    assert fz_max > 1e-3                                      # the term is live in this fleet
    env.close()


def _sharded_dw_worker(rank, world, port, mode, out):
    """One rank of a 2-rank fleet sharing the single GPU (gloo stands in for RCCL, which wants one device
    per rank): mixed quad/hexa slab shard with neighbour downwash, against a whole-world run."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dronesim_amd.envs import CtrlAviary, Physics
    from dronesim_amd.fleet import Targets
    n, slab = 1536, 60.0
    xyzs, models = [], []
    for r in range(world):
        rng = np.random.default_rng(900 + r)
        xyzs.append(np.stack([rng.uniform(r * slab, (r + 1) * slab, n), rng.uniform(0, 40, n), rng.uniform(0.5, 12, n)], 1))
        models += ["robobee" if i % 2 == 0 else "hexa_6DOF" for i in range(n)]
    xyz_w = np.concatenate(xyzs)

    def fly(env, xyz, steps=12):
        tg = Targets(env.ctx, xyz.shape[0])
        tg.set(pos=f32(xyz).T + np.array([[0.5], [0.0], [0.2]], dtype=np.float32), yaw=0.3)
        for _ in range(steps):
            env.step_fused(tg)
        torch.cuda.synchronize()
        return env.state.rigid_aos(), env.state.mem_aos()

    mine = CtrlAviary(models[rank * n:(rank + 1) * n], n, initial_xyzs=xyzs[rank], physics=Physics.PYB_DW,
                      noise_seed=0, dict_io=False, dist=dist, downwash_exchange=mode)
    r_s, m_s = fly(mine, xyzs[rank])
    halo = mine._downwash.halo
    sent = halo.sent_per_step if halo is not None else None
    whole = CtrlAviary(models, world * n, initial_xyzs=xyz_w, physics=Physics.PYB_DW, noise_seed=0, dict_io=False)
    r_w, m_w = fly(whole, xyz_w)
    sl = slice(rank * n, (rank + 1) * n)
    moved = float(np.abs(r_w[sl, 0:3] - xyzs[rank]).max())
    out[rank] = (float(rel_err(r_s, r_w[sl], RIGID_SCALE).max()), float(np.abs(m_s - m_w[sl]).max()), sent, moved)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["allgather", "halo"])
def test_two_rank_sharded_downwash_equals_whole_fleet(gpu, mode):
    """The multi-GPU exchange step end to end on the kernels: each of two ranks flies its slab shard of a
    mixed fleet with neighbour downwash on, positions exchanged by all-gather or by halo send/recv; the
    shard's trajectory equals the same drones' trajectory in a single whole-fleet run."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_sharded_dw_worker, args=(2, port, mode, out), nprocs=2, join=True)
    for r in (0, 1):
        err_r, err_m, sent, moved = out[r]
        assert err_r < 1e-5 and err_m < 1e-4, (r, out[r])
        assert moved > 1e-3
        if mode == "halo":
            assert 0 < sent < 1536          # only the boundary strip travels


# ---------------------------------------------------------------------------
# config 3: waypoint-table tracking (examples/fly_INDI_TrajectoryTrack.py) and multi-step launches
# ---------------------------------------------------------------------------
def _traj_fleet(gpu, golden_dir, n, noise_seed=0):
    from dronesim_amd.envs import CtrlAviary
    from dronesim_amd.fleet import WaypointTargets
    g = np.load(os.path.join(golden_dir, "traj_track_waypoints.npz"))
    n_wp = g["target_pos"].shape[0]
    side = int(math.ceil(math.sqrt(n)))
    off = np.stack([np.arange(n) % side, np.arange(n) // side, np.zeros(n)], 1).astype(np.float64)
    xyz = g["gates"][0][None, :] + off                       # each drone starts at gate 0 + its grid offset
    wp0 = np.array([int((i * n_wp / 6) % n_wp) for i in range(n)])    # fly_INDI_TrajectoryTrack.py:187-189
    env = CtrlAviary(["robobee"], n, initial_xyzs=xyz, aggregate_phy_steps=2, noise_seed=noise_seed, dict_io=False)
    wps = WaypointTargets(env.ctx, n, g["target_pos"], g["target_vel"], g["target_acc"], g["target_yaw"],
                          wp_counters=wp0, offsets=off)
    return env, wps, g, off, wp0


def test_waypoint_tracking_vs_oracle(gpu, golden_dir):
    """Config 3 semantics: 2 sub-steps per control, control_timestep 2/240 (the reference plays the
    96 Hz table at 0.8x speed), waypoint row advances by one per control step and wraps."""
    n = 300
    env, wps, g, off, wp0 = _traj_fleet(gpu, golden_dir, n)
    t = params.builtin_type("robobee")
    O = orc.Oracle([t])
    rigid, mem = env.state.rigid_aos(), env.state.mem_aos()
    tab = np.concatenate([g["target_pos"], g["target_vel"], g["target_acc"], g["target_yaw"][:, None]], 1).astype(np.float32).astype(np.float64)
    dtc = float(np.float32(2 / 240))
    wp = wp0.copy()
    n_wp = tab.shape[0]
    wp[5] = n_wp - 3                                       # forces a wrap inside the run
    wps.counters[5] = n_wp - 3
    for k in range(40):
        r0, m0 = env.state.rigid_aos(), env.state.mem_aos()      # this step from the device's own previous state
        env.step_fused(wps, control_timestep=dtc, action=np.full((n, 4), 0.4, dtype=np.float32) if k == 0 else None)
        tgt = tab[wp].copy()
        tgt[:, 0:3] = f32(tgt[:, 0:3].astype(np.float32) + off.astype(np.float32))     # fp32 add, as the kernel does
        a6 = None
        if k == 0:
            a6 = np.zeros((n, 6)); a6[:, :4] = 0.4
        assert O.step(rigid, mem, tgt, 2, DT, dtc, action=a6) == 0       # the oracle's free-running trajectory
        r1, m1 = r0.copy(), m0.copy()
        assert O.step(r1, m1, tgt, 2, DT, dtc, action=a6) == 0           # ... and its step from the device's state
        assert_step_parity("waypoint_tracking", [t], None, r0, m0, tgt, env.state.rigid_aos(), env.state.mem_aos(),
                           r1, m1, DT, dtc, 2, action=None if a6 is None else a6[:, :4])
        wp = np.where(wp < n_wp - 1, wp + 1, 0)
    np.testing.assert_array_equal(wps.counters[:n].cpu().numpy(), wp)
    # accumulated drift between the two closed loops over the 40 steps (feedback keeps them together)
    assert rel_err(env.state.rigid_aos(), rigid, RIGID_SCALE).max() < 1e-3
    assert rel_err(env.state.mem_aos(), mem, MEM_SCALE).max() < 1e-3
    env.close()


def test_device_trajectory_sampler_vs_reference_table(gpu, golden_dir):
    """dsim_traj_sample == the reference trajGenerator's own samples (golden table), per drone with its
    own start time and yaw memory; then the fused step consumes the sampled targets."""
    nat, fleet = gpu
    g = np.load(os.path.join(golden_dir, "traj_track_waypoints.npz"))
    t = params.builtin_type("robobee")
    ctx = fleet.Context([t])
    n = 3
    starts = [0, 100, 400]                       # drone j starts at table row starts[j]
    tr = fleet.TrajectoryTargets(ctx, n, g["coeffs"], g["TS"], t0=[g["t"][s_] for s_ in starts],
                                 offsets=[[0, 0, 0], [10, 0, 0], [0, 5, 1]])
    # yaw memory of a sampler that starts mid-trajectory: the reference's state after sampling rows < start
    for j, s_ in enumerate(starts):
        ys = np.zeros(3)
        for k in range(s_):
            orc.traj_sample(g["coeffs"], g["TS"], g["t"][k], ys)
        tr.yaw_state[:, j] = torch.from_numpy(ys)
    off = np.array([[0, 0, 0], [10, 0, 0], [0, 5, 1.0]])
    for k in range(300):
        tr.sample(1.0 / 96)
        T = tr.fields(0, 10).T.double().cpu().numpy()
        for j, s_ in enumerate(starts):
            row = s_ + k
            np.testing.assert_allclose(T[j, 0:3], g["target_pos"][row] + off[j], rtol=0, atol=2e-6)   # fp32 output rounding
            np.testing.assert_allclose(T[j, 3:6], g["target_vel"][row], rtol=0, atol=1e-6)
            np.testing.assert_allclose(T[j, 6:9], g["target_acc"][row], rtol=0, atol=1e-6)
            assert abs(T[j, 9] - g["target_yaw"][row]) < 2e-5, (j, k)
    st = fleet.FleetState(ctx, n)
    rigid, mem, _ = random_fleet(np.random.default_rng(81), n)
    st.load_aos(rigid, mem)
    a = _args(nat, 2, DT, float(np.float32(2 / 240)))
    nat.check(ctx.lib.dsim_step(ctx.handle, _stream(ctx), n, st.view(), tr.view(), ctypes.byref(a)))
    tgt = tr.fields(0, 10).T.double().cpu().numpy()
    r0, m0 = rigid.copy(), mem.copy()
    assert orc.Oracle([t]).step(rigid, mem, tgt, 2, DT, float(np.float32(2 / 240))) == 0
    assert_step_parity("traj_sampler_step", [t], None, r0, m0, tgt, st.rigid_aos(), st.mem_aos(), rigid, mem, DT,
                       float(np.float32(2 / 240)), 2)
    ctx.close()


@pytest.mark.parametrize("noise_seed", [0, 77])
def test_multi_step_launch_equals_single_steps(gpu, golden_dir, noise_seed):
    """n_steps = K in one launch is bit-identical to K single-step launches (state, counters, noise stream)."""
    n = 1000
    envA, wpA, *_ = _traj_fleet(gpu, golden_dir, n, noise_seed)
    envB, wpB, *_ = _traj_fleet(gpu, golden_dir, n, noise_seed)
    a0 = np.full((n, 4), 0.4, dtype=np.float32)
    envA.step_fused(wpA, action=a0); envB.step_fused(wpB, action=a0)
    for _ in range(12):
        envA.step_fused(wpA)
    envB.step_fused(wpB, n_steps=5); envB.step_fused(wpB, n_steps=7)
    np.testing.assert_array_equal(envA.state.fields(0, 24).cpu().numpy(), envB.state.fields(0, 24).cpu().numpy())
    np.testing.assert_array_equal(wpA.counters.cpu().numpy(), wpB.counters.cpu().numpy())
    assert envA._env_steps == envB._env_steps == 13
    envA.close(); envB.close()


# ---------------------------------------------------------------------------
# size-independent properties at BASELINE sizes
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("n", [65536, 1 << 20])
def test_full_size_properties(gpu, n):
    nat, fleet = gpu
    t, ctx, st, tg, rigid, mem, tgt = _make(gpu, "robobee", n, seed=21)
    st2 = fleet.FleetState(ctx, n, "tile64")
    tg2 = fleet.Targets(ctx, n, "tile64")
    st2.load_aos(rigid, mem)
    tg2.set_fields(0, torch.from_numpy(np.ascontiguousarray(tgt.T)))
    # permuted copy of the same fleet
    perm = np.random.default_rng(22).permutation(n)
    st3 = fleet.FleetState(ctx, n)
    tg3 = fleet.Targets(ctx, n)
    st3.load_aos(rigid[perm], mem[perm])
    tg3.set_fields(0, torch.from_numpy(np.ascontiguousarray(tgt[perm].T)))
    idx = np.random.default_rng(23).choice(n, 2048, replace=False)
    O = orc.Oracle([t])
    for k in range(4):
        a = _args(nat, 5, DT, float(np.float32(5 / 240)))
        prev = st.fields(0, 24)[:, torch.from_numpy(idx).to(ctx.device)].T.double().cpu().numpy()
        for s_, t_ in ((st, tg), (st2, tg2), (st3, tg3)):
            nat.check(ctx.lib.dsim_step(ctx.handle, _stream(ctx), n, s_.view(), t_.view(), ctypes.byref(a)))
        # 2048 drones of the big fleet, every step at the per-step bar from the device's previous state
        got = st.fields(0, 24)[:, torch.from_numpy(idx).to(ctx.device)].T.double().cpu().numpy()
        r0, m0 = prev[:, :13].copy(), np.concatenate([prev[:, 13:24], np.zeros((len(idx), 2))], 1)
        r1, m1 = r0.copy(), m0.copy()
        assert O.step(r1, m1, tgt[idx], 5, DT, float(np.float32(5 / 240))) == 0
        assert_step_parity(f"full_size[{n}]", [t], None, r0, m0, tgt[idx], got[:, :13],
                           np.concatenate([got[:, 13:24], np.zeros((len(idx), 2))], 1), r1, m1, DT,
                           float(np.float32(5 / 240)), 5)
    A = st.fields(0, 24).cpu().numpy()
    B = st2.fields(0, 24).cpu().numpy()
    C = st3.fields(0, 24).cpu().numpy()
    assert np.isfinite(A).all()
    np.testing.assert_array_equal(A, B)                 # layout does not change a single bit
    np.testing.assert_array_equal(A[:, perm], C)        # drones are independent: permutation-equivariant
    qn = np.linalg.norm(A[3:7], axis=0)
    assert np.abs(qn - 1).max() < 1e-6                  # quaternion stays unit
    assert (A[20:24] >= 0).all() and (A[20:24] <= 1).all()   # PWM clip
    # accumulated drift of the same sample from the oracle's free-running 4 steps
    r, m, tg_ = rigid[idx].copy(), mem[idx].copy(), tgt[idx].copy()
    for k in range(4):
        assert O.step(r, m, tg_, 5, DT, float(np.float32(5 / 240))) == 0
    assert rel_err(A[:13, idx].T.astype(np.float64), r, RIGID_SCALE).max() < 4 * REL_TOL
    ctx.close()


def test_bench_size_fleet_properties(gpu):
    """BASELINE bench size (4 194 304 drones, 0.97 GB of state): checked on the device — both layouts
    agree bit for bit after 3 noisy steps, quaternions stay unit, PWM stays clipped, everything is
    finite — plus an oracle check of a 1 024-drone sample (noise replayed through the oracle's
    restatement of the same counter-based generator)."""
    nat, fleet = gpu
    n = 4194304
    t = params.builtin_type("robobee")
    ctx = fleet.Context([t])
    rigid, mem, tgt = random_fleet(np.random.default_rng(101), n)
    states, targets = [], []
    for layout in ("soa", "tile64"):
        st = fleet.FleetState(ctx, n, layout); tg = fleet.Targets(ctx, n, layout)
        st.load_aos(rigid, mem); tg.set_fields(0, torch.from_numpy(np.ascontiguousarray(tgt.T)))
        states.append(st); targets.append(tg)
    seed = 4242
    idx = np.sort(np.random.default_rng(102).choice(n, 1024, replace=False))
    idx_dev = torch.from_numpy(idx).to(ctx.device)
    O = orc.Oracle([t])
    for k in range(3):
        a = _args(nat, 1, DT, DT, seed=seed, step_index=k)
        prev = states[0].fields(0, 24)[:, idx_dev].T.double().cpu().numpy()
        for st, tg in zip(states, targets):
            nat.check(ctx.lib.dsim_step(ctx.handle, _stream(ctx), n, st.view(), tg.view(), ctypes.byref(a)))
        # a 1 024-drone sample at the per-step bar, noise replayed through the oracle's restatement of the generator
        got = states[0].fields(0, 24)[:, idx_dev].T.double().cpu().numpy()
        nz = np.zeros((len(idx), 1, 12))
        for q, i in enumerate(idx):
            u = O.noise_normals(seed, int(i), k, 4, fine=True)       # (one sub-step per launch: the fine lattice)
            nz[q, 0, 0:4], nz[q, 0, 6:10] = u[0:4] * 0.01, u[4:8] * 0.001
        pad2 = np.zeros((len(idx), 2))
        r0, m0 = prev[:, :13].copy(), np.concatenate([prev[:, 13:24], pad2], 1)
        r1, m1 = r0.copy(), m0.copy()
        assert O.step(r1, m1, tgt[idx], 1, DT, DT, noise=nz) == 0
        assert_step_parity("bench_size_4194304", [t], None, r0, m0, tgt[idx], got[:, :13],
                           np.concatenate([got[:, 13:24], pad2], 1), r1, m1, DT, DT, 1)
    A, B = states[0].fields(0, 24), states[1].fields(0, 24)
    assert bool(torch.equal(A, B))
    assert bool(torch.isfinite(A).all())
    assert float((A[3:7].square().sum(0).sqrt() - 1).abs().max()) < 1e-6
    assert float(A[20:24].min()) >= 0.0 and float(A[20:24].max()) <= 1.0
    ctx.close()


def test_long_hover_soak_at_bench_size(gpu):
    """4 194 304 drones hovering on their targets under rotor noise for 6 000 Env.steps (25 s of flight, 2.5e10
    drone-steps): the fleet stays finite, unit-quaternion, PWM-clipped and within millimetres of the targets, and
    its spread under the noise is statistically stationary (no slow drift or growth)."""
    from dronesim_amd.envs import CtrlAviary
    from dronesim_amd.fleet import Targets
    n = 4194304
    side = 2048
    ij = np.arange(n)
    xyz = np.stack([(ij % side) * 1.0, (ij // side) * 1.0, np.full(n, 0.5)], 1)
    env = CtrlAviary(["robobee"], n, initial_xyzs=xyz, aggregate_phy_steps=1, noise_seed=7, dict_io=False)
    assert env.state.layout == "tile64"
    tg = Targets(env.ctx, n, env.state.layout)
    tgt = torch.from_numpy(np.ascontiguousarray(xyz.T.astype(np.float32))).to(env.ctx.device)
    tg.set(pos=tgt, yaw=0.4)
    hover = params.builtin_type("robobee").hover_pwm
    env.step_fused(tg, action=np.full((n, 4), hover, dtype=np.float32))
    spreads = []
    for block in range(6):
        for _ in range(1000):
            env.step_fused(tg)
        A = env.state.fields(0, 24)
        assert bool(torch.isfinite(A).all())
        assert float((A[3:7].square().sum(0).sqrt() - 1).abs().max()) < 1e-6
        assert float(A[20:24].min()) >= 0.0 and float(A[20:24].max()) <= 1.0
        dev = (A[0:3] - tgt)
        spreads.append(float(dev.square().mean().sqrt()))
        assert float(dev.abs().max()) < 0.05, (block, float(dev.abs().max()))
    assert max(spreads[2:]) < 1.5 * min(spreads[2:]) + 1e-6, spreads      # stationary once the start-up transient is gone
    env.close()


def test_fleet_beyond_4GiB_of_state(gpu):
    """Maximum sizes: 50 331 648 drones = 4.8 GB of state in one block (offsets past 2^32 bytes), both layouts.
    Every drone starts identical (noise off, one broadcast target), so after three steps every drone of the
    fleet — first tile to last — must hold bit-identical values, equal to those of a 256-drone fleet."""
    nat, fleet = gpu
    n = 50331648
    ctx = fleet.Context([params.builtin_type("robobee")])

    def fly(n_, layout):
        st = fleet.FleetState(ctx, n_, layout)
        d = st.data if layout == "soa" else st.data.permute(1, 0, 2)          # [F, ...] view, filled on the device
        d[2] = 0.5; d[6] = 1.0; d[7] = 0.3; d[11] = 0.2; d[20:24] = 0.45
        tg = fleet.Targets(ctx, n_, layout, broadcast=True)
        tg.set(pos=np.array([0.4, -0.2, 0.9], dtype=np.float32), yaw=0.3)
        a = _args(nat, 2, DT, float(np.float32(2 / 240)), options=nat.OPT_BCAST_TGT)
        for k in range(3):
            nat.check(ctx.lib.dsim_step(ctx.handle, _stream(ctx), n_, st.view(), tg.view(), ctypes.byref(a)))
        torch.cuda.synchronize()
        return st

    ref = fly(256, "soa").data[:, 0].clone()                                  # [24] one drone's fields
    assert abs(float(ref[2]) - 0.5) > 1e-6                                    # it moved
    for layout in ("soa", "tile64"):
        st = fly(n, layout)
        assert st.data.numel() * 4 > (1 << 32)
        d = st.data if layout == "soa" else st.data.permute(1, 0, 2).reshape(24, -1)
        for f in range(24):
            lo, hi = float(d[f].min()), float(d[f].max())
            assert lo == hi == float(ref[f]), (layout, f, lo, hi, float(ref[f]))
        del st, d
        torch.cuda.empty_cache()
    ctx.close()


def test_graph_replay_equals_eager_steps(gpu, golden_dir):
    """hipGraph of 8 fused steps, replayed 3 times == 24 eager steps, bit for bit, rotor noise on (the
    env-step counter that seeds it is read from device memory inside the captured kernels)."""
    n = 4096
    envA, wpA, *_ = _traj_fleet(gpu, golden_dir, n, noise_seed=1234)
    envB, wpB, *_ = _traj_fleet(gpu, golden_dir, n, noise_seed=1234)
    a0 = np.full((n, 4), 0.4, dtype=np.float32)
    envA.step_fused(wpA, action=a0); envB.step_fused(wpB, action=a0)
    for _ in range(24):
        envA.step_fused(wpA)
    g = envB.capture_fused(wpB, steps=8)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    np.testing.assert_array_equal(envA.state.fields(0, 24).cpu().numpy(), envB.state.fields(0, 24).cpu().numpy())
    np.testing.assert_array_equal(wpA.counters.cpu().numpy(), wpB.counters.cpu().numpy())
    assert envA._env_steps == envB._env_steps == 25
    envB.step_fused(wpB); envA.step_fused(wpA)               # eager stepping continues seamlessly after a replay
    np.testing.assert_array_equal(envA.state.fields(0, 24).cpu().numpy(), envB.state.fields(0, 24).cpu().numpy())
    envA.close(); envB.close()


def test_graph_replay_hexa_fleet(gpu):
    """hipGraph capture also covers fleets with the morphing hexa (the deferred-WLS-fallback queue is reserved
    beforehand through dsim_reserve): replay == eager stepping, rotor noise on, mixed interleaved fleet."""
    from dronesim_amd.envs import CtrlAviary
    from dronesim_amd.fleet import Targets
    n = 3000
    rng = np.random.default_rng(123)
    xyz = np.stack([rng.uniform(0, 50, n), rng.uniform(0, 50, n), rng.uniform(1, 5, n)], 1)
    tid = (np.arange(n) % 2).astype(np.uint8)
    envs, tgts = [], []
    for _ in range(2):
        e = CtrlAviary(["robobee", "hexa_6DOF"], n, initial_xyzs=xyz, aggregate_phy_steps=1, noise_seed=99, dict_io=False,
                       type_ids=tid, layout="tile64")          # wave-tiled: the persistent LDS-DMA ring kernel under capture
        t = Targets(e.ctx, n); t.set(pos=f32(xyz + 0.2).T, yaw=0.2)
        envs.append(e); tgts.append(t)
    for _ in range(12):
        envs[0].step_fused(tgts[0])
    g = envs[1].capture_fused(tgts[1], steps=4)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    np.testing.assert_array_equal(envs[0].state.fields(0, 26).cpu().numpy(), envs[1].state.fields(0, 26).cpu().numpy())
    assert envs[0]._env_steps == envs[1]._env_steps == 12
    for e in envs:
        e.close()


@pytest.mark.parametrize("sub", [1, 2])        # 1: the straight-line single-sub-step kernels, 2: the looped ones
def test_chained_stepping(gpu, sub):
    """DSIM_OPT_CHAINED: last_vel / last_rates recomputed from the stored rigid state instead of being read,
    and not written; materialize() restores them.  Same trajectory as the plain mode and as the oracle."""
    from dronesim_amd.envs import CtrlAviary
    from dronesim_amd.fleet import Targets
    n = 2048
    rng = np.random.default_rng(95)
    xyz = np.stack([rng.uniform(-20, 20, n), rng.uniform(-20, 20, n), rng.uniform(1, 5, n)], 1)
    rpy = np.stack([rng.uniform(-0.3, 0.3, n), rng.uniform(-0.3, 0.3, n), rng.uniform(-3, 3, n)], 1)
    envs = [CtrlAviary(["robobee"], n, initial_xyzs=xyz, initial_rpys=rpy, aggregate_phy_steps=sub, noise_seed=5,
                       dict_io=False, chained=c) for c in (False, True)]
    tgts = []
    for e in envs:
        tg = Targets(e.ctx, n); tg.set(pos=f32(xyz + 0.3).T, yaw=0.5); tgts.append(tg)
    a0 = np.full((n, 4), 0.4, dtype=np.float32)
    for e, tg in zip(envs, tgts):
        e.step_fused(tg, action=a0)              # first step: explicit action -> not chained
        for _ in range(15):
            e.step_fused(tg)
    assert envs[1]._chain_live and not envs[0]._chain_live
    A = envs[0].state.fields(0, 24).cpu().numpy()
    raw = envs[1].state.data[:, :n].cpu().numpy().copy()    # (the block as it lies in HBM — plain SoA here —, not through an accessor)
    live = [f for f in range(24) if not 13 <= f < 19]
    assert np.abs(A[live] - raw[live]).max() < 2e-5         # same trajectory (recomputed R^T w may differ by an ulp)
    assert np.abs(A[13:19] - raw[13:19]).max() > 1e-3       # the six fields ARE stale in chained mode...
    B2 = envs[1].state.fields(0, 24).cpu().numpy()          # ...until materialized: the host accessors do that themselves
    assert not envs[1]._chain_live
    assert np.abs(A[13:19] - B2[13:19]).max() < 2e-5
    # and a non-chained operation after a chained run sees consistent memory: one more plain step agrees
    for e, tg in zip(envs, tgts):
        e._chained_enabled = False
        e.step_fused(tg)
    assert np.abs(envs[0].state.fields(0, 24).cpu().numpy() - envs[1].state.fields(0, 24).cpu().numpy()).max() < 5e-5
    for e in envs:
        e.close()


def test_hover_equilibrium_and_determinism(gpu):
    nat, fleet = gpu
    t = params.builtin_type("robobee")
    ctx = fleet.Context([t])
    n = 4096
    st = fleet.FleetState(ctx, n)
    rigid = np.zeros((n, 13)); rigid[:, 2] = 0.5; rigid[:, 6] = 1.0
    mem = np.zeros((n, 13)); mem[:, 7:11] = t.hover_pwm
    st.load_aos(rigid, mem)
    a = _args(nat, 240, DT, DT)
    nat.check(ctx.lib.dsim_physics(ctx.handle, _stream(ctx), n, st.view(), None, ctypes.byref(a)))
    out = st.rigid_aos()
    assert np.abs(out[:, 0:3] - rigid[:, 0:3]).max() < 1e-4     # thrust == weight: stays put for 1 s
    assert np.abs(out[:, 7:13]).max() < 1e-4
    assert (out == out[0]).all()                               # identical drones -> identical bits
    ctx.close()


# ---------------------------------------------------------------------------
# the reference-shaped Python surfaces
# ---------------------------------------------------------------------------
def _obs_rpy_tol(quat):
    """Per-row tolerance [n,3] of the rpy columns of an observation row against the oracle's fp64 Euler angles of
    the SAME fp32 quaternion (BaseAviary.py:729 p.getEulerFromQuaternion).  The angles are atan2 / asin of fp32
    products of quaternion components: 4 fp32 ulps of the products (|q|^2 scale) through the inverse functions'
    slopes — roll: 1 / hypot(ra, rb), pitch: 1 / sqrt(1 - sarg^2), yaw: 1 / hypot(ya, yb); the gimbal branch is
    2 atan2 of two components (slope 1 / hypot(x, y)) — plus 2 ulps of the angle itself."""
    x, y, z, w = (quat[:, k] for k in range(4))
    n2 = x * x + y * y + z * z + w * w
    sarg = -2.0 * (x * z - w * y)
    ra, rb = 2 * (y * z + w * x), w * w - x * x - y * y + z * z
    ya, yb = 2 * (x * y + w * z), w * w + x * x - y * y - z * z
    e = 4 * ulp32(n2)
    gimbal = np.abs(sarg) >= 0.99999
    tol = np.stack([e / np.maximum(np.hypot(ra, rb), 1e-30), e / np.sqrt(np.maximum(1 - sarg ** 2, 1e-30)),
                    e / np.maximum(np.hypot(ya, yb), 1e-30)], 1)
    tol[gimbal, 0] = 0.0
    tol[gimbal, 1] = 0.0
    tol[gimbal, 2] = 2 * 4 * ulp32(np.hypot(x, y)[gimbal]) / np.maximum(np.hypot(x, y)[gimbal], 1e-30)
    return tol + 2 * ulp32(math.pi)


def _check_obs_rows(label, O, rows, rigid32, last6, type_id, types):
    """rows [n, W] from the device == orc_state_vector (oracle/dsim_oracle.c) of the same fp32 state: ALL columns —
    pos, quat, vel, ang_v and the echoed action are copies (bit-exact), rpy within _obs_rpy_tol per row."""
    want = O.state_vector(rigid32, last6, type_id)
    W = want.shape[1]
    tid = np.zeros(len(rows), dtype=np.int64) if type_id is None else np.asarray(type_id).astype(np.int64)
    na = np.array([t.n_act for t in types])[tid]
    copies = [0, 1, 2, 3, 4, 5, 6, 10, 11, 12, 13, 14, 15]
    np.testing.assert_array_equal(rows[:, copies], want[:, copies])
    for j in range(W - 16):
        live = na > j
        np.testing.assert_array_equal(rows[live, 16 + j], want[live, 16 + j])
    tol = _obs_rpy_tol(rigid32[:, 3:7])
    d = np.abs(rows[:, 7:10] - want[:, 7:10])
    d[:, 2] = np.minimum(d[:, 2], np.abs(d[:, 2] - 2 * math.pi))     # yaw within rounding of +-pi
    ratio = d / tol
    from tests.util import WORST
    WORST[label] = max(WORST.get(label, 0.0), float(ratio.max()))
    assert (d <= tol).all(), (label, float(ratio.max()), np.unravel_index(ratio.argmax(), ratio.shape))
    return want


@pytest.mark.parametrize("fleet_kind", ["quad", "hexa", "mixed"])
@pytest.mark.parametrize("layout", ["soa", "tile64"])
def test_observation_rows_vs_oracle_state_vector(gpu, fleet_kind, layout):
    """P5 / f1: dsim_observe (row-major) and dsim_observe_soa (field-major log slab) against orc_state_vector
    (BaseAviary.py:764-790), every one of the 20 / 22 columns, on attitudes that reach every branch of Bullet's
    getEulerFromQuaternion: the whole sphere, both signs of w, both gimbal branches (|sarg| >= 0.99999), the
    ill-conditioned band just outside the clamp, non-unit quaternions; with and without a separate last_action."""
    nat, fleet = gpu
    names = {"quad": ["robobee"], "hexa": ["hexa_6DOF"], "mixed": ["tello", "hexa_6DOF"]}[fleet_kind]
    types = [params.builtin_type(m) for m in names]
    ctx = fleet.Context(types)
    O = orc.Oracle(types)
    rng = np.random.default_rng(7)
    quat = attitude_zoo(rng)
    n = quat.shape[0]
    W = 16 + ctx.n_act
    rigid, mem, _ = random_fleet(rng, n, n_act=6)
    rigid[:, 3:7] = quat
    sarg = -2.0 * (quat[:, 0] * quat[:, 2] - quat[:, 3] * quat[:, 1])
    assert (sarg >= 0.99999).sum() > 50 and (sarg <= -0.99999).sum() > 50 and (quat[:, 3] < 0).sum() > 500
    assert (np.abs(np.linalg.norm(quat, axis=1) - 1) > 0.05).sum() > 200
    tid = (np.arange(n) % len(types)).astype(np.uint8) if len(types) > 1 else None
    if tid is not None:
        mem[tid == 0, 11:13] = 0.0
    if ctx.n_act == 4:
        mem[:, 11:13] = 0.0
    st = fleet.FleetState(ctx, n, layout)
    st.load_aos(rigid, mem)
    last = f32(rng.uniform(0, 1, (n, 6)))
    if ctx.n_act == 4:
        last[:, 4:6] = 0.0
    last_dev = torch.zeros((ctx.n_act, st.n_pad), device=ctx.device)
    last_dev[:, :n] = torch.from_numpy(np.ascontiguousarray(last[:, : ctx.n_act].T)).float()
    for la, la_host in ((last_dev, last), (None, mem[:, 7:13])):          # env's last_clipped_action / the stored cmd
        rows = torch.full((n, W), -7.0, device=ctx.device)
        nat.check(ctx.lib.dsim_observe(ctx.handle, _stream(ctx), n, st.view(), la.data_ptr() if la is not None else None,
                                       rows.data_ptr(), W))
        slab = torch.full((W, st.n_pad), -7.0, device=ctx.device)
        nat.check(ctx.lib.dsim_observe_soa(ctx.handle, _stream(ctx), n, st.view(), la.data_ptr() if la is not None else None,
                                           slab.data_ptr(), W))
        torch.cuda.synchronize()
        R, S = rows.double().cpu().numpy(), slab[:, :n].T.double().cpu().numpy()
        np.testing.assert_array_equal(R, S)                              # the two entry points agree bit for bit
        _check_obs_rows(f"observe[{fleet_kind},{layout}]", O, R, rigid, la_host, tid, types)
    ctx.close()


def test_env_and_controller_surfaces(gpu):
    """Example-style loop (examples/fly_INDI.py:217-239) through CtrlAviary.step + INDIControl.
    computeControlFromState, against the oracle doing the same calls: every step from the device's own previous
    state at the per-step bar, the observation rows (all 20 columns incl. rpy) against orc_state_vector, and the
    accumulated drift of the closed loop bounded separately."""
    from dronesim_amd.control import INDIControl
    from dronesim_amd.envs import CtrlAviary
    n = 3
    xyz = np.array([[0.0, 1.0, 0.5], [1.0, 0.0, 0.7], [-1.0, 0.5, 1.0]])
    env = CtrlAviary(["robobee"] * n, n, initial_xyzs=xyz, initial_rpys=np.zeros((n, 3)),
                     aggregate_phy_steps=5, noise_seed=0)
    ctrl = INDIControl("robobee", num_drones=n)
    obs = env.reset()
    assert set(obs.keys()) == {"0", "1", "2"} and obs["0"]["state"].shape == (20,)
    assert obs["1"]["neighbors"].shape == (n,)
    np.testing.assert_allclose(obs["2"]["state"][0:3], xyz[2], atol=1e-6)
    np.testing.assert_array_equal(obs["0"]["state"][16:20], 0.0)        # last_clipped_action starts at 0
    t = params.builtin_type("robobee")
    O = orc.Oracle([t])
    rigid = np.concatenate([xyz, np.tile([0, 0, 0, 1.0], (n, 1)), np.zeros((n, 6))], 1)   # the oracle's free-running loop
    mem = O.reset_mem(n)
    last = np.zeros((n, 6))
    action = {str(i): np.array([0.4, 0.4, 0.4, 0.4]) for i in range(n)}
    dtc = float(np.float32(5 / 240))
    tgt = f32(np.tile([0, 0, 0.5, 0, 0, 0, 0, 0, 0, 0.4], (n, 1)))
    for k in range(12):
        r0 = env.state.rigid_aos()
        act_host = f32(np.stack([action[str(i)] for i in range(n)]))     # what the device receives
        obs, reward, done, info = env.step(action)
        assert reward == -1 and done is False and info == {"answer": 42}
        states = np.stack([obs[str(i)]["state"] for i in range(n)])
        # Env.step from the device's previous state, at the bar
        r1, last1 = r0.copy(), last.copy()
        a6 = np.zeros((n, 6)); a6[:, :4] = act_host
        O.physics(r1, mem.copy(), 5, DT, action=a6, last_action=last1)
        assert_step_parity("env_step_loop", [t], None, r0, mem, tgt, env.state.rigid_aos(), None, r1, None, DT, dtc, 5,
                           control=False, action=act_host)
        # the observation == orc_state_vector of the device's new state, all 20 columns
        _check_obs_rows("env_step_loop obs", O, states, env.state.rigid_aos(), last1, None, [t])
        # computeControlFromState on that observation, from the controller's own previous memory, at the bar
        m0 = ctrl.state.mem_aos()
        cmd, pos_e, yaw_e = ctrl.computeControlFromState(dtc, states, target_pos=np.array([0, 0, 0.5]),
                                                         target_rpy=np.array([0, 0, 0.4]))
        action = {str(i): cmd[i].cpu().numpy() for i in range(n)}
        s32 = np.concatenate([states[:, 0:7], states[:, 10:16]], 1)
        m1 = m0.copy()
        rc, pe, ye = O.control(s32, m1, tgt, dtc)
        assert rc == 0
        assert_control_parity("env_step_loop control", [t], None, s32, m0, tgt, ctrl.state.mem_aos(), m1, dtc)
        np.testing.assert_array_equal(cmd.cpu().numpy(), ctrl.state.mem_aos()[:, 7:11].astype(np.float32))
        assert (np.abs(pos_e.double().cpu().numpy() - pe) <= ulp32(np.maximum(np.abs(tgt[:, 0:3]), np.abs(s32[:, 0:3])))).all()
        # the oracle's own closed loop (drift reference)
        a6f = np.zeros((n, 6)); a6f[:, :4] = 0.4 if k == 0 else mem[:, 7:11]
        O.physics(rigid, mem, 5, DT, action=a6f, last_action=last)
        O.control(f32(rigid), mem, tgt, dtc)
    assert rel_err(states[:, [0, 1, 2, 3, 4, 5, 6, 10, 11, 12, 13, 14, 15]], rigid, RIGID_SCALE).max() < 1e-3
    # single-drone call returns the reference's shapes
    ctrl1 = INDIControl("robobee")       # one controller per drone, as in the reference (fly_INDI.py:210)
    c1, pe1, ye1 = ctrl1.computeControlFromState(dtc, states[0], target_pos=np.array([0, 0, 0.5]))
    assert c1.shape == (4,) and pe1.shape == (3,) and isinstance(ye1, float)
    env.close()


@pytest.mark.parametrize("form", ["host action", "device rows", "device rows, streaming", "soa layout, noise"])
@pytest.mark.parametrize("mode", ["velocity", "rpyt"])
def test_action_adaptor_envs_vs_oracle(gpu, mode, form):
    """VelocityAviary / RPYTAviary: control inside step() on the current state, then the physics — ONE launch
    (k_adaptor_fast) for a homogeneous fleet: the action field-major (a host array goes through the env's buffer) or as
    the [N, 4] device tensor the caller holds (DSIM_OPT_ACTION_ROWS), Env.step's observation rows written by the same
    launch; default and streaming cache policy, both layouts, noise on and off."""
    from dronesim_amd.envs import RPYTAviary, VelocityAviary
    nat = gpu[0]
    n = 700
    rng = np.random.default_rng(71)
    xyz = np.stack([rng.uniform(-5, 5, n), rng.uniform(-5, 5, n), rng.uniform(1, 5, n)], 1)
    rpy = np.stack([rng.uniform(-0.2, 0.2, n), rng.uniform(-0.2, 0.2, n), rng.uniform(-3, 3, n)], 1)
    cls = VelocityAviary if mode == "velocity" else RPYTAviary
    seed = 9 if "noise" in form else 0
    env = cls(["robobee"], n, initial_xyzs=xyz, initial_rpys=rpy, aggregate_phy_steps=5, noise_seed=seed, dict_io=False,
              layout="soa" if "soa" in form else "tile64", options=nat.OPT_STREAM_ON if "streaming" in form else nat.OPT_STREAM_OFF)
    t = params.builtin_type("robobee")
    O = orc.Oracle([t])
    dtc = float(np.float32(5 * (1.0 / 240)))
    for k in range(6):
        if mode == "velocity":
            act = np.concatenate([rng.uniform(-1, 1, (n, 3)), rng.uniform(0, 0.3, (n, 1))], 1)
            act[0, 0:3] = 0.0                                  # zero vector: unit vector := 0 (VelocityAviary.py:243-246)
        else:
            act = np.concatenate([rng.uniform(-0.5, 0.5, (n, 3)), rng.uniform(0.3, 0.6, (n, 1))], 1)
        act = f32(act)
        r0, m0 = env.state.rigid_aos(), env.state.mem_aos()    # every step from the device's previous state
        a_t = torch.from_numpy(act.astype(np.float32))
        obs, reward, done, info = env.step(a_t.to(env.ctx.device) if "device" in form or k % 2 else a_t)
        got_r, got_m = env.state.rigid_aos(), env.state.mem_aos()
        # (a) the law inside _preprocessAction, on the state BEFORE the physics: the oracle's adaptor step with no
        # sub-steps is exactly that control call
        rc0, mem = r0.copy(), m0.copy()
        assert O.adaptor_step(0 if mode == "velocity" else 1, rc0, mem, act, 0, DT, dtc) == 0
        tgt = np.concatenate([r0[:, 0:3], np.zeros((n, 7))], 1)       # velocity mode tracks (own position, commanded velocity)
        if mode == "velocity":
            nrm = np.linalg.norm(act[:, 0:3], axis=1, keepdims=True)
            tgt[:, 3:6] = t.max_speed_kmh / 3.6 * np.abs(act[:, 3:4]) * np.divide(act[:, 0:3], nrm, out=np.zeros((n, 3)), where=nrm > 0)
        assert_control_parity(f"adaptor_env[{mode}] control", [t], None, r0, m0, tgt, got_m, mem, dtc)
        # (b) the physics with the command the DEVICE computed (its fp32 rounding is judged in (a), not again here)
        rigid = r0.copy()
        a6 = np.zeros((n, 6)); a6[:, :4] = got_m[:, 7:11]
        O.physics(rigid, got_m.copy(), 5, DT, action=a6, noise=_noise_block(O, [t], None, n, seed, k, 5) if seed else None)
        assert_step_parity(f"adaptor_env[{mode}] physics", [t], None, r0, got_m, tgt, got_r, None, rigid, None, DT, dtc, 5,
                           control=False, action=got_m[:, 7:11], extra_terms=noise_terms([t], None, n, DT, 5) if seed else None)
        # Env.step's return value: the rows of the NEW state with the applied command echoed, every column
        _check_obs_rows(f"adaptor_env[{mode}] rows", O, obs.double().cpu().numpy(), f32(got_r), a6, None, [t])
    env.close()


@pytest.mark.parametrize("model", ["robobee", "tello"])
@pytest.mark.parametrize("mode", ["vel", "rpyt"])
def test_action_adaptors_vs_reference_golden(gpu, golden_dir, model, mode):
    """dsim_step_adaptor against what the reference's own VelocityAviary / RPYTAviary._preprocessAction
    returned on the same states (tests/golden/env_side.npz): command and controller memory."""
    nat, fleet = gpu
    G = np.load(os.path.join(golden_dir, "env_side.npz"))
    g = lambda k: G[f"{model}_ad_{k}"]
    t = params.builtin_type(model)
    ctx = fleet.Context([t])
    sv = g("state")
    n = sv.shape[0]
    rigid = np.concatenate([sv[:, 0:7], sv[:, 10:16]], 1)
    mem = np.zeros((n, 13))
    mem[:, 0:3], mem[:, 3:6], mem[:, 6], mem[:, 7:11] = g("last_vel"), g("last_rates"), g("last_thrust"), g("cmd")
    st = fleet.FleetState(ctx, n)
    r32, m32, a32 = f32(rigid), f32(mem), f32(g(f"{mode}_action"))
    st.load_aos(r32, m32)
    act = torch.zeros((4, st.n_pad), device=ctx.device)
    act[:, :n] = torch.from_numpy(np.ascontiguousarray(a32.T)).float()
    last = torch.zeros((6, st.n_pad), device=ctx.device)
    dtc = float(np.float32(5 / 240))
    a = _args(nat, 5, DT, dtc)
    nat.check(ctx.lib.dsim_step_adaptor(ctx.handle, _stream(ctx), n, st.view(), act.data_ptr(),
                                        nat.ADAPT_VELOCITY if mode == "vel" else nat.ADAPT_RPYT, last.data_ptr(),
                                        ctypes.byref(a)))
    torch.cuda.synchronize()
    got = st.mem_aos()
    # (1) oracle on the same fp32 inputs; (3) the reference's numbers + the measured input-rounding effect, per case.
    # The law runs on the state BEFORE the physics (which does not touch the controller memory), so the memory
    # increments are those of one computeControl call.
    o32, rr = m32.copy(), r32.copy()
    assert orc.Oracle([t]).adaptor_step(0 if mode == "vel" else 1, rr, o32, a32, 5, DT, dtc) == 0
    tgt = np.concatenate([r32[:, 0:3], np.zeros((n, 7))], 1)
    if mode == "vel":
        nrm = np.linalg.norm(a32[:, 0:3], axis=1, keepdims=True)
        tgt[:, 3:6] = t.max_speed_kmh / 3.6 * np.abs(a32[:, 3:4]) * np.divide(a32[:, 0:3], nrm, out=np.zeros((n, 3)), where=nrm > 0)
    assert_control_parity(f"adaptor_golden[{model},{mode}] vs oracle(fp32 in)", [t], None, r32, m32, tgt, got, o32, dtc)
    gold = o32.copy()
    gold[:, 7:11], gold[:, 3:6], gold[:, 6] = g(f"{mode}_cmd_out"), g(f"{mode}_last_rates_out"), g(f"{mode}_last_thrust_out")
    gold[:, 0:3] = g(f"{mode}_last_vel_out")
    assert_control_parity(f"adaptor_golden[{model},{mode}] vs reference", [t], None, r32, m32, tgt, got, gold, dtc,
                          slack=np.abs(o32 - gold))
    np.testing.assert_allclose(last[:4, :n].T.cpu().numpy(), got[:, 7:11], rtol=0, atol=0)   # echoed into the env's action buffer
    ctx.close()


def test_device_logger_matches_reference_layout(gpu, tmp_path):
    """Logger: states[N,20,T] / controls[N,12,T] / timestamps[N,T] and the np.savez keys of the
    reference (Logger.py:53-86, 134-139, 152-157); every logged row equals orc_state_vector (the oracle's
    _getDroneStateVector) of the state the device held at that step, all 20 columns incl. rpy — and the flight being
    logged tumbles (asymmetric PWM), so roll, pitch and yaw all leave zero."""
    from dronesim_amd.envs import CtrlAviary
    from dronesim_amd.utils import Logger
    n = 5
    xyz = np.array([[i, 0.0, 1.0 + 0.1 * i] for i in range(n)])
    rpy0 = np.array([[0.1 * i, -0.05 * i, 0.7 * i - 1.5] for i in range(n)])
    env = CtrlAviary(["robobee"] * n, n, initial_xyzs=xyz, initial_rpys=rpy0, aggregate_phy_steps=5, noise_seed=0)
    log = Logger(logging_freq_hz=48, env=env, duration_sec=1)
    t = params.builtin_type("robobee")
    O = orc.Oracle([t])
    want_rows, obs_rows = [], []
    action = {str(i): np.array([0.5, 0.52 + 0.01 * i, 0.5, 0.47]) for i in range(n)}
    for k in range(10):
        obs, *_ = env.step(action)
        ctrl = np.zeros((12, n)); ctrl[0:3] = xyz.T; ctrl[5] = 0.4
        log.log(timestamp=k * 5 / 240, control=ctrl)
        last6 = np.zeros((n, 6)); last6[:, :4] = np.stack([np.float32(action[str(i)]) for i in range(n)])
        rows = np.stack([obs[str(i)]["state"] for i in range(n)])
        want_rows.append(_check_obs_rows("logger rows", O, rows, env.state.rigid_aos(), last6, None, [t]))
        obs_rows.append(rows)
    ts, st, ct = log.arrays()
    assert ts.shape == (n, 10) and st.shape == (n, 20, 10) and ct.shape == (n, 12, 10)
    want = np.stack(want_rows, 2)
    assert np.abs(want[:, 7:10, -1]).min() > 1e-3                      # every Euler angle is live in this flight
    # the log == the oracle's rows: copies exact (also == what env.step returned), rpy within the per-row bound
    cp = [0, 1, 2, 3, 4, 5, 6, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19]
    np.testing.assert_array_equal(st[:, cp, :], np.stack(obs_rows, 2)[:, cp, :])
    np.testing.assert_array_equal(st[:, cp, :], want[:, cp, :])
    for k in range(10):
        tol = _obs_rpy_tol(want[:, 3:7, k])
        assert (np.abs(st[:, 7:10, k] - want[:, 7:10, k]) <= tol).all(), k
    np.testing.assert_allclose(ts[3], np.arange(10) * 5 / 240)
    np.testing.assert_allclose(ct[2, 0:3, 4], xyz[2], atol=1e-6)
    path = log.save(str(tmp_path) + "/", "flight", drones=slice(1, 3))
    z = np.load(path)
    assert set(z.files) == {"timestamps", "states", "controls"} and z["states"].shape == (2, 20, 10)
    env.close()


def test_fleet_examples_fly(gpu):
    """examples/*_fleet.py — the loops of the reference's four example scripts on fleets — run and fly sensibly:
    the trajectory trackers stay on the gates' lap (table and device sampler agree), the velocity env reaches the
    commanded velocity, the 6-DOF hexas track a lateral circle level, without a single WLS fallback."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def load(name):
        spec = importlib.util.spec_from_file_location(name, os.path.join(root, "examples", name + ".py"))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        return m

    tt = load("fly_INDI_TrajectoryTrack_fleet")
    rel_a = tt.main(["--num_drones", "600", "--duration_sec", "3", "--targets", "table"])
    g = np.load(os.path.join(root, "tests", "golden", "traj_track_waypoints.npz"))
    # drones start at gate 0 with their targets spread over the whole lap (as in the reference), so the first
    # seconds are a saturated chase: only require that everybody stays around the lap
    lo, hi = g["target_pos"].min(0) - 4.0, g["target_pos"].max(0) + 4.0
    assert np.isfinite(rel_a).all() and ((rel_a >= lo) & (rel_a <= hi)).all()
    rel_b = tt.main(["--num_drones", "600", "--duration_sec", "3", "--targets", "sampler"])
    # device sampler, every drone at t = 0 of the same lap: they fly in formation along it
    want = g["target_pos"][min(int(3.0 * 96), len(g["t"]) - 1)]
    assert np.abs(rel_b - rel_b.mean(0)).max() < 2e-2 and np.linalg.norm(rel_b.mean(0) - want) < 0.5     # (rotor noise is on)
    vel, want = load("fly_INDI_velocity_fleet").main(["--num_drones", "500", "--duration_sec", "4"])
    assert np.abs(vel - want).max() < 0.1 * np.linalg.norm(want) + 0.02
    err_xy, err_z, tilt, fallbacks = load("fly_hexa_6DOF_fleet").main(["--num_drones", "700", "--duration_sec", "4"])
    assert err_xy.max() < 0.8 and err_z.max() < 0.05 and np.degrees(tilt.max()) < 3.0 and fallbacks == 0
    # the fourth shipped airframe through the reference-shaped two-call loop (the quad controller class on six actuators)
    err_xy, err_z, tilt = load("fly_hexa_6DOF_simple_fleet").main(["--num_drones", "700", "--duration_sec", "6"])
    assert np.isfinite(err_xy).all() and np.median(err_xy) < 0.5 and err_z.max() < 0.3 and np.degrees(tilt.max()) < 30.0


def test_c_caller_without_python_or_torch(gpu, tmp_path):
    """The boundary is a C-ABI: a plain C program (tests/c_abi_smoke.c) links the library, flies a
    fleet for 5 s and checks it reached the hover target."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "c_abi_smoke")
    lib_dir = os.path.join(root, "dronesim_amd")
    subprocess.check_call(["gcc", "-O2", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                           os.path.join(root, "tests", "c_abi_smoke.c"), "-o", exe, "-L" + lib_dir, "-ldronesim_amd",
                           "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib", "-lm"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "c_abi_smoke" in out.stdout


def test_two_quad_types_keep_their_own_gains(gpu):
    """robobee + tello in one fleet (type_id): each drone is flown with ITS type's constants and gains
    (the reference's class-attribute Gains would leak the last-constructed type's gains to all quads;
    documented deviation)."""
    nat, fleet = gpu
    n = 1500
    types = [params.builtin_type("robobee"), params.builtin_type("tello")]
    ctx = fleet.Context(types)
    assert ctx.n_fields == 24
    st = fleet.FleetState(ctx, n, "tile64")
    tg = fleet.Targets(ctx, n, "tile64")
    rigid, mem, tgt = random_fleet(np.random.default_rng(97), n)
    tid = (np.random.default_rng(98).uniform(size=n) < 0.4).astype(np.uint8)
    st.load_aos(rigid, mem)
    tg.set_fields(0, torch.from_numpy(np.ascontiguousarray(tgt.T)))
    tid_dev = torch.zeros(st.n_pad, dtype=torch.uint8, device=ctx.device); tid_dev[:n] = torch.from_numpy(tid)
    a = _args(nat, 5, DT, float(np.float32(5 / 240)), type_id=tid_dev)
    nat.check(ctx.lib.dsim_step(ctx.handle, _stream(ctx), n, st.view(), tg.view(), ctypes.byref(a)))
    r0, m0 = rigid.copy(), mem.copy()
    assert orc.Oracle(types).step(rigid, mem, tgt, 5, DT, float(np.float32(5 / 240)), type_id=tid) == 0
    assert_step_parity("two_quad_types", types, tid, r0, m0, tgt, st.rigid_aos(), st.mem_aos(), rigid, mem, DT,
                       float(np.float32(5 / 240)), 5)
    ctx.close()


@pytest.mark.parametrize("n_types", [3, 4, 8])      # 3, 4: the LDS-staged mixed kernel (4 resp. 5 waves per tile); 8: the general one
def test_randomised_airframes_vs_oracle(gpu, n_types):
    """Nothing of robobee/tello is baked into the kernels: six made-up quad types and two made-up hexa types (random
    mass, inertia, kf/km, pwm map and limits, rotor geometry, G1, gains, damping, gravity) in one interleaved fleet,
    fused step with in-kernel noise, against the oracle built from the same type table."""
    import dataclasses
    nat, fleet = gpu
    rng = np.random.default_rng(2024)
    types = []
    for k in range(8):
        hexa = k in (2, 5)
        t = dataclasses.replace(params.builtin_type("hexa_6DOF" if hexa else "robobee"), name=f"random{k}")
        na = t.n_act
        t.mass = float(rng.uniform(0.05, 2.0)); t.ctrl_mass = t.mass
        t.inertia = tuple(float(x) for x in rng.uniform(2e-5, 3e-3, 3))
        t.kf = float(rng.uniform(1e-9, 4e-8)); t.km = float(t.kf * rng.uniform(0.005, 0.1))
        sc = np.zeros(6); sc[:na] = rng.uniform(8000, 25000, na); t.pwm2rpm_scale = sc
        cn = np.zeros(6); cn[:na] = rng.uniform(0, 1500, na); t.pwm2rpm_const = cn
        lo = np.zeros(6); lo[:na] = rng.uniform(0.0, 0.15, na); t.pwm_min = lo
        hi = np.zeros(6); hi[:na] = rng.uniform(0.8, 1.0, na); t.pwm_max = hi
        rp = np.asarray(t.rotor_pos, dtype=float).copy(); rp[:na] *= rng.uniform(0.5, 2.0); rp[:na] += rng.normal(0, 0.005, (na, 3)); t.rotor_pos = rp
        G1 = np.asarray(t.G1, dtype=float).copy(); G1[:, :na] *= rng.uniform(0.6, 1.6, (G1.shape[0], 1)); t.G1 = G1
        t.kp_pos, t.kd_pos = float(rng.uniform(0.8, 2.0)), float(rng.uniform(1.5, 3.0))
        t.att_gain = tuple(float(x) for x in rng.uniform(4, 12, 3)); t.rate_gain = tuple(float(x) for x in rng.uniform(6, 20, 3))
        t.gravity = float(rng.uniform(3.0, 12.0))
        t.lin_damping, t.ang_damping = float(rng.uniform(0.0, 0.1)), float(rng.uniform(0.0, 0.1))
        t.alloc = None; t.__post_init__()                      # allocation matrices follow the new G1
        types.append(t)
    types = types[:n_types]
    n = 4000
    ctx = fleet.Context(types)
    layout = "tile64" if n_types == 4 else "soa"        # 4 types, wave-tiled: the LDS-DMA ring with 5 waves per tile
    st, tg = fleet.FleetState(ctx, n, layout), fleet.Targets(ctx, n, layout)
    rigid, mem, tgt = random_fleet(rng, n, n_act=6, tilt=0.3, rate=1.0)
    tid = rng.integers(0, n_types, n).astype(np.uint8)
    is_hexa = np.isin(tid, (2, 5))
    mem[~is_hexa, 11:13] = 0.0
    lo_all = np.stack([np.asarray(t.pwm_min)[:6] for t in types])[tid]; hi_all = np.stack([np.asarray(t.pwm_max)[:6] for t in types])[tid]
    mem[:, 7:13] = np.clip(mem[:, 7:13], lo_all, np.maximum(hi_all, lo_all))
    mem[~is_hexa, 11:13] = 0.0
    st.load_aos(rigid, mem)
    tg.set_fields(0, torch.from_numpy(np.ascontiguousarray(tgt.T)))
    tid_dev = torch.zeros(st.n_pad, dtype=torch.uint8, device=ctx.device); tid_dev[:n] = torch.from_numpy(tid)
    O = orc.Oracle(types)
    for sub in (1, 3):                  # both sub-step forms
        seed, sidx = 31 + sub, 2
        a = _args(nat, sub, DT, float(np.float32(sub / 240)), seed=seed, step_index=sidx, type_id=tid_dev)
        nat.check(ctx.lib.dsim_step(ctx.handle, _stream(ctx), n, st.view(), tg.view(), ctypes.byref(a)))
        nz = np.zeros((n, sub, 12))
        for i in range(n):
            na = 6 if is_hexa[i] else 4
            for s_ in range(sub):
                u = O.noise_normals(seed, i, sidx * sub + s_, na, fine=(sub == 1))
                nz[i, s_, 0:na], nz[i, s_, 6:6 + na] = rotor_noise(types[int(tid[i])], u)
        r0, m0 = rigid.copy(), mem.copy()
        assert O.step(rigid, mem, tgt, sub, DT, float(np.float32(sub / 240)), noise=nz, type_id=tid) == 0
        assert_step_parity(f"random_airframes[{n_types},{sub}]", types, tid, r0, m0, tgt, st.rigid_aos(), st.mem_aos(),
                           rigid, mem, DT, float(np.float32(sub / 240)), sub)
        rigid, mem = f32(rigid), f32(mem)
        st.load_aos(rigid, mem)         # continue from the oracle's state (rounded to what the device can hold)
    ctx.close()


@pytest.mark.parametrize("layout", ["soa", "tile64"])
@pytest.mark.parametrize("fleet_kind", ["quad", "hexa", "mixed"])
def test_kernels_stay_inside_their_views(gpu, layout, fleet_kind):
    """Every entry point writes only inside the views it was handed: state, targets, action/echo, observation and
    force buffers are carved out of larger allocations whose guard bands (before and after) must keep their
    sentinel pattern after reset, fused / physics / control steps (ragged fleet: fast kernel + tail kernel),
    observation and downwash."""
    nat, fleet = gpu
    names = {"quad": ["robobee"], "hexa": ["hexa_6DOF"], "mixed": ["robobee", "hexa_6DOF"]}[fleet_kind]
    types = [params.builtin_type(m) for m in names]
    ctx = fleet.Context(types)
    n, n_pad, G = 1000, 1024, 8192
    F = ctx.n_fields
    dev = ctx.device
    SENT = 12345.678

    def carve(numel):
        buf = torch.full((numel + 2 * G,), SENT, dtype=torch.float32, device=dev)
        return buf, buf[G:G + numel]

    def mk_view(t, nf):
        v = nat.View(); v.base = t.data_ptr(); v.n_pad = n_pad; v.n_fields = nf
        if layout == "soa":
            v.block, v.field_stride, v.block_stride = n_pad, n_pad, n_pad * nf
        else:
            v.block, v.field_stride, v.block_stride = 64, 64, 64 * nf
        return v

    bufs = {}
    for name, numel in (("state", F * n_pad), ("targets", 10 * n_pad), ("action", 6 * n_pad), ("echo", 6 * n_pad),
                        ("obs", n * (16 + ctx.n_act)), ("force", 3 * n_pad), ("pos_e", 3 * n_pad), ("yaw_e", n_pad),
                        ("init", 3 * n_pad)):
        bufs[name] = carve(numel)
    for k in ("targets", "action", "init"):
        bufs[k][1].zero_()
    bufs["action"][1].fill_(0.45)
    bufs["init"][1].copy_(torch.rand(3 * n_pad, device=dev) * 20 + 1)
    sv, tv = mk_view(bufs["state"][1], F), mk_view(bufs["targets"][1], 10)
    tid = None
    if fleet_kind == "mixed":
        tid = torch.zeros(n_pad, dtype=torch.uint8, device=dev); tid[:n] = torch.from_numpy((np.arange(n) % 2).astype(np.uint8))
    tp = tid.data_ptr() if tid is not None else None
    s = _stream(ctx)
    zeros = torch.zeros(3 * n_pad, device=dev)
    nat.check(ctx.lib.dsim_reset(ctx.handle, s, n, sv, bufs["init"][1].data_ptr(), zeros.data_ptr(), None, None, tp))
    a = _args(nat, 2, DT, float(np.float32(2 / 240)), seed=5, type_id=tid)
    for k in range(3):
        a.step_index = k
        nat.check(ctx.lib.dsim_step(ctx.handle, s, n, sv, tv, ctypes.byref(a)))
    a1 = _args(nat, 1, DT, DT, seed=5, type_id=tid)
    nat.check(ctx.lib.dsim_step(ctx.handle, s, n, sv, tv, ctypes.byref(a1)))                   # single-sub-step kernels
    a.action = bufs["action"][1].data_ptr()
    nat.check(ctx.lib.dsim_step(ctx.handle, s, n, sv, tv, ctypes.byref(a)))                    # explicit action: general kernel
    nat.check(ctx.lib.dsim_physics(ctx.handle, s, n, sv, bufs["echo"][1].data_ptr(), ctypes.byref(a)))
    a.action = None
    nat.check(ctx.lib.dsim_control(ctx.handle, s, n, sv, tv, ctypes.byref(a), bufs["pos_e"][1].data_ptr(),
                                   bufs["yaw_e"][1].data_ptr()))
    nat.check(ctx.lib.dsim_observe(ctx.handle, s, n, sv, None, bufs["obs"][1].data_ptr(), 16 + ctx.n_act))
    g = nat.DownwashArgs()
    nx = ny = 8
    assert ctx.lib.dsim_downwash_prebin_ok(n, nx, ny) == 1          # bucket form: the step kernel can fill it ahead
    ws = torch.empty((ctx.lib.dsim_downwash_workspace(n, nx, ny),), dtype=torch.int32, device=dev)
    g.pos_all, g.m, g.m_pad = None, n, n
    g.xmin, g.ymin, g.cell, g.nx, g.ny = 0.0, 0.0, 5.0, nx, ny
    g.workspace, g.workspace_len, g.type_id, g.local_offset = ws.data_ptr(), ws.numel(), tp, 0
    nat.check(ctx.lib.dsim_downwash(ctx.handle, s, n, sv, ctypes.byref(g), bufs["force"][1].data_ptr()))
    # the remaining entry points: field-major observation, adjacency, and (quads) adaptor / chained / trajectory sampler
    extra = {}
    for name, numel in (("obs_soa", (16 + ctx.n_act) * n_pad), ("adj_list", 4 * n_pad), ("cmd_out", ctx.n_act * n_pad),
                        ("obs2", n * (16 + ctx.n_act))):
        extra[name] = carve(numel)
    # fused step that also fills the next neighbour grid, then the downwash call that relies on it
    a2 = _args(nat, 1, DT, DT, seed=5, type_id=tid)
    a2.ext_force = bufs["force"][1].data_ptr()
    a2.bin_next = ctypes.addressof(g)
    nat.check(ctx.lib.dsim_step(ctx.handle, s, n, sv, tv, ctypes.byref(a2)))
    g.prebinned = 1
    nat.check(ctx.lib.dsim_downwash(ctx.handle, s, n, sv, ctypes.byref(g), bufs["force"][1].data_ptr()))
    g.prebinned = 0
    # Env.step with the observation rows fused, computeControl with the command handed out as a plain array
    a3 = _args(nat, 2, DT, float(np.float32(2 / 240)), seed=5, type_id=tid, action=bufs["action"][1])
    a3.obs_out, a3.obs_width = extra["obs2"][1].data_ptr(), 16 + ctx.n_act
    nat.check(ctx.lib.dsim_physics(ctx.handle, s, n, sv, bufs["echo"][1].data_ptr(), ctypes.byref(a3)))
    a3.action, a3.obs_out, a3.obs_width = None, None, 0
    nat.check(ctx.lib.dsim_control2(ctx.handle, s, n, sv, tv, ctypes.byref(a3), bufs["pos_e"][1].data_ptr(),
                                    bufs["yaw_e"][1].data_ptr(), extra["cmd_out"][1].data_ptr()))
    nat.check(ctx.lib.dsim_observe_soa(ctx.handle, s, n, sv, None, extra["obs_soa"][1].data_ptr(), 16 + ctx.n_act))
    # the ground-plane instances of the general kernels (fused, Env.step with observation rows, and the adaptors below)
    ap = _args(nat, 2, DT, float(np.float32(2 / 240)), seed=5, type_id=tid, options=nat.OPT_PLANE)
    nat.check(ctx.lib.dsim_step(ctx.handle, s, n, sv, tv, ctypes.byref(ap)))
    ap.action = bufs["action"][1].data_ptr()
    ap.obs_out, ap.obs_width = extra["obs2"][1].data_ptr(), 16 + ctx.n_act
    nat.check(ctx.lib.dsim_physics(ctx.handle, s, n, sv, bufs["echo"][1].data_ptr(), ctypes.byref(ap)))
    cnt_buf = torch.full((n_pad + 2 * G,), -7, dtype=torch.int32, device=dev)
    lst_buf = torch.full((4 * n_pad + 2 * G,), -7, dtype=torch.int32, device=dev)
    nat.check(ctx.lib.dsim_adjacency(ctx.handle, s, n, sv, ctypes.byref(g), 5.0, cnt_buf[G:].data_ptr(), lst_buf[G:].data_ptr(), 4))
    if fleet_kind == "quad":
        a.action = bufs["action"][1].data_ptr()
        for mode in (nat.ADAPT_VELOCITY, nat.ADAPT_RPYT):
            for opt in (0, nat.OPT_PLANE):
                a.options = opt
                nat.check(ctx.lib.dsim_step_adaptor(ctx.handle, s, n, sv, bufs["action"][1].data_ptr(), mode,
                                                    bufs["echo"][1].data_ptr(), ctypes.byref(a)))
        a.options = 0
        a.action = None
        a1.options = nat.OPT_CHAINED
        nat.check(ctx.lib.dsim_step(ctx.handle, s, n, sv, tv, ctypes.byref(a1)))
        nat.check(ctx.lib.dsim_materialize(ctx.handle, s, n, sv))
        gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "traj_track_waypoints.npz"))
        co = torch.from_numpy(np.ascontiguousarray(gold["coeffs"])).double().to(dev)
        ts = torch.from_numpy(gold["TS"]).double().to(dev)
        tt = torch.full((n_pad + 2 * G,), -3.0, dtype=torch.float64, device=dev); tt[G:G + n_pad] = 0.5
        ys = torch.full((3 * n_pad + 2 * G,), -3.0, dtype=torch.float64, device=dev); ys[G:G + 3 * n_pad] = 0.0; ys[G + n_pad:G + 2 * n_pad] = 1.0
        nat.check(ctx.lib.dsim_traj_sample(ctx.handle, s, n, co.data_ptr(), ts.data_ptr(), len(gold["TS"]) - 1,
                                           tt[G:].data_ptr(), 1 / 48, ys[G:].data_ptr(), None, tv))
        torch.cuda.synchronize()
        assert bool((tt[:G] == -3.0).all()) and bool((tt[-G:] == -3.0).all())
        assert bool((ys[:G] == -3.0).all()) and bool((ys[-G:] == -3.0).all())
    torch.cuda.synchronize()
    assert bool((cnt_buf[:G] == -7).all()) and bool((cnt_buf[-G:] == -7).all())
    assert bool((lst_buf[:G] == -7).all()) and bool((lst_buf[-G:] == -7).all())
    bufs.update(extra)
    for name, (buf, inner) in bufs.items():
        assert bool((buf[:G] == SENT).all()) and bool((buf[-G:] == SENT).all()), name
    st_inner = bufs["state"][1]
    assert bool(torch.isfinite(st_inner.reshape(-1)[: 13 * 64] if layout == "tile64" else st_inner[:n]).all())
    ctx.close()


def test_abi_argument_errors(gpu):
    nat, fleet = gpu
    t = params.builtin_type("robobee")
    ctx = fleet.Context([t])
    st = fleet.FleetState(ctx, 64)
    tg = fleet.Targets(ctx, 64)
    bad = st.view(); bad.n_pad = 65
    a = _args(nat, 1, DT, DT)
    assert ctx.lib.dsim_step(ctx.handle, _stream(ctx), 64, bad, tg.view(), ctypes.byref(a)) == -2   # DSIM_E_LAYOUT
    assert ctx.lib.dsim_step(ctx.handle, _stream(ctx), 0, st.view(), tg.view(), ctypes.byref(a)) == -1
    assert ctx.lib.dsim_step(None, _stream(ctx), 64, st.view(), tg.view(), ctypes.byref(a)) == -1
    a.dt_phys = 0.0
    assert ctx.lib.dsim_step(ctx.handle, _stream(ctx), 64, st.view(), tg.view(), ctypes.byref(a)) == -1
    assert b"layout" in ctx.lib.dsim_strerror(-2)
    ctx.close()


# ---------------------------------------------------------------------------
# round 2: the two-call loop's fast forms, neighbour lists in the observation, prebinned neighbour grid
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("layout", ["soa", "tile64"])
@pytest.mark.parametrize("noise", [0, 9])
def test_env_step_fused_observation_and_zero_copy_command(gpu, layout, noise):
    """The reference-shaped loop on a fleet in tensor mode: env.step(cmd) is ONE launch (k_physics_fast) that also
    writes the [N,20] observation rows, computeControlFromState(None) one launch (k_control_fast) whose command
    array goes back into env.step without a copy.  Every step against the oracle from the device's previous state;
    observation rows (all 20 columns) against orc_state_vector; and the same flight through the general kernels
    (ragged fleet) gives the same rows."""
    from dronesim_amd.control import INDIControl
    from dronesim_amd.envs import CtrlAviary
    nat, fleet = gpu
    n = 2048
    rng = np.random.default_rng(17)
    xyz = np.stack([rng.uniform(-30, 30, n), rng.uniform(-30, 30, n), rng.uniform(1, 6, n)], 1)
    rpy = np.stack([rng.uniform(-0.4, 0.4, n), rng.uniform(-0.4, 0.4, n), rng.uniform(-3, 3, n)], 1)
    env = CtrlAviary(["robobee"], n, initial_xyzs=xyz, initial_rpys=rpy, aggregate_phy_steps=2, noise_seed=noise,
                     dict_io=False, layout=layout)
    ctrl = INDIControl("robobee", env=env)
    t = params.builtin_type("robobee")
    O = orc.Oracle([t])
    dtc = float(np.float32(2 / 240))
    tpos = f32(xyz + rng.uniform(-0.5, 0.5, (n, 3)))
    tgt = np.concatenate([tpos, np.zeros((n, 6)), np.full((n, 1), float(np.float32(0.3)))], 1)
    cmd = torch.full((n, 4), 0.45, device=env.ctx.device)
    for k in range(6):
        r0 = env.state.rigid_aos()
        act = f32(cmd.double().cpu().numpy())
        obs, reward, done, info = env.step(cmd)
        assert obs.shape == (n, 20) and reward == -1 and done is False
        if k > 0:
            assert env._action_keep is cmd                    # the controller's array went in by pointer, no copy
        a6 = np.zeros((n, 6)); a6[:, :4] = act
        nz = None
        if noise:
            nz = np.zeros((n, 2, 12))
            for i in range(n):
                for s_ in range(2):
                    u = O.noise_normals(noise, i, k * 2 + s_, 4)
                    nz[i, s_, 0:4], nz[i, s_, 6:10] = u[0:4] * 0.01, u[4:8] * 0.001
        r1, last = r0.copy(), np.zeros((n, 6))
        O.physics(r1, env.state.mem_aos(), 2, DT, action=a6, noise=nz, last_action=last)
        assert_step_parity(f"two_call_loop physics[{layout},{noise}]", [t], None, r0, env.state.mem_aos(), tgt,
                           env.state.rigid_aos(), None, r1, None, DT, dtc, 2, control=False, action=act)
        rows = obs.double().cpu().numpy()
        _check_obs_rows(f"two_call_loop obs[{layout}]", O, rows, env.state.rigid_aos(), last, None, [t])
        # == the stand-alone observation kernel: copies bit for bit, the Euler angles to the last place or two (two
        # compilations of the same polynomial atan2 / asin; each is held to _obs_rpy_tol against the oracle)
        alone = env.observe().double().cpu().numpy()
        cp = [c for c in range(20) if not 7 <= c < 10]
        np.testing.assert_array_equal(rows[:, cp], alone[:, cp])
        _check_obs_rows(f"two_call_loop obs[{layout}]", O, alone, env.state.rigid_aos(), last, None, [t])
        m0 = env.state.mem_aos()
        cmd, pos_e, yaw_e = ctrl.computeControlFromState(dtc, None, target_pos=torch.from_numpy(tpos.T.copy()).float().to(env.ctx.device),
                                                         target_rpy=np.array([0, 0, 0.3]))
        m1 = m0.copy()
        rc, pe, ye = O.control(env.state.rigid_aos(), m1, tgt, dtc)
        assert rc == 0
        assert_control_parity(f"two_call_loop control[{layout}]", [t], None, env.state.rigid_aos(), m0, tgt,
                              env.state.mem_aos(), m1, dtc)
        np.testing.assert_array_equal(cmd.cpu().numpy(), env.state.mem_aos()[:, 7:11].astype(np.float32))
    # a per-drone target tensor is copied on every call — written in place, also through a view, it is picked up — and
    # only fleet.frozen(t), the caller's promise not to write it, lets the fleet-sized copy be skipped (the loop of
    # examples/fly_INDI.py passes the same target every call)
    from dronesim_amd.fleet import frozen
    tp = torch.from_numpy(tpos.T.copy()).float().to(env.ctx.device)
    held = ctrl._targets
    ctrl.computeControlFromState(dtc, None, target_pos=tp, target_rpy=np.array([0, 0, 0.3]))
    np.testing.assert_array_equal(held.fields(0, 3)[:, :n].cpu().numpy(), tp.cpu().numpy())
    tp[1, 5:9] += 2.0                                             # in place, through a view
    _, pos_e, _ = ctrl.computeControlFromState(dtc, None, target_pos=tp, target_rpy=np.array([0, 0, 0.3]))
    np.testing.assert_array_equal(held.fields(0, 3)[:, :n].cpu().numpy(), tp.cpu().numpy())
    want = tp.cpu().numpy().T - env.state.rigid_aos()[:, 0:3].astype(np.float32)
    np.testing.assert_allclose(pos_e.cpu().numpy(), want, rtol=0, atol=1e-5)
    fz = frozen(tp)
    ctrl.computeControlFromState(dtc, None, target_pos=fz, target_rpy=np.array([0, 0, 0.3]))
    before = tp.clone()
    tp += 1.0                                                     # the promise broken on purpose: the block keeps what it copied
    ctrl.computeControlFromState(dtc, None, target_pos=fz, target_rpy=np.array([0, 0, 0.3]))
    np.testing.assert_array_equal(held.fields(0, 3)[:, :n].cpu().numpy(), before.cpu().numpy())
    ctrl.computeControlFromState(dtc, None, target_pos=tp, target_rpy=np.array([0, 0, 0.3]))     # a plain tensor again: copied
    np.testing.assert_array_equal(held.fields(0, 3)[:, :n].cpu().numpy(), tp.cpu().numpy())
    env.close()


def test_fused_observation_general_kernels_agree(gpu):
    """dsim_physics with obs_out on a ragged mixed fleet handed over interleaved (stored type-major: the run kernels write
    the rows themselves, in the caller's numbering): the rows equal dsim_observe's — the copies bit for bit, the Euler angles
    (evaluated by another kernel's instruction stream) within rounding."""
    from dronesim_amd.envs import CtrlAviary
    n = 333
    rng = np.random.default_rng(19)
    xyz = np.stack([rng.uniform(-5, 5, n), rng.uniform(-5, 5, n), rng.uniform(1, 3, n)], 1)
    tid = (np.arange(n) % 2).astype(np.uint8)
    env = CtrlAviary(["robobee", "hexa_6DOF"], n, initial_xyzs=xyz, aggregate_phy_steps=3, noise_seed=0, dict_io=False,
                     type_ids=tid)
    act = torch.from_numpy(rng.uniform(0.3, 0.6, (n, 6)).astype(np.float32)).to(env.ctx.device)
    for _ in range(3):
        obs, *_ = env.step(act)
        a_, b_ = obs.cpu().numpy(), env.observe().cpu().numpy()
        cp = [c for c in range(22) if not 7 <= c < 10]
        np.testing.assert_array_equal(a_[:, cp], b_[:, cp])
        np.testing.assert_allclose(a_[:, 7:10], b_[:, 7:10], rtol=0, atol=2e-6)
    O = orc.Oracle(env.types)
    last = np.zeros((n, 6)); last[:, :] = act.double().cpu().numpy(); last[tid == 0, 4:6] = 0.0
    _check_obs_rows("fused_obs_general", O, obs.double().cpu().numpy(), env.state.rigid_aos(), last, tid, env.types)
    env.close()


def test_neighbor_lists_in_the_fleet_observation(gpu):
    """CtrlAviary(..., neighbors_k=K) in tensor mode: the observation carries, per drone, the count and the indices
    of the drones within NEIGHBOURHOOD_RADIUS — the sparse form of the reference's adjacency row
    (CtrlAviary.py:225-231, BaseAviary.py:901-921), from the device-side grid query; checked against the O(N^2) rule."""
    from dronesim_amd.envs import CtrlAviary, FleetObs
    n, radius, K = 2500, 4.0, 48
    rng = np.random.default_rng(23)
    xyz = np.stack([rng.uniform(0, 60, n), rng.uniform(0, 60, n), rng.uniform(0.5, 8, n)], 1)
    env = CtrlAviary(["robobee"], n, neighbourhood_radius=radius, initial_xyzs=xyz, noise_seed=0, dict_io=False, neighbors_k=K)
    obs = env.reset()
    assert isinstance(obs, FleetObs) and obs.state.shape == (n, 20) and obs.neighbor_list.shape == (K, n)
    obs, *_ = env.step(torch.full((n, 4), 0.5, device=env.ctx.device))
    p = obs.state[:, 0:3].cpu().numpy()
    d2 = ((p[:, None, :] - p[None, :, :]) ** 2).sum(-1)
    adj = (d2 < np.float32(radius) ** 2) & ~np.eye(n, dtype=bool)
    np.testing.assert_array_equal(obs.neighbor_count.cpu().numpy(), adj.sum(1))
    L = obs.neighbor_list.cpu().numpy()
    for i in rng.choice(n, 150, replace=False):
        got = set(int(x) for x in L[:, i] if x >= 0)
        assert got == set(np.nonzero(adj[i])[0].tolist()) or (adj[i].sum() > K and len(got) == K)
    env.close()


def test_action_adaptor_envs_refuse_physics_modes_they_do_not_fly(gpu):
    from dronesim_amd.envs import Physics, RPYTAviary, VelocityAviary
    for cls in (VelocityAviary, RPYTAviary):
        for ph in (Physics.PYB_DRAG, Physics.PYB_GND, Physics.PYB_DW, Physics.PYB_GND_DRAG_DW):
            with pytest.raises(NotImplementedError):
                cls(["robobee"], 4, initial_xyzs=np.zeros((4, 3)) + 1.0, physics=ph)


def test_prebinned_neighbour_grid_is_dropped_when_the_state_changes_behind_it(gpu):
    """step_fused() lets the step kernel fill the next step's neighbour grid; a physics-only step, a reset or a host
    write to the state in between must not leave stale cells behind: the next downwash force is the brute-force one
    on the CURRENT positions."""
    from dronesim_amd.envs import CtrlAviary, Physics
    from dronesim_amd.fleet import Targets
    n = 768
    rng = np.random.default_rng(29)
    xyz = np.stack([rng.uniform(0, 25, n), rng.uniform(0, 25, n), rng.uniform(0.5, 9, n)], 1)
    tid = (np.arange(n) % 2).astype(np.uint8)
    env = CtrlAviary(["robobee", "hexa_6DOF"], n, initial_xyzs=xyz, physics=Physics.PYB_DW, noise_seed=0, dict_io=False,
                     type_ids=tid)
    tg = Targets(env.ctx, n); tg.set(pos=f32(xyz).T, yaw=0.0)
    O = orc.Oracle(env.types)

    def force_ok():
        f = env._downwash.compute()[2, :n]
        f = (f if env.order is None else env.order.to_caller(f, 0)).cpu().numpy()      # (the force is per storage slot)
        r = env.state.rigid_aos()
        ref = O.downwash(r, r[:, 0:3], type_id=tid)
        assert_downwash("downwash prebinned grid", f, ref, env.types, tid, r[:, 0:3], r[:, 0:3])
        return ref

    for _ in range(3):
        env.step_fused(tg)                                   # prebins for the next compute()
    assert env._downwash._prebin_version is not None
    ref0 = force_ok()                                        # consumes the prebinned grid
    env.step_fused(tg)
    env.step(torch.full((n, 6), 0.47, device=env.ctx.device))      # Env.step moves the drones behind the prebinned grid
    force_ok()
    env.step_fused(tg)
    moved = env.state.fields(0, 3).clone(); moved[0] += 3.0; moved[2] = moved[2].flip(0)
    env.state.set_fields(0, moved)                                 # host write behind it
    ref1 = force_ok()
    assert np.abs(ref1 - ref0).max() > 1e-3
    env.step_fused(tg)
    env.reset()
    force_ok()
    env.close()


def test_wls_fallback_queue_of_a_large_fleet(gpu):
    """A saturating target step on a large hexa fleet sends a large share of it through the active-set loop at once
    (k_wls_fallback: chip-sized grid, LDS work area): counted, finite, within the PWM box, and equal to the oracle's
    full wls_alloc on a sample."""
    nat, fleet = gpu
    n = 131072
    t = params.builtin_type("hexa_6DOF")
    ctx = fleet.Context([t])
    rng = np.random.default_rng(37)
    rigid, mem, tgt = random_fleet(rng, n, n_act=6, tilt=0.3, rate=1.0)
    mem[:, 7:13] = f32(rng.uniform(0.0, 1.0, (n, 6)))            # commands anywhere in the box: bounds bind
    tgt[:, 0:3] = f32(rigid[:, 0:3] + rng.uniform(-40, 40, (n, 3)))
    rigid[:, 10:13] = f32(rng.uniform(-25, 25, (n, 3)))          # violent rates: the first WLS iteration leaves the box
    st, tg = fleet.FleetState(ctx, n), fleet.Targets(ctx, n)
    st.load_aos(rigid, mem)
    tg.set_fields(0, torch.from_numpy(np.ascontiguousarray(tgt.T)))
    a = _args(nat, 0, DT, DT)
    torch.cuda.synchronize()
    import time as _t
    t0 = _t.perf_counter()
    nat.check(ctx.lib.dsim_control(ctx.handle, _stream(ctx), n, st.view(), tg.view(), ctypes.byref(a), None, None))
    torch.cuda.synchronize()
    el = _t.perf_counter() - t0
    fb = _query(nat, ctx, 0)
    assert fb > n // 20, fb                                      # a real queue, tens of thousands long
    assert _query(nat, ctx, 1) == 0
    assert el < 2.0, el                                          # worked off by the whole chip, not by 32 workgroups
    got = st.mem_aos()
    assert np.isfinite(got).all() and got[:, 7:13].min() >= 0.0 and got[:, 7:13].max() <= 1.0
    idx = rng.choice(n, 3000, replace=False)
    o = mem[idx].copy()
    assert orc.Oracle([t]).control(rigid[idx], o, tgt[idx], DT)[0] == 0
    # the active-set iterations solve lstsq problems whose rows are scaled by gamma Wv (up to 1e8): the fp32 inputs nu
    # enter with that conditioning, so the increments are compared where the allocation is well inside the box and
    # the clip decides the rest identically
    both_clipped = ((got[idx, 7:13] <= 0) | (got[idx, 7:13] >= 1)) & ((o[:, 7:13] <= 0) | (o[:, 7:13] >= 1))
    err = np.abs(got[idx, 7:13] - o[:, 7:13])
    assert np.median(err) < 1e-5 and (err[both_clipped] == 0).all()
    # A bound for EVERY sampled drone (round 2 accepted a q99): the device evaluates the virtual control nu in fp32 —
    # its dominant terms are rates / dt, so a few ulps of the rates move nu by what the fp32 arithmetic may — and the
    # active-set solution is piecewise linear in nu.  The oracle's own allocation, re-run on inputs moved by +-4 fp32
    # ulps, shows per drone how far such a change carries the command: the device must stay within a small multiple.
    from tests.util import WORST
    spread = np.zeros_like(err)
    O2 = orc.Oracle([t])
    for trial in range(8):
        rp, mp = rigid[idx].copy(), mem[idx].copy()
        rp[:, 7:13] += rng.choice([-4.0, 4.0], rp[:, 7:13].shape) * ulp32(rp[:, 7:13])
        mp[:, 0:6] += rng.choice([-4.0, 4.0], mp[:, 0:6].shape) * ulp32(mp[:, 0:6])
        assert O2.control(rp, mp, tgt[idx], DT)[0] == 0
        spread = np.maximum(spread, np.abs(mp[:, 7:13] - o[:, 7:13]))
    tol = 4.0 * spread.max(1, keepdims=True) + REL_TOL * np.abs(o[:, 7:13] - mem[idx, 7:13]) + 4 * ulp32(1.0)
    ratio = err / tol
    WORST["wls fallback queue: per-drone bound"] = float(ratio.max())
    assert ratio.max() <= 1.0, (float(ratio.max()), int(np.argmax(ratio.max(1))), float(err.max()), float(np.quantile(tol, 0.99)))
    assert np.quantile(tol, 0.99) < 5e-3                          # ... and that bound is itself small for all but a few
    # a second call finds the queue emptied by the first
    nat.check(ctx.lib.dsim_control(ctx.handle, _stream(ctx), n, st.view(), tg.view(), ctypes.byref(a), None, None))
    assert np.isfinite(st.mem_aos()).all()
    ctx.close()


def test_bench_launches_its_own_ranks(gpu):
    """`python bench.py --gpus 2` (no launcher around it) starts two ranks itself, before the parent touches the GPU;
    with DSIM_BENCH_BACKEND=gloo both may share this box's one device.  One JSON line, n_gpus = 2, whole-job value."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DSIM_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "2",
                        "--workload", "config4"], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["dist"]["world_size"] == 2 and d["dist"]["backend"] == "gloo"
    assert d["config"]["drones_per_gpu"] == 65536 and d["value"] > 1e8 and d["steps_timed"] >= 10
    assert d["scaling"] == "weak" and "also" not in d


def test_bench_settled_form_times_launches_right_behind_the_settling_load(gpu):
    """`bench.py --settle-seconds S` (Fleet._timed_settled): the device is kept under the workload for S seconds, the timed
    launches follow without a gap, device time from events in the stream; the line says so, counts whole regions of --steps
    launches and agrees with the plain form to within what the device's clocks do to a short run."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = {}
    for settle in ("0", "0.05"):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "config4", "--substeps", "5", "--steps", "10",
                            "--warmup", "2", "--settle-seconds", settle, "--no-also", "--no-cpu-baseline"],
                           capture_output=True, text=True, timeout=600, env=env, cwd=root)
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1
        out[settle] = json.loads(lines[0])
    a, b = out["0"], out["0.05"]
    assert a["settle_seconds"] == 0.0 and b["settle_seconds"] == 0.05
    for d in (a, b):
        assert d["steps"] == 10 and d["steps_timed"] == 10 * d["timed_regions"] and d["steps_timed"] * d["ms_per_step"] >= 15.0     # (the count of regions comes from the first one's duration)
        assert d["value"] > 1e8 and d["config"]["phys_substeps"] == 5 and d["config"]["drones_per_gpu"] == 65536
    assert 0.5 < a["roofline"]["launch_us"] / b["roofline"]["launch_us"] < 2.0


def test_ground_plane_watch_counts_what_pybullet_would_have_caught(gpu):
    """Plane contact is not modelled; the library counts instead every drone-step that ends with the
    vehicle's collision cylinder (robobee.urdf:72-77: radius 0.15 m, 0.1 m long) at or below z = 0: exact against the
    device's own trajectory for falling and tumbling drones, zero for a hovering fleet, through the fused step, the
    physics-only step and the mixed-fleet kernel."""
    from dronesim_amd.envs import CtrlAviary
    from dronesim_amd.fleet import Targets
    nat, fleet = gpu
    t = params.builtin_type("robobee")

    def expected(rigid_f32):
        q, z = rigid_f32[:, 3:7].astype(np.float32), rigid_f32[:, 2].astype(np.float32)
        r22 = np.float32(1) - np.float32(2) * (q[:, 0] * q[:, 0] + q[:, 1] * q[:, 1])
        reach = np.float32(t.collision_below) * np.abs(r22) + np.float32(t.collision_radius) * np.sqrt(np.maximum(np.float32(1) - r22 * r22, np.float32(0)))
        return z <= reach

    # (a) free fall from staggered heights, some tilted (the rim reaches the plane before the centre would): Env.step only
    n = 320
    xyz = np.stack([np.arange(n) * 1.0, np.zeros(n), np.linspace(0.06, 0.9, n)], 1)
    rpy = np.zeros((n, 3)); rpy[::3, 0] = 0.7; rpy[1::3, 1] = -1.2
    env = CtrlAviary(["robobee"], n, initial_xyzs=xyz, initial_rpys=rpy, aggregate_phy_steps=2, noise_seed=0, dict_io=False)
    assert env.ground_contacts() == 0
    want = 0
    zero = torch.zeros((n, 4), device=env.ctx.device)
    for k in range(30):
        env.step(zero)
        hit = expected(env.state.rigid_aos())
        want += int(hit.sum())
    assert 0 < want < 30 * n and env.ground_contacts() == want
    # (b) the fused step (fast kernel) on the same env keeps counting
    tg = Targets(env.ctx, n); tg.set(pos=f32(xyz).T, yaw=0.0)
    for k in range(5):
        env.step_fused(tg)
        want += int(expected(env.state.rigid_aos()).sum())
    assert env.ground_contacts() == want
    env.close()
    # (c) a fleet hovering on its targets one metre up never touches it; mixed fleet, LDS-staged kernel
    n = 4096
    xyz = np.stack([np.arange(n) % 64, np.arange(n) // 64, np.full(n, 1.0)], 1).astype(np.float64)
    tid = (np.arange(n) % 2).astype(np.uint8)
    env = CtrlAviary(["robobee", "hexa_6DOF"], n, initial_xyzs=xyz, noise_seed=3, dict_io=False, type_ids=tid, layout="tile64")
    tg = Targets(env.ctx, n, "tile64"); tg.set(pos=f32(xyz).T, yaw=0.2)
    hover = np.where(tid[:, None] == 0, t.hover_pwm, params.builtin_type("hexa_6DOF").hover_pwm) * np.ones((n, 6))
    env.step_fused(tg, action=hover.astype(np.float32))
    for _ in range(200):
        env.step_fused(tg)
    assert env.ground_contacts() == 0 and float(env.state.fields(2, 1).min()) > 0.5      # (start-up dip of ~0.1 m)
    env.close()


# ---------------------------------------------------------------------------
# DSIM_OPT_PLANE: the ground plane of the reference's world (BaseAviary.py:660) — the product-defined contact model of
# oracle/dsim_oracle.c:orc_plane_contact on the device (dsim_device.h:plane_contact), against the oracle
# ---------------------------------------------------------------------------
from tests.util import PLANE_SWEEPS  # noqa: E402  (DSIM_PLANE_ITERS: every sweep updates (v, w) twelve times)


def _near_ground_fleet(t, n, seed, n_act=4):
    """Seeded fleet around the plane: a third resting on it, a third arriving (tilted, descending, spinning), a third
    flying clear of the 0.02 m margin."""
    rng = np.random.default_rng(seed)
    rigid, mem, tgt = random_fleet(rng, n, n_act=n_act, tilt=0.6, speed=1.5, rate=2.0, spread=20.0)
    h = t.rest_height
    kind = np.arange(n) % 3
    rest = kind == 0
    rigid[rest, 2] = h + rng.uniform(-2e-3, 2e-3, rest.sum())                 # within the penetration / margin band
    rigid[rest, 3:7] = np.stack([orc.quat_from_euler([0.0, 0.0, y]) for y in rng.uniform(-3, 3, rest.sum())])
    rigid[rest, 7:10] = rng.uniform(-0.3, 0.3, (rest.sum(), 3)) * np.array([1.0, 1.0, 0.2])    # sliding
    rigid[rest, 10:13] = rng.uniform(-0.2, 0.2, (rest.sum(), 3))
    arrive = kind == 1
    rigid[arrive, 2] = h + rng.uniform(-0.01, 0.12, arrive.sum())             # tilted rims reach the plane at different heights
    rigid[arrive, 9] = -np.abs(rigid[arrive, 9])
    rigid[kind == 2, 2] = rng.uniform(0.5, 3.0, (kind == 2).sum())
    return f32(rigid), mem, tgt, kind


@pytest.mark.parametrize("model,sub", [("robobee", 1), ("robobee", 5), ("tello", 2), ("hexa_6DOF", 1), ("hexa_6DOF", 4)])
def test_plane_contact_vs_oracle(gpu, model, sub):
    """Env.step (explicit action) and the fused step with DSIM_OPT_PLANE from the same near-ground state, per drone
    and field on the increments.  Tolerance: the step's own bar (tests/util.py) with the contact terms added to the
    magnitudes (plane_terms) and PLANE_SWEEPS more roundings per sub-step — stated here, checked per case."""
    from tests.util import plane_terms
    nat, fleet = gpu
    t = params.builtin_type(model)
    na = t.n_act
    n = 1800
    ctx = fleet.Context([t])
    st = fleet.FleetState(ctx, n, "soa", 256)
    tg = fleet.Targets(ctx, n, "soa", pad=256)
    rigid, mem, tgt, kind = _near_ground_fleet(t, n, 131 + sub, na)
    st.load_aos(rigid, mem)
    tg.set_fields(0, torch.from_numpy(np.ascontiguousarray(tgt.T)))
    O = orc.Oracle([t])
    dtc = float(np.float32(sub / 240.0))
    k_contact = K_ULP * sub * (1 + PLANE_SWEEPS)
    # (a) Env.step with an explicit action: idle, hover and full thrust
    rng = np.random.default_rng(5)
    act = f32(t.hover_pwm * rng.choice([0.0, 1.0, 1.6], (n, 1)) * np.ones((1, na)))
    act_dev = torch.zeros((na, st.n_pad), device=ctx.device); act_dev[:, :n] = torch.from_numpy(act.T).float()
    echo = torch.zeros((na, st.n_pad), device=ctx.device)
    a = _args(nat, sub, DT, dtc, options=nat.OPT_PLANE, action=act_dev)
    nat.check(ctx.lib.dsim_physics(ctx.handle, _stream(ctx), n, st.view(), echo.data_ptr(), ctypes.byref(a)))
    a6 = np.zeros((n, 6)); a6[:, :na] = act
    ref = rigid.copy()
    O.physics(ref, mem, sub, DT, action=a6, options=nat.OPT_PLANE)
    free = rigid.copy()
    O.physics(free, mem, sub, DT, action=a6)
    touched = np.abs(free - ref).max(1) > 1e-6
    assert touched[kind == 0].mean() > 0.3 and not touched[kind == 2].any() and 0.15 < touched[kind == 1].mean() < 0.95
    got = st.rigid_aos()
    assert_step_parity(f"plane_physics[{model},{sub}]", [t], None, rigid, mem, tgt, got, None, ref, None, DT, dtc, sub,
                       control=False, k=k_contact, action=act, extra_terms=plane_terms([t], None, rigid, dtc))
    # drones clear of the margin take the same arithmetic as without the option: the plain bar holds for them
    fly = kind == 2
    assert_step_parity(f"plane_physics[{model},{sub}] clear of the plane", [t], None, rigid[fly], mem[fly], tgt[fly], got[fly],
                       None, ref[fly], None, DT, dtc, sub, control=False, action=act[fly])
    # nothing ends the step deeper in the plane than it started, beyond the solver's residual
    def lowest(r):
        r22 = 1.0 - 2.0 * (r[:, 3] ** 2 + r[:, 4] ** 2)
        return r[:, 2] - (t.rest_height * np.abs(r22) + t.collision_radius * np.sqrt(np.maximum(1.0 - r22 * r22, 0.0)))
    # (the eight body-fixed rim points see an edge at most r sin(tilt) (1 - cos 22.5 deg) late)
    r22 = 1.0 - 2.0 * (got[:, 3] ** 2 + got[:, 4] ** 2)
    late = 0.0762 * t.collision_radius * np.sqrt(np.maximum(1.0 - r22 * r22, 0.0))
    assert (lowest(got) >= np.minimum(lowest(rigid), 0.0) - late - 2e-3).all()
    # (b) the fused step (physics + INDI law) from the device's state
    r0, m0 = st.rigid_aos(), st.mem_aos()
    a2 = _args(nat, sub, DT, dtc, options=nat.OPT_PLANE)
    nat.check(ctx.lib.dsim_step(ctx.handle, _stream(ctx), n, st.view(), tg.view(), ctypes.byref(a2)))
    r1, m1 = r0.copy(), m0.copy()
    assert O.step(r1, m1, tgt, sub, DT, dtc, options=nat.OPT_PLANE) == 0
    assert_step_parity(f"plane_fused[{model},{sub}]", [t], None, r0, m0, tgt, st.rigid_aos(), st.mem_aos(), r1, m1, DT, dtc,
                       sub, k=k_contact, extra_terms=plane_terms([t], None, r0, dtc))
    ctx.close()


def test_plane_touchdown_flight_config1(gpu):
    """BASELINE configs[0]'s flight with the plane the reference's world has: a robobee starting 0.1 m up under the
    INDI law sinks onto the plane while the thrust state winds up, rests ON it (instead of passing through z = 0 to
    -0.14 m), lifts off and reaches its target.  CtrlAviary switches the plane on by itself for
    such a fleet; every step is also judged against the oracle from the device's own previous state."""
    from dronesim_amd.control import INDIControl
    from dronesim_amd.envs import CtrlAviary
    from tests.util import plane_terms
    nat, fleet = gpu
    t = params.builtin_type("robobee")
    env = CtrlAviary(["robobee"], 1, initial_xyzs=np.array([[0.0, 0.0, 0.1]]), aggregate_phy_steps=5, noise_seed=0)
    assert env.ground_plane
    ctrl = INDIControl("robobee")
    O = orc.Oracle([t])
    dtc = float(np.float32(5 / 240.0))
    target = np.array([0.0, 0.0, 1.0])
    obs, _, _, _ = env.step({"0": np.zeros(4)})
    zs = []
    for k in range(48 * 6):
        cmd, _, _ = ctrl.computeControlFromState(control_timestep=dtc, state=obs["0"]["state"], target_pos=target,
                                                 target_rpy=np.zeros(3))
        r0 = env.state.rigid_aos()
        obs, _, _, _ = env.step({"0": cmd})
        ref = r0.copy()
        a6 = np.zeros((1, 6)); a6[0, :4] = f32(cmd)
        O.physics(ref, np.zeros((1, 13)), 5, DT, action=a6, options=nat.OPT_PLANE)
        assert_step_parity("plane_touchdown_flight", [t], None, r0, np.zeros((1, 13)), np.zeros((1, 10)), env.state.rigid_aos(),
                           None, ref, None, DT, dtc, 5, control=False, k=K_ULP * 5 * (1 + PLANE_SWEEPS), action=f32(cmd)[None, :],
                           extra_terms=plane_terms([t], None, r0, dtc))
        zs.append(float(obs["0"]["state"][2]))
    zs = np.array(zs)
    assert t.collision_below - 2e-3 < zs.min() < t.collision_below + 0.01         # rests on the plane, not below it
    on_ground = (zs < t.collision_below + 5e-3).sum() * dtc
    assert 0.03 < on_ground < 1.5
    assert abs(zs[-1] - 1.0) < 0.05 and env.ground_contacts() > 0
    env.close()
    # the same flight without the plane goes through the floor (what the watch counter reports)
    env = CtrlAviary(["robobee"], 1, initial_xyzs=np.array([[0.0, 0.0, 0.1]]), aggregate_phy_steps=5, noise_seed=0, ground_plane=False)
    ctrl = INDIControl("robobee")
    obs, _, _, _ = env.step({"0": np.zeros(4)})
    zmin = 1.0
    for k in range(48 * 2):
        cmd, _, _ = ctrl.computeControlFromState(control_timestep=dtc, state=obs["0"]["state"], target_pos=target, target_rpy=np.zeros(3))
        obs, _, _, _ = env.step({"0": cmd})
        zmin = min(zmin, float(obs["0"]["state"][2]))
    assert zmin < -0.05
    env.close()


@pytest.mark.parametrize("mode", ["velocity", "rpyt"])
def test_plane_action_adaptor_envs(gpu, mode):
    """VelocityAviary / RPYTAviary over the plane: the law on the current state, then the physics with the contact
    solve — a fleet parked on the ground and commanded down / idle stays on it, and the step matches the oracle."""
    from dronesim_amd.envs import RPYTAviary, VelocityAviary
    from tests.util import plane_terms
    nat, fleet = gpu
    t = params.builtin_type("tello")
    n = 48
    rng = np.random.default_rng(17)
    xyz = np.stack([np.arange(n) * 1.0, np.zeros(n), np.where(np.arange(n) % 2 == 0, t.collision_below, 0.5)], 1)
    cls = VelocityAviary if mode == "velocity" else RPYTAviary
    env = cls(["tello"], n, initial_xyzs=xyz, aggregate_phy_steps=2, noise_seed=0, ground_plane=True)
    O = orc.Oracle([t])
    dtc = float(np.float32(2 / 240.0))
    for k in range(25):
        if mode == "velocity":
            act = np.concatenate([rng.uniform(-1, 1, (n, 2)), -np.ones((n, 1)), np.full((n, 1), 0.2)], 1)   # descend
        else:
            act = np.concatenate([np.zeros((n, 3)), np.full((n, 1), 0.2)], 1)                                # low thrust, level
        act = f32(act)
        r0, m0 = env.state.rigid_aos(), env.state.mem_aos()
        env.step({str(i): act[i] for i in range(n)})
        got_r, got_m = env.state.rigid_aos(), env.state.mem_aos()
        ref = r0.copy()
        a6 = np.zeros((n, 6)); a6[:, :4] = got_m[:, 7:11]
        O.physics(ref, got_m.copy(), 2, DT, action=a6, options=nat.OPT_PLANE)
        assert_step_parity(f"plane_adaptor_env[{mode}]", [t], None, r0, got_m, np.zeros((n, 10)), got_r, None, ref, None, DT, dtc, 2,
                           control=False, k=K_ULP * 2 * (1 + PLANE_SWEEPS), action=got_m[:, 7:11],
                           extra_terms=plane_terms([t], None, r0, dtc))
    z = env.state.rigid_aos()[:, 2]
    assert (z[::2] > t.collision_below - 2e-3).all() and (z[::2] < t.collision_below + 5e-3).all()    # parked: still on the plane
    assert (z > t.collision_below - 2e-3).all()                                                       # nobody below it
    env.close()


def test_plane_option_routing_and_refusals(gpu):
    """The plane works for every airframe kind and storage (general kernels); the add-on formulas that are written for
    four-rotor links still refuse the hexa, with or without it."""
    nat, fleet = gpu
    t4, t6 = params.builtin_type("robobee"), params.builtin_type("hexa_6DOF")
    ctx = fleet.Context([t4, t6])
    n = 700
    tid = (np.arange(n) % 2).astype(np.uint8)
    for layout in ("soa", "tile64"):
        st = fleet.FleetState(ctx, n, layout, 256)
        tg = fleet.Targets(ctx, n, layout, pad=256)
        rigid, mem, tgt = random_fleet(np.random.default_rng(3), n, n_act=6, tilt=0.4)
        rigid[:, 2] = f32(np.where(tid == 0, t4.rest_height, t6.rest_height) + np.random.default_rng(4).uniform(-1e-3, 0.05, n))
        rigid[:, 9] = -np.abs(rigid[:, 9])
        mem[tid == 0, 11:13] = 0.0
        st.load_aos(rigid, mem)
        tg.set_fields(0, torch.from_numpy(np.ascontiguousarray(tgt.T)))
        type_dev = torch.zeros(st.n_pad, dtype=torch.uint8, device=ctx.device); type_dev[:n] = torch.from_numpy(tid)
        a = _args(nat, 2, DT, float(np.float32(2 / 240)), options=nat.OPT_PLANE, type_id=type_dev, seed=0)
        nat.check(ctx.lib.dsim_step(ctx.handle, _stream(ctx), n, st.view(), tg.view(), ctypes.byref(a)))
        O = orc.Oracle([t4, t6])
        r1, m1 = rigid.copy(), mem.copy()
        assert O.step(r1, m1, tgt, 2, DT, float(np.float32(2 / 240)), options=nat.OPT_PLANE, type_id=tid) == 0
        from tests.util import plane_terms
        assert_step_parity(f"plane_mixed_fleet[{layout}]", [t4, t6], tid, rigid, mem, tgt, st.rigid_aos(), st.mem_aos(), r1, m1,
                           DT, float(np.float32(2 / 240)), 2, k=K_ULP * 2 * (1 + PLANE_SWEEPS),
                           extra_terms=plane_terms([t4, t6], tid, rigid, float(np.float32(2 / 240))))
        a.options = nat.OPT_PLANE | nat.OPT_DRAG
        assert ctx.lib.dsim_step(ctx.handle, _stream(ctx), n, st.view(), tg.view(), ctypes.byref(a)) == -5     # DSIM_E_UNSUPPORTED
    ctx.close()


def test_plane_with_waypoints_multi_step_launches_and_fused_rows(gpu, golden_dir):
    """The plane instances of the general kernels carry every option of the general path: a fleet that starts ON the
    ground and tracks the waypoint table with n_steps = K per launch ends bit-identical to K single-step launches
    (state, waypoint counters, contact counter), noise on; and Env.step's fused observation rows over the plane equal
    the stand-alone observation kernel's."""
    from dronesim_amd.envs import CtrlAviary
    from dronesim_amd.fleet import WaypointTargets
    nat, fleet = gpu
    g = np.load(os.path.join(golden_dir, "traj_track_waypoints.npz"))
    t = params.builtin_type("robobee")
    n = 700                                                  # ragged: general kernels with or without the plane
    off = np.stack([np.arange(n) % 30 * 1.5, np.arange(n) // 30 * 1.5, np.zeros(n)], 1)
    xyz = off + np.array([0.0, 0.0, t.rest_height])          # parked on the plane
    wp0 = np.arange(n) % g["target_pos"].shape[0]
    envs, wps = [], []
    for _ in range(2):
        env = CtrlAviary(["robobee"], n, initial_xyzs=xyz, aggregate_phy_steps=2, noise_seed=11, dict_io=False, ground_plane=True)
        envs.append(env)
        wps.append(WaypointTargets(env.ctx, n, g["target_pos"], g["target_vel"], g["target_acc"], g["target_yaw"],
                                   wp_counters=wp0, offsets=off))
    a0 = np.full((n, 4), 0.1, dtype=np.float32)
    for env, wp in zip(envs, wps):
        env.step_fused(wp, action=a0)
    for _ in range(12):
        envs[0].step_fused(wps[0])
    envs[1].step_fused(wps[1], n_steps=5); envs[1].step_fused(wps[1], n_steps=7)
    np.testing.assert_array_equal(envs[0].state.fields(0, 24).cpu().numpy(), envs[1].state.fields(0, 24).cpu().numpy())
    np.testing.assert_array_equal(wps[0].counters.cpu().numpy(), wps[1].counters.cpu().numpy())
    assert envs[0].ground_contacts() == envs[1].ground_contacts() > 0
    z = envs[0].state.rigid_aos()[:, 2]
    assert (z > t.rest_height - 3e-3).all()                  # nobody went through the floor on the way up
    # Env.step with the observation rows fused (general kernel + observation kernel behind it) over the plane
    env = envs[0]
    act = torch.full((n, 4), 0.2, device=env.ctx.device)
    obs, _, _, _ = env.step(act)
    alone = env.observe()
    cp = [c for c in range(20) if not 7 <= c < 10]
    np.testing.assert_array_equal(obs.cpu().numpy()[:, cp], alone.cpu().numpy()[:, cp])
    np.testing.assert_allclose(obs.cpu().numpy()[:, 7:10], alone.cpu().numpy()[:, 7:10], rtol=0, atol=1e-6)
    for e in envs:
        e.close()


@pytest.mark.parametrize("fleet_kind", ["quad", "hexa", "mixed"])
def test_tuning_options_do_not_change_results(gpu, fleet_kind):
    """DSIM_OPT_STREAM_ON / _OFF select the streaming or the default cache policy of the same kernel: bit-identical
    states.  The option bits of the measured-and-rejected kernel forms of rounds 1-2 are IGNORED by the library: same bits again."""
    nat, fleet = gpu
    names = {"quad": ["robobee"], "hexa": ["hexa_6DOF"], "mixed": ["robobee", "hexa_6DOF"]}[fleet_kind]
    types = [params.builtin_type(m) for m in names]
    n = 2048
    na = max(t.n_act for t in types)
    rigid, mem, tgt = random_fleet(np.random.default_rng(5), n, n_act=na, tilt=0.3, rate=1.0)
    tid = (np.arange(n) % 2).astype(np.uint8) if fleet_kind == "mixed" else None
    if tid is not None:
        mem[tid == 0, 11:13] = 0.0
    out = {}
    for name, opt in (("on", nat.OPT_STREAM_ON), ("off", nat.OPT_STREAM_OFF),
                      ("generic", nat.OPT_STREAM_OFF | nat.VAR_GENERIC | nat.VAR_MIXED_V1 | nat.VAR_MIXED_RING | nat.VAR_RUNS_SEPARATE)):
        if name == "generic" and fleet_kind != "mixed":
            continue
        ctx = fleet.Context(types)
        st = fleet.FleetState(ctx, n, "tile64")
        tg = fleet.Targets(ctx, n, "tile64")
        st.load_aos(rigid, mem)
        tg.set_fields(0, torch.from_numpy(np.ascontiguousarray(tgt.T)))
        tdev = None
        if tid is not None:
            tdev = torch.zeros(st.n_pad, dtype=torch.uint8, device=ctx.device); tdev[:n] = torch.from_numpy(tid)
        a = _args(nat, 2, DT, float(np.float32(2 / 240)), options=opt, seed=21, step_index=3, type_id=tdev)
        nat.check(ctx.lib.dsim_step(ctx.handle, _stream(ctx), n, st.view(), tg.view(), ctypes.byref(a)))
        out[name] = (st.rigid_aos(), st.mem_aos())
        ctx.close()
    np.testing.assert_array_equal(out["on"][0], out["off"][0])
    np.testing.assert_array_equal(out["on"][1], out["off"][1])
    if "generic" in out:
        np.testing.assert_array_equal(out["generic"][0], out["off"][0])
        np.testing.assert_array_equal(out["generic"][1], out["off"][1])


# ---------------------------------------------------------------------------
# every template instance of the step kernels: each combination the launcher can dispatch is run once —
# streaming and default cache policy give the same bits, and the result meets the oracle at the step's bar
# ---------------------------------------------------------------------------
def _noise_block(O, types, tid, n, seed, step_index, sub, fine=None):
    """[n, sub, 12] scaled normals of a launch of `sub` sub-steps: the fine lattice for one sub-step, the coarse one for several
    (include/dronesim_amd.h: DSIM_OPT_NOISE_FINE / _COARSE), unless `fine` says which."""
    nz = np.zeros((n, sub, 12))
    for i in range(n):
        na = types[0 if tid is None else int(tid[i])].n_act
        for s_ in range(sub):
            u = O.noise_normals(seed, i, step_index * sub + s_, na, fine=(sub == 1) if fine is None else fine)
            nz[i, s_, 0:na], nz[i, s_, 6:6 + na] = rotor_noise(types[0 if tid is None else int(tid[i])], u)
    return nz


def _sweep_case(gpu, label, types, tid, n, sub, seed, options, action=None, n_steps=1, runs=None, layout="tile64", pad=256,
                fleet_kw=None, dt_phys=None, waypoints=False):
    """One dsim_step launch (both streaming policies, bit-identical) against the oracle at the step bar.  fleet_kw: keywords for
    random_fleet (envelope= ...); dt_phys: the physics period (default 1/240 s; dt_ctrl = sub x dt_phys); waypoints: the targets
    come from a three-row waypoint table + per-drone offsets (examples/fly_INDI_TrajectoryTrack.py:242-245) — the EXT instances of
    the fast kernel, also for a single Env.step per launch."""
    nat, fleet = gpu
    na = max(t.n_act for t in types)
    DT = float(np.float32(dt_phys)) if dt_phys is not None else globals()["DT"]
    rigid, mem, tgt = random_fleet(np.random.default_rng(n + sub + seed), n, n_act=na, **(fleet_kw or dict(tilt=0.3, rate=1.0)))
    if tid is not None:
        for k, t in enumerate(types):
            mem[tid == k, 7 + t.n_act:13] = 0.0
    elif na == 4:
        mem[:, 11:13] = 0.0
    if options & nat.OPT_CHAINED:
        # the chained form does not READ last_vel / last_rates: it takes them to be what a previous step left — the
        # velocity and the body rates of the stored state (computeControl stores them at the end of every call)
        mem[:, 0:3] = rigid[:, 7:10]
        for i in range(n):
            R = np.array(orc.matrix_from_quat(rigid[i, 3:7])).reshape(3, 3)
            mem[i, 3:6] = f32(R.T @ rigid[i, 10:13])
        mem = f32(mem)
    dtc = float(np.float32(sub / 240)) if dt_phys is None else float(np.float32(sub * DT))
    sidx = 4
    got = {}
    wp_rows = wp_cnt = wp_off = None
    if waypoints:
        assert n_steps == 1
        wrng = np.random.default_rng(seed + 99)
        wp_rows = wrng.uniform(-0.5, 0.5, (3, 10)).astype(np.float32)
        wp_cnt = wrng.integers(0, 3, n).astype(np.int32)
        wp_off = (tgt[:, 0:3].astype(np.float32) - wp_rows[wp_cnt, 0:3]).astype(np.float32)
        tgt = np.concatenate([(wp_rows[wp_cnt, 0:3] + wp_off).astype(np.float64), wp_rows[wp_cnt, 3:10].astype(np.float64)], 1)   # (fp32 sum, as the kernel forms it)
    for pol in (nat.OPT_STREAM_ON, nat.OPT_STREAM_OFF):
        ctx = fleet.Context(types)
        st, tg = fleet.FleetState(ctx, n, layout, pad), fleet.Targets(ctx, n, layout, pad=pad)
        st.load_aos(rigid, mem)
        tg.set_fields(0, torch.from_numpy(np.ascontiguousarray(tgt.T)))
        tdev = None
        if tid is not None:
            tdev = torch.zeros(st.n_pad, dtype=torch.uint8, device=ctx.device); tdev[:n] = torch.from_numpy(tid)
        adev = None
        if action is not None:
            adev = torch.zeros((na, st.n_pad), device=ctx.device); adev[:, :n] = torch.from_numpy(np.ascontiguousarray(action.T)).float()
        a = _args(nat, sub, DT, dtc, options=options | pol, seed=seed, step_index=sidx, type_id=tdev, action=adev)
        a.n_steps = n_steps
        if waypoints:
            wp = fleet.WaypointTargets(ctx, n, wp_rows[:, 0:3], wp_rows[:, 3:6], wp_rows[:, 6:9], wp_rows[:, 9], wp_counters=wp_cnt, offsets=wp_off, pad=pad)
            wp.fill(a)
        arr = None
        if runs is not None:
            arr = (nat.TypeRun * len(runs))()
            for k, (f, c, ty) in enumerate(runs):
                arr[k].first, arr[k].count, arr[k].type = f, c, ty
            a.runs, a.n_runs = ctypes.addressof(arr), len(runs)
        nat.check(ctx.lib.dsim_step(ctx.handle, _stream(ctx), n, st.view(), tg.view(), ctypes.byref(a)))
        if options & nat.OPT_CHAINED:
            nat.check(ctx.lib.dsim_materialize(ctx.handle, _stream(ctx), n, st.view()))
        got[pol] = (st.rigid_aos(), st.mem_aos())
        ctx.close()
    np.testing.assert_array_equal(got[nat.OPT_STREAM_ON][0], got[nat.OPT_STREAM_OFF][0], err_msg=label)
    np.testing.assert_array_equal(got[nat.OPT_STREAM_ON][1], got[nat.OPT_STREAM_OFF][1], err_msg=label)
    O = orc.Oracle(types)
    r, m = rigid.copy(), mem.copy()
    for k in range(n_steps):
        r0, m0 = r.copy(), m.copy()
        nz = _noise_block(O, types, tid, n, seed, sidx + k, sub) if seed else None
        a6 = None
        if action is not None and k == 0:
            a6 = np.zeros((n, 6)); a6[:, :na] = action
        assert O.step(r, m, tgt, sub, DT, dtc, noise=nz, type_id=tid, action=a6, options=options & (nat.OPT_DRAG | nat.OPT_GROUND)) == 0
    if n_steps == 1:
        assert_step_parity(label, types, tid, rigid, mem, tgt, got[nat.OPT_STREAM_OFF][0], got[nat.OPT_STREAM_OFF][1], r, m,
                           DT, dtc, sub, action=action, noise=bool(seed))
    else:      # several Env.steps in one launch: the bar of the LAST step from the oracle's previous state, widened by the count
        assert_step_parity(label, types, tid, r0, m0, tgt, got[nat.OPT_STREAM_OFF][0], got[nat.OPT_STREAM_OFF][1], r, m,
                           DT, dtc, sub, k=K_ULP * sub * 4 * n_steps)


@pytest.mark.parametrize("seed", [0, 7])
@pytest.mark.parametrize("sub", [1, 2])
def test_every_instance_of_the_single_type_step_kernels(gpu, sub, seed):
    nat, fleet = gpu
    rb, hx = params.builtin_type("robobee"), params.builtin_type("hexa_6DOF")
    n = 512
    act4 = f32(np.random.default_rng(1).uniform(0.3, 0.7, (n, 4)))
    act6 = f32(np.random.default_rng(2).uniform(0.3, 0.7, (n, 6)))
    # k_step_fast<NOISE, NT, EXT, CH, SUB, ACT>: plain, explicit action, chained, several steps per launch (+ chained)
    _sweep_case(gpu, f"sweep fast plain[{sub},{seed}]", [rb], None, n, sub, seed, 0)
    _sweep_case(gpu, f"sweep fast action[{sub},{seed}]", [rb], None, n, sub, seed, 0, action=act4)
    _sweep_case(gpu, f"sweep fast chained[{sub},{seed}]", [rb], None, n, sub, seed, nat.OPT_CHAINED)
    _sweep_case(gpu, f"sweep fast 3 steps[{sub},{seed}]", [rb], None, n, sub, seed, 0, n_steps=3)
    _sweep_case(gpu, f"sweep fast 3 steps chained[{sub},{seed}]", [rb], None, n, sub, seed, nat.OPT_CHAINED, n_steps=3)
    # k_step_hexa<NOISE, NT, S1, ACT>
    _sweep_case(gpu, f"sweep hexa plain[{sub},{seed}]", [hx], None, n, sub, seed, 0)
    _sweep_case(gpu, f"sweep hexa action[{sub},{seed}]", [hx], None, n, sub, seed, 0, action=act6)
    # hexa_6DOF_simple (morphing-hexa physics, the quad law on six actuators): k_step_run<2, ..> for a fleet of it and for
    # runs that hold it (one launch per run), the general kernel with an explicit action
    hs = params.builtin_type("hexa_6DOF_simple")
    _sweep_case(gpu, f"sweep hexa_simple plain[{sub},{seed}]", [hs], None, n, sub, seed, 0)
    _sweep_case(gpu, f"sweep hexa_simple action[{sub},{seed}]", [hs], None, n, sub, seed, 0, action=act6)
    tid3 = np.array([0] * 150 + [1] * 170 + [2] * 192, dtype=np.uint8)
    _sweep_case(gpu, f"sweep runs of three kinds[{sub},{seed}]", [rb, hx, hs], tid3, n, sub, seed, 0,
                runs=[(0, 150, 0), (150, 170, 1), (320, 192, 2)])
    _sweep_case(gpu, f"sweep three kinds per lane[{sub},{seed}]", [rb, hx, hs], (np.arange(n) % 3).astype(np.uint8), n, sub, seed, 0)
    # type-major runs: k_step_runs<NOISE, NT, S1, ACT> (all runs in one launch); aligned runs, and runs that begin and end
    # inside tiles.  k_step_run<HEXA, NOISE, NT, S1> (one launch per run) serves a fleet that is ONE run: a homogeneous fleet
    # with the downwash force as input, and more runs than one launch holds (nine runs of two types here)
    tid = np.repeat(np.array([0, 1], dtype=np.uint8), 256)
    tid2 = np.array([0] * 200 + [1] * 312, dtype=np.uint8)
    _sweep_case(gpu, f"sweep runs one launch[{sub},{seed}]", [rb, hx], tid, n, sub, seed, 0, runs=[(0, 256, 0), (256, 256, 1)])
    _sweep_case(gpu, f"sweep runs sharing a tile, one launch[{sub},{seed}]", [rb, hx], tid2, n, sub, seed, 0,
                runs=[(0, 200, 0), (200, 312, 1)])
    cuts = [0, 50, 120, 180, 256, 300, 350, 420, 470, 512]
    tid9 = np.concatenate([np.full(b - a_, k % 2, dtype=np.uint8) for k, (a_, b) in enumerate(zip(cuts[:-1], cuts[1:]))])
    _sweep_case(gpu, f"sweep nine runs, one launch each[{sub},{seed}]", [rb, hx], tid9, n, sub, seed, 0,
                runs=[(a_, b - a_, k % 2) for k, (a_, b) in enumerate(zip(cuts[:-1], cuts[1:]))])


@pytest.mark.parametrize("seed", [0, 7])
@pytest.mark.parametrize("sub", [1, 2])
@pytest.mark.parametrize("n_types", [2, 3, 4])
def test_every_instance_of_the_mixed_fleet_kernels(gpu, n_types, sub, seed):
    """k_step_mixed4 / k_step_mixed3 <NOISE, NT, waves | types, S1, ...>: both forms the product ships, 2-4 types, on the
    layouts that select them."""
    import dataclasses
    nat, fleet = gpu
    rb, hx, te = params.builtin_type("robobee"), params.builtin_type("hexa_6DOF"), params.builtin_type("tello")
    types = [rb, hx, te, dataclasses.replace(rb, name="rb2", kp_pos=1.3, mass=0.8)][:n_types]
    n = 640
    tid = np.random.default_rng(n_types).integers(0, n_types, n).astype(np.uint8)
    for form, opt, layout in (("v4", 0, "tile64"), ("v3 soa", 0, "soa")):
        _sweep_case(gpu, f"sweep mixed {form}[{n_types},{sub},{seed}]", types, tid, n, sub, seed, opt, layout=layout)


@pytest.mark.parametrize("seed", [0, 7])
@pytest.mark.parametrize("sub", [1, 2])
@pytest.mark.parametrize("n_types", [2, 3, 4])
def test_every_binning_instance_of_the_mixed_kernel(gpu, n_types, sub, seed):
    """k_step_mixed4<.., BIN = true>: the step kernel that also fills the NEXT neighbour grid (dsim_step_args.bin_next).
    Same step as the oracle's, streaming on or off, and the downwash evaluated from the pre-binned grid equals the
    brute-force sum over the NEW positions."""
    import dataclasses
    nat, fleet = gpu
    rb, hx, te = params.builtin_type("robobee"), params.builtin_type("hexa_6DOF"), params.builtin_type("tello")
    types = [rb, hx, te, dataclasses.replace(rb, name="rb2", kp_pos=1.3, mass=0.8)][:n_types]
    n = 1024
    rng = np.random.default_rng(10 * n_types + sub)
    tid = rng.integers(0, n_types, n).astype(np.uint8)
    rigid, mem, tgt = random_fleet(rng, n, n_act=6, tilt=0.3, rate=1.0)
    rigid[:, 0] = f32(rng.uniform(1, 79, n)); rigid[:, 1] = f32(rng.uniform(1, 59, n)); rigid[:, 2] = f32(rng.uniform(1, 15, n))
    for k, t in enumerate(types):
        mem[tid == k, 7 + t.n_act:13] = 0.0
    dtc = float(np.float32(sub / 240))
    O = orc.Oracle(types)
    got = {}
    for pol in (nat.OPT_STREAM_ON, nat.OPT_STREAM_OFF):
        ctx = fleet.Context(types)
        st, tg = fleet.FleetState(ctx, n, "tile64"), fleet.Targets(ctx, n, "tile64")
        st.load_aos(rigid, mem)
        tg.set_fields(0, torch.from_numpy(np.ascontiguousarray(tgt.T)))
        tdev = torch.zeros(st.n_pad, dtype=torch.uint8, device=ctx.device); tdev[:n] = torch.from_numpy(tid)
        g = nat.DownwashArgs()
        nx, ny = 16, 12
        assert ctx.lib.dsim_downwash_prebin_ok(n, nx, ny) == 1
        ws = torch.empty((ctx.lib.dsim_downwash_workspace(n, nx, ny),), dtype=torch.int32, device=ctx.device)
        g.pos_all, g.m, g.m_pad = None, n, n
        g.xmin, g.ymin, g.cell, g.nx, g.ny = 0.0, 0.0, 5.0, nx, ny
        g.workspace, g.workspace_len, g.type_id, g.local_offset = ws.data_ptr(), ws.numel(), tdev.data_ptr(), 0
        force = torch.zeros((3, st.n_pad), device=ctx.device)
        s_ = _stream(ctx)
        nat.check(ctx.lib.dsim_downwash(ctx.handle, s_, n, st.view(), ctypes.byref(g), force.data_ptr()))   # builds grid 0, zeroes grid 1
        f0 = force.cpu().numpy()[2, :n].astype(np.float64)
        a = _args(nat, sub, DT, dtc, options=pol, seed=seed, step_index=2, type_id=tdev)
        a.ext_force, a.bin_next = force.data_ptr(), ctypes.addressof(g)
        nat.check(ctx.lib.dsim_step(ctx.handle, s_, n, st.view(), tg.view(), ctypes.byref(a)))
        g.prebinned = 1
        force2 = torch.zeros((3, st.n_pad), device=ctx.device)
        nat.check(ctx.lib.dsim_downwash(ctx.handle, s_, n, st.view(), ctypes.byref(g), force2.data_ptr()))
        got[pol] = (st.rigid_aos(), st.mem_aos(), force2.cpu().numpy()[2, :n].astype(np.float64))
        ctx.close()
    on, off = got[nat.OPT_STREAM_ON], got[nat.OPT_STREAM_OFF]
    # (not bitwise here: the bucket order, hence the summation order of the force that enters the step, follows an atomic scatter)
    np.testing.assert_allclose(on[0], off[0], rtol=2e-6, atol=1e-7); np.testing.assert_allclose(on[1], off[1], rtol=2e-5, atol=2e-6)
    # the step itself, with the downwash force of the old positions held over the sub-steps
    ref0 = O.downwash(rigid, rigid[:, 0:3], type_id=tid)
    assert_downwash("downwash binning sweep", f0, ref0, types, tid, rigid[:, 0:3], rigid[:, 0:3])
    r, m = rigid.copy(), mem.copy()
    nz = _noise_block(O, types, tid, n, seed, 2, sub) if seed else None
    ext = np.zeros((n, 3)); ext[:, 2] = f32(f0)
    assert O.step(r, m, tgt, sub, DT, dtc, noise=nz, type_id=tid, ext_force=ext) == 0
    assert_step_parity(f"sweep mixed v4 binning[{n_types},{sub},{seed}]", types, tid, rigid, mem, tgt, off[0], off[1], r, m, DT, dtc, sub)
    # the grid the step kernel filled: the force from it == brute force over the device's new positions
    ref1 = O.downwash(off[0], off[0][:, 0:3], type_id=tid)
    assert_downwash("downwash binning sweep, prebinned", off[2], ref1, types, tid, off[0][:, 0:3], off[0][:, 0:3])
    assert (ref1 < 0).sum() > n // 4


@pytest.mark.parametrize("seed", [0, 7])
def test_every_instance_of_the_two_call_adaptor_and_general_kernels(gpu, seed):
    """What the sweeps above leave: the streaming instances of k_physics_fast / k_control_fast, and the general and
    adaptor kernels on a fleet of TWO QUAD types (per-lane types with four actuators: <.., UNIFORM = false, 4>), with
    and without the plane.  Streaming on/off bitwise; each against the oracle."""
    import dataclasses
    nat, fleet = gpu
    rb = params.builtin_type("robobee")
    rb2 = dataclasses.replace(rb, name="rb2", kp_pos=1.3, mass=0.8)
    te = params.builtin_type("tello")
    dtc = float(np.float32(2 / 240))

    def run(types, tid, n, fn, layout="tile64", near_ground=False):
        rigid, mem, tgt = random_fleet(np.random.default_rng(n + seed), n, n_act=4, tilt=0.3, rate=1.0)
        if near_ground:
            rigid[:, 2] = f32(np.array([types[0 if tid is None else int(k)].rest_height for k in (tid if tid is not None else np.zeros(n))])
                              + np.random.default_rng(3).uniform(-1e-3, 0.04, n))
            rigid[:, 9] = -np.abs(rigid[:, 9])
        mem[:, 11:13] = 0.0
        out = {}
        for pol in (nat.OPT_STREAM_ON, nat.OPT_STREAM_OFF):
            ctx = fleet.Context(types)
            st, tg = fleet.FleetState(ctx, n, layout), fleet.Targets(ctx, n, layout)
            st.load_aos(rigid, mem)
            tg.set_fields(0, torch.from_numpy(np.ascontiguousarray(tgt.T)))
            tdev = None
            if tid is not None:
                tdev = torch.zeros(st.n_pad, dtype=torch.uint8, device=ctx.device); tdev[:n] = torch.from_numpy(tid)
            out[pol] = fn(ctx, st, tg, tdev, pol)
            ctx.close()
        for x, y in zip(out[nat.OPT_STREAM_ON], out[nat.OPT_STREAM_OFF]):
            np.testing.assert_array_equal(x, y)
        return rigid, mem, tgt, out[nat.OPT_STREAM_OFF]

    O1 = orc.Oracle([rb])
    n = 512
    act = f32(np.random.default_rng(9).uniform(0.3, 0.7, (n, 4)))

    # ---- k_physics_fast<NOISE, NT, OBS> and k_control_fast<NT, WANT_YAW> (homogeneous quads, whole tiles); arows: the
    # action as the [N, 4] array Env.step is handed (DSIM_OPT_ACTION_ROWS)
    for with_obs, arows in ((False, False), (True, False), (False, True), (True, True)):
        def phys(ctx, st, tg, tdev, pol):
            if arows:
                adev = torch.from_numpy(act.astype(np.float32)).to(ctx.device).contiguous()
                pol = pol | nat.OPT_ACTION_ROWS
            else:
                adev = torch.zeros((4, st.n_pad), device=ctx.device); adev[:, :n] = torch.from_numpy(np.ascontiguousarray(act.T)).float()
            echo = torch.zeros((4, st.n_pad), device=ctx.device)
            obs = torch.zeros((n, 20), device=ctx.device)
            a = _args(nat, 2, DT, dtc, options=pol, seed=seed, step_index=1, action=adev)
            if with_obs:
                a.obs_out, a.obs_width = obs.data_ptr(), 20
            nat.check(ctx.lib.dsim_physics(ctx.handle, _stream(ctx), n, st.view(), echo.data_ptr(), ctypes.byref(a)))
            return st.rigid_aos(), echo.cpu().numpy(), obs.cpu().numpy()
        rigid, mem, tgt, (gr, ge, go) = run([rb], None, n, phys)
        r = rigid.copy()
        a6 = np.zeros((n, 6)); a6[:, :4] = act
        O1.physics(r, mem, 2, DT, action=a6, noise=_noise_block(O1, [rb], None, n, seed, 1, 2) if seed else None)
        assert_step_parity(f"sweep physics_fast[{with_obs},{arows},{seed}]", [rb], None, rigid, mem, tgt, gr, None, r, None, DT, dtc, 2,
                           control=False, action=act)
        np.testing.assert_array_equal(ge[:, :n].T, act)
        if with_obs:
            np.testing.assert_array_equal(go[:, 0:7], gr[:, 0:7].astype(np.float32))
    for want_yaw in (False, True):
        def ctrl(ctx, st, tg, tdev, pol):
            pe = torch.zeros((3, st.n_pad), device=ctx.device); ye = torch.zeros((st.n_pad,), device=ctx.device)
            cmd = torch.zeros((4, st.n_pad), device=ctx.device)
            a = _args(nat, 0, dtc, dtc, options=pol)
            nat.check(ctx.lib.dsim_control2(ctx.handle, _stream(ctx), n, st.view(), tg.view(), ctypes.byref(a), pe.data_ptr(),
                                            ye.data_ptr() if want_yaw else None, cmd.data_ptr()))
            return st.mem_aos(), pe.cpu().numpy(), cmd.cpu().numpy()
        rigid, mem, tgt, (gm, gpe, gcmd) = run([rb], None, n, ctrl)
        m = mem.copy()
        rc, pe, ye = O1.control(rigid, m, tgt, dtc)
        assert rc == 0
        assert_control_parity(f"sweep control_fast[{want_yaw}]", [rb], None, rigid, mem, tgt, gm, m, dtc)
        np.testing.assert_array_equal(gcmd[:, :n].T, gm[:, 7:11].astype(np.float32))

    # ---- two quad types, per-lane: general step (explicit action), lean step (a quad-only mixed table takes it), Env.step, control,
    # the adaptors — and the same over the plane
    types = [rb, te]
    n2 = 600
    tid = (np.arange(n2) % 2).astype(np.uint8)
    O2 = orc.Oracle(types)
    act2 = f32(np.random.default_rng(8).uniform(0.3, 0.7, (n2, 4)))
    from tests.util import plane_terms
    for plane in (0, nat.OPT_PLANE):
        kw = {}
        if plane:
            kw = dict(near_ground=True)
        k_bar = K_ULP * 2 * (1 + PLANE_SWEEPS) if plane else None

        def step_action(ctx, st, tg, tdev, pol):
            adev = torch.zeros((4, st.n_pad), device=ctx.device); adev[:, :n2] = torch.from_numpy(np.ascontiguousarray(act2.T)).float()
            a = _args(nat, 2, DT, dtc, options=pol | plane, seed=seed, step_index=1, type_id=tdev, action=adev)
            nat.check(ctx.lib.dsim_step(ctx.handle, _stream(ctx), n2, st.view(), tg.view(), ctypes.byref(a)))
            return st.rigid_aos(), st.mem_aos()
        rigid, mem, tgt, (gr, gm) = run(types, tid, n2, step_action, **kw)
        r, m = rigid.copy(), mem.copy()
        a6 = np.zeros((n2, 6)); a6[:, :4] = act2
        nz = _noise_block(O2, types, tid, n2, seed, 1, 2) if seed else None
        assert O2.step(r, m, tgt, 2, DT, dtc, noise=nz, type_id=tid, action=a6, options=plane) == 0
        assert_step_parity(f"sweep gen two quads[{plane},{seed}]", types, tid, rigid, mem, tgt, gr, gm, r, m, DT, dtc, 2, action=act2,
                           k=k_bar, extra_terms=plane_terms(types, tid, rigid, dtc) if plane else None)

        def phys2(ctx, st, tg, tdev, pol):
            adev = torch.zeros((4, st.n_pad), device=ctx.device); adev[:, :n2] = torch.from_numpy(np.ascontiguousarray(act2.T)).float()
            echo = torch.zeros((4, st.n_pad), device=ctx.device)
            a = _args(nat, 2, DT, dtc, options=pol | plane, seed=seed, step_index=1, type_id=tdev, action=adev)
            nat.check(ctx.lib.dsim_physics(ctx.handle, _stream(ctx), n2, st.view(), echo.data_ptr(), ctypes.byref(a)))
            return (st.rigid_aos(),)
        rigid, mem, tgt, (gr,) = run(types, tid, n2, phys2, **kw)
        r = rigid.copy()
        O2.physics(r, mem, 2, DT, action=a6, noise=nz, type_id=tid, options=plane)
        assert_step_parity(f"sweep physics two quads[{plane},{seed}]", types, tid, rigid, mem, tgt, gr, None, r, None, DT, dtc, 2,
                           control=False, action=act2, k=k_bar, extra_terms=plane_terms(types, tid, rigid, dtc) if plane else None)

        for mode, mname in ((nat.ADAPT_VELOCITY, "velocity"), (nat.ADAPT_RPYT, "rpyt")):
            av = f32(np.concatenate([np.random.default_rng(6).uniform(-1, 1, (n2, 3)), np.full((n2, 1), 0.25)], 1))

            def adapt(ctx, st, tg, tdev, pol):
                adev = torch.zeros((4, st.n_pad), device=ctx.device); adev[:, :n2] = torch.from_numpy(np.ascontiguousarray(av.T)).float()
                echo = torch.zeros((4, st.n_pad), device=ctx.device)
                a = _args(nat, 2, DT, dtc, options=pol | plane, seed=seed, step_index=1, type_id=tdev)
                nat.check(ctx.lib.dsim_step_adaptor(ctx.handle, _stream(ctx), n2, st.view(), adev.data_ptr(), mode, echo.data_ptr(),
                                                    ctypes.byref(a)))
                return st.rigid_aos(), st.mem_aos()
            rigid, mem, tgt, (gr, gm) = run(types, tid, n2, adapt, **kw)
            # the law (on the state before the physics), then the physics with the command the DEVICE computed
            rc0, m = rigid.copy(), mem.copy()
            assert O2.adaptor_step(0 if mode == nat.ADAPT_VELOCITY else 1, rc0, m, av, 0, DT, dtc, type_id=tid) == 0
            r = rigid.copy()
            c6 = np.zeros((n2, 6)); c6[:, :4] = gm[:, 7:11]
            O2.physics(r, gm.copy(), 2, DT, action=c6, noise=nz, type_id=tid, options=plane)
            assert_step_parity(f"sweep adaptor two quads[{mname},{plane},{seed}]", types, tid, rigid, gm, tgt, gr, None, r, None, DT, dtc, 2,
                               control=False, action=gm[:, 7:11], k=k_bar,
                               extra_terms=plane_terms(types, tid, rigid, dtc) if plane else None)
            assert np.abs(gm[:, 7:11] - m[:, 7:11]).max() < 1e-3

    def lean(ctx, st, tg, tdev, pol):
        a = _args(nat, 2, DT, dtc, options=pol, seed=seed, step_index=1, type_id=tdev)
        nat.check(ctx.lib.dsim_step(ctx.handle, _stream(ctx), n2, st.view(), tg.view(), ctypes.byref(a)))
        return st.rigid_aos(), st.mem_aos()
    rigid, mem, tgt, (gr, gm) = run(types, tid, n2, lean)
    r, m = rigid.copy(), mem.copy()
    assert O2.step(r, m, tgt, 2, DT, dtc, noise=_noise_block(O2, types, tid, n2, seed, 1, 2) if seed else None, type_id=tid) == 0
    assert_step_parity(f"sweep lean two quads[{seed}]", types, tid, rigid, mem, tgt, gr, gm, r, m, DT, dtc, 2)

    def ctrl2(ctx, st, tg, tdev, pol):
        pe = torch.zeros((3, st.n_pad), device=ctx.device); ye = torch.zeros((st.n_pad,), device=ctx.device)
        a = _args(nat, 0, dtc, dtc, options=pol, type_id=tdev)
        nat.check(ctx.lib.dsim_control(ctx.handle, _stream(ctx), n2, st.view(), tg.view(), ctypes.byref(a), pe.data_ptr(), ye.data_ptr()))
        return (st.mem_aos(),)
    rigid, mem, tgt, (gm,) = run(types, tid, n2, ctrl2)
    m = mem.copy()
    assert O2.control(rigid, m, tgt, dtc, type_id=tid)[0] == 0
    assert_control_parity("sweep control two quads", types, tid, rigid, mem, tgt, gm, m, dtc)


@pytest.mark.parametrize("seed", [0, 7])
def test_every_instance_of_the_tail_kernels(gpu, seed):
    """Ragged fleets (n_pad not a multiple of 256: the general kernels serve the tail, or everything below one tile) of
    one type and of quads + hexas, plain and with an explicit action, and computeControl on them."""
    import dataclasses
    nat, fleet = gpu
    rb, hx = params.builtin_type("robobee"), params.builtin_type("hexa_6DOF")
    n = 300                                                       # n_pad = 320: one whole tile + a 64-drone tail
    act4 = f32(np.random.default_rng(1).uniform(0.3, 0.7, (n, 4)))
    act6 = f32(np.random.default_rng(2).uniform(0.3, 0.7, (n, 6)))
    tid = (np.arange(n) % 2).astype(np.uint8)
    for layout in ("soa", "tile64"):
        _sweep_case(gpu, f"sweep tail quad[{seed}]", [rb], None, n, 2, seed, 0, layout=layout, pad=64)
        _sweep_case(gpu, f"sweep tail quad action[{seed}]", [rb], None, n, 2, seed, 0, action=act4, n_steps=2, layout=layout, pad=64)
        _sweep_case(gpu, f"sweep tail hexa[{seed}]", [hx], None, n, 2, seed, 0, layout=layout, pad=64)
        _sweep_case(gpu, f"sweep tail hexa action[{seed}]", [hx], None, n, 2, seed, 0, action=act6, layout=layout, pad=64)
        five = [rb, hx] + [dataclasses.replace(rb, name=f"rb{k}", kp_pos=1.0 + 0.1 * k) for k in range(3)]
        _sweep_case(gpu, f"sweep generic mixed, five types[{seed}]", five, (np.arange(n) % 5).astype(np.uint8), n, 2, seed, 0,
                    layout=layout, pad=64)
        _sweep_case(gpu, f"sweep mixed action[{seed}]", [rb, hx], tid, n, 2, seed, 0, action=act6, layout=layout, pad=64)
    if seed:
        return
    # Env.step over the plane on the mixed fleet, no noise; computeControl on the ragged quad fleet
    from tests.util import plane_terms
    types = [rb, hx]
    rigid, mem, tgt = random_fleet(np.random.default_rng(4), n, n_act=6, tilt=0.3)
    rigid[:, 2] = f32(np.where(tid == 0, rb.rest_height, hx.rest_height) + np.random.default_rng(5).uniform(-1e-3, 0.04, n))
    mem[tid == 0, 11:13] = 0.0
    dtc = float(np.float32(2 / 240))
    ctx = fleet.Context(types)
    st = fleet.FleetState(ctx, n, "soa", 64)
    st.load_aos(rigid, mem)
    tdev = torch.zeros(st.n_pad, dtype=torch.uint8, device=ctx.device); tdev[:n] = torch.from_numpy(tid)
    adev = torch.zeros((6, st.n_pad), device=ctx.device); adev[:, :n] = torch.from_numpy(np.ascontiguousarray(act6.T)).float()
    echo = torch.zeros((6, st.n_pad), device=ctx.device)
    a = _args(nat, 2, DT, dtc, options=nat.OPT_PLANE, type_id=tdev, action=adev)
    nat.check(ctx.lib.dsim_physics(ctx.handle, _stream(ctx), n, st.view(), echo.data_ptr(), ctypes.byref(a)))
    r = rigid.copy()
    a6 = act6.astype(np.float64).copy(); a6[tid == 0, 4:] = 0.0
    orc.Oracle(types).physics(r, mem, 2, DT, action=a6, type_id=tid, options=nat.OPT_PLANE)
    assert_step_parity("sweep physics plane mixed", types, tid, rigid, mem, tgt, st.rigid_aos(), None, r, None, DT, dtc, 2, control=False,
                       action=a6, k=K_ULP * 2 * (1 + PLANE_SWEEPS), extra_terms=plane_terms(types, tid, rigid, dtc))
    ctx.close()
    rigid, mem, tgt = random_fleet(np.random.default_rng(6), n, tilt=0.4)
    ctx = fleet.Context([rb])
    st, tg = fleet.FleetState(ctx, n, "soa", 64), fleet.Targets(ctx, n, "soa", pad=64)
    st.load_aos(rigid, mem)
    tg.set_fields(0, torch.from_numpy(np.ascontiguousarray(tgt.T)))
    pe = torch.zeros((3, st.n_pad), device=ctx.device); ye = torch.zeros((st.n_pad,), device=ctx.device)
    a = _args(nat, 0, dtc, dtc)
    nat.check(ctx.lib.dsim_control(ctx.handle, _stream(ctx), n, st.view(), tg.view(), ctypes.byref(a), pe.data_ptr(), ye.data_ptr()))
    m = mem.copy()
    assert orc.Oracle([rb]).control(rigid, m, tgt, dtc)[0] == 0
    assert_control_parity("sweep control ragged quad", [rb], None, rigid, mem, tgt, st.mem_aos(), m, dtc)
    ctx.close()
