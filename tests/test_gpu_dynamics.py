"""Row D1 on the device: Physics.DYN — BaseAviary._dynamics (BaseAviary.py:1767-1828) inside the step() loop
(:510-547) — through the C-ABI (dsim_physics / dsim_step with DSIM_OPT_DYN) and through the host surfaces
(CtrlAviary(physics=Physics.DYN)), against

  * the oracle (oracle/dsim_oracle.c:orc_dynamics, orc_dyn_physics_batch, orc_dyn_step_batch), itself equal to the
    reference's own functions at 1e-12 (tests/test_oracle_dynamics.py), from the same fp32-representable inputs, judged on
    the INCREMENT of every field per launch (tests/util.py): |d_gpu - d_oracle| <= 1e-4 |d_oracle| + k ulp32(M);
  * tests/golden/dynamics.npz directly: outputs of the reference's functions (single calls, and 240 / 120-sub-step
    flights through the reference's own BaseAviary.step loop).

k: the model re-reads the Euler angles from the quaternion and rebuilds the quaternion from the summed angles EVERY sub-step
(:729, 1812, 1817) — two atan2, an asin and three sincos where the Bullet step has one polynomial — so the quaternion
fields get K_DYN roundings per sub-step instead of K_ULP.
"""
import ctypes
import math
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from dronesim_amd import params  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from tests.util import (K_ULP, TINY, WORST, f32, increment_ratio, random_fleet, step_terms, tilt_gain, ulp32)  # noqa: E402

pytestmark = pytest.mark.gpu

DT = float(np.float32(1.0 / 240.0))
K_DYN = 4.0


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device; the product has no CPU fallback")
    from dronesim_amd import _native as nat
    from dronesim_amd import fleet
    return nat, fleet


def _dyn_terms(types, tid, rigid, rates, mem, tgt, dt, dt_ctrl, sub, control, action):
    """Magnitudes entering the fp32 update of every field ([n,13] rigid, [n,3] rates, [n,13] mem): util.step_terms'
    (thrust and weight cancel in the velocity, the rotor moments pairwise in the rates), the rates in place of the
    angular velocity, and the gyroscopic term rates x J rates / J_min."""
    tr, tm = step_terms(types, tid, rigid, mem, tgt, dt, dt_ctrl, sub, control, action)
    n = rigid.shape[0]
    t_idx = np.zeros(n, dtype=np.int64) if tid is None else np.asarray(tid).astype(np.int64)
    jr = np.array([max(t.inertia) / min(t.inertia) for t in types])[t_idx]
    w = np.abs(rates).max(1)
    trr = tr[:, 10:13] + (jr * w * w * dt)[:, None]
    tr = tr.copy()
    tr[:, 10:13] = TINY        # ang_v is a placeholder or R rates (compared separately)
    return tr, trr, tm


def _assert_dyn(label, types, tid, prev_rigid, prev_rates, prev_mem, tgt, got, ref, dt, dt_ctrl, sub, control=False, action=None,
                body_rates=False):
    """got / ref = (rigid, rates, mem | None).  Per field: |got - ref| <= 1e-4 |increment| + k ulp32(M), with two things that
    are particular to this model:
      * the quaternion is one object: its four components are judged against the norm of the quaternion's increment;
      * the model re-reads pitch = asin(sarg) from the quaternion every sub-step (BaseAviary.py:729): half an ulp of sarg is
        ulp32(1) / (2 cos pitch) in pitch, and half of that in the rebuilt quaternion — the conditioning of the REFERENCE's
        own formulation (in fp64 it amplifies 1e-16 the same way), charged per drone: k (1 + 0.5 / |cos pitch|) on the
        quaternion fields, and through the thrust direction (|acc| dt, |acc| dt^2) on velocity and position."""
    k = K_DYN * max(1, sub)
    tr, trr, tm = _dyn_terms(types, tid, prev_rigid, prev_rates, prev_mem, tgt, dt, dt_ctrl, max(1, sub), control, action)
    cosp = np.minimum(_cos_pitch(prev_rigid[:, 3:7]), _cos_pitch(ref[0][:, 3:7]))
    cond = 1.0 + 0.5 / np.maximum(cosp, 1e-3)
    kq = np.ones((prev_rigid.shape[0], 10)) * k
    kq[:, 3:7] *= cond[:, None]
    d_ref = ref[0][:, :10] - prev_rigid[:, :10]
    rel = np.abs(d_ref)
    rel[:, 3:7] = np.linalg.norm(d_ref[:, 3:7], axis=1, keepdims=True)
    M = np.maximum(np.maximum(np.maximum(np.abs(prev_rigid[:, :10]), np.abs(ref[0][:, :10])), tr[:, :10]), TINY)
    tol = 1e-4 * rel + kq * ulp32(M)
    q_err = k * (cond - 1.0) * ulp32(1.0)                                     # what the pitch conditioning leaves in the quaternion
    tol[:, 7:10] += (q_err * tr[:, 7] * max(1, sub))[:, None]                # ... turns the thrust: |acc| dt per sub-step
    tol[:, 0:3] += (q_err * tr[:, 7] * max(1, sub) * dt * max(1, sub))[:, None]
    parts = {"rigid": np.abs(got[0][:, :10] - ref[0][:, :10]) / tol,
             "rates": increment_ratio(got[1], ref[1], prev_rates, trr, k)}
    if body_rates:        # R(quat) rates: three products of the rates (known to k ulps) with entries of R(quat) — quadratic in a
        # quaternion known to k cond ulp32(1): 2 k cond ulp32(1) |rates| — and their fp32 sum
        Mw = np.maximum(np.abs(ref[1]).max(1, keepdims=True), TINY)
        parts["ang_v"] = np.abs(got[0][:, 10:13] - ref[0][:, 10:13]) / (
            1e-4 * np.abs(ref[0][:, 10:13]) + (k * cond)[:, None] * (2.0 * ulp32(1.0) * Mw + ulp32(Mw)))
    else:
        np.testing.assert_array_equal(got[0][:, 10:13], ref[0][:, 10:13])      # the placeholder, bit for bit
    if control:
        kk = K_ULP * max(1, sub) * tilt_gain(types, tid, ref[0]) * (K_DYN / K_ULP) * cond[:, None]
        # (the law differentiates the reported angular velocity and the new velocity: their own k ulps enter through 1 / dt_ctrl,
        # which is what step_terms' rate / acceleration terms carry)
        parts["mem"] = increment_ratio(got[2], ref[2], prev_mem, tm, kk)
    worst, which = max((float(v.max()), name) for name, v in parts.items())
    if worst > 1.0:
        v = parts[which]
        i, f = (int(x) for x in np.unravel_index(v.argmax(), v.shape))
        print(label, "worst", worst, which, "drone", i, "field", f, "\nprev", prev_rigid[i], prev_rates[i], prev_mem[i], "\ngot", got[0][i], got[1][i],
              None if got[2] is None else got[2][i], "\nref", ref[0][i], ref[1][i], None if ref[2] is None else ref[2][i],
              "\nrpy", orc.euler_from_quat(prev_rigid[i, 3:7]), "cond", cond[i])
    WORST[label] = max(WORST.get(label, 0.0), worst)
    assert worst <= 1.0, (label, worst, which)
    return worst


def _cos_pitch(q):
    sarg = -2.0 * (q[:, 0] * q[:, 2] - q[:, 3] * q[:, 1]) / np.maximum((q * q).sum(1), 1e-300)
    return np.sqrt(np.maximum(1.0 - np.minimum(sarg * sarg, 1.0), 0.0))


def _args(nat, rates, sub, options, dt_ctrl=None, action=None, type_id=None):
    a = nat.StepArgs()
    a.phys_substeps, a.dt_phys, a.dt_ctrl = sub, DT, DT * max(1, sub) if dt_ctrl is None else dt_ctrl
    a.options = nat.OPT_DYN | options
    a.noise_seed = 12345                  # ignored: the model has no noise
    a.dyn_rpy_rates = rates.data_ptr()
    a.action = action.data_ptr() if action is not None else None
    a.type_id = type_id.data_ptr() if type_id is not None else None
    return a


def _soa(a, n_pad, dev):
    t = torch.zeros((a.shape[1], n_pad), dtype=torch.float32)
    t[:, : a.shape[0]] = torch.from_numpy(np.ascontiguousarray(a.T)).float()
    return t.to(dev)


@pytest.mark.parametrize("tag", ["robobee_x", "tello_x", "robobee_plus", "tello_hb"])
def test_dyn_kernel_vs_reference_single_calls(gpu, golden_dir, tag):
    """One sub-step of Env.step on Physics.DYN for the golden's seeded states: the kernel against the oracle on the same
    fp32-rounded inputs (the bar), and against what the REFERENCE's function produced on the fp64 inputs (the bar plus the
    measured effect of rounding the inputs, per case and field)."""
    nat, fleet = gpu
    G = np.load(os.path.join(golden_dir, "dynamics.npz"))
    g = lambda k: G[f"{tag}_{k}"]
    t = params.builtin_type(tag.split("_")[0])
    t.dyn_mixer = params.DYN_MIXER_PLUS if bool(g("mixer_plus")) else params.DYN_MIXER_X
    keep = np.array([i for i in range(g("pos").shape[0]) if i != 3])          # case 3: a quat / rpy pair the engine never holds
    n = len(keep)
    ctx = fleet.Context([t])
    O = orc.Oracle([t])
    for layout in ("soa", "tile64"):
        st = fleet.FleetState(ctx, n, layout)
        rigid = f32(np.concatenate([g("pos")[keep], g("quat")[keep], g("vel")[keep], np.zeros((n, 3))], 1))
        rates0 = f32(g("rates")[keep])
        mem = O.reset_mem(n)
        st.load_aos(rigid, mem)
        rates = _soa(rates0, st.n_pad, ctx.device)
        pwm = f32(g("pwm")[keep])
        act = _soa(np.concatenate([pwm, np.zeros((n, 2))], 1)[:, :4], st.n_pad, ctx.device)
        echo = torch.zeros((4, st.n_pad), device=ctx.device)
        # (the streaming instances, k_dyn<.., NT = true, ..>, are what a large fleet runs: forced here on the tags of one mixer)
        a = _args(nat, rates, 1, nat.OPT_STREAM_ON if tag.endswith("_x") else nat.OPT_STREAM_OFF, action=act)
        nat.check(ctx.lib.dsim_physics(ctx.handle, ctx.stream_ptr(), n, st.view(), echo.data_ptr(), ctypes.byref(a)))
        torch.cuda.synchronize()
        got_r, got_w = st.rigid_aos(), rates[:, :n].T.double().cpu().numpy()
        ref_r, ref_w = rigid.copy(), rates0.copy()
        act6 = np.zeros((n, 6)); act6[:, :4] = pwm
        assert O.dyn_physics(ref_r, ref_w, mem, 1, DT, action=act6) == 0
        _assert_dyn(f"dyn_single[{tag},{layout}] vs oracle(fp32 in)", [t], None, rigid, rates0, mem, np.zeros((1, 10)),
                    (got_r, got_w, None), (ref_r, ref_w, None), DT, DT, 1, action=act6[:, :4])
        np.testing.assert_array_equal(echo[:, :n].T.cpu().numpy(), pwm.astype(np.float32))
        # against the reference's own outputs (fp64 inputs, dt = 1/240 exactly): slack = |oracle(fp32 in) - golden|
        gold = np.concatenate([g("pos_out")[keep], g("quat_out")[keep], g("vel_out")[keep]], 1)
        slack = np.abs(ref_r[:, :10] - gold)
        tr, trr, _ = _dyn_terms([t], None, rigid, rates0, mem, np.zeros((1, 10)), DT, DT, 1, False, act6[:, :4])
        M = np.maximum(np.maximum(np.maximum(np.abs(rigid[:, :10]), np.abs(gold)), tr[:, :10]), TINY)
        tol = 1e-4 * np.abs(gold - rigid[:, :10]) + K_DYN * ulp32(M) + slack
        assert (np.abs(got_r[:, :10] - gold) <= tol).all(), (np.abs(got_r[:, :10] - gold) / tol).max()
        Mw = np.maximum(np.maximum(np.maximum(np.abs(rates0), np.abs(g("rates_out")[keep])), trr), TINY)
        tolw = 1e-4 * np.abs(g("rates_out")[keep] - rates0) + K_DYN * ulp32(Mw) + np.abs(ref_w - g("rates_out")[keep])
        assert (np.abs(got_w - g("rates_out")[keep]) <= tolw).all()
        np.testing.assert_array_equal(got_r[:, 10:13], g("ang_v_out")[keep])       # (-1, -1, -1)
    ctx.close()


@pytest.mark.parametrize("tag", ["flight5", "flight1"])
def test_dyn_flights_vs_reference_step_loop(gpu, golden_dir, tag):
    """The golden's flights through the reference's own BaseAviary.step loop (3 robobees, 240 / 120 sub-steps of
    manoeuvring at up to 3 rad/s): every device Env.step against the oracle from the device's own previous state (the bar),
    and the free-running device trajectory against the REFERENCE's recorded states (accumulated drift)."""
    nat, fleet = gpu
    G = np.load(os.path.join(golden_dir, "dynamics.npz"))
    aggr, init, pwm, states = int(G[f"{tag}_aggr"]), G[f"{tag}_init"], G[f"{tag}_pwm"], G[f"{tag}_states"]
    t = params.builtin_type("robobee")
    ctx = fleet.Context([t])
    O = orc.Oracle([t])
    n = init.shape[0]
    st = fleet.FleetState(ctx, n)
    rigid0 = f32(np.concatenate([init[:, 0:7], init[:, 10:16]], 1))
    mem = O.reset_mem(n)
    st.load_aos(rigid0, mem)
    rates = torch.zeros((3, st.n_pad), device=ctx.device)
    obs = torch.zeros((n, 20), device=ctx.device)
    echo = torch.zeros((4, st.n_pad), device=ctx.device)
    drift = 0.0
    for k in range(pwm.shape[0]):
        prev_r, prev_w = st.rigid_aos(), rates[:, :n].T.double().cpu().numpy()
        p32 = f32(pwm[k])
        act = _soa(p32, st.n_pad, ctx.device)
        a = _args(nat, rates, aggr, nat.OPT_STREAM_ON if k % 2 else nat.OPT_STREAM_OFF, action=act)      # (both cache policies: the same results)
        a.obs_out, a.obs_width = obs.data_ptr(), 20
        nat.check(ctx.lib.dsim_physics(ctx.handle, ctx.stream_ptr(), n, st.view(), echo.data_ptr(), ctypes.byref(a)))
        torch.cuda.synchronize()
        got_r, got_w = st.rigid_aos(), rates[:, :n].T.double().cpu().numpy()
        ref_r, ref_w = prev_r.copy(), prev_w.copy()
        act6 = np.zeros((n, 6)); act6[:, :4] = p32
        assert O.dyn_physics(ref_r, ref_w, mem, aggr, DT, action=act6) == 0
        _assert_dyn(f"dyn_flight[{tag}] per Env.step", [t], None, prev_r, prev_w, mem, np.zeros((1, 10)), (got_r, got_w, None),
                    (ref_r, ref_w, None), DT, DT * aggr, aggr, action=act6[:, :4])
        # the observation rows Env.step returns (BaseAviary.py:780-790): the new state, rpy of the stored quaternion, the
        # placeholder angular velocity, the clipped action echoed
        o = obs.double().cpu().numpy()
        np.testing.assert_array_equal(o[:, 0:7], got_r[:, 0:7]); np.testing.assert_array_equal(o[:, 10:13], got_r[:, 7:10])
        np.testing.assert_array_equal(o[:, 13:16], -1.0); np.testing.assert_array_equal(o[:, 16:20], p32)
        ref = states[k]
        rpy = np.stack([orc.euler_from_quat(q) for q in got_r[:, 3:7]])
        dq = np.minimum(np.abs(got_r[:, 3:7] - ref[:, 3:7]).max(1), np.abs(got_r[:, 3:7] + ref[:, 3:7]).max(1))     # q and -q: one attitude
        d = np.abs(np.concatenate([got_r[:, 0:3] - ref[:, 0:3], got_r[:, 7:10] - ref[:, 10:13], got_w - ref[:, 16:19], dq[:, None]], 1))
        drift = max(drift, float(d.max()))
        # the rows' rpy = fp32 Euler angles of the stored quaternion: roll and yaw are conditioned by 1 / cos(pitch) (:729)
        e = np.abs(o[:, 7:10] - rpy)
        e = np.minimum(e, 2 * math.pi - e)
        assert (e <= 4 * ulp32(4.0) * (1.0 + 1.0 / np.maximum(_cos_pitch(got_r[:, 3:7]), 1e-3))[:, None]).all()
    WORST[f"dyn_flight[{tag}] drift from the reference's states / 1e-3"] = drift / 1e-3
    assert drift < 1e-3, drift            # fp32 against the reference's fp64 over the whole flight (states of order 1-5)
    ctx.close()


@pytest.mark.parametrize("sub,layout,n,body", [(1, "soa", 1000, False), (5, "tile64", 4096, True), (2, "soa", 300, True)])
def test_dyn_fused_step_mixed_quad_types_vs_oracle(gpu, sub, layout, n, body):
    """dsim_step on Physics.DYN: Env.step then computeControl in one launch, robobees and tellos interleaved (per-lane
    type ids), ragged fleet sizes, per-drone targets; an explicit action for the physics part of the first step; both
    forms of the reported angular velocity."""
    nat, fleet = gpu
    types = [params.builtin_type("robobee"), params.builtin_type("tello")]
    types[1].dyn_mixer = params.DYN_MIXER_PLUS
    ctx = fleet.Context(types)
    O = orc.Oracle(types)
    rng = np.random.default_rng(3 + sub)
    rigid, mem, tgt = random_fleet(rng, n, tilt=0.6)
    tid = (rng.integers(0, 2, n)).astype(np.uint8)
    st = fleet.FleetState(ctx, n, layout, pad=64)
    tg = fleet.Targets(ctx, n, layout, pad=64)
    st.load_aos(rigid, mem)
    tg.set_fields(0, torch.from_numpy(np.ascontiguousarray(tgt.T)))
    tid_dev = torch.zeros(st.n_pad, dtype=torch.uint8, device=ctx.device); tid_dev[:n] = torch.from_numpy(tid)
    rates0 = f32(rng.uniform(-1.5, 1.5, (n, 3)))
    rates = _soa(rates0, st.n_pad, ctx.device)
    opt = nat.OPT_DYN_BODY_RATES if body else 0
    dt_ctrl = float(np.float32(DT * sub))
    for k in range(4):
        prev_r, prev_m, prev_w = st.rigid_aos(), st.mem_aos(), rates[:, :n].T.double().cpu().numpy()
        act = f32(rng.uniform(-0.1, 1.1, (n, 4))) if k == 0 else None
        act_dev = _soa(act, st.n_pad, ctx.device) if act is not None else None
        a = _args(nat, rates, sub, opt | (nat.OPT_STREAM_ON if k % 2 else nat.OPT_STREAM_OFF), dt_ctrl=dt_ctrl, action=act_dev, type_id=tid_dev)
        nat.check(ctx.lib.dsim_step(ctx.handle, ctx.stream_ptr(), n, st.view(), tg.view(), ctypes.byref(a)))
        torch.cuda.synchronize()
        got = (st.rigid_aos(), rates[:, :n].T.double().cpu().numpy(), st.mem_aos())
        ref_r, ref_w, ref_m = prev_r.copy(), prev_w.copy(), prev_m.copy()
        act6 = None
        if act is not None:
            act6 = np.zeros((n, 6)); act6[:, :4] = act
        assert O.dyn_step(ref_r, ref_w, ref_m, tgt, sub, DT, dt_ctrl, options=opt, type_id=tid, action=act6) == 0
        _assert_dyn(f"dyn_fused[{sub},{layout},{'body' if body else 'ref'}]", types, tid, prev_r, prev_w, prev_m, tgt, got,
                    (ref_r, ref_w, ref_m), DT, dt_ctrl, sub, control=True, action=None if act is None else act, body_rates=body)
    ctx.close()


def test_dyn_env_and_indi_close_the_loop_on_config1(gpu):
    """BASELINE configs[0] (examples/fly_INDI.py: one robobee from (0, 1, 0.5) to (0, 0, 0.5), yaw target 0.4 + k / 200,
    48 Hz control over 5 physics sub-steps, initial action 0.4) on Physics.DYN through the two reference-shaped calls
    obs = env.step(action); action = ctrl.computeControlFromState(obs).  With the reference's placeholder ang_v the INDI
    rate loop reads (-1, -1, -1) rad/s and the flight tumbles — reproduced against the oracle step by step while it is
    finite; with dyn_ang_vel="body_rates" the loop closes and the drone arrives."""
    from dronesim_amd.control import INDIControl
    from dronesim_amd.envs import CtrlAviary, Physics
    nat, _ = gpu
    t = params.builtin_type("robobee")
    O = orc.Oracle([t])
    for mode, steps in (("body_rates", 720), ("reference", 60)):
        env = CtrlAviary(["robobee"], 1, initial_xyzs=np.array([[0.0, 1.0, 0.5]]), physics=Physics.DYN, aggregate_phy_steps=5,
                         noise_seed=0, dyn_ang_vel=mode)
        ctrl = INDIControl("robobee", env=env)
        opt = nat.OPT_DYN_BODY_RATES if mode == "body_rates" else 0
        action = {"0": np.array([0.4, 0.4, 0.4, 0.4])}
        rigid = env.state.rigid_aos(); rates = np.zeros((1, 3)); mem = env.state.mem_aos()
        dt_ctrl = float(np.float32(5 / 240))
        worst_pos = 0.0
        for k in range(steps):
            obs, reward, done, info = env.step(action)
            assert reward == -1 and done is False and info == {"answer": 42}
            s = obs["0"]["state"]
            if mode == "reference":
                np.testing.assert_array_equal(s[13:16], -1.0)
            cmd, pos_e, yaw_e = ctrl.computeControlFromState(5 / 240, s, np.array([0.0, 0.0, 0.5]),
                                                             target_rpy=np.array([0.0, 0.0, 0.4 + k / 200]))
            action = {"0": cmd}
            # oracle from the device's previous state
            prev_r, prev_w, prev_m = rigid.copy(), rates.copy(), mem.copy()
            a6 = np.zeros((1, 6)); a6[0, :4] = f32(np.array([0.4] * 4)) if k == 0 else prev_m[0, 7:11]
            tgt = f32(np.array([[0, 0, 0.5, 0, 0, 0, 0, 0, 0, 0.4 + k / 200]]))
            ref_r, ref_w, ref_m = prev_r.copy(), prev_w.copy(), prev_m.copy()
            O.dyn_step(ref_r, ref_w, ref_m, tgt, 5, DT, dt_ctrl, options=opt, action=a6)
            rigid, mem = env.state.rigid_aos(), env.state.mem_aos()
            rates = env.rpy_rates.T.double().cpu().numpy()
            if np.isfinite(ref_r).all() and np.abs(ref_w).max() < 50:
                _assert_dyn(f"dyn_config1[{mode}]", [t], None, prev_r, prev_w, prev_m, tgt, (rigid, rates, mem), (ref_r, ref_w, ref_m),
                            DT, dt_ctrl, 5, control=True, action=a6[:, :4], body_rates=mode == "body_rates")
            worst_pos = max(worst_pos, float(np.abs(rigid[0, :3]).max()))
        if mode == "body_rates":
            assert np.abs(rigid[0, :3] - [0, 0, 0.5]).max() < 5e-3 and np.abs(rigid[0, 7:10]).max() < 5e-3    # arrived, at rest
            assert abs(mem[0, 7:11].mean() - t.hover_pwm) < 2e-3
        else:
            rpy = orc.euler_from_quat(rigid[0, 3:7])
            assert np.abs(rpy[:2]).max() > 0.5                                 # tumbling: the reference's behaviour
        env.close()


def test_dyn_fleet_env_fused_and_graph(gpu):
    """Fleet-sized use through the host class: 4 096 tellos on Physics.DYN (body rates), step_fused per control step and the
    same steps replayed from one captured hipGraph: bit-identical state, and the fleet holds its hover targets."""
    from dronesim_amd.envs import CtrlAviary, Physics
    from dronesim_amd.fleet import Targets
    n = 4096
    xyz = np.stack([np.arange(n) % 64, np.arange(n) // 64, np.full(n, 1.0)], 1).astype(np.float64)
    outs = []
    for use_graph in (False, True):
        env = CtrlAviary(["tello"], n, initial_xyzs=xyz, physics=Physics.DYN, aggregate_phy_steps=5, dyn_ang_vel="body_rates")
        tg = Targets(env.ctx, n)
        tg.set(pos=(xyz + [0.1, -0.1, 0.2]).T, yaw=np.full(n, 0.3))
        env.step_fused(tg, action=np.full((n, 4), 0.4))
        if use_graph:
            g = env.capture_fused(tg, 10)
            for _ in range(6):
                g.replay()
        else:
            for _ in range(60):
                env.step_fused(tg)
        torch.cuda.synchronize()
        outs.append((env.state.rigid_aos(), env.rpy_rates.cpu().numpy().copy()))
        env.close()
    np.testing.assert_array_equal(outs[0][0], outs[1][0]); np.testing.assert_array_equal(outs[0][1], outs[1][1])
    r = outs[0][0]
    assert np.isfinite(r).all() and np.abs(r[:, :3] - (xyz + [0.1, -0.1, 0.2])).max() < 0.5      # on its way, 1.27 s in


def test_dyn_refusals(gpu):
    """What the mode does not combine with is refused, never dropped (include/dronesim_amd.h, DSIM_OPT_DYN)."""
    nat, fleet = gpu
    from dronesim_amd.envs import CtrlAviary, Physics, VelocityAviary
    ctx = fleet.Context([params.builtin_type("robobee")])
    st = fleet.FleetState(ctx, 256)
    tg = fleet.Targets(ctx, 256)
    rates = torch.zeros((3, 256), device=ctx.device)
    a = _args(nat, rates, 1, 0)
    a.dyn_rpy_rates = None
    assert ctx.lib.dsim_physics(ctx.handle, ctx.stream_ptr(), 256, st.view(), None, ctypes.byref(a)) == -1        # DSIM_E_ARG
    for opt in (nat.OPT_DRAG, nat.OPT_GROUND, nat.OPT_PLANE, nat.OPT_CHAINED, nat.OPT_ACTION_ROWS):
        a = _args(nat, rates, 1, opt)
        assert ctx.lib.dsim_step(ctx.handle, ctx.stream_ptr(), 256, st.view(), tg.view(), ctypes.byref(a)) == -5   # DSIM_E_UNSUPPORTED
        assert ctx.lib.dsim_physics(ctx.handle, ctx.stream_ptr(), 256, st.view(), None, ctypes.byref(a)) == -5
    a = _args(nat, rates, 1, 0)
    a.ext_force = rates.data_ptr()
    assert ctx.lib.dsim_step(ctx.handle, ctx.stream_ptr(), 256, st.view(), tg.view(), ctypes.byref(a)) == -5
    act = torch.zeros((4, 256), device=ctx.device)
    a = _args(nat, rates, 1, 0)
    assert ctx.lib.dsim_step_adaptor(ctx.handle, ctx.stream_ptr(), 256, st.view(), act.data_ptr(), 0, None, ctypes.byref(a)) == -5
    ctx.close()
    hx = fleet.Context([params.builtin_type("hexa_6DOF")])
    sh = fleet.FleetState(hx, 256)
    a = _args(nat, rates, 1, 0)
    assert hx.lib.dsim_physics(hx.handle, hx.stream_ptr(), 256, sh.view(), None, ctypes.byref(a)) == -5
    hx.close()
    xyz = np.zeros((2, 3)); xyz[:, 2] = 1
    with pytest.raises(NotImplementedError):
        CtrlAviary(["hexa_6DOF"], 2, initial_xyzs=xyz, physics=Physics.DYN)
    with pytest.raises(ValueError):
        CtrlAviary(["robobee"], 2, initial_xyzs=xyz, physics=Physics.DYN, ground_plane=True)
    with pytest.raises(NotImplementedError):
        VelocityAviary(["robobee"], 2, initial_xyzs=xyz, physics=Physics.DYN)
