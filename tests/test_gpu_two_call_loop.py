"""GPU parity tests of the reference-shaped two-call loop (HIP path through the C-ABI vs the fp64 oracle, tests/util.py's bars): the
reference-shaped two-call loop — obs = env.step(action); action = ctrl.computeControlFromState(obs)
(examples/fly_INDI.py:223-239, examples/fly_hexa_6DOF.py:214-221) — on every fleet kind through the run kernels
(k_physics_runs / k_control_runs), and the explicit-action instances of k_step_runs.
"""
import ctypes
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from dronesim_amd import params  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from tests.test_gpu_parity import _args, _check_obs_rows, _stream  # noqa: E402
from tests.util import (K_ULP, assert_control_parity, assert_downwash, assert_step_parity, f32, random_fleet, rotor_noise)  # noqa: E402

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DT = float(np.float32(1.0 / 240.0))


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device; the product has no CPU fallback")
    from dronesim_amd import _native as nat
    from dronesim_amd import fleet
    return nat, fleet


def _noise_by_id(O, types, tid, ids, seed, step_index, sub):
    """[n, sub, 12] scaled normals of the in-kernel generator, keyed by ids[i] (dsim_step_args.drone_id)."""
    n = len(ids)
    nz = np.zeros((n, sub, 12))
    for i in range(n):
        na = types[0 if tid is None else int(tid[i])].n_act
        for s_ in range(sub):
            u = O.noise_normals(seed, int(ids[i]), step_index * sub + s_, na, fine=(sub == 1))
            nz[i, s_, 0:na], nz[i, s_, 6:6 + na] = rotor_noise(types[0 if tid is None else int(tid[i])], u)
    return nz


def _downwash_part(O, types, tid, r0, r1, sub):
    """[n, 13]: how much of one Env.step's increment is the neighbour-downwash force's doing (per drone: the larger of the
    force at the start and at the end of the step, over the mass, times the time it acts) — that part is itself an fp32
    sum of exp() terms and only known to REL_TOL (tests/util.py:increment_ratio).  Plus, for the sub-steps behind the
    first: the device's and the oracle's positions differ by fp32 rounding there, and the term of a nearly vertical close
    pair (alpha ~ 1 / dz^2) turns that into a force difference of its own — measured by moving every receiver two fp32
    ulps up and down against the others."""
    from tests.util import REL_TOL, ulp32
    n = r0.shape[0]
    m = np.array([t.mass for t in types])[tid] if tid is not None else np.full(n, types[0].mass)
    f0, f1 = O.downwash(r0, r0[:, 0:3], type_id=tid), O.downwash(r1, r1[:, 0:3], type_id=tid)
    f = np.maximum(np.abs(f0), np.abs(f1))
    sens = np.zeros(n)
    if sub > 1:
        for sign in (1.0, -1.0):
            rp = r1.copy()
            rp[:, 2] += sign * 2 * ulp32(r1[:, 2])
            sens = np.maximum(sens, np.abs(O.downwash(rp, r1[:, 0:3], type_id=tid) - f1))
    part = np.zeros((n, 13))
    part[:, 7:10] = ((f + sens / REL_TOL) / m * DT * sub)[:, None]
    part[:, 0:3] = part[:, 7:10] * DT * sub
    return part


def _runs_arr(nat, runs):
    arr = (nat.TypeRun * len(runs))()
    for k, (f, c, ty) in enumerate(runs):
        arr[k].first, arr[k].count, arr[k].type = f, c, ty
    return arr


def _two_call_case(gpu, label, types, tid, runs, n, sub, seed, layout="tile64", pad=256, ext=False, ids=False, obs=True,
                   want_yaw=True, align_obs=0, caller_io=False):
    """One Env.step (dsim_physics: action in, state + echo + observation rows out) and one computeControl (dsim_control2)
    on a fleet stored as `runs`, streaming on and off (same bits), each against the oracle at the step's bar.
    caller_io: DSIM_OPT_CALLER_IO — action, rows, command and errors are indexed by drone_id[i]."""
    nat, fleet = gpu
    na = max(t.n_act for t in types)
    W = 16 + na
    rng = np.random.default_rng(n + 3 * sub + seed)
    rigid, mem, tgt = random_fleet(rng, n, n_act=na, tilt=0.3, rate=1.0)
    if tid is not None:
        for k, t in enumerate(types):
            mem[tid == k, 7 + t.n_act:13] = 0.0
    act = f32(rng.uniform(-0.1, 1.1, (n, na)))                      # some of it outside [pwm_min, pwm_max]: clipped in-kernel
    if tid is not None:
        for k, t in enumerate(types):
            act[tid == k, t.n_act:] = 0.0
    force = f32(np.stack([np.zeros(n), np.zeros(n), -rng.uniform(0.0, 0.3, n)], 1)) if ext else None
    id_arr = rng.permutation(n).astype(np.int32) if ids else np.arange(n, dtype=np.int32)
    dtc = float(np.float32(sub / 240))
    got = {}
    for pol in (nat.OPT_STREAM_ON, nat.OPT_STREAM_OFF):
        ctx = fleet.Context(types)
        st, tg = fleet.FleetState(ctx, n, layout, pad), fleet.Targets(ctx, n, layout, pad=pad)
        st.load_aos(rigid, mem)
        tg.set_fields(0, torch.from_numpy(np.ascontiguousarray(tgt.T)))
        dev = ctx.device
        tdev = None
        if tid is not None:
            tdev = torch.zeros(st.n_pad, dtype=torch.uint8, device=dev); tdev[:n] = torch.from_numpy(tid)
        adev = torch.zeros((na, st.n_pad), device=dev)
        io = torch.from_numpy(id_arr.astype(np.int64)).to(dev) if caller_io else torch.arange(n, device=dev)
        adev[:, io] = torch.from_numpy(np.ascontiguousarray(act.T)).float().to(dev)      # slot i's action sits at column io[i]
        echo = torch.full((na, st.n_pad), -3.0, device=dev)
        obs_buf = torch.full((n * W + 4,), -7.0, device=dev)
        rows = obs_buf[align_obs: align_obs + n * W].view(n, W)      # align_obs = 1: a row block that is only 4-byte aligned
        fdev = None
        if ext:
            fdev = torch.zeros((3, st.n_pad), device=dev); fdev[:, :n] = torch.from_numpy(np.ascontiguousarray(force.T)).float()
        iddev = None
        if ids:
            full = np.arange(st.n_pad, dtype=np.int32); full[:n] = id_arr
            iddev = torch.from_numpy(full).to(dev)
        io_opt = nat.OPT_CALLER_IO if caller_io else 0
        a = _args(nat, sub, DT, dtc, options=pol | io_opt, seed=seed, step_index=5, type_id=tdev, action=adev)
        arr = _runs_arr(nat, runs) if runs is not None else None
        if arr is not None:
            a.runs, a.n_runs = ctypes.addressof(arr), len(runs)
        a.ext_force = fdev.data_ptr() if fdev is not None else None
        a.drone_id = iddev.data_ptr() if iddev is not None else None
        if obs:
            a.obs_out, a.obs_width = rows.data_ptr(), W
        nat.check(ctx.lib.dsim_physics(ctx.handle, _stream(ctx), n, st.view(), echo.data_ptr(), ctypes.byref(a)))
        g_rigid = st.rigid_aos()
        g_echo = echo[:, :n].T.double().cpu().numpy()
        g_rows = rows[io].double().cpu().numpy()                     # (row of slot i)
        guard = obs_buf.cpu().numpy()
        # computeControl on the new state
        pe = torch.full((3, st.n_pad), -5.0, device=dev); ye = torch.full((st.n_pad,), -5.0, device=dev)
        cmd = torch.full((na, st.n_pad), -5.0, device=dev)
        c = _args(nat, 0, dtc, dtc, options=pol | io_opt, type_id=tdev)
        c.drone_id = iddev.data_ptr() if iddev is not None else None
        if arr is not None:
            c.runs, c.n_runs = ctypes.addressof(arr), len(runs)
        nat.check(ctx.lib.dsim_control2(ctx.handle, _stream(ctx), n, st.view(), tg.view(), ctypes.byref(c), pe.data_ptr(),
                                        ye.data_ptr() if want_yaw else None, cmd.data_ptr()))
        got[pol] = (g_rigid, g_echo, g_rows, st.mem_aos(), pe[:, io].T.double().cpu().numpy(), ye[io].double().cpu().numpy(),
                    cmd[:, io].T.double().cpu().numpy(), guard)
        assert ctx.query(nat.QUERY_WLS_FAILURES) == 0
        ctx.close()
    for x, y in zip(got[nat.OPT_STREAM_ON], got[nat.OPT_STREAM_OFF]):
        np.testing.assert_array_equal(x, y, err_msg=label)
    g_rigid, g_echo, g_rows, g_mem, g_pe, g_ye, g_cmd, guard = got[nat.OPT_STREAM_OFF]
    O = orc.Oracle(types)
    # ---- Env.step: the clipped action, the physics, the echo, the rows
    a6 = np.zeros((n, 6)); a6[:, :na] = act
    r = rigid.copy()
    nz = _noise_by_id(O, types, tid, id_arr, seed, 5, sub) if seed else None
    last = np.zeros((n, 6))
    O.physics(r, mem, sub, DT, action=a6, noise=nz, type_id=tid, last_action=last, ext_force=force)
    assert_step_parity(label + " physics", types, tid, rigid, mem, tgt, g_rigid, None, r, None, DT, dtc, sub, control=False,
                       action=np.clip(act, 0.0, 1.0))
    np.testing.assert_array_equal(g_echo, last[:, :na], err_msg=label + " echo")      # every row, zeros behind a quad's own four
    if obs:
        _check_obs_rows(label + " rows", O, g_rows, g_rigid, last, tid, types)
        np.testing.assert_array_equal(g_rows[:, 16:W], last[:, :na])                      # ... the same in the rows
        assert (guard[:align_obs] == -7.0).all() and (guard[align_obs + n * W:] == -7.0).all()
    # ---- computeControl on the device's own new state
    m = mem.copy()
    rc, pe, ye = O.control(g_rigid, m, tgt, dtc, type_id=tid)
    assert rc == 0
    assert_control_parity(label + " control", types, tid, g_rigid, mem, tgt, g_mem, m, dtc)
    np.testing.assert_array_equal(g_cmd, g_mem[:, 7:7 + na])                              # cmd_out: the command, every row
    np.testing.assert_allclose(g_pe, pe, rtol=0, atol=1e-5)
    if want_yaw:
        d = np.abs(g_ye - ye)
        assert np.minimum(d, np.abs(d - 2 * np.pi)).max() < 2e-5
    return g_rigid


@pytest.mark.parametrize("seed", [0, 7])
@pytest.mark.parametrize("sub", [1, 2])
def test_two_call_loop_on_runs_vs_oracle(gpu, sub, seed):
    """k_physics_runs<NOISE, NT, OBS> / k_control_runs<NT, WANT_YAW>: homogeneous hexa fleets (one run), type-major quad +
    hexa fleets with aligned runs and with runs that begin and end inside tiles and waves, two quad types (20-wide rows on
    the run path), the downwash force input, the noise stream keyed by the caller's index, ragged tails."""
    import dataclasses
    nat, fleet = gpu
    rb, hx, te = params.builtin_type("robobee"), params.builtin_type("hexa_6DOF"), params.builtin_type("tello")
    s = f"[{sub},{seed}]"
    # one type, no runs given: the library makes ONE run of it
    _two_call_case(gpu, "two-call hexa" + s, [hx], None, None, 512, sub, seed)
    _two_call_case(gpu, "two-call hexa soa ragged" + s, [hx], None, None, 300, sub, seed, layout="soa", pad=64, want_yaw=False)
    _two_call_case(gpu, "two-call hexa no rows" + s, [hx], None, None, 512, sub, seed, obs=False)
    _two_call_case(gpu, "two-call quad ext force" + s, [rb], None, None, 512, sub, seed, ext=True)
    _two_call_case(gpu, "two-call quad ragged" + s, [rb], None, None, 300, sub, seed, pad=64)
    # type-major quad + hexa: aligned runs; runs sharing a tile AND a wave (200 is not a multiple of 64)
    tid = np.repeat(np.array([0, 1], dtype=np.uint8), 256)
    _two_call_case(gpu, "two-call runs" + s, [rb, hx], tid, [(0, 256, 0), (256, 256, 1)], 512, sub, seed, ids=True)
    # DSIM_OPT_CALLER_IO: the per-drone arrays beside the state in the caller's numbering (a random permutation)
    _two_call_case(gpu, "two-call runs, caller io" + s, [rb, hx], tid, [(0, 256, 0), (256, 256, 1)], 512, sub, seed, ids=True,
                   caller_io=True)
    _two_call_case(gpu, "two-call hexa, caller io, ragged" + s, [hx], None, None, 300, sub, seed, ids=True, caller_io=True, pad=64)
    _two_call_case(gpu, "two-call two quad types, caller io" + s, [rb, te], np.array([0] * 300 + [1] * 212, dtype=np.uint8),
                   [(0, 300, 0), (300, 212, 1)], 512, sub, seed, ids=True, caller_io=True, layout="soa", want_yaw=False)
    tid2 = np.array([0] * 200 + [1] * 312, dtype=np.uint8)
    _two_call_case(gpu, "two-call runs sharing a wave" + s, [rb, hx], tid2, [(0, 200, 0), (200, 312, 1)], 512, sub, seed, ids=True,
                   ext=True)
    _two_call_case(gpu, "two-call runs sharing a wave, caller io" + s, [rb, hx], tid2, [(0, 200, 0), (200, 312, 1)], 512, sub, seed,
                   ids=True, ext=True, caller_io=True)
    _two_call_case(gpu, "two-call runs sharing a wave, soa" + s, [rb, hx], tid2, [(0, 200, 0), (200, 312, 1)], 512, sub, seed,
                   layout="soa", want_yaw=False)
    # the caller's numbering without observation rows (k_physics_runs_io<.., OBS = false, ..>: the action gather alone)
    _two_call_case(gpu, "two-call runs sharing a wave, caller io, no rows" + s, [rb, hx], tid2, [(0, 200, 0), (200, 312, 1)], 512, sub,
                   seed, ids=True, caller_io=True, obs=False)
    # three runs, the middle one inside one wave; ragged end
    tid3 = np.array([1] * 70 + [0] * 30 + [1] * 337, dtype=np.uint8)
    _two_call_case(gpu, "two-call three runs" + s, [rb, hx], tid3, [(0, 70, 1), (70, 30, 0), (100, 337, 1)], 437, sub, seed, pad=64)
    # two QUAD types as runs: 20-wide rows through the same kernels
    tid4 = np.array([0] * 300 + [1] * 212, dtype=np.uint8)
    _two_call_case(gpu, "two-call two quad types" + s, [rb, te], tid4, [(0, 300, 0), (300, 212, 1)], 512, sub, seed)
    # hexa_6DOF_simple: morphing-hexa physics + the quad law on six actuators — alone, and as one of three kinds of runs
    hs = params.builtin_type("hexa_6DOF_simple")
    _two_call_case(gpu, "two-call hexa_simple" + s, [hs], None, None, 512, sub, seed)
    tid5 = np.array([0] * 150 + [1] * 170 + [2] * 192, dtype=np.uint8)
    _two_call_case(gpu, "two-call three kinds" + s, [rb, hx, hs], tid5, [(0, 150, 0), (150, 170, 1), (320, 192, 2)], 512, sub, seed,
                   ids=True, caller_io=True)
    # a row block that is not 8-byte aligned: the rows come from the observation kernel behind the step
    _two_call_case(gpu, "two-call hexa misaligned rows" + s, [hx], None, None, 512, sub, seed, align_obs=1)
    del dataclasses


@pytest.mark.parametrize("seed", [0, 7])
@pytest.mark.parametrize("sub", [1, 2])
def test_step_runs_with_an_explicit_action(gpu, sub, seed):
    """k_step_runs<NOISE, NT, S1, ACT = true>: the fused step of a type-major fleet with an explicit action for the physics
    part (the first iteration of the example loop, fly_INDI.py:214) — one run, two runs, runs sharing a tile."""
    from tests.test_gpu_parity import _sweep_case
    nat, fleet = gpu
    rb, hx = params.builtin_type("robobee"), params.builtin_type("hexa_6DOF")
    n = 512
    act6 = f32(np.random.default_rng(2).uniform(0.3, 0.7, (n, 6)))
    tid = np.repeat(np.array([0, 1], dtype=np.uint8), 256)
    a = act6.copy(); a[tid == 0, 4:] = 0.0
    _sweep_case(gpu, f"sweep runs action[{sub},{seed}]", [rb, hx], tid, n, sub, seed, 0, action=a, runs=[(0, 256, 0), (256, 256, 1)])
    tid2 = np.array([0] * 200 + [1] * 312, dtype=np.uint8)
    a = act6.copy(); a[tid2 == 0, 4:] = 0.0
    _sweep_case(gpu, f"sweep runs sharing a tile, action[{sub},{seed}]", [rb, hx], tid2, n, sub, seed, 0, action=a,
                runs=[(0, 200, 0), (200, 312, 1)])


def test_two_call_physics_fills_the_next_neighbour_grid(gpu):
    """dsim_physics honours bin_next: the Env.step launch of the two-call loop appends the NEW positions to the next
    neighbour grid, and the downwash evaluated from that pre-binned grid equals the brute-force sum over the new positions
    (mixed type-major fleet, the downwash force as input)."""
    nat, fleet = gpu
    rb, hx = params.builtin_type("robobee"), params.builtin_type("hexa_6DOF")
    types = [rb, hx]
    n = 1024
    rng = np.random.default_rng(11)
    tid = np.array([0] * 500 + [1] * 524, dtype=np.uint8)
    runs = [(0, 500, 0), (500, 524, 1)]
    rigid, mem, tgt = random_fleet(rng, n, n_act=6, tilt=0.3, rate=1.0)
    rigid[:, 0] = f32(rng.uniform(1, 79, n)); rigid[:, 1] = f32(rng.uniform(1, 59, n)); rigid[:, 2] = f32(rng.uniform(1, 15, n))
    mem[tid == 0, 11:13] = 0.0
    act = f32(rng.uniform(0.3, 0.7, (n, 6))); act[tid == 0, 4:] = 0.0
    O = orc.Oracle(types)
    ctx = fleet.Context(types)
    st = fleet.FleetState(ctx, n, "tile64")
    st.load_aos(rigid, mem)
    dev = ctx.device
    tdev = torch.zeros(st.n_pad, dtype=torch.uint8, device=dev); tdev[:n] = torch.from_numpy(tid)
    adev = torch.zeros((6, st.n_pad), device=dev); adev[:, :n] = torch.from_numpy(np.ascontiguousarray(act.T)).float()
    echo = torch.zeros((6, st.n_pad), device=dev)
    g = nat.DownwashArgs()
    nx, ny = 16, 12
    assert ctx.lib.dsim_downwash_prebin_ok(n, nx, ny) == 1
    ws = torch.empty((ctx.lib.dsim_downwash_workspace(n, nx, ny),), dtype=torch.int32, device=dev)
    g.pos_all, g.m, g.m_pad = None, n, n
    g.xmin, g.ymin, g.cell, g.nx, g.ny = 0.0, 0.0, 5.0, nx, ny
    g.workspace, g.workspace_len, g.type_id, g.local_offset = ws.data_ptr(), ws.numel(), tdev.data_ptr(), 0
    force = torch.zeros((3, st.n_pad), device=dev)
    s_ = _stream(ctx)
    nat.check(ctx.lib.dsim_downwash(ctx.handle, s_, n, st.view(), ctypes.byref(g), force.data_ptr()))
    f0 = force.cpu().numpy()[2, :n].astype(np.float64)
    a = _args(nat, 2, DT, float(np.float32(2 / 240)), seed=3, step_index=2, type_id=tdev, action=adev)
    arr = _runs_arr(nat, runs)
    a.runs, a.n_runs = ctypes.addressof(arr), 2
    a.ext_force, a.bin_next = force.data_ptr(), ctypes.addressof(g)
    rows = torch.zeros((n, 22), device=dev)
    a.obs_out, a.obs_width = rows.data_ptr(), 22
    nat.check(ctx.lib.dsim_physics(ctx.handle, s_, n, st.view(), echo.data_ptr(), ctypes.byref(a)))
    g.prebinned = 1
    force2 = torch.zeros((3, st.n_pad), device=dev)
    nat.check(ctx.lib.dsim_downwash(ctx.handle, s_, n, st.view(), ctypes.byref(g), force2.data_ptr()))
    new = st.rigid_aos()
    f1 = force2.cpu().numpy()[2, :n].astype(np.float64)
    # the step itself, with the force of the OLD positions held over the sub-steps
    r = rigid.copy()
    ext = np.zeros((n, 3)); ext[:, 2] = f32(f0)
    a6 = act.astype(np.float64)
    nz = _noise_by_id(O, types, tid, np.arange(n), 3, 2, 2)
    O.physics(r, mem, 2, DT, action=a6, noise=nz, type_id=tid, ext_force=ext)
    assert_step_parity("two-call physics + binning", types, tid, rigid, mem, tgt, new, None, r, None, DT, float(np.float32(2 / 240)), 2,
                       control=False, action=act)
    ref1 = O.downwash(new, new[:, 0:3], type_id=tid)
    assert_downwash("two-call physics, prebinned grid", f1, ref1, types, tid, new[:, 0:3], new[:, 0:3])
    assert (ref1 < 0).sum() > n // 4
    # a physics call WITHOUT bin_next moves the drones behind the grid: the library must not trust prebinned = 1 any more
    a.bin_next = None
    nat.check(ctx.lib.dsim_physics(ctx.handle, s_, n, st.view(), echo.data_ptr(), ctypes.byref(a)))
    nat.check(ctx.lib.dsim_downwash(ctx.handle, s_, n, st.view(), ctypes.byref(g), force2.data_ptr()))
    new2 = st.rigid_aos()
    ref2 = O.downwash(new2, new2[:, 0:3], type_id=tid)
    assert_downwash("two-call physics, stale grid dropped", force2.cpu().numpy()[2, :n].astype(np.float64), ref2, types, tid,
                    new2[:, 0:3], new2[:, 0:3])
    ctx.close()


@pytest.mark.parametrize("kind", ["hexa", "hexa_simple", "mixed", "mixed_downwash"])
def test_env_step_then_computeControl_loop_on_every_fleet_kind(gpu, kind):
    """The reference-shaped surfaces end to end (examples/fly_hexa_6DOF.py:214-221): obs = env.step(action); action =
    ctrl.computeControlFromState(obs) for 12 iterations on a hexa fleet, an interleaved quad + hexa fleet (stored
    type-major behind the caller's numbering) and the same with the neighbour-downwash term — every iteration's state
    increment and controller memory against the oracle driven from the device's own previous state, the observation rows
    against the state vector of the new state, in the CALLER's numbering."""
    nat, fleet = gpu
    from dronesim_amd.control import INDIControl
    from dronesim_amd.envs import CtrlAviary, Physics
    from dronesim_amd.fleet import frozen
    rb, hx = params.builtin_type("robobee"), params.builtin_type("hexa_6DOF")
    n = 1500
    rng = np.random.default_rng(17)
    if kind == "hexa":
        models, tid, types = ["hexa_6DOF"], None, [hx]
    elif kind == "hexa_simple":      # examples/fly_hexa_6DOF_simple.py:202-221
        models, tid, types = ["hexa_6DOF_simple"], None, [params.builtin_type("hexa_6DOF_simple")]
    else:
        models, tid, types = ["robobee", "hexa_6DOF"], (np.arange(n) % 2).astype(np.uint8), [rb, hx]
    xyz = np.stack([rng.uniform(0, 60, n), rng.uniform(0, 60, n), rng.uniform(2, 12, n)], 1)
    sub = 2
    env = CtrlAviary(models, n, initial_xyzs=xyz, aggregate_phy_steps=sub, noise_seed=0, dict_io=False, type_ids=tid,
                     physics=Physics.PYB_DW if kind == "mixed_downwash" else Physics.PYB, layout="tile64")
    ctrl = INDIControl(models[-1], env=env)
    na = env.n_act
    tpos = f32(xyz + rng.uniform(-0.5, 0.5, (n, 3)))
    tp = frozen(torch.from_numpy(tpos.astype(np.float32)).to(env.ctx.device))
    O = orc.Oracle(types)
    dtc = float(np.float32(sub / 240))
    tg = np.concatenate([tpos, np.zeros((n, 6)), np.full((n, 1), np.float32(0.2))], 1)
    action = torch.full((n, na), 0.45, device=env.ctx.device)
    if tid is not None:
        action[torch.from_numpy(tid == 0).to(env.ctx.device), 4:] = 0.0
    for k in range(12):
        r0, m0 = env.state.rigid_aos(), env.state.mem_aos()
        a6 = np.zeros((n, 6)); a6[:, :na] = action.double().cpu().numpy()
        obs, _, _, _ = env.step(action)
        r1 = env.state.rigid_aos()
        r = r0.copy()
        last = np.zeros((n, 6))
        fz = None
        if kind == "mixed_downwash":                 # the term per physics sub-step, as the reference loops it
            fz = O.downwash(r0, r0[:, 0:3], type_id=tid)
            assert O.physics_downwash(r, m0, sub, DT, action=a6, type_id=tid, last_action=last) == 0
        else:
            O.physics(r, m0, sub, DT, action=a6, type_id=tid, last_action=last)
        assert_step_parity(f"env.step loop[{kind}]", types, tid, r0, m0, tg, r1, None, r, None, DT, dtc, sub, control=False,
                           action=a6[:, :na], part_rigid=_downwash_part(O, types, tid, r0, r, sub) if fz is not None else None)
        _check_obs_rows(f"env.step loop rows[{kind}]", O, obs.double().cpu().numpy(), r1, last, tid, types)
        action, pos_e, yaw_e = ctrl.computeControlFromState(dtc, None, target_pos=tp, target_rpy=np.array([0, 0, 0.2]))
        m1 = env.state.mem_aos()
        m = m0.copy()
        rc, pe, ye = O.control(r1, m, tg, dtc, type_id=tid)
        assert rc == 0
        assert_control_parity(f"computeControl loop[{kind}]", types, tid, r1, m0, tg, m1, m, dtc)
        np.testing.assert_array_equal(action.double().cpu().numpy(), m1[:, 7:7 + na])
        np.testing.assert_allclose(pos_e.double().cpu().numpy(), pe, rtol=0, atol=1e-5)
    assert env.ctx.query(nat.QUERY_WLS_FAILURES) == 0
    env.close()


def test_caller_io_is_refused_where_the_run_kernels_do_not_serve(gpu):
    """DSIM_OPT_CALLER_IO without drone_id, on the fused step, on the adaptors, with the plane option: refused, never ignored."""
    nat, fleet = gpu
    rb = params.builtin_type("robobee")
    n = 256
    ctx = fleet.Context([rb])
    st, tg = fleet.FleetState(ctx, n), fleet.Targets(ctx, n)
    ids = torch.arange(st.n_pad, dtype=torch.int32, device=ctx.device)
    act = torch.zeros((4, st.n_pad), device=ctx.device)
    a = _args(nat, 1, DT, DT, options=nat.OPT_CALLER_IO, action=act)
    sp = _stream(ctx)
    assert ctx.lib.dsim_physics(ctx.handle, sp, n, st.view(), None, ctypes.byref(a)) == -1          # no drone_id: DSIM_E_ARG
    a.drone_id = ids.data_ptr()
    assert ctx.lib.dsim_physics(ctx.handle, sp, n, st.view(), None, ctypes.byref(a)) == 0           # one type = one run: served
    assert ctx.lib.dsim_step(ctx.handle, sp, n, st.view(), tg.view(), ctypes.byref(a)) == -5          # DSIM_E_UNSUPPORTED
    assert ctx.lib.dsim_step_adaptor(ctx.handle, sp, n, st.view(), act.data_ptr(), nat.ADAPT_RPYT, None, ctypes.byref(a)) == -5
    a.options |= nat.OPT_PLANE
    assert ctx.lib.dsim_physics(ctx.handle, sp, n, st.view(), None, ctypes.byref(a)) == -5
    torch.cuda.synchronize()
    ctx.close()


def test_reordered_fleet_over_the_plane_is_translated_by_the_host(gpu):
    """Where the run kernels do not serve a fleet stored behind the caller's numbering (here: the ground-plane option, the
    general kernels), the host class translates rows, commands and errors itself: same results as the same fleet stored in
    the caller's own order."""
    from dronesim_amd.control import INDIControl
    from dronesim_amd.envs import CtrlAviary
    from dronesim_amd.fleet import frozen
    n = 600
    rng = np.random.default_rng(23)
    tid = rng.integers(0, 2, n).astype(np.uint8)
    xyz = np.stack([rng.uniform(0, 30, n), rng.uniform(0, 30, n), rng.uniform(0.04, 0.4, n)], 1)
    out = {}
    for storage in ("auto", "caller"):
        env = CtrlAviary(["robobee", "hexa_6DOF"], n, initial_xyzs=xyz, aggregate_phy_steps=2, noise_seed=0, dict_io=False,
                         type_ids=tid, storage=storage, ground_plane=True)
        assert not env._caller_io
        ctrl = INDIControl("hexa_6DOF", env=env)
        tp = frozen(torch.from_numpy(f32(xyz + [0, 0, 0.5]).astype(np.float32)).to(env.ctx.device))
        cmd = torch.full((n, 6), 0.3, device=env.ctx.device)
        for _ in range(6):
            obs, _, _, _ = env.step(cmd)
            cmd, pos_e, yaw_e = ctrl.computeControlFromState(2 * DT, None, target_pos=tp)
        out[storage] = (obs.cpu().numpy(), cmd.cpu().numpy(), pos_e.cpu().numpy(), yaw_e.cpu().numpy())
        env.close()
    for a_, c_ in zip(out["auto"], out["caller"]):
        np.testing.assert_allclose(a_, c_, rtol=3e-4, atol=3e-5)


def test_rccl_runs_at_world_size_one(gpu):
    """The collective backend of the multi-GPU path (RCCL: torch.distributed backend "nccl") on the hardware that is here: a
    child started by torch.distributed.run — one rank — makes the process group on the device, runs the default bench line
    through its N-rank code path (barrier, MAX all-reduce and the per-rank all-gather on DEVICE tensors), then the
    device-side all-gather of the downwash term's positions and one grouped isend / irecv batch on slices of halo-sized
    buffers, and destroys the group.  Proves that the stack's RCCL loads and moves device memory here (and that
    HSA_ENABLE_IPC_MODE_LEGACY=0 is what it wants); it is not a scaling measurement."""
    import json
    import socket
    import subprocess
    import sys
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(DSIM_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--workload", "config4", "--steps", "20",
           "--warmup", "3", "--no-also", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=420, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["dist"]["backend"] == "nccl" and d["dist"]["world_size"] == 1 and d["n_gpus"] == 1
    st = d["rccl_selftest"]
    assert st["backend"] == "nccl" and st["all_gather_positions_ok"] and st["grouped_isend_irecv_ok"], st
    assert st["all_reduce_on"].startswith("cuda")
    assert d["value"] > 1e8


@pytest.mark.parametrize("surface", ["step_fused", "two_call"])
def test_downwash_is_evaluated_per_physics_substep(gpu, surface):
    """Physics.PYB_DW with AGGR_PHY_STEPS = 5 (what examples/fly_INDI.py's defaults select): the reference refreshes the
    positions and applies _downwash INSIDE the sub-step loop (BaseAviary.py:510-536), so the force follows the drones through
    the sub-steps.  The env launches [query -> one sub-step] pairs; checked per Env.step against the oracle that evaluates
    the term per sub-step (orc_physics_downwash_batch), noise on, from the device's own previous state — and the result
    must differ measurably from holding the first sub-step's force (the round-3 behaviour)."""
    nat, fleet = gpu
    from dronesim_amd.control import INDIControl
    from dronesim_amd.envs import CtrlAviary, Physics
    from dronesim_amd.fleet import Targets, frozen
    rb, hx = params.builtin_type("robobee"), params.builtin_type("hexa_6DOF")
    types = [rb, hx]
    n, sub, seed = 900, 5, 11
    rng = np.random.default_rng(29)
    tid = (np.arange(n) % 2).astype(np.uint8)
    # dense columns of drones, the upper ones descending fast on the lower ones: the term changes within one Env.step
    xyz = np.stack([rng.uniform(0, 12, n), rng.uniform(0, 12, n), rng.uniform(1.0, 6.0, n)], 1)
    vel = np.zeros((n, 3)); vel[:, 2] = rng.uniform(-6.0, 2.0, n)
    env = CtrlAviary(["robobee", "hexa_6DOF"], n, initial_xyzs=xyz, initial_vels=vel, aggregate_phy_steps=sub, noise_seed=seed,
                     dict_io=False, type_ids=tid, physics=Physics.PYB_DW, layout="tile64")
    O = orc.Oracle(types)
    dtc = float(np.float32(sub / 240))
    tpos = f32(xyz)
    tg_np = np.concatenate([tpos, np.zeros((n, 6)), np.zeros((n, 1))], 1)
    tg = Targets(env.ctx, n, "tile64")
    tg.set(pos=tpos.T, yaw=0.0)
    ctrl = INDIControl("hexa_6DOF", env=env) if surface == "two_call" else None
    if ctrl is not None:
        env._housekeeping()          # (the controller's reset wrote the memory of a fresh fleet; positions and velocities again)
    action = torch.full((n, 6), 0.45, device=env.ctx.device)
    action[torch.from_numpy(tid == 0).to(env.ctx.device), 4:] = 0.0
    held_differs = 0.0
    for k in range(4):
        r0, m0 = env.state.rigid_aos(), env.state.mem_aos()
        a6 = action.double().cpu().numpy() if (surface == "two_call" or k == 0) else None
        if surface == "two_call":
            env.step(action)
        else:
            env.step_fused(tg, action=action if k == 0 else None)
        r1 = env.state.rigid_aos()
        nz = _noise_by_id(O, types, tid, np.arange(n), seed, k, sub)
        r = r0.copy()
        assert O.physics_downwash(r, m0, sub, DT, action=a6, noise=nz, type_id=tid) == 0
        # what holding the force of the first sub-step would have given (round 3)
        rh = r0.copy()
        fz = O.downwash(r0, r0[:, 0:3], type_id=tid)
        ext = np.zeros((n, 3)); ext[:, 2] = fz
        O.physics(rh, m0, sub, DT, action=a6, noise=nz, type_id=tid, ext_force=ext)
        held_differs = max(held_differs, float(np.abs(rh[:, 9] - r[:, 9]).max()))
        applied = a6[:, :6] if a6 is not None else m0[:, 7:13]
        assert_step_parity(f"downwash per sub-step[{surface}]", types, tid, r0, m0, tg_np, r1, None, r, None, DT, dtc, sub,
                           control=False, action=applied, part_rigid=_downwash_part(O, types, tid, r0, r, sub))
        if surface == "two_call":
            action, _, _ = ctrl.computeControlFromState(dtc, None, target_pos=frozen(torch.from_numpy(tpos.astype(np.float32)).to(env.ctx.device)))
    assert held_differs > 1e-4          # m/s: the case tells the two semantics apart by far more than the bar
    with pytest.raises(NotImplementedError):
        env.capture_fused(tg, 4)
    env.close()


@pytest.mark.parametrize("kind", ["hexa", "mixed"])
def test_placement_of_rows_on_hexa_and_reordered_fleets_changes_nothing(gpu, kind):
    """Placement by trial (dronesim_amd/placement.py) beyond quad fleets: the observation rows and the controller's outputs
    of a large hexa fleet / an interleaved quad + hexa fleet are allocated straight from the driver (dsim_dev_alloc) by
    timing real zero-sub-step passes behind a snapshot of the state block — states, rows and commands bit for bit those of an
    env with plainly allocated arrays; the search reports what it cost and never touches PyTorch's allocator cache."""
    nat, fleet = gpu
    from dronesim_amd.control import INDIControl
    from dronesim_amd.envs import CtrlAviary
    from dronesim_amd.fleet import frozen
    nd = 1 << 20                                        # rows 92 MB: above placement.MIN_BYTES
    rng = np.random.default_rng(3)
    xyz = np.stack([np.arange(nd) % 1024, np.arange(nd) // 1024, rng.uniform(1.0, 3.0, nd)], 1).astype(np.float64)
    models, tid = (["hexa_6DOF"], None) if kind == "hexa" else (["robobee", "hexa_6DOF"], (np.arange(nd) % 2).astype(np.uint8))
    res = []
    reserved0 = torch.cuda.memory_reserved()
    for p in (True, False):
        e = CtrlAviary(models, nd, initial_xyzs=xyz, aggregate_phy_steps=1, noise_seed=4, dict_io=False, type_ids=tid, placement=p)
        ctrl = INDIControl("hexa_6DOF", env=e)
        tp = frozen(torch.from_numpy(f32(xyz + 0.1).astype(np.float32)).to(e.ctx.device))
        cmd = torch.full((nd, 6), 0.45, device=e.ctx.device)
        if tid is not None:
            cmd[torch.from_numpy(tid == 0).to(e.ctx.device), 4:] = 0.0
        for _ in range(3):
            obs, _, _, _ = e.step(cmd)
            cmd, pos_e, yaw_e = ctrl.computeControlFromState(1 / 240, None, target_pos=tp, target_rpy=np.array([0.0, 0.0, 0.3]))
        res.append((e.state.data.clone(), obs.clone(), cmd.clone(), pos_e.clone(), yaw_e.clone(), e.ground_contacts()))
        if p:
            log = list(e.ctx.placement_log)
            assert [r["array"] for r in log if r["array"] != "observation rows" and not r["array"].startswith("state block")] in (["computeControl outputs"], ["computeControl outputs", "computeControl targets"])      # (targets: only from 64 MB on)
            rows = [r for r in log if r["array"] == "observation rows"]
            assert len(rows) == 1 and rows[0]["memory"].startswith("driver") and rows[0]["candidates"] >= 2
            assert rows[0]["candidates"] * rows[0]["bytes"] <= rows[0]["peak_bytes"] <= rows[0]["budget_bytes"]     # (+ the ballast strides)
            assert 0 < rows[0]["seconds"] < 30
            outs = [r for r in log if r["array"] == "computeControl outputs"][0]
            # the command lives in the room behind the placed rows — unless the controller's own arrays served the launch better
            assert (cmd.data_ptr() == e._written_tail.data_ptr()) == outs["placed"].startswith("behind the env's observation rows")
            assert outs["behind_the_rows_pass_us"] > 0 and outs["plain_pass_us"] > 0
        else:
            assert e.ctx.placement_log == []
        e.close()
    for x, y in zip(res[0][:5], res[1][:5]):
        assert torch.equal(x, y)
    assert res[0][5] == res[1][5]


@pytest.mark.parametrize("mode", ["velocity", "rpyt"])
def test_every_instance_of_the_one_launch_adaptor_step(gpu, mode):
    """k_adaptor_fast<MODE, NOISE, NT> (+ its run-time switches: rows fused or not, action layout) through dsim_step_adaptor: the action field-major or row-major
    (DSIM_OPT_ACTION_ROWS), the observation rows of the NEW state fused or not, noise, both cache policies — control part
    and physics part each against the oracle (VelocityAviary.py:221-264, RPYTAviary.py:181-193), rows against
    orc_state_vector; a ragged fleet inside whole tiles."""
    from tests.test_gpu_parity import _noise_block
    from tests.util import noise_terms
    nat, fleet = gpu
    t = params.builtin_type("robobee")
    O = orc.Oracle([t])
    ctx = fleet.Context([t])
    n, sub = 700, 2
    dtc = float(np.float32(sub / 240))
    m_id = nat.ADAPT_VELOCITY if mode == "velocity" else nat.ADAPT_RPYT
    rng = np.random.default_rng(5)
    for seed in (0, 3):
        for pol in (nat.OPT_STREAM_ON, nat.OPT_STREAM_OFF):
            for with_obs in (True, False):
                for arows in (True, False):
                    rigid, mem, _ = random_fleet(rng, n, n_act=4, tilt=0.3, rate=1.0)
                    rigid, mem = f32(rigid), f32(mem)
                    st = fleet.FleetState(ctx, n, "tile64")
                    assert st.n_pad % 256 == 0
                    st.load_aos(rigid, mem)
                    if mode == "velocity":
                        act = np.concatenate([rng.uniform(-1, 1, (n, 3)), rng.uniform(0, 0.3, (n, 1))], 1)
                        act[0, 0:3] = 0.0
                    else:
                        act = np.concatenate([rng.uniform(-0.5, 0.5, (n, 3)), rng.uniform(0.3, 0.6, (n, 1))], 1)
                    act = f32(act)
                    if arows:
                        adev = torch.from_numpy(act.astype(np.float32)).to(ctx.device).contiguous()
                    else:
                        adev = torch.zeros((4, st.n_pad), device=ctx.device)
                        adev[:, :n] = torch.from_numpy(np.ascontiguousarray(act.T)).float()
                    echo = torch.zeros((4, st.n_pad), device=ctx.device)
                    obs = torch.full((n, 20), -7.0, device=ctx.device)
                    a = _args(nat, sub, DT, dtc, options=pol | (nat.OPT_ACTION_ROWS if arows else 0), seed=seed, step_index=4)
                    if with_obs:
                        a.obs_out, a.obs_width = obs.data_ptr(), 20
                    nat.check(ctx.lib.dsim_step_adaptor(ctx.handle, _stream(ctx), n, st.view(), adev.data_ptr(), m_id,
                                                        echo.data_ptr(), ctypes.byref(a)))
                    torch.cuda.synchronize()
                    got_r, got_m = st.rigid_aos(), st.mem_aos()
                    label = f"adaptor_fast[{mode},{seed},{pol},{with_obs},{arows}]"
                    rc0, m_ref = rigid.copy(), mem.copy()
                    assert O.adaptor_step(0 if mode == "velocity" else 1, rc0, m_ref, act, 0, DT, dtc) == 0
                    tgt = np.concatenate([rigid[:, 0:3], np.zeros((n, 7))], 1)
                    if mode == "velocity":
                        nrm = np.linalg.norm(act[:, 0:3], axis=1, keepdims=True)
                        tgt[:, 3:6] = t.max_speed_kmh / 3.6 * np.abs(act[:, 3:4]) * np.divide(act[:, 0:3], nrm, out=np.zeros((n, 3)),
                                                                                              where=nrm > 0)
                    assert_control_parity(label + " control", [t], None, rigid, mem, tgt, got_m, m_ref, dtc)
                    r_ref = rigid.copy()
                    a6 = np.zeros((n, 6)); a6[:, :4] = got_m[:, 7:11]
                    O.physics(r_ref, got_m.copy(), sub, DT, action=a6, noise=_noise_block(O, [t], None, n, seed, 4, sub) if seed else None)
                    assert_step_parity(label + " physics", [t], None, rigid, got_m, tgt, got_r, None, r_ref, None, DT, dtc, sub,
                                       control=False, action=got_m[:, 7:11], extra_terms=noise_terms([t], None, n, DT, sub) if seed else None)
                    np.testing.assert_array_equal(echo[:, :n].T.cpu().numpy(), got_m[:, 7:11].astype(np.float32))
                    if with_obs:
                        _check_obs_rows(label + " rows", O, obs.double().cpu().numpy(), f32(got_r), a6, None, [t])
                    else:
                        assert float(obs.min()) == -7.0 and float(obs.max()) == -7.0
    # a homogeneous fleet that is NOT in whole tiles (n_pad = 704) goes to the general kernels (k_adaptor<.., UNIFORM = true>);
    # the rows then come from the observation kernel behind the step
    for seed in (0, 3):
        rigid, mem, _ = random_fleet(rng, n, n_act=4, tilt=0.3, rate=1.0)
        rigid, mem = f32(rigid), f32(mem)
        st = fleet.FleetState(ctx, n, "soa", 64)
        assert st.n_pad % 256 != 0
        st.load_aos(rigid, mem)
        act = f32(np.concatenate([rng.uniform(-0.5, 0.5, (n, 3)), rng.uniform(0.05, 0.3, (n, 1))], 1))
        adev = torch.zeros((4, st.n_pad), device=ctx.device)
        adev[:, :n] = torch.from_numpy(np.ascontiguousarray(act.T)).float()
        echo = torch.zeros((4, st.n_pad), device=ctx.device)
        obs = torch.full((n, 20), -7.0, device=ctx.device)
        a = _args(nat, sub, DT, dtc, seed=seed, step_index=4)
        a.obs_out, a.obs_width = obs.data_ptr(), 20
        nat.check(ctx.lib.dsim_step_adaptor(ctx.handle, _stream(ctx), n, st.view(), adev.data_ptr(), m_id, echo.data_ptr(), ctypes.byref(a)))
        torch.cuda.synchronize()
        got_r, got_m = st.rigid_aos(), st.mem_aos()
        rc0, m_ref = rigid.copy(), mem.copy()
        assert O.adaptor_step(0 if mode == "velocity" else 1, rc0, m_ref, act, 0, DT, dtc) == 0
        tgt = np.concatenate([rigid[:, 0:3], np.zeros((n, 7))], 1)
        if mode == "velocity":
            nrm = np.linalg.norm(act[:, 0:3], axis=1, keepdims=True)
            tgt[:, 3:6] = t.max_speed_kmh / 3.6 * np.abs(act[:, 3:4]) * np.divide(act[:, 0:3], nrm, out=np.zeros((n, 3)), where=nrm > 0)
        assert_control_parity(f"adaptor_gen[{mode},{seed}] control", [t], None, rigid, mem, tgt, got_m, m_ref, dtc)
        r_ref = rigid.copy()
        a6 = np.zeros((n, 6)); a6[:, :4] = got_m[:, 7:11]
        O.physics(r_ref, got_m.copy(), sub, DT, action=a6, noise=_noise_block(O, [t], None, n, seed, 4, sub) if seed else None)
        assert_step_parity(f"adaptor_gen[{mode},{seed}] physics", [t], None, rigid, got_m, tgt, got_r, None, r_ref, None, DT, dtc, sub,
                           control=False, action=got_m[:, 7:11], extra_terms=noise_terms([t], None, n, DT, sub) if seed else None)
        _check_obs_rows(f"adaptor_gen[{mode},{seed}] rows", O, obs.double().cpu().numpy(), f32(got_r), a6, None, [t])
    # the general kernels take the action field-major only
    t2 = params.builtin_type("tello")
    ctx2 = fleet.Context([t, t2])
    st2 = fleet.FleetState(ctx2, 512, "tile64")
    tid = torch.zeros(st2.n_pad, dtype=torch.uint8, device=ctx2.device)
    a = _args(nat, 1, DT, DT, options=nat.OPT_ACTION_ROWS, type_id=tid)
    adev = torch.zeros((512, 4), device=ctx2.device)
    assert ctx2.lib.dsim_step_adaptor(ctx2.handle, _stream(ctx2), 512, st2.view(), adev.data_ptr(), m_id, None, ctypes.byref(a)) == -5     # DSIM_E_UNSUPPORTED
    ctx.close(); ctx2.close()


def test_env_step_takes_the_action_rows_as_the_caller_holds_them(gpu):
    """CtrlAviary.step(action): an [N, 4] float32 device tensor goes to the launch as it is (DSIM_OPT_ACTION_ROWS,
    k_physics_fast with StepK.action_rows), anything else through the env's field-major buffer — same state, same rows, bit for bit;
    the caller's tensor is not touched."""
    from dronesim_amd.envs import CtrlAviary
    n = 1024
    rng = np.random.default_rng(3)
    xyz = np.stack([rng.uniform(-5, 5, n), rng.uniform(-5, 5, n), rng.uniform(1, 5, n)], 1)
    envs = [CtrlAviary(["robobee"], n, initial_xyzs=xyz, aggregate_phy_steps=2, noise_seed=5, dict_io=False, layout="tile64")
            for _ in range(2)]
    for k in range(4):
        act = rng.uniform(0.2, 0.8, (n, 4)).astype(np.float32)
        act[0] = [-1.0, 2.0, 0.5, 0.5]                        # clipped inside (CtrlAviary.py:258-263)
        dev = torch.from_numpy(act).to(envs[0].ctx.device)
        keep = dev.clone()
        o0 = envs[0].step(dev)[0]
        assert envs[0]._rows_in
        o1 = envs[1].step(act)[0]
        assert not envs[1]._rows_in
        assert torch.equal(o0, o1) and torch.equal(dev, keep)
        np.testing.assert_array_equal(envs[0].state.rigid_aos(), envs[1].state.rigid_aos())
        np.testing.assert_array_equal(o0[:, 16:20].cpu().numpy(), np.clip(act, 0.0, 1.0))
    for e in envs:
        e.close()


@pytest.mark.gpu
def test_env_step_of_a_quad_fleet_over_several_substeps_every_looped_instance(gpu):
    """Env.step of a homogeneous quad fleet in whole tiles with the examples' five sub-steps (examples/fly_INDI.py:139-141,
    BaseAviary.py:510-545) runs on the LOOPED instances of k_physics_fast (body-frame sub-step loop, Box-Muller tables):
    noise on / off x streaming policy x rows fused or not, against the oracle at the step bar; the fine lattice and a single
    sub-step keep to the plain instances (checked through the same bar); a zero-sub-step pass changes nothing."""
    nat, fleet = gpu
    from tests.test_gpu_parity import _noise_block
    from tests.util import noise_terms
    t = params.builtin_type("robobee")
    ctx = fleet.Context([t])
    O = orc.Oracle([t])
    n, rng = 512, np.random.default_rng(77)
    for sub in (5, 2, 1):
        for seed in (0, 5):
            for pol in (nat.OPT_STREAM_ON, nat.OPT_STREAM_OFF):
                for with_obs in (True, False):
                    for fine in ((False, True) if (seed and sub == 5) else (False,)):
                        rigid, mem, _ = random_fleet(rng, n, n_act=4, tilt=0.3, rate=1.0)
                        rigid, mem = f32(rigid), f32(mem)
                        st = fleet.FleetState(ctx, n, "tile64")
                        st.load_aos(rigid, mem)
                        act = f32(rng.uniform(-0.1, 1.1, (n, 4)))
                        adev = torch.zeros((4, st.n_pad), device=ctx.device)
                        adev[:, :n] = torch.from_numpy(np.ascontiguousarray(act.T)).float()
                        echo = torch.zeros((4, st.n_pad), device=ctx.device)
                        obs = torch.full((n, 20), -7.0, device=ctx.device)
                        a = _args(nat, sub, DT, DT * sub, options=pol | (nat.OPT_NOISE_FINE if fine else 0), seed=seed, step_index=3,
                                  action=adev)
                        if with_obs:
                            a.obs_out, a.obs_width = obs.data_ptr(), 20
                        nat.check(ctx.lib.dsim_physics(ctx.handle, _stream(ctx), n, st.view(), echo.data_ptr(), ctypes.byref(a)))
                        torch.cuda.synchronize()
                        got_r = st.rigid_aos()
                        label = f"env_step_looped[{sub},{seed},{pol},{with_obs},{fine}]"
                        a6 = np.zeros((n, 6)); a6[:, :4] = np.clip(act, 0, 1)
                        nz = None
                        if seed:
                            if fine:
                                nz = np.zeros((n, sub, 12))
                                for i in range(n):
                                    for s_ in range(sub):
                                        u = O.noise_normals(seed, i, 3 * sub + s_, 4, fine=True)
                                        nz[i, s_, 0:4], nz[i, s_, 6:10] = u[0:4] * 0.01, u[4:8] * 0.001
                            else:
                                nz = _noise_block(O, [t], None, n, seed, 3, sub)
                        r_ref = rigid.copy()
                        O.physics(r_ref, mem.copy(), sub, DT, action=a6, noise=nz)
                        tgt = np.concatenate([rigid[:, 0:3], np.zeros((n, 7))], 1)
                        assert_step_parity(label, [t], None, rigid, mem, tgt, got_r, None, r_ref, None, DT, DT * sub, sub,
                                           control=False, action=a6[:, :4], extra_terms=noise_terms([t], None, n, DT, sub) if seed else None)
                        np.testing.assert_array_equal(echo[:, :n].T.cpu().numpy(), a6[:, :4].astype(np.float32))
                        np.testing.assert_array_equal(st.mem_aos(), mem)                  # controller memory untouched
                        if with_obs:
                            _check_obs_rows(label + " rows", O, obs.double().cpu().numpy(), f32(got_r), a6, None, [t])
                        else:
                            assert float(obs.min()) == -7.0 and float(obs.max()) == -7.0
    # the neutral pass of the placement trials: zero sub-steps write the state back bit for bit (the plain instance)
    rigid, mem, _ = random_fleet(rng, n, n_act=4, tilt=0.3, rate=1.0)
    st = fleet.FleetState(ctx, n, "tile64")
    st.load_aos(f32(rigid), f32(mem))
    before = st.rigid_aos()
    a = _args(nat, 0, DT, DT, seed=5)
    nat.check(ctx.lib.dsim_physics(ctx.handle, _stream(ctx), n, st.view(), None, ctypes.byref(a)))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(st.rigid_aos(), before)
    ctx.close()
