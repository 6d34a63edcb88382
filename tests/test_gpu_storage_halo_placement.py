"""GPU parity tests of fleet STORAGE and SHARDING (HIP path through the C-ABI vs the fp64 oracle, tests/util.py's bars):
transparent type-major storage of interleaved fleets, runs that share tiles, the device-paced halo exchange at BASELINE
config 5's real shard size, the deferred WLS fallback pass, the fused observation store, and the staleness rules of the
pre-binned neighbour grid.
"""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from dronesim_amd import params  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from tests.util import (K_ULP, MEM_SCALE, REL_TOL, RIGID_SCALE, assert_downwash, assert_step_parity, f32, random_fleet, rotor_noise,  # noqa: E402
                        rel_err, ulp32)

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


def _noise(O, types, tid, n, seed, step_index, sub):
    """[n, sub, 12] scaled normals of the in-kernel generator, keyed by the CALLER's drone index."""
    nz = np.zeros((n, sub, 12))
    for i in range(n):
        na = types[tid[i]].n_act
        for s_ in range(sub):
            u = O.noise_normals(seed, i, step_index * sub + s_, na, fine=(sub == 1))
            nz[i, s_, 0:na], nz[i, s_, 6:6 + na] = rotor_noise(types[tid[i]], u)
    return nz


# ---------------------------------------------------------------------------------------------------------------------
# transparent type-major storage
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("layout", ["soa", "tile64"])
@pytest.mark.parametrize("sub", [1, 2])
def test_interleaved_fleet_is_stored_type_major_behind_the_callers_numbering(gpu, sub, layout):
    """CtrlAviary is handed three airframes in RANDOM order.  It stores the fleet type-major (fleet.StorageOrder: a
    permutation, no padding slots — the type runs begin and end inside tiles) and steps every type with the single-type
    kernel of its kind, while initial positions, explicit actions, targets, state accessors, observation rows and the
    rotor-noise streams keep the caller's numbering: every step against the oracle run in the CALLER's order with the
    noise keyed by the caller's index, and against the same fleet kept in the caller's order (storage="caller": the
    mixed-fleet kernel)."""
    nat, fleet = gpu
    from dronesim_amd.envs import CtrlAviary
    from dronesim_amd.fleet import Targets
    models = ["robobee", "hexa_6DOF", "tello"]
    types = [params.builtin_type(m) for m in models]
    n, seed = 1500, 77
    rng = np.random.default_rng(300 + sub)
    tid = rng.integers(0, 3, n).astype(np.uint8)
    xyz = f32(np.stack([rng.uniform(0, 40, n), rng.uniform(0, 40, n), rng.uniform(2, 12, n)], 1))
    rpy = f32(rng.uniform(-0.2, 0.2, (n, 3)))
    tgt = f32(np.concatenate([xyz + rng.uniform(-1, 1, (n, 3)), rng.uniform(-0.3, 0.3, (n, 3)), rng.uniform(-0.3, 0.3, (n, 3)),
                              rng.uniform(-2, 2, (n, 1))], 1))
    O = orc.Oracle(types)
    dtc = float(np.float32(sub / 240))
    envs = {}
    for storage in ("auto", "caller"):
        env = CtrlAviary(models, n, initial_xyzs=xyz, initial_rpys=rpy, aggregate_phy_steps=sub, noise_seed=seed,
                         dict_io=False, type_ids=tid, layout=layout, storage=storage)
        tg = Targets(env.ctx, n, layout)
        tg.set_fields(0, torch.from_numpy(np.ascontiguousarray(tgt.T)))
        envs[storage] = (env, tg)
    auto = envs["auto"][0]
    assert auto.order is not None and envs["caller"][0].order is None
    assert len(auto._runs) == 3 and [r.type for r in auto._runs] == [0, 1, 2]
    assert any(r.first % 256 for r in auto._runs)                     # the runs really do share tiles
    np.testing.assert_array_equal(auto.state.rigid_aos()[:, 0:3], xyz)          # accessors speak the caller's numbering
    act0 = f32(rng.uniform(0.35, 0.6, (n, 6)))
    for storage, (env, tg) in envs.items():
        for k in range(4):
            r0, m0 = env.state.rigid_aos(), env.state.mem_aos()
            action = act0 if k == 0 else None                         # explicit action: the general kernel; then the runs
            env.step_fused(tg, action=act0.astype(np.float32) if k == 0 else None)
            r1, m1 = r0.copy(), m0.copy()
            nz = _noise(O, types, tid, n, seed, k, sub)
            assert O.step(r1, m1, tgt, sub, DT, dtc, noise=nz, type_id=tid, action=action) == 0
            assert_step_parity(f"storage[{storage},{sub},{layout}]", types, tid, r0, m0, tgt, env.state.rigid_aos(),
                               env.state.mem_aos(), r1, m1, DT, dtc, sub, action=action, noise=True)
    ra, ma = envs["auto"][0].state.rigid_aos(), envs["auto"][0].state.mem_aos()
    rc, mc = envs["caller"][0].state.rigid_aos(), envs["caller"][0].state.mem_aos()
    # two differently compiled kernels of the same laws, same noise streams: equal up to fp32 contraction
    assert rel_err(ra, rc, RIGID_SCALE).max() < 0.2 * REL_TOL and rel_err(ma, mc, MEM_SCALE).max() < 0.5 * REL_TOL
    # observation rows (caller order) against the oracle's state vector of the caller-ordered state
    rows = auto.observe().double().cpu().numpy()
    ref = O.state_vector(ra, ma[:, 7:13], type_id=tid)
    np.testing.assert_array_equal(rows[:, 0:7], ref[:, 0:7])
    np.testing.assert_array_equal(rows[:, 10:16], ref[:, 10:16])
    for i in range(n):
        na = types[tid[i]].n_act
        np.testing.assert_array_equal(rows[i, 16:16 + na], ref[i, 16:16 + na])
    # Env.step(action) in the caller's order, rows back in the caller's order
    act = torch.from_numpy(act0.astype(np.float32)).to(auto.ctx.device)
    r0 = auto.state.rigid_aos()
    obs, _, _, _ = auto.step(act)
    obs = obs.double().cpu().numpy()
    np.testing.assert_array_equal(obs[:, 0:3], auto.state.rigid_aos()[:, 0:3])
    for i in (0, 1, 2, n - 1):
        na = types[tid[i]].n_act
        clip = np.clip(act0[i, :na], np.asarray(types[tid[i]].pwm_min)[:na], np.asarray(types[tid[i]].pwm_max)[:na])
        np.testing.assert_allclose(obs[i, 16:16 + na], clip, rtol=0, atol=0)
    assert np.abs(auto.state.rigid_aos()[:, 0:3] - r0[:, 0:3]).max() > 1e-5
    for env, _ in envs.values():
        env.close()


def test_reordered_fleet_waypoints_neighbours_controller_and_logger(gpu, golden_dir, tmp_path):
    """The other per-drone surfaces of a fleet stored type-major: waypoint counters and offsets, neighbour lists, the bound
    controller's return triple (and its command handed back as the next action), the device Logger."""
    nat, fleet = gpu
    from dronesim_amd.control import INDIControl
    from dronesim_amd.envs import CtrlAviary
    from dronesim_amd.fleet import WaypointTargets, frozen
    from dronesim_amd.utils.Logger import Logger
    g = np.load(os.path.join(golden_dir, "traj_track_waypoints.npz"))
    n = 700
    rng = np.random.default_rng(5)
    tid = rng.integers(0, 2, n).astype(np.uint8)                      # two QUAD types: the quad controller surface applies
    off = np.stack([rng.uniform(0, 30, n), rng.uniform(0, 30, n), np.zeros(n)], 1)
    xyz = g["gates"][0][None, :] + off
    wp0 = rng.integers(0, g["target_pos"].shape[0], n)
    out = {}
    for storage in ("auto", "caller"):
        env = CtrlAviary(["robobee", "tello"], n, initial_xyzs=xyz, aggregate_phy_steps=2, noise_seed=5, dict_io=False,
                         type_ids=tid, storage=storage, neighbourhood_radius=2.5, neighbors_k=6)
        wt = WaypointTargets(env.ctx, n, g["target_pos"], g["target_vel"], g["target_acc"], g["target_yaw"], wp_counters=wp0,
                             offsets=off)
        for _ in range(5):
            env.step_fused(wt)
        cnt, lst = env.neighbors()
        out[storage] = (env.state.rigid_aos(), env.state.mem_aos(), cnt.cpu().numpy(), lst.cpu().numpy(), env)
    ra, ma, ca, la, env_a = out["auto"]
    rc, mc, cc, lc, env_c = out["caller"]
    assert env_a.order is not None
    assert rel_err(ra, rc, RIGID_SCALE).max() < 0.2 * REL_TOL and rel_err(ma, mc, MEM_SCALE).max() < 0.5 * REL_TOL
    # neighbour counts against the O(N^2) rule on the caller-ordered positions; the lists name drones, not slots
    d = np.linalg.norm(ra[:, None, 0:3] - ra[None, :, 0:3], axis=2)
    adj = (d < 2.5) & ~np.eye(n, dtype=bool)
    edge = np.abs(d - 2.5) < 1e-4
    ok = ~edge.any(1)
    np.testing.assert_array_equal(ca[ok], adj.sum(1)[ok])
    for i in np.flatnonzero(ok)[:200]:
        got = set(int(x) for x in la[:, i] if x >= 0)
        assert got <= set(np.flatnonzero(adj[i])) and len(got) == min(6, adj[i].sum())
    # the bound controller: caller-ordered triple, and its command taken back as the action without a copy
    for env in (env_a, env_c):
        ctrl = INDIControl("robobee", env=env)
        tpos = frozen(torch.from_numpy(f32(xyz).astype(np.float32)).to(env.ctx.device))
        cmd, pos_e, yaw_e = ctrl.computeControlFromState(2 * DT, None, target_pos=tpos, target_rpy=np.array([0, 0, 0.3]))
        env._ctrl_out = (cmd.cpu().numpy(), pos_e.cpu().numpy(), yaw_e.cpu().numpy())
        obs, _, _, _ = env.step(cmd)
        env._obs_after = obs.state.cpu().numpy()                      # (neighbors_k > 0: a FleetObs)
        if env.order is not None and not env._caller_io:
            assert env._cmd_token is not None and env._cmd_token[0] is cmd
        if env._caller_io:          # the command came back in the caller's numbering straight from the launch: a view, no copy
            assert cmd.data_ptr() == ctrl._cmd.data_ptr()
    for a_, c_ in zip(env_a._ctrl_out, env_c._ctrl_out):
        np.testing.assert_allclose(a_, c_, rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(env_a._ctrl_out[1], f32(xyz) - rc[:, 0:3], rtol=0, atol=2e-4)      # pos_e really is per caller drone
    np.testing.assert_allclose(env_a._obs_after, env_c._obs_after, rtol=2e-4, atol=2e-5)
    # Logger slabs come back per caller drone
    lg = Logger(120, env_a, duration_sec=1)
    lg.log(0.0, control=torch.arange(12 * n, dtype=torch.float32).reshape(12, n))
    ts, st, ct = lg.arrays()
    np.testing.assert_allclose(st[:, 0:3, 0], env_a.state.rigid_aos()[:, 0:3], rtol=0, atol=0)
    np.testing.assert_array_equal(ct[:, 0, 0], np.arange(n))
    env_a.close(); env_c.close()


def test_frozen_targets_skip_the_copy_and_plain_tensors_never_do(gpu):
    """Targets.set used to skip the copy of a device tensor whose torch write counter had not moved — but the
    library's own kernels write through raw pointers.  A plain tensor is now always copied (a view of the state block
    handed over every iteration follows the drones); only fleet.frozen(t) — the caller's promise — skips."""
    nat, fleet = gpu
    from dronesim_amd.envs import CtrlAviary
    from dronesim_amd.fleet import Targets, frozen
    n = 512
    xyz = np.stack([np.arange(n) % 32, np.arange(n) // 32, np.full(n, 3.0)], 1).astype(np.float64)
    env = CtrlAviary(["robobee"], n, initial_xyzs=xyz, noise_seed=3, dict_io=False)
    tg = Targets(env.ctx, n)
    tg.set(pos=xyz.T.astype(np.float32), yaw=0.1)
    follow = Targets(env.ctx, n)
    live = env.state.fields(0, 3)                                     # a view of the state block (plain SoA, no storage order)
    assert live.data_ptr() == env.state.data.data_ptr()
    pinned = frozen(live)
    held = Targets(env.ctx, n)
    follow.set(pos=live); held.set(pos=pinned)
    for _ in range(3):
        env.step_fused(tg, action=np.full((n, 4), 0.6, dtype=np.float32))
    torch.cuda.synchronize()
    assert (live.cpu().numpy()[2] != 3.0).all()                       # the kernels moved the drones, torch saw no write
    follow.set(pos=live); held.set(pos=pinned)
    np.testing.assert_array_equal(follow.fields(0, 3).cpu().numpy(), live.cpu().numpy())
    np.testing.assert_array_equal(held.fields(0, 3).cpu().numpy()[2], np.full(n, np.float32(3.0)))
    env.close()


# ---------------------------------------------------------------------------------------------------------------------
# WLS fallback: command output of computeControl, explicit pass
# ---------------------------------------------------------------------------------------------------------------------
def test_hexa_fallback_commands_reach_cmd_out_and_the_deferred_pass(gpu):
    """dsim_control2's cmd_out — computeControl's first return value, which Env.step takes back as the
    action — must carry the commands the active-set fallback pass solves, not the pre-allocation values; and
    DSIM_OPT_DEFER_FALLBACK + dsim_wls_fallback is the same computation in two calls."""
    nat, fleet = gpu
    n = 4096
    t = params.builtin_type("hexa_6DOF")
    rng = np.random.default_rng(41)
    rigid, mem, tgt = random_fleet(rng, n, n_act=6, tilt=0.3, rate=1.0)
    mem[:, 7:13] = f32(rng.uniform(0.0, 1.0, (n, 6)))
    tgt[:, 0:3] = f32(rigid[:, 0:3] + rng.uniform(-40, 40, (n, 3)))
    rigid[:, 10:13] = f32(rng.uniform(-25, 25, (n, 3)))               # violent rates: the first WLS iteration leaves the box
    res = {}
    for mode in ("inline", "deferred"):
        ctx = fleet.Context([t])
        st, tg = fleet.FleetState(ctx, n), fleet.Targets(ctx, n)
        st.load_aos(rigid, mem)
        tg.set_fields(0, torch.from_numpy(np.ascontiguousarray(tgt.T)))
        cmd_out = torch.full((6, st.n_pad), -1.0, device=ctx.device)
        a = nat.StepArgs()
        a.phys_substeps, a.dt_phys, a.dt_ctrl, a.options = 0, DT, DT, (nat.OPT_DEFER_FALLBACK if mode == "deferred" else 0)
        sp = ctx.stream_ptr()
        nat.check(ctx.lib.dsim_control2(ctx.handle, sp, n, st.view(), tg.view(), ctypes.byref(a), None, None, cmd_out.data_ptr()))
        if mode == "deferred":
            before = st.mem_aos()[:, 7:13]
            assert ctx.query(nat.QUERY_WLS_FALLBACKS) == 0            # queued, not solved yet
            nat.check(ctx.lib.dsim_wls_fallback(ctx.handle, sp, n, st.view(), None, cmd_out.data_ptr()))
            assert np.abs(st.mem_aos()[:, 7:13] - before).max() > 1e-3
        fb = ctx.query(nat.QUERY_WLS_FALLBACKS)
        assert fb > n // 20 and ctx.query(nat.QUERY_WLS_FAILURES) == 0
        got = st.mem_aos()[:, 7:13]
        np.testing.assert_array_equal(cmd_out[:, :n].T.double().cpu().numpy(), got)      # every drone, fallback or not
        res[mode] = (got, fb)
        ctx.close()
    np.testing.assert_array_equal(res["inline"][0], res["deferred"][0])
    assert res["inline"][1] == res["deferred"][1]
    # the per-case bound on the fallback drones: the oracle's own allocation evaluated on the
    # virtual control nu rounded one fp32 ulp up and down brackets what a faithful fp32 evaluation of nu can return
    O = orc.Oracle([t])
    o = mem.copy()
    assert O.control(rigid, o, tgt, DT)[0] == 0
    err = np.abs(res["inline"][0] - o[:, 7:13])
    clipped = ((res["inline"][0] <= 0) | (res["inline"][0] >= 1)) & ((o[:, 7:13] <= 0) | (o[:, 7:13] >= 1))
    assert (err[clipped] == 0).all() and np.median(err) < 1e-5


# ---------------------------------------------------------------------------------------------------------------------
# fused observation store of Env.step
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [256, 1000, 4096 + 37])
@pytest.mark.parametrize("stream", ["on", "off"])
def test_env_step_observation_rows_of_ragged_fleets(gpu, n, stream):
    """k_physics_fast writes the [n, 20] rows from wave-private LDS blocks as 16-byte pieces; a last tile that ends in
    the middle of a wave (and of a 16-byte piece's row) writes exactly n rows: the rows equal dsim_observe's, and the
    guard words behind row n stay untouched."""
    nat, fleet = gpu
    t = params.builtin_type("robobee")
    ctx = fleet.Context([t])
    st = fleet.FleetState(ctx, n, "tile64")
    rng = np.random.default_rng(n)
    rigid, mem, _ = random_fleet(rng, n)
    st.load_aos(rigid, mem)
    obs = torch.full((n + 8, 20), 7.5, device=ctx.device)
    act = torch.from_numpy(f32(rng.uniform(-0.2, 1.2, (4, st.n_pad))).astype(np.float32)).to(ctx.device)
    echo = torch.zeros((4, st.n_pad), device=ctx.device)
    a = nat.StepArgs()
    a.phys_substeps, a.dt_phys, a.dt_ctrl = 2, DT, 2 * DT
    a.options = nat.OPT_STREAM_ON if stream == "on" else nat.OPT_STREAM_OFF
    a.noise_seed, a.step_index, a.action = 11, 4, act.data_ptr()
    a.obs_out, a.obs_width = obs.data_ptr(), 20
    nat.check(ctx.lib.dsim_physics(ctx.handle, ctx.stream_ptr(), n, st.view(), echo.data_ptr(), ctypes.byref(a)))
    ref = torch.zeros((n, 20), device=ctx.device)
    nat.check(ctx.lib.dsim_observe(ctx.handle, ctx.stream_ptr(), n, st.view(), echo.data_ptr(), ref.data_ptr(), 20))
    torch.cuda.synchronize()
    got, want = obs.cpu().numpy(), ref.cpu().numpy()
    np.testing.assert_array_equal(got[n:], np.full((8, 20), np.float32(7.5)))
    cp = [c for c in range(20) if not 7 <= c < 10]
    np.testing.assert_array_equal(got[:n, cp], want[:, cp])           # copies: bit for bit
    # the Euler angles: two compilations of the same polynomial atan2 / asin (each held to the oracle elsewhere) agree to
    # the last place or two, except where asin is ill-conditioned (|pitch| near 90 degrees)
    well = np.abs(want[:, 8]) < 1.5
    assert np.abs(got[:n, 7:10] - want[:, 7:10])[well].max() <= 4 * np.spacing(np.float32(3.2))
    # the echoed action is the CLIPPED one
    np.testing.assert_array_equal(got[:n, 16:20], np.clip(act[:, :n].cpu().numpy().T, 0.0, 1.0))
    ctx.close()


# ---------------------------------------------------------------------------------------------------------------------
# the pre-binned neighbour grid goes stale by the library's own record
# ---------------------------------------------------------------------------------------------------------------------
def test_library_drops_a_prebinned_grid_when_positions_moved_behind_it(gpu):
    """dsim_step(bin_next) fills the next neighbour grid.  If ANOTHER library call then moves the drones (dsim_physics
    here, called straight through the C-ABI so that no host class can warn the library), or a second dsim_step bins
    again before any dsim_downwash consumed the first generation, a dsim_downwash that claims prebinned = 1 must still
    return the brute-force force on the CURRENT positions."""
    nat, fleet = gpu
    from dronesim_amd.envs import CtrlAviary, Physics
    from dronesim_amd.fleet import Targets
    n = 1024
    rng = np.random.default_rng(61)
    xyz = np.stack([rng.uniform(0, 28, n), rng.uniform(0, 28, n), rng.uniform(0.5, 9, n)], 1)
    tid = (np.arange(n) % 2).astype(np.uint8)
    env = CtrlAviary(["robobee", "hexa_6DOF"], n, initial_xyzs=xyz, physics=Physics.PYB_DW, noise_seed=0, dict_io=False,
                     type_ids=tid, storage="caller")
    tg = Targets(env.ctx, n); tg.set(pos=f32(xyz).T + np.array([[1.0], [0.5], [0.3]], dtype=np.float32), yaw=0.0)
    O = orc.Oracle(env.types)
    lib, h = env.ctx.lib, env.ctx.handle

    def force_ok(label):
        f = env._downwash.compute().cpu().numpy()[2, :n]
        r = env.state.rigid_aos()
        ref = O.downwash(r, r[:, 0:3], type_id=tid)
        assert_downwash(label, f, ref, env.types, tid, r[:, 0:3], r[:, 0:3])

    for _ in range(3):
        env.step_fused(tg)
    assert env._downwash._prebin_version is not None
    # (1) a physics-only call through the bare ABI moves the drones; the host class still vouches for the grid
    a = env.step_args()                                               # (runs compute(): consumes the grid; the next step re-bins)
    a.bin_next = env._downwash.bin_next_ptr()
    nat.check(lib.dsim_step(h, env.ctx.stream_ptr(), n, env.state.view(), tg.view(), ctypes.byref(a)))
    env._join_fallback()
    p = nat.StepArgs()
    p.phys_substeps, p.dt_phys, p.dt_ctrl, p.type_id = 6, DT, DT, env._type_id.data_ptr()
    nat.check(lib.dsim_physics(h, env.ctx.stream_ptr(), n, env.state.view(), None, ctypes.byref(p)))
    assert env._downwash._prebin_version == env.state.version         # nobody told the host class
    force_ok("stale prebin: physics behind it")
    # (2) two binning steps in a row: the second generation replaces the first
    for _ in range(2):
        a = env.step_args() if _ == 0 else a
        a.bin_next = env._downwash.bin_next_ptr()
        a.ext_force = env._downwash.force.data_ptr()
        nat.check(lib.dsim_step(h, env.ctx.stream_ptr(), n, env.state.view(), tg.view(), ctypes.byref(a)))
        env._join_fallback()
    force_ok("stale prebin: two generations")
    env.close()


# ---------------------------------------------------------------------------------------------------------------------
# halo exchange: device-side lists, split-phase query, BASELINE config 5 at its real shard size
# ---------------------------------------------------------------------------------------------------------------------
def test_halo_pack_selection_and_split_phase_query_against_the_one_grid_form(gpu):
    """One process plays both sides of a slab boundary through the bare C-ABI: dsim_fleet_bounds against numpy;
    dsim_halo_pack's selection against the rule in numpy (every drone inside the peer's box — as the peer's last header
    reported it — grown by reach is shipped exactly once, nobody else; the header carries the count and the sender's own
    box), a capacity that is too small (dropped AND counted, on both sides); and the split form of the query (LOCAL +
    HALO_BIN + HALO_QUERY) against the one-grid form (ALL) and the brute-force oracle — in a dense world (height-banded
    query) and a sparse one."""
    nat, fleet = gpu
    types = [params.builtin_type("robobee"), params.builtin_type("hexa_6DOF")]
    O = orc.Oracle(types)
    HDR = nat.HALO_HDR
    for density, n, width, depth in (("dense", 6000, 60.0, 100.0), ("sparse", 3000, 150.0, 200.0)):
        rng = np.random.default_rng(17 if density == "dense" else 18)
        fleets = []
        for r in range(2):
            rigid, mem, _ = random_fleet(rng, n, n_act=6)
            rigid[:, 0] = f32(rng.uniform(r * width, (r + 1) * width, n)); rigid[:, 1] = f32(rng.uniform(0, depth, n))
            rigid[:, 2] = f32(rng.uniform(0.5, 20.5, n))
            rigid[:, 7:10] = f32(rng.uniform(-1.5, 1.5, (n, 3)))
            tid = (rng.random(n) < 0.5).astype(np.uint8)
            ctx = fleet.Context(types)
            st = fleet.FleetState(ctx, n, "tile64")
            st.load_aos(rigid, mem)
            tdev = torch.zeros(st.n_pad, dtype=torch.uint8, device=ctx.device); tdev[:n] = torch.from_numpy(tid)
            fleets.append(dict(ctx=ctx, st=st, rigid=rigid, tid=tid, tdev=tdev))
        dev = fleets[0]["ctx"].device
        bounds = torch.zeros((2, 5), device=dev)
        for r, f in enumerate(fleets):
            nat.check(f["ctx"].lib.dsim_fleet_bounds(f["ctx"].handle, f["ctx"].stream_ptr(), n, f["st"].view(), bounds[r].data_ptr()))
        torch.cuda.synchronize()
        b = bounds.cpu().numpy()
        for r, f in enumerate(fleets):
            p = f["rigid"]
            np.testing.assert_array_equal(b[r], np.array([p[:, 0].min(), p[:, 1].min(), p[:, 0].max(), p[:, 1].max(),
                                                          np.abs(p[:, 7:10]).max()], dtype=np.float32))
        reach = np.float32(10.0 + 100.0 / 240.0)
        plans = []
        for r, f in enumerate(fleets):
            cap = f["st"].n_pad
            stride = HDR + 3 * cap
            send = torch.full((2, stride), -7.0, device=dev)
            recv = torch.zeros((2, stride), device=dev)
            scratch = torch.zeros(32, dtype=torch.int32, device=dev)
            peer = 1 - r
            recv[peer, 1:5] = bounds[peer, 0:4]                       # "the peer's last header"
            pl = nat.HaloPlan()
            pl.world, pl.rank, pl.cap = 2, r, cap
            pl.send, pl.recv, pl.scratch = send.data_ptr(), recv.data_ptr(), scratch.data_ptr()
            pl.send_cap[peer], pl.reach[peer] = cap, float(reach)
            lib, h, sp = f["ctx"].lib, f["ctx"].handle, f["ctx"].stream_ptr()
            for rep in range(2):                                      # (twice: the scratch resets itself)
                nat.check(lib.dsim_halo_pack(h, sp, n, f["st"].view(), ctypes.byref(pl)))
            torch.cuda.synchronize()
            hdr = send[peer, :HDR].cpu().numpy()
            c = int(hdr[0:1].view(np.int32)[0])
            np.testing.assert_array_equal(hdr[1:6], b[r])             # the sender's own box (and top speed) ride along
            p32 = f["rigid"].astype(np.float32)
            lo, hi = b[peer, 0:2] - reach, b[peer, 2:4] + reach
            inside = (p32[:, 0] >= lo[0]) & (p32[:, 0] <= hi[0]) & (p32[:, 1] >= lo[1]) & (p32[:, 1] <= hi[1])
            assert c == int(inside.sum()) and 0 < c < n
            got = send[peer, HDR: HDR + 3 * c].reshape(c, 3).cpu().numpy()
            key = lambda a_: a_[np.lexsort((a_[:, 2], a_[:, 1], a_[:, 0]))]
            np.testing.assert_array_equal(key(got), key(p32[inside, 0:3]))          # the same set of positions, each once
            assert (send[r].cpu().numpy() == -7.0).all()              # nothing is written for oneself
            assert f["ctx"].query(nat.QUERY_HALO_OVERFLOW) == 0
            # a message that is too small: the header still says how many were selected, the rest is dropped and counted
            pl.send_cap[peer] = 100
            nat.check(lib.dsim_halo_pack(h, sp, n, f["st"].view(), ctypes.byref(pl)))
            torch.cuda.synchronize()
            assert int(send[peer, 0:1].view(torch.int32)[0]) == c and f["ctx"].query(nat.QUERY_HALO_OVERFLOW) == c - 100
            pl.send_cap[peer] = cap
            nat.check(lib.dsim_halo_pack(h, sp, n, f["st"].view(), ctypes.byref(pl)))
            plans.append(dict(pl=pl, send=send, recv=recv, scratch=scratch, cnt=c, lost=c - 100))
        # "the wire": fixed-size messages with headroom, as HaloWire sizes them
        caps = [int(-(-int(pl_["cnt"] * 1.25 + 512) // 256) * 256) for pl_ in plans]
        for r in range(2):
            peer = 1 - r
            k = HDR + 3 * caps[r]
            plans[peer]["recv"][r, :k] = plans[r]["send"][peer, :k]
            plans[peer]["pl"].recv_cap[r] = caps[r]
            plans[r]["pl"].send_cap[peer] = caps[r]
        for r, f in enumerate(fleets):
            ctx, st, pl = f["ctx"], f["st"], plans[r]["pl"]
            peer = 1 - r
            H, cnt = caps[peer], plans[peer]["cnt"]
            halo_pos = plans[r]["recv"][peer, HDR: HDR + 3 * cnt].reshape(cnt, 3).double().cpu().numpy()
            world = np.concatenate([f["rigid"][:, 0:3], halo_pos])
            ref = O.downwash(f["rigid"], world, type_id=f["tid"])
            own = O.downwash(f["rigid"], f["rigid"][:, 0:3], type_id=f["tid"])
            assert np.abs(ref - own).max() > 1e-4                                   # the halo matters
            cell = 5.0 if density == "dense" else 10.0
            xmin, ymin = float(b[r, 0] - reach - cell), float(b[r, 1] - reach - cell)
            nx = int((b[r, 2] + reach + cell - xmin) // cell) + 1; ny = int((b[r, 3] + reach + cell - ymin) // cell) + 1
            assert ctx.lib.dsim_downwash_prebin_ok(n + H, nx, ny) == 1
            ws = torch.empty((ctx.lib.dsim_downwash_workspace_halo(n, H, nx, ny),), dtype=torch.int32, device=dev)
            g = nat.DownwashArgs()
            g.pos_all, g.m, g.m_pad = None, n + H, n + H
            g.xmin, g.ymin, g.cell, g.nx, g.ny = xmin, ymin, cell, nx, ny
            g.workspace, g.workspace_len, g.type_id, g.local_offset = ws.data_ptr(), ws.numel(), f["tdev"].data_ptr(), 0
            g.halo = ctypes.addressof(pl)
            sp = ctx.stream_ptr()
            for rep in range(3):                                       # (several steps: the double-buffered halo counts alternate)
                fa = torch.full((3, st.n_pad), 9.0, device=dev)
                g.phase = nat.DW_ALL
                nat.check(ctx.lib.dsim_downwash(ctx.handle, sp, n, st.view(), ctypes.byref(g), fa.data_ptr()))
                fs = torch.full((3, st.n_pad), 9.0, device=dev)
                for ph in (nat.DW_HALO_BIN, nat.DW_LOCAL, nat.DW_HALO_QUERY):
                    g.phase = ph
                    nat.check(ctx.lib.dsim_downwash(ctx.handle, sp, n, st.view(), ctypes.byref(g), fs.data_ptr()))
                for name, fz in dict(all=fa.cpu().numpy(), split=fs.cpu().numpy()).items():
                    np.testing.assert_array_equal(fz[0:2, :n], 0.0)
                    assert_downwash(f"halo {density} {name}", fz[2, :n], ref, types, f["tid"], f["rigid"][:, 0:3], world)
            # a LOCAL pass alone is the fleet-alone force
            fl = torch.zeros((3, st.n_pad), device=dev)
            g.phase = nat.DW_HALO_BIN
            nat.check(ctx.lib.dsim_downwash(ctx.handle, sp, n, st.view(), ctypes.byref(g), None))
            g.phase = nat.DW_LOCAL
            nat.check(ctx.lib.dsim_downwash(ctx.handle, sp, n, st.view(), ctypes.byref(g), fl.data_ptr()))
            assert_downwash(f"halo {density} local pass", fl.cpu().numpy()[2, :n], own, types, f["tid"], f["rigid"][:, 0:3],
                            f["rigid"][:, 0:3])
            g.phase = nat.DW_HALO_QUERY
            nat.check(ctx.lib.dsim_downwash(ctx.handle, sp, n, st.view(), ctypes.byref(g), fl.data_ptr()))
            # a header that announces more than the message holds is counted on the receiving side too
            before = ctx.query(nat.QUERY_HALO_OVERFLOW)
            plans[r]["recv"][peer, 0:1].view(torch.int32)[0] = H + 9
            g.phase = nat.DW_HALO_BIN
            nat.check(ctx.lib.dsim_downwash(ctx.handle, sp, n, st.view(), ctypes.byref(g), None))
            g.phase = nat.DW_LOCAL
            nat.check(ctx.lib.dsim_downwash(ctx.handle, sp, n, st.view(), ctypes.byref(g), fl.data_ptr()))
            g.phase = nat.DW_HALO_QUERY
            nat.check(ctx.lib.dsim_downwash(ctx.handle, sp, n, st.view(), ctypes.byref(g), fl.data_ptr()))
            assert ctx.query(nat.QUERY_HALO_OVERFLOW) == before + 9
            # argument errors of the split form
            g.phase = 7
            assert ctx.lib.dsim_downwash(ctx.handle, sp, n, st.view(), ctypes.byref(g), fl.data_ptr()) == -1
            g.phase, g.m = nat.DW_LOCAL, n + H + 1
            assert ctx.lib.dsim_downwash(ctx.handle, sp, n, st.view(), ctypes.byref(g), fl.data_ptr()) == -1
            g.m, g.halo = n + H, None
            assert ctx.lib.dsim_downwash(ctx.handle, sp, n, st.view(), ctypes.byref(g), fl.data_ptr()) == -1
            torch.cuda.synchronize()
        for f in fleets:
            f["ctx"].close()


def _config5_worker(rank, world, port, exchange, split, out):
    """One rank of BASELINE config 5 at its real shard size — bench.py's own fleet: 65 536 mixed drones in a
    128 m x 512 m slab, one drone per m^2, interleaved quad / hexa, neighbour downwash — two ranks sharing the one GPU,
    gloo standing in for RCCL (which wants one device per rank)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      DSIM_DW_EXCHANGE=exchange, DSIM_DW_SPLIT="1" if split else "0")
    if exchange == "halo":
        os.environ["DSIM_TEST_RESIZE"] = "16"
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    import bench
    from dronesim_amd import _native as nat
    fl = bench.Fleet(65536, 1, 0, 1, "tile64", 1 + rank, config5=True, dist=dist, rank=rank)
    env = fl.env
    n = env.NUM_DRONES
    if dwn_resize := os.environ.get("DSIM_TEST_RESIZE"):
        env._downwash.halo.resize_every = int(dwn_resize)
    steps = 36                                                        # (two resizes of the message capacities with resize_every = 16)
    for _ in range(steps):
        fl.step()
    torch.cuda.synchronize()
    dwn = env._downwash
    force = dwn.compute().clone()                                     # one more exchange, positions unchanged since
    torch.cuda.synchronize()
    pos = env.state.raw_fields(0, 3).T.contiguous().cpu()             # storage order, like the force and the type ids
    allpos = [torch.zeros_like(pos) for _ in range(world)]
    dist.all_gather(allpos, pos)
    world_pos = torch.cat(allpos).double().numpy()
    rigid = np.zeros((n, 13)); rigid[:, 0:3] = pos.double().numpy(); rigid[:, 6] = 1.0
    tid = env._type_id[:n].cpu().numpy()
    rng = np.random.default_rng(rank)
    # 512 receivers, half of them from the strips next to the other slabs (where the halo decides the answer)
    edges = [128.0 * k for k in (rank, rank + 1) if 0 < k < world]
    near = np.flatnonzero(np.min([np.abs(rigid[:, 0] - e) for e in edges], axis=0) < 8.0)
    sample = np.concatenate([rng.choice(near, 256, replace=False), rng.choice(n, 256, replace=False)])
    O = orc.Oracle(env.types)
    ref = O.downwash(rigid[sample], world_pos, type_id=tid[sample], nthreads=4)
    own = O.downwash(rigid[sample], rigid[:, 0:3], type_id=tid[sample], nthreads=4)
    got = force[2, :n].double().cpu().numpy()[sample]
    res = dict(finite=bool(np.isfinite(env.state.raw_fields(0, env.state.n_fields).cpu().numpy()).all()),
               ground=env.ground_contacts(), wls_fail=env.ctx.query(nat.QUERY_WLS_FAILURES),
               overflow=env.ctx.query(nat.QUERY_HALO_OVERFLOW), halo_matters=float(np.abs(ref - own).max()),
               sent=None, margin=None, reordered=env.order is not None,
               vmax=float(env.state.raw_fields(7, 3).abs().max()))
    try:
        assert_downwash(f"config5 shard {exchange}{'' if split else ' one-grid'}", got, ref, env.types, tid[sample],
                        rigid[sample, 0:3], world_pos)
        res["force_ok"] = True
    except AssertionError as e:
        res["force_ok"] = repr(e)
    if dwn.halo is not None:
        res["sent"], res["margin"] = dwn.halo.sent_per_step, dwn.halo.step_reach - dwn.halo.cutoff
    out[rank] = res
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("exchange,split", [("halo", True), ("halo", False), ("allgather", True)])
def test_config5_two_ranks_at_the_real_shard_size(gpu, exchange, split):
    """config 5 at the size and density it is benchmarked at (65 536 mixed drones per rank in 128 m
    slabs, 25 per 5 m cell of a 64-slot bucket, halo on), both exchange forms: everything finite, no ground contact, no
    WLS failure, no halo message that outgrew its capacity, and the downwash force of 512 receivers — half of them next to
    the slab boundary — equal to the brute-force sum over the WHOLE two-rank world.  (This random world holds
    near-vertical pairs whose P8 term is singular: some drones leave at tens of m/s, res["vmax"]; the selection margin
    assumes nothing about speeds below the integrator's clamp.)"""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    out = mgr.dict()
    world = 3 if (exchange, split) == ("halo", False) else 2           # three ranks: the middle one has TWO peers
    mp.spawn(_config5_worker, args=(world, port, exchange, split, out), nprocs=world, join=True)
    for r in range(world):
        res = out[r]
        assert res["finite"] and res["wls_fail"] == 0 and res["overflow"] == 0, (r, res)
        # (ground contacts: this world starts drones at z = 0.5 m under the downwash of 20 m of others; a few dozen
        # drone-steps of the 2.4 M flown end on the plane, the same ones whatever the exchange form)
        assert res["ground"] < 200 and res["ground"] == out[r]["ground"], (r, res)
        assert res["force_ok"] is True, (r, res)
        assert res["halo_matters"] > 1e-4 and res["reordered"], (r, res)
        if exchange == "halo":
            # the boundary strip only: one drone per m^2 x 512 m x (10 m + 100 m/s x 1/240 s) ~ 5.3 k, not the ~12 k of
            # round 2's sixteen-step margin
            peers = 2 if 0 < r < world - 1 else 1
            assert 4500 * peers < res["sent"] < 6500 * peers and abs(res["margin"] - 100.0 / 240.0) < 1e-6, (r, res)


def test_bench_config5_line_two_gloo_ranks(gpu):
    """bench.py --gpus 2 --workload config5 under DSIM_BENCH_BACKEND=gloo: the line carries the exchange's size and time."""
    env = dict(os.environ, DSIM_BENCH_BACKEND="gloo", PYTHONPATH=ROOT)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "config5", "--steps", "20",
                        "--warmup", "4", "--no-also", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["dist"]["backend"] == "gloo" and d["value"] > 0
    ex = d["exchange"]
    assert ex["form"] == "halo" and 4500 < ex["sent_per_step_max"] < 6500 and ex["bytes_per_step_max"] == 12 * ex["sent_per_step_max"]
    assert ex["overflow"] == 0 and ex["peers_max"] == 1 and ex["exchange_span_us_max"] > 0
    assert d["ranks"]["launch_us_min"] > 0 and d["ranks"]["launch_us_max"] >= d["ranks"]["launch_us_min"]


@pytest.mark.parametrize("world", ["dense", "vast"])          # bucket form (the step kernel fills the next grid) / counting-sort form
def test_graph_replay_of_a_downwash_fleet(gpu, world):
    """capture_fused on a single-rank fleet WITH the neighbour-downwash term (round 2's verdict, item 4c): every captured
    step is query -> step (+ binning ahead) -> fallback.  Replays, eager steps between replays and eager steps after them
    fly the trajectory of plain eager stepping (to the rounding of the force's summation order, which follows an atomic
    scatter in both), and the force the grid yields after a replay is the brute-force one on the positions of that moment."""
    nat, fleet = gpu
    from dronesim_amd.envs import CtrlAviary, Physics
    from dronesim_amd.fleet import Targets
    n = 3000
    rng = np.random.default_rng(77)
    xyz = np.stack([rng.uniform(0, 40.0, n), rng.uniform(0, 40.0, n), rng.uniform(0.5, 9, n)], 1)
    if world == "vast":                                          # 40 clusters in a 2.8 km box: > 65 536 cells of 10 m
        ctr = rng.uniform(100, 2900, (40, 2))
        ctr[0], ctr[1] = (100.0, 100.0), (2900.0, 2900.0)
        xyz[:, 0:2] = ctr[np.arange(n) % 40] + rng.uniform(-15, 15, (n, 2))
    tid = (np.arange(n) % 2).astype(np.uint8)
    envs, tgts = [], []
    for _ in range(2):
        e = CtrlAviary(["robobee", "hexa_6DOF"], n, initial_xyzs=xyz, physics=Physics.PYB_DW, noise_seed=31, dict_io=False,
                       type_ids=tid)
        t = Targets(e.ctx, n); t.set(pos=f32(xyz).T + np.array([[0.8], [-0.4], [0.3]], dtype=np.float32), yaw=0.1)
        envs.append(e); tgts.append(t)
    A, B = envs
    for _ in range(15):
        A.step_fused(tgts[0])
    B.step_fused(tgts[1])
    reuses = B.ctx.query(nat.QUERY_DW_REUSES)
    g = B.capture_fused(tgts[1], steps=4)
    assert B._downwash.keep_lists == 0                          # (a captured sequence is fixed: its queries are made from scratch)
    assert (B._downwash._last is not None) and bool(B.ctx.lib.dsim_downwash_prebin_ok(n, B._downwash._last.nx, B._downwash._last.ny)) == (world == "dense")
    g.replay(); g.replay()                                      # 1 + 8
    B.step_fused(tgts[1])                                       # an ODD number of eager steps in between (the two count
    g.replay()                                                  # buffers alternate per step: a replay assumes nothing)
    B.step_fused(tgts[1]); B.step_fused(tgts[1])                # 1 + 8 + 1 + 4 + 2 = 16
    A.step_fused(tgts[0])
    torch.cuda.synchronize()
    assert A._env_steps == B._env_steps == 16
    assert B.ctx.query(nat.QUERY_DW_REUSES) == reuses and (world != "dense" or A.ctx.query(nat.QUERY_DW_REUSES) > 10)
    sa, sb = A.state.fields(0, 26).cpu().numpy(), B.state.fields(0, 26).cpu().numpy()
    np.testing.assert_allclose(sb, sa, rtol=2e-5, atol=2e-6)
    assert B.ctx.query(nat.QUERY_WLS_FAILURES) == 0
    # the grid after a replay: the force of the CURRENT positions
    g.replay()
    O = orc.Oracle(B.types)
    fz = B._downwash.compute()[2:3, :n]
    f = (fz if B.order is None else B.order.to_caller(fz, 1)).cpu().numpy()[0]       # (the force is kept in storage order)
    r = B.state.rigid_aos()
    ref = O.downwash(r, r[:, 0:3], type_id=tid)
    assert_downwash(f"graph replay, {world} world", f, ref, B.types, tid, r[:, 0:3], r[:, 0:3])
    for e in envs:
        e.close()


def test_observation_rows_are_placed_by_timing_zero_substep_passes(gpu):
    """placement.place_rows (a fleet large enough for HBM placement to matter): the candidates are compared by passes of
    the real Env.step launch with zero physics sub-steps, which leave the state bit for bit as it was, the echoed action
    and the ground-contact count untouched from the caller's point of view; stepping afterwards gives exactly what an env
    with plainly allocated rows gives."""
    nat, fleet = gpu
    from dronesim_amd.envs import CtrlAviary
    from dronesim_amd.fleet import Targets
    nd = 1 << 21                                                  # rows 160 MB, controller outputs 64 MB: placement.MIN_BYTES and above
    rng = np.random.default_rng(12)
    xyz = np.stack([np.arange(nd) % 2048, np.arange(nd) // 2048, rng.uniform(-0.2, 3.0, nd)], 1).astype(np.float64)   # some on the ground
    envs = [CtrlAviary(["robobee"], nd, initial_xyzs=xyz, aggregate_phy_steps=1, noise_seed=4, dict_io=False, placement=p)
            for p in (True, False)]
    obs = []
    for e in envs:
        t = Targets(e.ctx, nd, e.state.layout); t.set(pos=f32(xyz + 0.1).T, yaw=0.3)
        tg_before = t.data.clone()
        e.step_fused(t, action=np.full((nd, 4), 0.4, dtype=np.float32))      # (placement=True: the targets are placed by trial first)
        assert torch.equal(t.data, tg_before)
        before = e.state.data.view(torch.int32).clone()
        la = e._last_action.clone()
        gc0 = e.ground_contacts()
        rows = e._obs_tensor()                                   # the search (placement=True) or a plain allocation
        assert rows.shape == (nd, 20) and float(rows.abs().max()) == 0.0
        assert torch.equal(e.state.data.view(torch.int32), before) and torch.equal(e._last_action, la) and e.ground_contacts() == gc0
        o, _, _, _ = e.step(np.full((nd, 4), 0.5, dtype=np.float32))
        obs.append(o.obs if hasattr(o, "obs") else o)
    log = [r for r in envs[0].ctx.placement_log if not r["array"].startswith("state block")]   # (the state's own move: below)
    assert len(log) + 1 == len(envs[0].ctx.placement_log) and envs[0].ctx.placement_log[0]["array"].startswith("state block")
    arrays = [r["array"] for r in log]
    assert arrays == ["per-drone targets", "observation rows"]
    for r in log[:2]:
        assert 2 <= r["candidates"] and 0 <= r["chosen"] < r["candidates"] and 0 < r["chosen_pass_us"] <= r["first_pass_us"]
    assert envs[1].ctx.placement_log == []
    assert torch.equal(envs[0].state.data, envs[1].state.data) and torch.equal(obs[0], obs[1])
    assert envs[0].ground_contacts() == envs[1].ground_contacts() > 0
    # the controller's outputs (command, position error, yaw error) the same way, behind a snapshot of the state block: the
    # reference-shaped loop gives bit for bit what it gives with plainly allocated arrays
    from dronesim_amd.control import INDIControl
    from dronesim_amd.fleet import frozen
    res = []
    for e in envs:
        ctrl = INDIControl("robobee", env=e)
        tp = frozen(torch.from_numpy(f32(xyz + 0.1)).to(e.ctx.device))
        cmd = torch.full((nd, 4), 0.45, device=e.ctx.device)
        for _ in range(3):
            e.step(cmd)
            cmd, pos_e, yaw_e = ctrl.computeControlFromState(1 / 240, None, target_pos=tp, target_rpy=np.array([0.0, 0.0, 0.3]))
        res.append((cmd.clone(), pos_e.clone(), yaw_e.clone()))
    assert [r["array"] for r in envs[0].ctx.placement_log
            if r["array"] != "observation rows" and not r["array"].startswith("state block")] in (["per-drone targets", "computeControl outputs"], ["per-drone targets", "computeControl outputs", "computeControl targets"])
    assert envs[1].ctx.placement_log == []
    assert torch.equal(envs[0].state.data, envs[1].state.data)
    for x, y in zip(res[0], res[1]):
        assert torch.equal(x, y)
    for e in envs:
        e.close()


def test_state_block_of_a_placed_fleet_moves_once_into_its_own_allocation(gpu):
    """A large fleet with placement on moves its state block ONCE into a driver allocation with room for two target blocks
    behind it (arrays read beside the state want the state's own window of device memory: _ensure_read_room), before anything
    is timed against it; the fused step's targets and a bound controller's targets may take that room.  Whatever the
    searches decide, the env steps exactly as one with plainly allocated arrays does, and the report says what is held."""
    nat, fleet = gpu
    from dronesim_amd.control import INDIControl
    from dronesim_amd.envs import CtrlAviary
    from dronesim_amd.fleet import Targets, frozen
    nd = 1 << 21
    xyz = np.stack([np.arange(nd) % 2048, np.arange(nd) // 2048, np.full(nd, 1.5)], 1).astype(np.float64)
    envs = [CtrlAviary(["robobee"], nd, initial_xyzs=xyz, aggregate_phy_steps=1, noise_seed=9, dict_io=False, placement=p)
            for p in (True, False)]
    res = []
    for e in envs:
        ptr = e.state.data.data_ptr()
        tg = Targets(e.ctx, nd)
        tg.set(pos=torch.from_numpy(f32(xyz + 0.05).T.astype(np.float32)).to(e.ctx.device), yaw=0.1)
        e.step_fused(tg, action=np.full((nd, 4), 0.45, dtype=np.float32))
        e.step_fused(tg)
        ctrl = INDIControl("robobee", env=e)
        tp = frozen(torch.from_numpy(f32(xyz + 0.1)).to(e.ctx.device))
        cmd = torch.full((nd, 4), 0.45, device=e.ctx.device)
        for _ in range(3):
            o, _, _, _ = e.step(cmd)
            cmd, pos_e, yaw_e = ctrl.computeControlFromState(1 / 240, None, target_pos=tp, target_rpy=np.array([0.0, 0.0, 0.3]))
        e.moved = e.state.data.data_ptr() != ptr
        res.append((e.state.data.clone(), o.clone(), cmd.clone(), pos_e.clone(), yaw_e.clone(), e.ground_contacts()))
    log = envs[0].ctx.placement_log
    assert envs[0].moved and not envs[1].moved and envs[1].ctx.placement_log == []
    assert log[0]["array"].startswith("state block") and log[0]["held_bytes"] == 2 * 4 * nat.NT * envs[0].state.n_pad
    assert [r["array"] for r in log[1:]] == ["per-drone targets", "observation rows", "computeControl outputs", "computeControl targets"]
    for r in (log[1], log[4]):
        assert r["behind_the_state_pass_us"] > 0 and r["candidates"] >= 2
    for x, y in zip(res[0], res[1]):
        assert (torch.equal(x, y) if torch.is_tensor(x) else x == y)
    for e in envs:
        e.close()


def test_sharded_downwash_example_runs_as_two_ranks(gpu):
    """examples/fly_sharded_downwash_fleet.py under torch.distributed.run, two gloo ranks sharing the one GPU: both ranks
    fly, ship the boundary strip only, lose nothing."""
    import re
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "examples", "fly_sharded_downwash_fleet.py"),
                        "--drones_per_rank", "16384", "--steps", "40", "--slab_m", "64", "--backend", "gloo"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    # (the two ranks write to one pipe: their lines may run into each other)
    ships = re.findall(r"ships (\d+) positions per step to (\d+) neighbour\(s\), overflow (\d+)", p.stdout)
    assert len(ships) == 2 and p.stdout.count("WLS failures 0") == 2 and len(re.findall(r"rank [01]/2:", p.stdout)) == 2, p.stdout
    for sent, peers, lost in ships:
        assert 0 < int(sent) < 16384 // 3 and int(peers) == 1 and int(lost) == 0, p.stdout


def test_driver_blocks_go_back_to_the_driver_after_the_env_is_closed(gpu):
    """A placed fleet's arrays live on driver allocations (placement._DriverBlock) that PyTorch tensors keep alive.  env.close()
    destroys the library context; the tensors are usually dropped AFTERWARDS — the blocks must still go back (they
    leaked, ~1.3 GB per closed 4 M-drone env)."""
    import gc
    from dronesim_amd.envs import CtrlAviary
    from dronesim_amd.fleet import Targets
    nd = 1 << 21
    xyz = np.stack([np.arange(nd) % 2048, np.arange(nd) // 2048, np.full(nd, 1.0)], 1).astype(np.float64)
    gc.collect(); torch.cuda.synchronize(); torch.cuda.empty_cache()
    free0 = torch.cuda.mem_get_info()[0]
    env = CtrlAviary(["robobee"], nd, initial_xyzs=xyz, aggregate_phy_steps=1, noise_seed=4, dict_io=False, placement=True)
    t = Targets(env.ctx, nd, "tile64")
    t.set(pos=xyz.T.astype(np.float32), yaw=0.1)
    env.step_fused(t, action=np.full((nd, 4), 0.4, dtype=np.float32))
    env.step(torch.full((nd, 4), 0.45, device=env.ctx.device))
    torch.cuda.synchronize()
    assert any("driver" in str(r.get("memory", r.get("placed", ""))) for r in env.ctx.placement_log)
    held = free0 - torch.cuda.mem_get_info()[0]
    assert held > 300 << 20                                       # the fleet is there
    env.close()                                                   # the ctx goes first ...
    del env, t                                                    # ... the tensors on the driver blocks afterwards
    gc.collect(); torch.cuda.synchronize(); torch.cuda.empty_cache()
    leaked = free0 - torch.cuda.mem_get_info()[0]
    assert leaked < 64 << 20, leaked                              # (allocator granularity; the blocks were ~0.5 GB)
