"""Kept candidate lists of the neighbour downwash (dsim_downwash_args.keep; formula P8, BaseAviary.py:1736-1763): one BUILD
query writes per-cell candidate lists, the REUSE queries that follow gather current positions and run the pair loops only.
The claim under test is EXACTNESS FOR ANY MOTION: whatever the drones did since the BUILD — crept inside the skin, swapped
heights, jumped across the world, left the grid, became NaN, crowded into one cell — the force equals the brute-force sum of
the oracle on the CURRENT positions, at the same bar as the plain query."""
import numpy as np
import pytest
import torch

from dronesim_amd import params
from oracle import oracle as orc
from tests.util import assert_downwash, f32, random_fleet

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device; the product has no CPU fallback")
    from dronesim_amd import _native as nat
    from dronesim_amd import fleet
    return nat, fleet


def _world(rng, n, side, zmax=20.5):
    rigid, mem, _ = random_fleet(rng, n, n_act=6)
    rigid[:, 0] = f32(rng.uniform(0, side, n)); rigid[:, 1] = f32(rng.uniform(0, side, n)); rigid[:, 2] = f32(rng.uniform(0.5, zmax, n))
    return rigid, mem


def _setup(gpu, n, side, seed, keep=8, skin=0.25, zmax=20.5, patch=None):
    nat, fleet = gpu
    from dronesim_amd.downwash import Downwash
    types = [params.builtin_type("robobee"), params.builtin_type("hexa_6DOF")]
    ctx = fleet.Context(types)
    st = fleet.FleetState(ctx, n)
    rng = np.random.default_rng(seed)
    rigid, mem = _world(rng, n, side, zmax)
    if patch is not None:
        k, lo, hi = patch
        rigid[:k, 0] = f32(rng.uniform(lo, hi, k)); rigid[:k, 1] = f32(rng.uniform(lo, hi, k))
    st.load_aos(rigid, mem)
    tid = (rng.random(n) < 0.5).astype(np.uint8)
    tid_dev = torch.zeros(st.n_pad, dtype=torch.uint8, device=ctx.device); tid_dev[:n] = torch.from_numpy(tid)
    # (keep_movers: the host's pacing of the BUILDs by the device's reports is switched off — these tests say which query does what)
    dw = Downwash(ctx, st, tid_dev, keep_lists=keep, keep_skin=skin, keep_movers=1 << 30)
    return nat, ctx, st, dw, types, tid, rigid, rng


def _check(label, dw, st, types, tid, pos, n):
    """moves the fleet to `pos` (a host write: nothing refreshed the lists' positions, the query does it itself) and compares"""
    pos = f32(pos)                    # (what the device holds: the callers add to their copy in fp64)
    st.set_fields(0, torch.from_numpy(np.ascontiguousarray(pos.T)).float())
    f = dw.compute().cpu().numpy()
    rigid = np.zeros((n, 13)); rigid[:, 0:3] = pos
    finite = np.isfinite(pos).all(1)
    ref = orc.Oracle(types).downwash(rigid[finite], pos[finite], type_id=tid[finite])
    assert_downwash(label, f[2, :n][finite], ref, types, tid[finite], pos[finite], pos[finite])
    assert np.all(f[2, :n][~finite] == 0.0) and np.all(f[0:2, :n] == 0.0)
    return ref


def _reuses(nat, ctx):
    return ctx.query(nat.QUERY_DW_REUSES), ctx.query(nat.QUERY_DW_MOVERS)


def test_reuse_equals_bruteforce_through_every_kind_of_motion(gpu):
    n, side = 5200, 80.0                                                                  # 0.8 drones per m^2: the banded query
    nat, ctx, st, dw, types, tid, rigid, rng = _setup(gpu, n, side, 41)
    pos = rigid[:, 0:3].astype(np.float64).copy()
    pos = f32(pos)
    _check("kept lists: the BUILD query", dw, st, types, tid, pos, n)
    assert dw._last.cell == 5.25 and dw._last.keep == nat.DW_KEEP_REUSE                   # what the next query will be
    assert _reuses(nat, ctx) == (0, 0)
    # 1: everybody creeps inside the skin (|d| <= 0.24 m in total over three queries), heights reshuffle locally
    for k in range(3):
        pos = f32(pos + rng.uniform(-0.045, 0.045, pos.shape))
        _check(f"kept lists: creep {k}", dw, st, types, tid, pos, n)
    r, m = _reuses(nat, ctx)
    assert r == 3 and m == 0
    # 1b: a dozen leave the skin: the movers ride in front of every workgroup's tile
    pos = pos.copy()
    pos[700:712] += f32(rng.uniform(0.3, 0.9, (12, 3)))
    pos[712:720, 0:2] = f32([12.0, 60.0]) + f32(rng.uniform(0, 2, (8, 2)))                # eight of them meet in one cell
    _check("kept lists: a few movers", dw, st, types, tid, pos, n)
    r, m = _reuses(nat, ctx)
    assert r == 4 and m == 20
    # 2: some leave the skin by a little, some jump across the world, some leave the grid, two swap places, one is NaN
    pos = pos.copy()
    pos[10:30] += f32(rng.uniform(0.3, 0.6, (20, 3)))
    pos[100:110, 0:2] = f32(rng.uniform(0, side, (10, 2)))
    pos[200:204, 0] = f32([-40.0, side + 55.0, -3.0, side + 7.0]); pos[204, 1] = -25.0
    pos[[300, 301]] = pos[[301, 300]]
    pos[400] = np.nan
    pos[500:520, 2] = f32(rng.uniform(0.5, 20.5, 20))                                     # new heights: the band order is stale
    _check("kept lists: movers", dw, st, types, tid, pos, n)
    r, m = _reuses(nat, ctx)
    assert r == 5 and 20 + 70 <= m <= 20 + 90
    # 3: many movers (more than the tile keeps in front): the overflow list is read from memory, in batches
    pos = pos.copy()
    pos[1000:1400, 0:2] += f32(rng.uniform(-3, 3, (400, 2)))
    pos[400] = pos[401] + f32([0.0, 0.0, 0.5])                                             # back from NaN, right above a neighbour
    _check("kept lists: many movers", dw, st, types, tid, pos, n)
    # 4: thirty-odd of them in ONE cell, above and below its residents
    pos = pos.copy()
    pos[2000:2040, 0:2] = f32([41.0, 41.0]) + f32(rng.uniform(0, 3, (40, 2)))
    _check("kept lists: a crowd arrives in one cell", dw, st, types, tid, pos, n)
    r, m = _reuses(nat, ctx)
    assert r == 7 and dw._last.keep == nat.DW_KEEP_BUILD
    # the ninth query BUILDs again: everybody is back in a list
    _check("kept lists: rebuilt", dw, st, types, tid, pos, n)
    assert _reuses(nat, ctx) == (7, m)
    pos = f32(pos + rng.uniform(-0.02, 0.02, pos.shape))
    _check("kept lists: after the rebuild", dw, st, types, tid, pos, n)
    assert _reuses(nat, ctx) == (8, m)                                                     # nobody outside the new skin
    ctx.close()


@pytest.mark.parametrize("heights", ["flat", "two_layers", "ties"])
def test_reuse_with_degenerate_heights(gpu, heights):
    n, side = 5200, 80.0
    nat, ctx, st, dw, types, tid, rigid, rng = _setup(gpu, n, side, {"flat": 5, "two_layers": 6, "ties": 7}[heights])
    z = {"flat": np.full(n, 3.0), "two_layers": np.where(rng.random(n) < 0.5, 2.0, 2.5) + rng.uniform(0, 1e-3, n),
         "ties": np.round(rng.uniform(0.5, 6.5, n) * 2) / 2}[heights]
    pos = f32(rigid[:, 0:3]); pos[:, 2] = f32(z)
    _check(f"kept lists[{heights}]: build", dw, st, types, tid, pos, n)
    for k in range(3):
        pos = pos.copy()
        pos[:, 0:2] += f32(rng.uniform(-0.05, 0.05, (n, 2)))
        if heights != "flat":
            pos[:, 2] += f32(rng.choice([-0.05, 0.0, 0.05], n))                           # ties break and re-form
        ref = _check(f"kept lists[{heights}]: reuse {k}", dw, st, types, tid, pos, n)
        assert (np.abs(ref).max() == 0.0) == (heights == "flat")
    assert _reuses(nat, ctx)[0] == 3
    ctx.close()


def test_crowded_neighbourhoods_leave_unbanded_lists_and_full_buckets(gpu):
    """1 500 of 4 000 drones in a 24 m square (2.6 per m^2: more than the banded tile holds, cells with more drones than
    a bucket: those sit in the overflow list from the BUILD on) — the lists of that corner are unbanded and longer than one fill."""
    n, side = 4000, 70.0
    nat, ctx, st, dw, types, tid, rigid, rng = _setup(gpu, n, side, 43, patch=(1500, 20.0, 44.0))
    pos = f32(rigid[:, 0:3])
    _check("kept lists crowded: build", dw, st, types, tid, pos, n)
    for k in range(3):
        pos = f32(pos + rng.uniform(-0.06, 0.06, pos.shape))
        if k == 1:
            pos[0:12, 0:2] += f32([9.0, -7.0])                                            # residents of the crowd move house
        _check(f"kept lists crowded: reuse {k}", dw, st, types, tid, pos, n)
    r, m = _reuses(nat, ctx)
    assert r == 3 and m > 3 * 100                                                          # the buckets' overflow is there every time
    ctx.close()


def test_empty_cells_and_a_sparse_rim(gpu):
    """A fleet with a hole in the middle and stragglers far outside: cells without receivers make lists too, and a drone
    that flies into one (or beyond the grid) is served there."""
    n, side = 5200, 80.0
    nat, ctx, st, dw, types, tid, rigid, rng = _setup(gpu, n, side, 44)
    pos = f32(rigid[:, 0:3])
    hole = (np.abs(pos[:, 0] - 40) < 9) & (np.abs(pos[:, 1] - 40) < 9)
    pos[hole, 0] += 18.5
    pos[:6, 0:2] = f32([[-11, -11], [93, 40], [40, 94], [-9, 88], [91, 92], [40, -10]])     # (beyond the cut-off from everybody)
    _check("kept lists hole: build", dw, st, types, tid, pos, n)
    pos = pos.copy()
    pos[50:58, 0:2] = f32([40.0, 40.0]) + f32(rng.uniform(-4, 4, (8, 2)))                 # into the hole
    pos[58:60, 0:2] = f32([[-12.5, -11.5], [99, 44]])                                      # out to the stragglers, one beyond the grid
    pos[0, 0:2] = f32([40.0, 41.0])                                                       # a straggler comes home
    _check("kept lists hole: arrivals", dw, st, types, tid, pos, n)
    pos = f32(pos + rng.uniform(-0.03, 0.03, pos.shape))
    _check("kept lists hole: reuse 2", dw, st, types, tid, pos, n)
    assert _reuses(nat, ctx)[0] == 2
    ctx.close()


def _fly(gpu, keep, steps, planted, sub=1):
    from dronesim_amd.envs import CtrlAviary, Physics
    from dronesim_amd.fleet import Targets
    n = 4096
    rng = np.random.default_rng(77)
    xyz = np.stack([rng.uniform(0, 64, n), rng.uniform(0, 64, n), rng.uniform(0.5, 20.5, n)], 1)
    if planted:          # near-vertical pairs a few millimetres apart in xy: the P8 term is singular, the lower drone is thrown out
        for k in range(6):
            xyz[2 * k + 1] = xyz[2 * k] + [1e-3 * (k + 1), 0.0, -(0.004 + 0.001 * k)]
    tid = (np.arange(n) % 2).astype(np.uint8)
    env = CtrlAviary(["robobee", "hexa_6DOF"], n, initial_xyzs=xyz, physics=Physics.PYB_DW, noise_seed=3, dict_io=False,
                     type_ids=tid, aggregate_phy_steps=sub, downwash_keep=keep)
    tg = Targets(env.ctx, n); tg.set(pos=f32(xyz).T, yaw=0.2)
    out = []
    for k in range(steps):
        env.step_fused(tg)
        if k % 5 == 4 or k == steps - 1:
            out.append(env.state.rigid_aos().copy())
    nat = gpu[0]
    stats = (env.ctx.query(nat.QUERY_DW_REUSES), env.ctx.query(nat.QUERY_DW_MOVERS))
    env.close()
    return out, stats


@pytest.mark.parametrize("planted", [False, True])
def test_a_flight_with_kept_lists_is_the_flight_without(gpu, planted):
    """Env-level: the step kernels refresh the lists' positions (no binning, no refresh launch); 23 Env.steps with one BUILD in
    six against the same flight with a BUILD-free plain query every step.  The two differ by the order of a sum only."""
    ref, s0 = _fly(gpu, 0, 23, planted)
    got, s1 = _fly(gpu, 6, 23, planted)
    assert s0 == (0, 0) and s1[0] == 23 - 4                                                # steps 0, 6, 12, 18 BUILD
    if planted:
        assert s1[1] > 10                                                                  # the ejected drones left the skin
        v = np.abs(ref[-1][:12, 7:10]).max()
        assert v > 5.0, v                                                                  # ... at speed
    for a, b in zip(ref, got):
        assert np.isfinite(b).all()
        # (positions to a few ulp of 64 m, velocities to 1e-5 of the ejection speeds)
        np.testing.assert_allclose(b[:, 0:3], a[:, 0:3], rtol=0, atol=2e-4)
        np.testing.assert_allclose(b[:, 7:10], a[:, 7:10], rtol=2e-4, atol=2e-4)


def test_sub_stepped_flight_with_kept_lists(gpu):
    """aggregate_phy_steps = 3: one query per physics SUB-step (the reference refreshes the positions per sub-step), every one of
    them a link of the same BUILD / REUSE chain."""
    ref, _ = _fly(gpu, 0, 8, False, sub=3)
    got, s1 = _fly(gpu, 6, 8, False, sub=3)
    assert s1[0] == 24 - 4
    for a, b in zip(ref, got):
        np.testing.assert_allclose(b[:, 0:3], a[:, 0:3], rtol=0, atol=2e-4)
        np.testing.assert_allclose(b[:, 7:10], a[:, 7:10], rtol=2e-4, atol=2e-4)


def test_keep_is_ignored_where_the_lists_do_not_apply(gpu):
    """a sparse world (the one-wave query), a world given as pos_all: plain queries, same results, no REUSE counted"""
    nat, fleet = gpu
    n, side = 3000, 300.0
    nat, ctx, st, dw, types, tid, rigid, rng = _setup(gpu, n, side, 45)
    pos = f32(rigid[:, 0:3])
    for k in range(3):
        pos = f32(pos + rng.uniform(-0.05, 0.05, pos.shape))
        _check(f"kept lists sparse {k}", dw, st, types, tid, pos, n)
    assert _reuses(nat, ctx) == (0, 0) and dw._last.keep == nat.DW_KEEP_OFF
    ctx.close()


def _sequence(gpu, keep, storage):
    """an env driven through everything that can happen between two queries: fused steps, the two-call loop's Env.step, a host
    write to the state, a reset — the downwash force checked against the brute-force oracle after each, the states returned"""
    from dronesim_amd.envs import CtrlAviary, Physics
    from dronesim_amd.fleet import Targets
    nat = gpu[0]
    n = 2048
    rng = np.random.default_rng(91)
    xyz = np.stack([rng.uniform(0, 45, n), rng.uniform(0, 45, n), rng.uniform(0.5, 12, n)], 1)
    tid = (np.arange(n) % 2).astype(np.uint8)
    env = CtrlAviary(["robobee", "hexa_6DOF"], n, initial_xyzs=xyz, physics=Physics.PYB_DW, noise_seed=5, dict_io=False,
                     type_ids=tid, downwash_keep=keep, storage=storage)
    tg = Targets(env.ctx, n); tg.set(pos=f32(xyz).T, yaw=0.1)
    O = orc.Oracle(env.types)
    states = []

    def check(label):
        f = env._downwash.compute()[2, :n]
        f = (f if env.order is None else env.order.to_caller(f, 0)).cpu().numpy()          # (the force is per storage slot)
        r = env.state.rigid_aos()
        ref = O.downwash(r, r[:, 0:3], type_id=tid)
        assert_downwash(f"kept lists sequence[{keep},{storage}]: {label}", f, ref, env.types, tid, r[:, 0:3], r[:, 0:3])
        states.append(r.copy())

    for _ in range(3):
        env.step_fused(tg)
    check("fused steps")
    check("two queries in a row")
    env.step_fused(tg); env.step_fused(tg)
    cmd = torch.full((n, 6), 0.45, device=env.ctx.device)
    for _ in range(3):
        env.step(cmd)                                                                      # Env.step of the two-call loop
    check("Env.step")
    env.step_fused(tg)
    moved = env.state.fields(0, 3).clone(); moved[0] += 2.5; moved[2] = moved[2].flip(0)
    env.state.set_fields(0, moved)                                                         # a host write behind the lists
    check("host write")
    for _ in range(2):
        env.step_fused(tg)
    env.reset()
    check("reset")
    for _ in range(5):
        env.step_fused(tg)
    check("flight after the reset")
    stats = (env.ctx.query(nat.QUERY_DW_REUSES), env.ctx.query(nat.QUERY_DW_MOVERS))
    env.close()
    return states, stats


@pytest.mark.parametrize("storage", ["auto", "caller"])
def test_everything_that_can_happen_between_two_queries(gpu, storage):
    ref, s0 = _sequence(gpu, 0, storage)
    got, s1 = _sequence(gpu, 4, storage)
    assert s0[0] == 0 and s1[0] >= 10
    for a, b in zip(ref, got):
        np.testing.assert_allclose(b[:, 0:3], a[:, 0:3], rtol=0, atol=1e-4)
        np.testing.assert_allclose(b[:, 7:10], a[:, 7:10], rtol=1e-4, atol=1e-4)


def test_the_library_answers_a_reuse_it_cannot_serve_as_a_build(gpu):
    """REUSE is a request: without lists of this very grid (none made yet, forgotten by dsim_downwash_reset, another buffer) the call is
    answered as a BUILD; with a buffer too small for the lists, as a plain query.  Same force every time."""
    n, side = 5200, 80.0
    nat, ctx, st, dw, types, tid, rigid, rng = _setup(gpu, n, side, 46)
    pos = f32(rigid[:, 0:3])
    _check("kept lists requests: build", dw, st, types, tid, pos, n)
    a = dw._last
    assert a.keep == nat.DW_KEEP_REUSE
    nat.check(ctx.lib.dsim_downwash_reset(ctx.handle))                                    # the library forgets its lists
    pos = f32(pos + rng.uniform(-0.05, 0.05, pos.shape))
    _check("kept lists requests: reuse after a reset", dw, st, types, tid, pos, n)
    assert _reuses(nat, ctx)[0] == 0                                                      # ... it was a BUILD
    pos = f32(pos + rng.uniform(-0.05, 0.05, pos.shape))
    _check("kept lists requests: reuse", dw, st, types, tid, pos, n)
    assert _reuses(nat, ctx)[0] == 1
    keep_ws = dw._keep_ws
    dw._keep_ws = torch.empty_like(keep_ws)                                               # another buffer: not where the lists are
    a.keep_ws = dw._keep_ws.data_ptr()
    pos = f32(pos + rng.uniform(-0.05, 0.05, pos.shape))
    _check("kept lists requests: reuse from another buffer", dw, st, types, tid, pos, n)
    assert _reuses(nat, ctx)[0] == 1
    a.keep_ws_len = 1000                                                                  # too small: a plain query
    a.keep = nat.DW_KEEP_REUSE
    pos = f32(pos + rng.uniform(-0.05, 0.05, pos.shape))
    _check("kept lists requests: buffer too small", dw, st, types, tid, pos, n)
    assert _reuses(nat, ctx)[0] == 1
    ctx.close()


def _march(gpu, keep, shear):
    from dronesim_amd.envs import CtrlAviary, Physics
    from dronesim_amd.fleet import Targets
    nat = gpu[0]
    n = 4096
    rng = np.random.default_rng(93)
    xyz = np.stack([rng.uniform(0, 64, n), rng.uniform(0, 64, n), rng.uniform(0.5, 20.5, n)], 1)
    tid = (np.arange(n) % 2).astype(np.uint8)
    env = CtrlAviary(["robobee", "hexa_6DOF"], n, initial_xyzs=xyz, physics=Physics.PYB_DW, noise_seed=7, dict_io=False,
                     type_ids=tid, downwash_keep=keep)
    # the whole fleet on the march, 2 cm a step — as one formation, or (shear) every other drone the opposite way: no common drift
    sign = np.where(np.arange(n) % 2 == 0, 1.0, -1.0) if shear else np.ones(n)
    v = f32(np.stack([4.0 * sign, -1.0 * sign, np.zeros(n)]))
    env.state.set_fields(7, torch.from_numpy(v).float())
    tg = Targets(env.ctx, n); tg.set(pos=f32(xyz + 0.75 * v.T).T, vel=v, yaw=0.0)
    kinds = []
    for k in range(70):
        env.step_fused(tg)
        kinds.append(int(env._downwash._last.keep))                                        # (what the NEXT query will be)
    r = env.state.rigid_aos().copy()
    stats = (env.ctx.query(nat.QUERY_DW_REUSES), env.ctx.query(nat.QUERY_DW_MOVERS))
    env.close()
    return r, stats, kinds


def test_a_fleet_on_the_march_keeps_its_lists(gpu):
    """Env-level: the skin moves with the fleet (the drift of a sample of it, predicted a step ahead): 70 steps at 4 m/s — 1.2 m, twelve
    skins — with the BUILDs the host was given (one query in 32) and next to nobody in the overflow list; the flight is the flight
    with plain queries."""
    ref, s0, _ = _march(gpu, 0, False)
    got, s1, kinds = _march(gpu, 32, False)
    assert s0 == (0, 0)
    assert kinds.count(1) <= 3 and s1[0] >= 66, kinds
    assert s1[1] / s1[0] < 16, s1
    np.testing.assert_allclose(got[:, 0:3], ref[:, 0:3], rtol=0, atol=2e-4)
    np.testing.assert_allclose(got[:, 7:10], ref[:, 7:10], rtol=2e-4, atol=2e-4)


def test_a_fleet_that_shears_paces_its_own_builds(gpu):
    """Every other drone flies the opposite way: there is no common drift to follow, every drone would leave the 0.1 m skin within six
    steps.  The device's report of how many are HALF WAY out (host memory, nothing synchronises; the host stays at most eight queries
    ahead of it) makes the host BUILD every few queries instead of the K = 32 it was given — once the period is learnt nobody reaches
    the overflow list — and the flight is the flight with plain queries."""
    ref, s0, _ = _march(gpu, 0, True)
    got, s1, kinds = _march(gpu, 32, True)
    assert s0 == (0, 0)
    builds = [i for i, k in enumerate(kinds) if k == 1]
    assert len(builds) >= 5 and max(j - i for i, j in zip(builds[:-1], builds[1:])) <= 20, kinds
    assert s1[0] >= 30                                                                     # ... and most queries still came from lists
    assert s1[1] <= 2 * 4096, s1                           # the overflow list held the fleet at most twice (the first time round, before the period was learnt)
    np.testing.assert_allclose(got[:, 0:3], ref[:, 0:3], rtol=0, atol=2e-4)
    np.testing.assert_allclose(got[:, 7:10], ref[:, 7:10], rtol=2e-4, atol=2e-4)


def test_a_formation_in_flight_leaves_no_skin(gpu):
    """The skin moves with the fleet: the whole fleet translating at 4 m/s (2 cm a step, the 0.1 m skin six steps wide), a thousand
    drones of it loitering — the marchers stay in the lists for the whole K = 32, the loiterers drift out of the moving skin into the
    overflow list, and the force is the brute-force one every step; then the formation stops, turns and flies on."""
    n, side = 5200, 80.0
    nat, ctx, st, dw, types, tid, rigid, rng = _setup(gpu, n, side, 47, keep=32, skin=0.1)
    pos = f32(rigid[:, 0:3])
    _check("kept lists formation: build", dw, st, types, tid, pos, n)
    v = np.array([4.0, -1.0, 0.5]) / 240.0
    lo = np.zeros(n, bool); lo[::50] = True                                              # 104 loiterers
    for k in range(20):
        if k == 12:
            v = np.array([0.0, 0.0, 0.0])                                                 # the formation stops ...
        if k == 15:
            v = np.array([-3.0, 2.0, 0.0]) / 240.0                                        # ... and flies on elsewhere
        pos = pos + np.where(lo[:, None], 0.0, v[None, :]) + rng.uniform(-0.002, 0.002, pos.shape)
        _check(f"kept lists formation: step {k}", dw, st, types, tid, pos, n)
    r, m = _reuses(nat, ctx)
    assert r == 20                                                                        # no BUILD in between
    # the formation has come 0.26 m and more: without the moving skin every one of its 5 096 drones would have been a mover from
    # the seventh step on; with it only the loiterers (and the turns' overshoot) are
    assert m < 20 * 400, m
    ctx.close()


@pytest.mark.parametrize("seed", range(6))
def test_random_worlds_random_motions(gpu, seed):
    """density 0.5 - 1.3 drones per m^2, skins of 0.05 - 0.3 m, a drifting, jittering fleet with stragglers, teleports and the odd NaN:
    every query of three BUILD periods against the brute-force oracle"""
    rng0 = np.random.default_rng(1000 + seed)
    n = int(rng0.integers(3000, 9000))
    side = float(np.sqrt(n / rng0.uniform(0.5, 1.3)))
    skin = float(rng0.choice([0.05, 0.1, 0.3]))
    nat, ctx, st, dw, types, tid, rigid, rng = _setup(gpu, n, side, 2000 + seed, keep=6, skin=skin, zmax=float(rng0.choice([4.0, 20.5])))
    pos = f32(rigid[:, 0:3])
    drift = rng0.uniform(-0.6, 0.6, 3) * skin
    for k in range(17):
        if k:
            pos = pos + drift + rng.uniform(-0.25, 0.25, pos.shape) * skin
            if k % 5 == 2:
                drift = rng0.uniform(-0.6, 0.6, 3) * skin                                  # the formation turns
            j = rng.integers(0, n, int(rng0.integers(0, 30)))
            pos[j, 0:2] = rng.uniform(-5, side + 5, (len(j), 2))                          # teleports, some beyond the grid
            if k == 7:
                pos[int(rng0.integers(0, n))] = np.nan
            if k == 9:
                pos = np.where(np.isnan(pos), 1.0, pos)
        _check(f"kept lists random[{seed}] query {k}", dw, st, types, tid, pos, n)
        pos = f32(pos)
    assert _reuses(nat, ctx)[0] >= 12 or dw._last.keep == nat.DW_KEEP_OFF                  # (a sparse draw takes the one-wave query: no lists)
    ctx.close()
