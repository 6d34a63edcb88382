"""The rotor noise ON THE DEVICE: dsim_noise_draw hands out the normals the step kernels draw — compared, draw by draw, with
the definition the oracle restates (both lattices, quad and hexa streams), and as a distribution (1e7 draws against N(0, 1):
the statistics tests/test_noise_distribution.py measures on the definition); and DSIM_OPT_NOISE_FINE through every class of
kernel that carries it — the single-sub-step fast instances, and the general kernels that take a fine-lattice launch with
several sub-steps — against the oracle fed the fine stream."""
import ctypes
import math

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from dronesim_amd import params  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from tests.util import assert_step_parity, noise_terms, random_fleet, rotor_noise  # noqa: E402

pytestmark = pytest.mark.gpu
DT = float(np.float32(1.0 / 240.0))


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device; the product has no CPU fallback")
    from dronesim_amd import _native as nat
    from dronesim_amd import fleet
    return nat, fleet


def _draw(nat, ctx, n, n_act, seed, step_index, sub, fine, drone_id=None):
    n_pad = (n + 255) // 256 * 256
    out = torch.zeros((sub, 2 * n_act, n_pad), device=ctx.device)
    nat.check(ctx.lib.dsim_noise_draw(ctx.handle, ctx.stream_ptr(), n, n_pad, n_act, seed, step_index, sub,
                                      nat.OPT_NOISE_FINE if fine else nat.OPT_NOISE_COARSE, drone_id.data_ptr() if drone_id is not None else None,
                                      out.data_ptr()))
    torch.cuda.synchronize()
    return out[:, :, :n]


@pytest.mark.parametrize("fine", [False, True])
@pytest.mark.parametrize("n_act", [4, 6])
def test_noise_draw_equals_the_definition(gpu, n_act, fine):
    """Every normal of 3 000 drones x 4 sub-steps: the device's fp32 evaluation (v_log_f32, v_sqrt_f32, v_cos_f32 / v_sin_f32
    on the lattice point's integers) against the oracle's fp64 one; with a drone_id array the stream follows the drone."""
    nat, fleet = gpu
    t = params.builtin_type("robobee" if n_act == 4 else "hexa_6DOF")
    ctx = fleet.Context([t])
    O = orc.Oracle([t])
    n, sub, seed, sidx = 3000, 4, 0xC0FFEE1234, 11
    got = _draw(nat, ctx, n, n_act, seed, sidx, sub, fine).permute(2, 0, 1).double().cpu().numpy()      # [n, sub, 2 n_act]
    ref = O.noise_batch(seed, 0, n, sidx * sub, sub, n_act, fine=fine)
    # (the radius sqrt(-2 ln u) near u = 1 is conditioned by 1 / (1 - u): a few ulps of log2 at 65 535 / 65 536)
    assert np.abs(got - ref).max() < (3e-5 if fine else 2e-6), np.abs(got - ref).max()
    ids = torch.from_numpy(np.random.default_rng(1).permutation(n).astype(np.int32)).to(ctx.device)
    pad = torch.zeros(((n + 255) // 256 * 256,), dtype=torch.int32, device=ctx.device); pad[:n] = ids
    got2 = _draw(nat, ctx, n, n_act, seed, sidx, sub, fine, drone_id=pad).permute(2, 0, 1).double().cpu().numpy()
    np.testing.assert_array_equal(got2, got[ids.cpu().numpy()])
    ctx.close()


@pytest.mark.parametrize("fine", [False, True])
def test_noise_distribution_on_the_device(gpu, fine):
    """1e7 normals drawn by the device: mean, variance, kurtosis, support, mass beyond 3 sigma and the Kolmogorov distance
    from N(0, 1) — the numbers tests/test_noise_distribution.py pins for the definition."""
    from scipy import stats
    nat, fleet = gpu
    ctx = fleet.Context([params.builtin_type("robobee")])
    n, sub = 262144, 5
    z = _draw(nat, ctx, n, 4, 0x5EED1234ABCD, 3, sub, fine).double().cpu().numpy().ravel()
    assert z.size >= 10_000_000
    assert abs(z.mean()) < 1e-3 and abs(z.var() - 1.0) < 2e-3
    zs = np.sort(z)
    cdf = stats.norm.cdf(zs)
    ks = max(float((np.arange(1, z.size + 1) / z.size - cdf).max()), float((cdf - np.arange(0, z.size) / z.size).max()))
    tail = float((np.abs(z) > 3.0).mean())
    if fine:
        assert abs((z ** 4).mean() - 2.9998) < 0.012 and 4.0 < np.abs(z).max() < 4.8547
        assert ks < 1.63 / math.sqrt(z.size) and abs(tail - 2.69980e-3) < 1e-4
    else:
        assert abs((z ** 4).mean() - 2.9767) < 0.01 and np.abs(z).max() < 3.5347
        assert ks < 3.0e-3 and 2.3e-3 < tail < 2.8e-3, (ks, tail)
    assert float((z == 0.0).mean()) == 0.0                # no lattice point is 0: no atom at zero on either lattice
    ctx.close()


def _fine_replay(O, seed, n, step_index, sub, t, ids=None):
    """[n, sub, 12] per-rotor noise values of a fine-lattice launch of type t (tests/util.py:rotor_noise)"""
    n_act = t.n_act
    nz = np.zeros((n, sub, 12))
    b = O.noise_batch(seed, 0, n, step_index * sub, sub, n_act, fine=True)
    if ids is not None:
        b = np.stack([O.noise_batch(seed, int(i), 1, step_index * sub, sub, n_act, fine=True)[0] for i in ids])
    for i in range(n):
        for s_ in range(sub):
            nz[i, s_, 0:n_act], nz[i, s_, 6:6 + n_act] = rotor_noise(t, b[i, s_])
    return nz


@pytest.mark.parametrize("model,sub,n", [("robobee", 1, 1024), ("robobee", 3, 1024), ("robobee", 2, 777), ("hexa_6DOF", 1, 1024),
                                         ("hexa_6DOF", 2, 512), ("hexa_6DOF_simple", 1, 512)])
def test_fine_lattice_through_the_fused_step(gpu, model, sub, n):
    """dsim_step with DSIM_OPT_NOISE_FINE: one sub-step -> the fast kernels' single-sub-step instances; several -> the
    general kernel (the looped fast instances carry the default lattice only) — either way the oracle fed the fine stream."""
    nat, fleet = gpu
    t = params.builtin_type(model)
    ctx = fleet.Context([t])
    O = orc.Oracle([t])
    rigid, mem, tgt = random_fleet(np.random.default_rng(4), n, n_act=t.n_act)
    st = fleet.FleetState(ctx, n, "tile64" if n % 256 == 0 else "soa", pad=64)
    tg = fleet.Targets(ctx, n, st.layout, pad=64)
    st.load_aos(rigid, mem)
    tg.set_fields(0, torch.from_numpy(np.ascontiguousarray(tgt.T)))
    seed, dtc = 0xABCDEF12345, float(np.float32(sub / 240))
    for step_index in (5, 6):
        r0, m0 = st.rigid_aos(), st.mem_aos()
        a = nat.StepArgs()
        a.phys_substeps, a.dt_phys, a.dt_ctrl, a.options = sub, DT, dtc, nat.OPT_NOISE_FINE
        a.noise_seed, a.step_index = seed, step_index
        nat.check(ctx.lib.dsim_step(ctx.handle, ctx.stream_ptr(), n, st.view(), tg.view(), ctypes.byref(a)))
        torch.cuda.synchronize()
        r1, m1 = r0.copy(), m0.copy()
        assert O.step(r1, m1, tgt, sub, DT, dtc, noise=_fine_replay(O, seed, n, step_index, sub, t)) == 0
        assert_step_parity(f"fine_noise_fused[{model},{sub}]", [t], None, r0, m0, tgt, st.rigid_aos(), st.mem_aos(), r1, m1, DT, dtc, sub,
                           extra_terms=noise_terms([t], None, n, DT, sub))
    # the coarse lattice on the same state gives another trajectory (the switch is honoured)
    b = nat.StepArgs.from_buffer_copy(a); b.options = nat.OPT_NOISE_COARSE
    keep = st.data.clone()
    nat.check(ctx.lib.dsim_step(ctx.handle, ctx.stream_ptr(), n, st.view(), tg.view(), ctypes.byref(b)))
    d0 = st.rigid_aos()
    st.data.copy_(keep)
    nat.check(ctx.lib.dsim_step(ctx.handle, ctx.stream_ptr(), n, st.view(), tg.view(), ctypes.byref(a)))
    assert np.abs(st.rigid_aos()[:, 7:10] - d0[:, 7:10]).max() > 1e-6
    ctx.close()


@pytest.mark.parametrize("sub", [1, 5])
def test_fine_lattice_through_the_two_call_loop_and_the_adaptors(gpu, sub):
    """CtrlAviary(noise="fine"): Env.step (k_physics_fast carries both lattices at any sub-step count), an interleaved quad +
    hexa fleet (run kernels at one sub-step, the general kernel beyond), and VelocityAviary (the general adaptor kernel) —
    every Env.step against the oracle fed the fine stream."""
    from dronesim_amd.envs import CtrlAviary, VelocityAviary
    nat, _ = gpu
    n = 512
    xyz = np.stack([np.arange(n) % 32, np.arange(n) // 32, np.full(n, 2.0)], 1).astype(np.float64)
    seed = 991
    # homogeneous quads, device-row actions
    env = CtrlAviary(["robobee"], n, initial_xyzs=xyz, aggregate_phy_steps=sub, noise_seed=seed, dict_io=False, noise="fine")
    t = env.types[0]
    O = orc.Oracle([t])
    rng = np.random.default_rng(0)
    for k in range(3):
        r0, m0 = env.state.rigid_aos(), env.state.mem_aos()
        act = torch.from_numpy(rng.uniform(0.35, 0.6, (n, 4)).astype(np.float32)).to(env.ctx.device)
        env.step(act)
        r1 = r0.copy()
        a6 = np.zeros((n, 6)); a6[:, :4] = act.cpu().numpy()
        O.physics(r1, m0, sub, DT, action=a6, noise=_fine_replay(O, seed, n, k, sub, params.builtin_type("robobee")))
        assert_step_parity(f"fine_noise_env_step[{sub}]", [t], None, r0, m0, np.zeros((1, 10)), env.state.rigid_aos(), None, r1, None, DT,
                           DT * sub, sub, control=False, action=a6[:, :4], extra_terms=noise_terms([t], None, n, DT, sub))
    env.close()
    # interleaved quads + hexas (type-major storage behind the caller's numbering: the noise follows the drone)
    tid = (np.arange(n) % 2).astype(np.uint8)
    env = CtrlAviary(["robobee", "hexa_6DOF"], n, initial_xyzs=xyz, aggregate_phy_steps=sub, noise_seed=seed, dict_io=False,
                     type_ids=tid, noise="fine")
    O = orc.Oracle(env.types)
    for k in range(2):
        r0, m0 = env.state.rigid_aos(), env.state.mem_aos()
        act = rng.uniform(0.35, 0.6, (n, 6)); act[tid == 0, 4:] = 0.0
        env.step(torch.from_numpy(act.astype(np.float32)).to(env.ctx.device))
        r1 = r0.copy()
        nz = np.zeros((n, sub, 12))
        for i in range(n):
            na = 4 if tid[i] == 0 else 6
            b = O.noise_batch(seed, i, 1, k * sub, sub, na, fine=True)[0]
            for s_ in range(sub):
                nz[i, s_, 0:na], nz[i, s_, 6:6 + na] = rotor_noise(env.types[int(tid[i])], b[s_])
        O.physics(r1, m0, sub, DT, action=act.astype(np.float32).astype(np.float64), noise=nz, type_id=tid)
        assert_step_parity(f"fine_noise_env_step_mixed[{sub}]", env.types, tid, r0, m0, np.zeros((1, 10)), env.state.rigid_aos(), None, r1, None,
                           DT, DT * sub, sub, control=False, action=act.astype(np.float32).astype(np.float64),
                           extra_terms=noise_terms(env.types, tid, n, DT, sub))
    env.close()
    # VelocityAviary
    env = VelocityAviary(["robobee"], n, initial_xyzs=xyz, aggregate_phy_steps=sub, noise_seed=seed, dict_io=False, noise="fine")
    O = orc.Oracle(env.types)
    t = env.types[0]
    for k in range(2):
        r0, m0 = env.state.rigid_aos(), env.state.mem_aos()
        act = np.concatenate([rng.normal(0, 1, (n, 3)), rng.uniform(0.1, 1, (n, 1))], 1).astype(np.float32)
        env.step(torch.from_numpy(act).to(env.ctx.device))
        r1, m1 = r0.copy(), m0.copy()
        # the adaptor's law on the current state, then the physics with the fine stream (orc_adaptor_step_batch is noise-free:
        # the control part through it with zero sub-steps, the physics through orc_physics_batch)
        O.adaptor_step(0, r1, m1, act.astype(np.float64), 0, DT, float(np.float32(DT * sub)))
        O.physics(r1, m1, sub, DT, noise=_fine_replay(O, seed, n, k, sub, params.builtin_type("robobee")))
        assert_step_parity(f"fine_noise_velocity_aviary[{sub}]", [t], None, r0, m0, np.zeros((1, 10)), env.state.rigid_aos(), None, r1, None,
                           DT, DT * sub, sub, control=False, action=m1[:, 7:11], extra_terms=noise_terms([t], None, n, DT, sub))
    env.close()
