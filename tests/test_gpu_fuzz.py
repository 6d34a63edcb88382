"""Call-sequence fuzzing of the host classes: CtrlAviary carries a dozen cached plans and
validity flags (chained stepping, prepared argument blocks, pre-binned neighbour grids, placed buffers, command tokens,
the caller-order I/O of reordered fleets) and every bug the advisor found in rounds 2-3 was a stale one of them.  Here
hypothesis draws random interleavings of

    reset / step(action) / step_fused(action | None, n_steps) / observe / neighbors / capture_fused + replay /
    materialize / computeControl / Targets.set(plain | frozen)

on a quad fleet (chained stepping on), a hexa fleet, an interleaved quad + hexa fleet (stored type-major behind the
caller's numbering) and the same with the neighbour-downwash term, and checks EVERY operation against an oracle-driven
model at the step's bar: the model is re-seated on the device's state before each operation (tests/util.py: increments
from the device's own previous state), applies the same operation with the fp64 oracle, and the two results must agree —
states, controller memory, observation rows, returned commands.
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
hyp = pytest.importorskip("hypothesis")
from hypothesis import HealthCheck, given, settings  # noqa: E402
from hypothesis import strategies as st  # noqa: E402

from dronesim_amd import params  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from tests.test_gpu_parity import _check_obs_rows  # noqa: E402
from tests.test_gpu_two_call_loop import _downwash_part, _noise_by_id  # noqa: E402
from tests.util import K_ULP, assert_control_parity, assert_step_parity, f32, noise_terms  # noqa: E402

pytestmark = pytest.mark.gpu
DT = float(np.float32(1.0 / 240.0))

FLEETS = {
    # name: (models, type ids or None, sub-steps, physics with downwash, chained)
    "quad": (["robobee"], None, 2, False, True),
    "hexa": (["hexa_6DOF"], None, 1, False, False),
    "hexa_simple": (["hexa_6DOF_simple"], None, 2, False, False),      # morphing-hexa physics, the quad law on six actuators
    "mixed": (["robobee", "hexa_6DOF"], "interleaved", 2, False, False),
    "mixed_dw": (["tello", "hexa_6DOF"], "random", 2, True, False),
}

OPS = st.lists(st.one_of(
    st.tuples(st.just("reset")),
    st.tuples(st.just("step"), st.sampled_from(["random", "last_cmd", "out_of_range"])),
    st.tuples(st.just("step_fused"), st.booleans(), st.sampled_from([1, 1, 1, 3])),
    st.tuples(st.just("observe")),
    st.tuples(st.just("neighbors")),
    st.tuples(st.just("graph"), st.sampled_from([2, 3])),
    st.tuples(st.just("materialize")),
    st.tuples(st.just("control"), st.booleans()),
    st.tuples(st.just("targets"), st.sampled_from(["plain", "frozen_new", "frozen_same", "constant"])),
), min_size=4, max_size=14)


class Loop:
    """One env + bound controller + the oracle-driven model of both."""

    def __init__(self, kind, seed):
        from dronesim_amd.control import INDIControl
        from dronesim_amd.envs import CtrlAviary, Physics
        from dronesim_amd.fleet import Targets
        models, tids, self.sub, self.dw, chained = FLEETS[kind]
        self.kind = kind
        self.rng = np.random.default_rng(seed)
        self.n = n = 300
        self.types = [params.builtin_type(m) for m in models]
        self.tid = None
        if tids == "interleaved":
            self.tid = (np.arange(n) % 2).astype(np.uint8)
        elif tids == "random":
            self.tid = self.rng.integers(0, 2, n).astype(np.uint8)
        self.xyz = np.stack([self.rng.uniform(0, 14, n), self.rng.uniform(0, 14, n), self.rng.uniform(2, 9, n)], 1)
        self.noise_seed = 5
        self.env = CtrlAviary(models, n, initial_xyzs=self.xyz, aggregate_phy_steps=self.sub, noise_seed=self.noise_seed,
                              dict_io=False, type_ids=self.tid, chained=chained, neighbourhood_radius=1.5,
                              physics=Physics.PYB_DW if self.dw else Physics.PYB)
        self.na = self.env.n_act
        self.ctrl = INDIControl(models[-1], env=self.env)
        self.env.reset()                                  # (the controller's own reset wrote controller memory: start over)
        self.O = orc.Oracle(self.types)
        self.tg = Targets(self.env.ctx, n)
        self.tgt = np.zeros((n, 10))
        self.set_targets("plain")
        self.dtc = float(np.float32(self.sub / 240))
        self.last = np.zeros((n, 6))                      # the env's last_clipped_action (BaseAviary.py:545)
        self.use_last = True                              # what the observation rows echo: it, or the controller's command
        self.env_steps = 0
        self.cmd = None                                   # the tensor the controller returned last
        self.graphs = {}
        self._frozen = None

    # ---- helpers
    def dev_state(self):
        return self.env.state.rigid_aos(), self.env.state.mem_aos()

    def noise(self, step_index):
        return _noise_by_id(self.O, self.types, self.tid, np.arange(self.n), self.noise_seed, step_index, self.sub)

    def model_physics(self, r, m, a6, track_last):
        last = self.last if track_last else None
        nz = self.noise(self.env_steps)
        if self.dw:
            assert self.O.physics_downwash(r, m, self.sub, DT, action=a6, noise=nz, type_id=self.tid, last_action=last) == 0
        else:
            self.O.physics(r, m, self.sub, DT, action=a6, noise=nz, type_id=self.tid, last_action=last)
        self.env_steps += 1

    def check_step(self, label, r0, m0, r, m, control, applied, k=None):
        gr, gm = self.dev_state()
        part = _downwash_part(self.O, self.types, self.tid, r0, r, self.sub) if self.dw else None
        assert_step_parity(f"fuzz {self.kind} {label}", self.types, self.tid, r0, m0, self.tgt, gr, gm if control else None, r,
                           m if control else None, DT, self.dtc, self.sub, control=control, action=applied, k=k, part_rigid=part,
                           extra_terms=noise_terms(self.types, self.tid, self.n, DT, self.sub))

    def action6(self, a):
        a6 = np.zeros((self.n, 6)); a6[:, :self.na] = a
        return a6

    def applied(self, a6, m0):
        """[n, n_act] the action the physics applies: the clipped explicit one, or the stored command."""
        return np.clip(a6[:, :self.na], 0.0, 1.0) if a6 is not None else m0[:, 7:7 + self.na]

    def set_targets(self, how):
        from dronesim_amd.fleet import frozen
        n = self.n
        if how == "constant":
            c = f32(self.rng.uniform(2, 10, 3))
            self.tg.set(pos=c, yaw=0.3)
            self.tgt[:, 0:3], self.tgt[:, 9] = c, np.float32(0.3)
            return
        if how == "frozen_same" and self._frozen is not None:
            # the same object again: nothing is copied if the block still holds it — and it is copied again if a plain set came
            # in between; either way the block holds the frozen tensor's positions afterwards
            self.tg.set(pos=self._frozen[0])
            self.tgt[:, 0:3] = self._frozen[1]
            return
        tp = f32(self.xyz + self.rng.uniform(-0.6, 0.6, (n, 3)))
        t = torch.from_numpy(np.ascontiguousarray(tp.T).astype(np.float32)).to(self.env.ctx.device)
        if how.startswith("frozen"):
            self._frozen = (frozen(t), tp)
            self.tg.set(pos=self._frozen[0], yaw=0.2)
        else:
            self.tg.set(pos=t, yaw=0.2)
        self.tgt[:, 0:3], self.tgt[:, 9] = tp, np.float32(0.2)

    # ---- the operations
    def op_reset(self):
        self.env.reset()
        self.ctrl.reset()
        self.env_steps, self.use_last = 0, True
        self.last[:] = 0.0
        r, m = self.dev_state()
        np.testing.assert_allclose(r[:, 0:3], f32(self.xyz), rtol=0, atol=0)
        assert np.abs(m - self.O.reset_mem(self.n, self.tid)).max() < 1e-7          # (0.3 as an fp32)
        self.graphs.clear()                               # (a graph captured before holds a noise counter of its own)

    def op_step(self, how):
        n, na = self.n, self.na
        if how == "last_cmd" and self.cmd is not None:
            act = self.cmd                                # the very tensor computeControl returned (zero-copy paths)
            a = act.double().cpu().numpy()
        else:
            lo, hi = (-0.3, 1.3) if how == "out_of_range" else (0.3, 0.6)
            a = f32(self.rng.uniform(lo, hi, (n, na)))
            if self.tid is not None:
                for k, t in enumerate(self.types):
                    a[self.tid == k, t.n_act:] = 0.0
            act = torch.from_numpy(a.astype(np.float32)).to(self.env.ctx.device)
        r0, m0 = self.dev_state()
        obs, _, _, _ = self.env.step(act)
        r, a6 = r0.copy(), self.action6(a)
        self.model_physics(r, m0, a6, True)
        self.use_last = True
        self.check_step("step", r0, m0, r, None, False, self.applied(a6, m0))
        _check_obs_rows(f"fuzz {self.kind} step rows", self.O, obs.double().cpu().numpy(), self.dev_state()[0], self.last, self.tid,
                        self.types)

    def op_step_fused(self, with_action, n_steps):
        if self.dw:
            n_steps = 1
        r0, m0 = self.dev_state()
        a6 = None
        act = None
        if with_action:
            a = f32(self.rng.uniform(0.3, 0.6, (self.n, self.na)))
            if self.tid is not None:
                for k, t in enumerate(self.types):
                    a[self.tid == k, t.n_act:] = 0.0
            a6, act = self.action6(a), a.astype(np.float32)
        self.env.step_fused(self.tg, action=act, n_steps=n_steps)
        r, m = r0.copy(), m0.copy()
        rp, mp = r0, m0
        for k in range(n_steps):
            rp, mp = r.copy(), m.copy()
            self.model_physics(r, m, a6 if k == 0 else None, False)
            assert self.O.control(r, m, self.tgt, self.dtc, type_id=self.tid)[0] == 0
        self.use_last = False
        if n_steps == 1:
            self.check_step("step_fused", r0, m0, r, m, True, self.applied(a6, m0))
        else:
            self.check_step("step_fused x3", rp, mp, r, m, True, mp[:, 7:7 + self.na], k=K_ULP * self.sub * 4 * n_steps)

    def op_observe(self):
        rows = self.env.observe().double().cpu().numpy()
        r, m = self.dev_state()
        _check_obs_rows(f"fuzz {self.kind} observe", self.O, rows, r, self.last if self.use_last else m[:, 7:13], self.tid, self.types)

    def op_neighbors(self):
        cnt, lst = self.env.neighbors(max_k=4)
        r, _ = self.dev_state()
        d = np.linalg.norm(r[:, None, 0:3] - r[None, :, 0:3], axis=2)
        adj = (d < 1.5) & ~np.eye(self.n, dtype=bool)
        ok = ~(np.abs(d - 1.5) < 1e-4).any(1)
        np.testing.assert_array_equal(cnt.cpu().numpy()[ok], adj.sum(1)[ok])
        la = lst.cpu().numpy()
        for i in np.flatnonzero(ok)[:60]:
            got = set(int(x) for x in la[:, i] if x >= 0)
            assert got <= set(np.flatnonzero(adj[i])) and len(got) == min(4, adj[i].sum())

    def op_graph(self, steps):
        if self.dw and self.sub > 1:
            with pytest.raises(NotImplementedError):
                self.env.capture_fused(self.tg, steps)
            return
        key = (steps, self.tg.data.data_ptr())
        if key not in self.graphs:
            self.graphs[key] = self.env.capture_fused(self.tg, steps)
        # (a graph holds the targets block's address, not its contents: targets set after the capture are honoured)
        r0, m0 = self.dev_state()
        self.graphs[key].replay()
        r, m = r0.copy(), m0.copy()
        rp, mp = r0, m0
        for _ in range(steps):
            rp, mp = r.copy(), m.copy()
            self.model_physics(r, m, None, False)
            assert self.O.control(r, m, self.tgt, self.dtc, type_id=self.tid)[0] == 0
        self.use_last = False
        self.check_step(f"graph x{steps}", rp, mp, r, m, True, mp[:, 7:7 + self.na], k=K_ULP * self.sub * 4 * steps)

    def op_materialize(self):
        r0, m0 = self.dev_state()                         # (the accessors materialise themselves: nothing may change)
        self.env.materialize()
        r1, m1 = self.dev_state()
        np.testing.assert_array_equal(r0, r1); np.testing.assert_array_equal(m0, m1)

    def op_control(self, want_frozen):
        from dronesim_amd.fleet import frozen
        tp = f32(self.xyz + self.rng.uniform(-0.6, 0.6, (self.n, 3)))
        t = torch.from_numpy(tp.astype(np.float32)).to(self.env.ctx.device)
        r0, m0 = self.dev_state()
        cmd, pos_e, yaw_e = self.ctrl.computeControlFromState(self.dtc, None, target_pos=frozen(t) if want_frozen else t,
                                                              target_rpy=np.array([0.0, 0.0, 0.25]))
        tgt = np.concatenate([tp, np.zeros((self.n, 6)), np.full((self.n, 1), np.float32(0.25))], 1)
        m = m0.copy()
        rc, pe, ye = self.O.control(r0, m, tgt, self.dtc, type_id=self.tid)
        assert rc == 0
        _, gm = self.dev_state()
        assert_control_parity(f"fuzz {self.kind} computeControl", self.types, self.tid, r0, m0, tgt, gm, m, self.dtc)
        np.testing.assert_array_equal(cmd.double().cpu().numpy(), gm[:, 7:7 + self.na])
        np.testing.assert_allclose(pos_e.double().cpu().numpy(), pe, rtol=0, atol=1e-5)
        self.cmd = cmd

    def run(self, ops):
        for op in ops:
            getattr(self, "op_" + op[0])(*op[1:]) if op[0] != "targets" else self.set_targets(op[1])
        assert self.env.ctx.query(1) == 0                 # DSIM_Q_WLS_FAILURES
        self.env.close()


# one long sequence that every fleet kind runs whatever hypothesis draws: every operation, stale-cache suspects back to back
# (fused steps around Env.step and computeControl, a graph replayed across eager steps and target changes, the command
# tensor handed back after the state moved, reset in the middle)
LONG = [("step_fused", True, 1), ("step_fused", False, 1), ("step_fused", False, 1), ("graph", 2), ("step_fused", False, 3),
        ("observe",), ("control", True), ("step", "last_cmd"), ("control", False), ("step", "last_cmd"), ("step", "out_of_range"),
        ("observe",), ("targets", "frozen_new"), ("step_fused", False, 1), ("targets", "frozen_same"), ("step_fused", False, 1),
        ("graph", 2), ("targets", "constant"), ("graph", 2), ("step_fused", False, 1), ("neighbors",), ("step_fused", False, 1),
        ("materialize",), ("step_fused", False, 1), ("reset",), ("step_fused", False, 1), ("step", "random"), ("control", True),
        ("step_fused", True, 3), ("neighbors",), ("step", "last_cmd"), ("observe",)]


@pytest.mark.parametrize("kind", list(FLEETS))
@settings(max_examples=100, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
@hyp.example(ops=LONG, seed=1)
@given(ops=OPS, seed=st.integers(0, 10_000))
def test_random_call_sequences_against_the_oracle_model(kind, ops, seed):
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device; the product has no CPU fallback")
    Loop(kind, seed).run(ops)


ADAPTOR_OPS = st.lists(st.one_of(
    st.tuples(st.just("reset")),
    st.tuples(st.just("step"), st.sampled_from(["host", "device", "device_same", "device_noncontiguous"])),
    st.tuples(st.just("observe")),
), min_size=3, max_size=12)


@pytest.mark.parametrize("mode", ["velocity", "rpyt"])
@settings(max_examples=40, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
@given(ops=ADAPTOR_OPS, seed=st.integers(0, 10_000))
def test_random_call_sequences_of_the_adaptor_envs(mode, ops, seed):
    """VelocityAviary / RPYTAviary: reset / step / observe in random order, the action handed over as a host array, as a
    fresh [N, 4] device tensor, as the SAME device tensor again (the prepared launch of the one-launch step is re-used) or as
    a non-contiguous view (goes through the env's buffer) — every step against the oracle (control part on the state before
    the physics, physics part with the command the device computed), the returned rows against orc_state_vector."""
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device; the product has no CPU fallback")
    from dronesim_amd.envs import RPYTAviary, VelocityAviary
    from tests.test_gpu_parity import _noise_block
    n, sub, noise_seed = 300, 2, 7
    rng = np.random.default_rng(seed)
    xyz = np.stack([rng.uniform(-5, 5, n), rng.uniform(-5, 5, n), rng.uniform(1, 5, n)], 1)
    env = (VelocityAviary if mode == "velocity" else RPYTAviary)(["robobee"], n, initial_xyzs=xyz, aggregate_phy_steps=sub,
                                                                   noise_seed=noise_seed, dict_io=False)
    t = params.builtin_type("robobee")
    O = orc.Oracle([t])
    dtc = float(np.float32(sub / 240))
    steps, same = 0, None
    for op in ops:
        if op[0] == "reset":
            env.reset()
            steps = 0
            np.testing.assert_array_equal(env.state.rigid_aos()[:, 0:3], f32(xyz))
            continue
        if op[0] == "observe":
            rows = env.observe().double().cpu().numpy()
            r, m = env.state.rigid_aos(), env.state.mem_aos()
            last = np.zeros((n, 6)); last[:, :4] = env._last_action[:, :n].T.double().cpu().numpy()
            _check_obs_rows(f"fuzz adaptor[{mode}] observe", O, rows, r, last, None, [t])
            continue
        if mode == "velocity":
            act = np.concatenate([rng.uniform(-1, 1, (n, 3)), rng.uniform(0, 0.3, (n, 1))], 1)
        else:
            act = np.concatenate([rng.uniform(-0.5, 0.5, (n, 3)), rng.uniform(0.3, 0.6, (n, 1))], 1)
        act = f32(act)
        a32 = torch.from_numpy(act.astype(np.float32))
        if op[1] == "host":
            handed = act.astype(np.float32)
        elif op[1] == "device":
            handed = a32.to(env.ctx.device)
        elif op[1] == "device_same":
            if same is None:
                same = torch.zeros((n, 4), device=env.ctx.device)
            same.copy_(a32)                                   # the same tensor object, new contents
            handed = same
        else:
            wide = torch.zeros((n, 8), device=env.ctx.device)
            wide[:, ::2] = a32.to(env.ctx.device)
            handed = wide[:, ::2]
        r0, m0 = env.state.rigid_aos(), env.state.mem_aos()
        obs, _, _, _ = env.step(handed)
        got_r, got_m = env.state.rigid_aos(), env.state.mem_aos()
        rc0, m_ref = r0.copy(), m0.copy()
        assert O.adaptor_step(0 if mode == "velocity" else 1, rc0, m_ref, act, 0, DT, dtc) == 0
        tgt = np.concatenate([r0[:, 0:3], np.zeros((n, 7))], 1)
        if mode == "velocity":
            nrm = np.linalg.norm(act[:, 0:3], axis=1, keepdims=True)
            tgt[:, 3:6] = t.max_speed_kmh / 3.6 * np.abs(act[:, 3:4]) * np.divide(act[:, 0:3], nrm, out=np.zeros((n, 3)), where=nrm > 0)
        assert_control_parity(f"fuzz adaptor[{mode}] control", [t], None, r0, m0, tgt, got_m, m_ref, dtc)
        r_ref = r0.copy()
        a6 = np.zeros((n, 6)); a6[:, :4] = got_m[:, 7:11]
        O.physics(r_ref, got_m.copy(), sub, DT, action=a6, noise=_noise_block(O, [t], None, n, noise_seed, steps, sub))
        assert_step_parity(f"fuzz adaptor[{mode}] physics", [t], None, r0, got_m, tgt, got_r, None, r_ref, None, DT, dtc, sub,
                           control=False, action=got_m[:, 7:11], extra_terms=noise_terms([t], None, n, DT, sub))
        _check_obs_rows(f"fuzz adaptor[{mode}] rows", O, obs.double().cpu().numpy(), f32(got_r), a6, None, [t])
        steps += 1
    env.close()
