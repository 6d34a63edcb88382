"""Batched INDI controller — the reference's ``BaseControl`` / ``INDIControl`` surface
(dronesim/control/BaseControl.py:13-149, dronesim/control/INDIControl.py:25-227) for a
whole fleet: same method names, argument meaning and return triple, one HIP launch
per call instead of one Python object per drone.

The per-drone controller memory (``last_vel``, ``last_rates``, ``last_thrust``,
``cmd``; INDIControl.py:109-146) lives in the fleet state tensor.  A controller is
either bound to an env (``INDIControl(env=env)`` — then it reads the env's state in
place, no copies) or stand-alone (it owns a state block and the caller passes
positions / quaternions / velocities explicitly, as with the reference).
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import Optional, Union

import numpy as np
import torch

from .. import _native as nat
from ..fleet import Context, FleetState, Frozen, Targets
from ..params import DroneType, builtin_type


@dataclass
class _ControlPlan:
    """The prepared launch of a repeated ``computeControl`` call of the reference-shaped loop (state read from the bound env,
    the same frozen target_pos, the same small host vectors compared by content, the same buffers): launched again as it is."""

    key: tuple
    state_view: object
    targets_view: object
    args_ref: object            # ctypes.byref(args)
    pos_e_ptr: int
    yaw_e_ptr: int
    cmd_ptr: int
    out: tuple                  # the (cmd, pos_e, yaw_e) triple handed out
    cmd_token: object           # what the env recognises when the command comes back as the next action, or None
    args: object                # nat.StepArgs (kept alive)

    def matches(self, key: tuple) -> bool:
        return self.key == key


def _as3(x, n, device) -> torch.Tensor:
    """-> [3, n] float32 device tensor from (3,), (n,3) or (3,n) input."""
    t = torch.as_tensor(np.asarray(x) if not torch.is_tensor(x) else x, dtype=torch.float32, device=device)
    if t.numel() == 3:
        return t.reshape(3, 1).expand(3, n)
    if t.shape == (n, 3):
        return t.T
    return t.reshape(3, n)


class BaseControl:
    """dronesim/control/BaseControl.py:13-149."""

    def __init__(self, drone_model: Union[str, DroneType] = "tello", g: float = 9.8, *, num_drones: int = 1,
                 env=None, device: int = 0, layout: str = "soa"):
        self.DRONE_MODEL = drone_model if isinstance(drone_model, str) else drone_model.name
        self.type = builtin_type(drone_model) if isinstance(drone_model, str) else drone_model
        self.GRAVITY = g * self.type.ctrl_mass       # BaseControl.py:36
        self.KF, self.KM = self.type.kf, self.type.km
        self.env = env
        if env is not None:
            self.ctx, self.state, self.n = env.ctx, env.state, env.NUM_DRONES
            self._type_id = env._type_id
        else:
            self.ctx = Context([self.type], device)
            self.n = num_drones
            self.state = FleetState(self.ctx, num_drones, layout)
            self._type_id = None
        self._targets = Targets(self.ctx, self.n, self.state.layout)
        self._pos_e = torch.zeros((3, self.state.n_pad), dtype=torch.float32, device=self.ctx.device)
        self._yaw_e = torch.zeros((self.state.n_pad,), dtype=torch.float32, device=self.ctx.device)
        # the command as a plain SoA [n_act][n_pad] array, written by the control launch: its transposed view is what
        # computeControl returns, and Env.step takes it back as the action without a copy
        self._cmd = torch.zeros((self.ctx.n_act, self.state.n_pad), dtype=torch.float32, device=self.ctx.device)
        self._outputs_placed = False     # large quad fleets: the three arrays above are re-allocated by trial at the first call
        self._plan = None                # the prepared launch of computeControl's repeated-call path
        self.reset()

    def reset(self):
        """BaseControl.reset (BaseControl.py:51-57)."""
        self.control_counter = 0

    def computeControlFromState(self, control_timestep, state, target_pos, target_vel=np.zeros(3),
                                target_acc=np.zeros(3), target_rpy=np.zeros(3), target_rpy_rates=np.zeros(3)):
        """BaseControl.computeControlFromState (BaseControl.py:61-103): slices
        ``state[0:3], [3:7], [10:13], [13:16]``.  ``state`` is one 20-vector or [N,20];
        ``None`` with a bound env means "the env's current device state"."""
        if state is None:
            if self.env is None:
                raise ValueError("state=None needs a controller bound to an env")
            return self.computeControl(control_timestep, None, None, None, None, target_pos, target_vel,
                                       target_acc, target_rpy, target_rpy_rates)
        s = torch.as_tensor(np.asarray(state) if not torch.is_tensor(state) else state)
        single = s.ndim == 1
        s = s.reshape(-1, s.shape[-1])
        out = self.computeControl(control_timestep, s[:, 0:3], s[:, 3:7], s[:, 10:13], s[:, 13:16], target_pos,
                                  target_vel, target_acc, target_rpy, target_rpy_rates)
        return self._maybe_single(out, single)

    def _maybe_single(self, out, single):
        if not single:
            return out
        cmd, pos_e, yaw_e = out
        return (cmd[0].double().cpu().numpy(), pos_e[0].double().cpu().numpy(), float(yaw_e[0]))

    def computeControl(self, *a, **k):
        raise NotImplementedError     # BaseControl.py:107-149


class INDIControl(BaseControl):
    """dronesim/control/INDIControl.py:25-490 for N drones of one quad type."""

    def reset(self):
        """INDIControl.reset (INDIControl.py:109-146): last_vel = last_rates = 0,
        last_thrust = 0, cmd = 0 (6DOF: 0.3 / 0.5)."""
        super().reset()
        st = self.state
        st.set_fields(13, torch.zeros((6, st.n), device=self.ctx.device))
        tid = getattr(self.env, "type_ids_caller", None) if self.env is not None else None
        if tid is not None and len(self.env.types) > 1:
            # a heterogeneous fleet (one controller object for the whole of it): every drone gets the reset values of ITS
            # type's controller class (INDIControl.py:127-129 / INDIControl_6DOF.py:232-234)
            types = self.env.types
            thrust = np.array([t.reset_thrust for t in types], dtype=np.float32)[tid]
            cmd = np.zeros((st.n_fields - 20, st.n), dtype=np.float32)
            for k, t in enumerate(types):
                cmd[: t.n_act, tid == k] = t.reset_cmd
            st.set_fields(19, torch.from_numpy(thrust).reshape(1, -1))
            st.set_fields(20, torch.from_numpy(cmd))
            return
        st.set_fields(19, torch.full((1, st.n), self.type.reset_thrust, device=self.ctx.device))
        cmd = torch.zeros((st.n_fields - 20, st.n), device=self.ctx.device)
        cmd[: self.type.n_act] = self.type.reset_cmd
        st.set_fields(20, cmd)

    def _place_targets(self, a) -> None:
        """Large fleets: the targets block this launch READS beside the state block it updates wants the SAME 16 GiB window as
        the state (tools/region_probe.py --arena, G: 136 us against 143 us with it in the outputs' window, 4 194 304 quads) —
        and where PyTorch puts it is either.  The launch has no neutral form: snapshot of the state block and of the outputs,
        real passes on candidates holding a copy of the targets (a 4 GiB walk, the fastest kept: as CtrlAviary does for the
        fused step's targets), snapshot back."""
        from .. import placement
        st, old = self.state, self._targets.data
        if not (self.ctx.placement and 4 * old.numel() >= placement.MIN_BYTES and not getattr(self.env, "_graph_made", False)):
            return
        snap = st.data.clone()
        outs = (self._cmd.clone(), self._pos_e.clone(), self._yaw_e.clone())
        view, tview, ref = st.view(), self._targets.view(), ctypes.byref(a)
        lib, h = self.ctx.lib, self.ctx.handle
        filled = set()

        def trial(c):
            if c.data_ptr() not in filled:           # (the first pass on a candidate is the untimed one)
                filled.add(c.data_ptr())
                c.copy_(old)
            tview.base = c.data_ptr()
            nat.check(lib.dsim_control2(h, self.ctx.stream_ptr(), self.n, view, tview, ref, self._pos_e.data_ptr(),
                                        self._yaw_e.data_ptr(), self._cmd.data_ptr()))
        # (the env's own predicate: no move of its state block for fleets the placement does not apply to)
        room = (self.env._take_read_room(old.numel()) if (self.env is not None and self.env._placement_applies(4 * old.numel())) else None)
        t_room = None
        if room is not None:                          # the block right behind the state, in the state's own allocation
            room = room.view(old.shape)
            view = st.view()
            t_room = placement._event_timer(trial, room, 3)
        keep = placement.place_rows(self.ctx.device, tuple(old.shape), trial, report=self.ctx.placement_log,
                                    label="computeControl targets", clearly=0.0, walk_bytes=4 << 30, ctx=self.ctx)
        if t_room is not None:
            rep = self.ctx.placement_log[-1]
            rep["behind_the_state_pass_us"] = round(t_room, 1)
            if t_room <= 1.01 * rep.get("chosen_pass_us", 0.0):
                keep = room
                rep["decided_by"] = "the block behind the state, in its allocation, is as fast as the walk's best: kept"
        keep.copy_(old)
        self._targets.data = keep
        st.data.copy_(snap)
        self._cmd.copy_(outs[0]); self._pos_e.copy_(outs[1]); self._yaw_e.copy_(outs[2])
        self._plan = None

    def _place_outputs(self, a) -> None:
        """Large fleets: where the arrays this launch WRITES (command, position error, yaw error) lie relative to the state
        block whose controller memory it updates is worth ~6 % of it (placement.py).  A controller bound to an env takes the
        room the env left behind its placed observation rows (one allocation, one search); a stand-alone one searches for
        itself — the launch has no neutral form, so: snapshot of the state block, real passes on a few candidates,
        snapshot back."""
        from .. import placement
        st, n_pad, na = self.state, self.state.n_pad, self.ctx.n_act
        if not self.ctx.placement:
            return
        if self.env is not None and self.env._written_tail is None and self.env._placement_applies(4 * self.n * (16 + na)):
            self.env._obs_tensor()                       # (the env places its rows, and the room behind them, now)
        tail = getattr(self.env, "_written_tail", None) if self.env is not None else None
        if tail is not None and tuple(tail.shape) == (na + 4, n_pad):
            # the room behind the env's placed rows (one allocation, one search) — unless the arrays this controller was
            # built with serve the launch better (never worse than no search): real passes on both behind a snapshot
            tail.zero_()
            own = (self._cmd, self._pos_e, self._yaw_e)
            snap = st.data.clone()
            view, tview, ref = st.view(), self._targets.view(), ctypes.byref(a)

            def passes(c, p, y):
                nat.check(self.ctx.lib.dsim_control2(self.ctx.handle, self.ctx.stream_ptr(), self.n, view, tview, ref,
                                                     p.data_ptr(), y.data_ptr(), c.data_ptr()))
            t_tail = placement._event_timer(lambda _: passes(tail[0:na], tail[na:na + 3], tail[na + 3]), None, 3)
            t_own = placement._event_timer(lambda _: passes(*own), None, 3)
            st.data.copy_(snap)
            entry = {"array": "computeControl outputs", "bytes": 4 * tail.numel(), "behind_the_rows_pass_us": round(t_tail, 1),
                     "plain_pass_us": round(t_own, 1)}
            if t_own < 0.99 * t_tail:
                for x in own:
                    x.zero_()
                entry["placed"] = "the plain allocation is faster than the room behind the env's observation rows: kept"
            else:
                tail.zero_()
                self._cmd, self._pos_e, self._yaw_e = tail[0:na], tail[na:na + 3], tail[na + 3]
                entry["placed"] = "behind the env's observation rows (one allocation, one search)"
            self.ctx.placement_log.append(entry)
            return
        if not (self.env is None and self._type_id is None and 4 * (na + 4) * n_pad >= placement.MIN_BYTES):
            return
        snap = st.data.clone()
        view, tview, ref = st.view(), self._targets.view(), ctypes.byref(a)
        lib, h = self.ctx.lib, self.ctx.handle

        def trial(c):      # c: [n_act + 4, n_pad] = cmd | pos_e (3) | yaw_e (1)
            nat.check(lib.dsim_control2(h, self.ctx.stream_ptr(), self.n, view, tview, ref, c[na:na + 3].data_ptr(),
                                        c[na + 3].data_ptr(), c[0:na].data_ptr()))
        keep = placement.place_rows(self.ctx.device, (na + 4, n_pad), trial, report=self.ctx.placement_log,
                                    label="computeControl outputs", ctx=self.ctx, stride_bytes=placement.STRIDE_BYTES)
        st.data.copy_(snap)
        self._cmd, self._pos_e, self._yaw_e = keep[0:na], keep[na:na + 3], keep[na + 3]

    def computeControl(self, control_timestep, cur_pos, cur_quat, cur_vel, cur_ang_vel, target_pos,
                       target_vel=np.zeros(3), target_acc=np.zeros(3), target_rpy=np.zeros(3),
                       target_rpy_rates=np.zeros(3)):
        """INDIControl.computeControl (INDIControl.py:154-227).  Returns
        ``(cmd [N,4] PWM, pos_e [N,3], yaw_e [N])`` as device tensors.  ``target_rpy_rates``
        is accepted and ignored, as in the reference (:404-410)."""
        self.control_counter += 1
        n, dev, st = self.n, self.ctx.device, self.state
        if self.env is not None:
            # the controller memory it is about to differentiate against must be the stored one: a chained
            # step_fused() sequence leaves last_vel / last_rates stale until materialized
            self.env.materialize()          # (also joins a deferred WLS fallback pass: this launch reads the commands)
            self.env._chain_ok = False
        # The loop of the reference's examples calls this every control period with the SAME targets
        # (examples/fly_INDI.py:229-239), and at the reference's own fleet sizes the Python in front of the launch is what an
        # iteration costs (4 096 quads: 13 us of it around a 4 us kernel).  When nothing that goes into the launch has changed
        # since the last call — state read from the bound env, the same frozen target_pos, the same small host vectors (compared
        # by content), the same buffers — the prepared argument block is launched again as it is.
        key = None
        if cur_pos is None and isinstance(target_pos, Frozen):
            small = []
            for x in (target_vel, target_acc, target_rpy):
                if torch.is_tensor(x) or np.size(x) != 3:
                    small = None
                    break
                small.append(np.asarray(x, dtype=np.float32).tobytes())
            if small is not None:
                key = (target_pos, float(control_timestep), small[0], small[1], small[2], st.data.data_ptr(),
                       self._targets.data.data_ptr(), self._targets.version, id(getattr(self.env, "_runs", None)),
                       getattr(self.env, "_tuning", 0), self._cmd.data_ptr() if self._cmd is not None else 0)
                plan = self._plan
                if plan is not None and plan.matches(key):
                    nat.check(self.ctx.lib.dsim_control2(self.ctx.handle, self.ctx.stream_ptr(), n, plan.state_view, plan.targets_view,
                                                         plan.args_ref, plan.pos_e_ptr, plan.yaw_e_ptr, plan.cmd_ptr))
                    if plan.cmd_token is not None:
                        self.env._cmd_token = plan.cmd_token
                    return plan.out
        if cur_pos is not None:                       # explicit state (stand-alone use)
            st.set_fields(0, _as3(cur_pos, n, dev))
            q = torch.as_tensor(np.asarray(cur_quat) if not torch.is_tensor(cur_quat) else cur_quat,
                                dtype=torch.float32, device=dev)
            st.set_fields(3, q.reshape(4, 1).expand(4, n) if q.numel() == 4 else (q.T if q.shape == (n, 4) else q))
            st.set_fields(7, _as3(cur_vel, n, dev))
            st.set_fields(10, _as3(cur_ang_vel, n, dev))
        if torch.is_tensor(target_rpy):
            yaw = target_rpy.to(dev, torch.float32)
            yaw = (yaw[..., 2].reshape(-1) if yaw.shape[-1] == 3 else yaw.reshape(3, -1)[2]).reshape(1, -1)
        else:                                             # only target_rpy[2] is read (INDIControl.py:341)
            r = np.asarray(target_rpy, dtype=np.float32)
            yaw = float(r.reshape(-1)[2]) if r.size == 3 else np.ascontiguousarray(
                r[..., 2] if r.shape[-1] == 3 else r.reshape(3, -1)[2]).reshape(1, -1)

        def as_target(x):      # a bare 3-vector stays one: Targets.set skips a constant it already holds, and a Frozen
            if isinstance(x, Frozen):      # tensor it has already copied (fleet.frozen)
                return x
            return x if (not torch.is_tensor(x) and np.size(x) == 3) else _as3(x, n, dev)
        self._targets.set(pos=as_target(target_pos), vel=as_target(target_vel), acc=as_target(target_acc), yaw=yaw)
        a = nat.StepArgs()
        a.phys_substeps, a.dt_phys, a.dt_ctrl = 0, float(control_timestep), float(control_timestep)
        a.options, a.noise_seed, a.step_index = 0, 0, 0
        a.noise_replay, a.action = None, None
        a.type_id = self._type_id.data_ptr() if self._type_id is not None else None
        a.options = getattr(self.env, "_tuning", 0)
        runs = getattr(self.env, "_runs", None) if self.env is not None else None
        if runs is not None:        # a fleet stored as runs of one type: the single-type bodies (k_control_runs)
            a.runs, a.n_runs = ctypes.addressof(runs), len(runs)
        caller_io = bool(getattr(self.env, "_caller_io", False)) if self.env is not None else False
        if caller_io:               # the triple comes back in the caller's numbering straight from the launch
            a.options |= nat.OPT_CALLER_IO
            a.drone_id = st.order.drone_id(st.n_pad).data_ptr()
        if not self._outputs_placed:
            self._outputs_placed = True
            if self.ctx.placement:
                # (the trials are real passes of this law: what they add to the WLS counters is not the fleet's history)
                before = [self.ctx.query(w) for w in (nat.QUERY_WLS_FALLBACKS, nat.QUERY_WLS_FAILURES)]
                self._place_outputs(a)
                self._place_targets(a)
                for w, b in zip((nat.QUERY_WLS_FALLBACKS, nat.QUERY_WLS_FAILURES), before):
                    self.ctx.query_offsets[w] = self.ctx.query_offsets.get(w, 0) + self.ctx.query(w) - b
        sview, tview = st.view(), self._targets.view()
        nat.check(self.ctx.lib.dsim_control2(self.ctx.handle, self.ctx.stream_ptr(), n, sview, tview, ctypes.byref(a),
                                             self._pos_e.data_ptr(), self._yaw_e.data_ptr(), self._cmd.data_ptr()))
        order = st.order
        if order is None or caller_io:
            out = (self._cmd[:, :n].T, self._pos_e[:, :n].T, self._yaw_e[:n])
            if key is not None:
                # (the key is completed with what this call settled: the buffers of the outputs and the targets' version)
                key = key[:6] + (self._targets.data.data_ptr(), self._targets.version) + key[8:10] + (self._cmd.data_ptr(),)
                self._plan = _ControlPlan(key, sview, tview, ctypes.byref(a), self._pos_e.data_ptr(), self._yaw_e.data_ptr(),
                                          self._cmd.data_ptr(), out, None, a)
            return out
        # a fleet stored in another order than the caller's: the triple goes back in the caller's numbering; the env
        # recognises the command tensor when it comes back as the next action and takes the storage-order array as is
        cmd = order.to_caller(self._cmd[:, :n], 1).T
        if self.env is not None:
            self.env._cmd_token = (cmd, cmd._version, self._cmd)
        return cmd, order.to_caller(self._pos_e[:, :n], 1).T, order.to_caller(self._yaw_e[:n], 0)
