"""The reference's two alternate action adaptors at fleet scale (dronesim/envs/VelocityAviary.py, RPYTAviary.py): the action
is turned into a PWM command by (part of) the INDI law INSIDE ``step()`` — on the current state — and the physics follows
(``dsim_step_adaptor``)."""
from __future__ import annotations

import ctypes

import torch

from .. import _native as nat
from .fleet_aviary import CtrlAviary, Physics, _StepPlan


class _AdaptorAviary(CtrlAviary):
    """Shared body of the two alternate action adaptors: the action is turned into a PWM command by
    (part of) the INDI law INSIDE step() — on the current state — and the physics follows."""

    _MODE = -1

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        if self.PHYSICS != Physics.PYB:
            # dsim_step_adaptor flies plain PYB (no drag / ground effect / downwash inputs): refuse instead of
            # silently ignoring the mode the caller asked for
            raise NotImplementedError(f"{type(self).__name__}: physics={self.PHYSICS} — the action-adaptor envs "
                                      "support Physics.PYB only")

    def step(self, action):
        self.materialize()
        self._chain_ok = False
        plan = self._step_plan
        if plan is not None and plan.matches(self._step_key(), action, self._is_action_rows):
            # (the prepared launch of the one-launch form, re-used while nothing it was built from has changed; the action is
            # a parameter of the call)
            plan.args.step_index = self._env_steps
            plan.action, plan.action_ptr = action, action.data_ptr()
            self._action_keep = action
            nat.check(self.ctx.lib.dsim_step_adaptor(self.ctx.handle, self.ctx.stream_ptr(), self.NUM_DRONES, plan.state_view,
                                                     action.data_ptr(), self._MODE, plan.echo_ptr, plan.args_ref))
            self._use_last_action = True
            self.step_counter += self.AGGR_PHY_STEPS
            self._env_steps += 1
            return plan.out, self._computeReward(), self._computeDone(), plan.info
        args = self.step_args(self.AGGR_PHY_STEPS * self.TIMESTEP)
        # A homogeneous fleet in whole tiles steps in ONE launch that takes the action as the caller holds it ([N, 4] rows
        # on the device: no transpose) and writes Env.step's observation rows itself (k_adaptor_fast).
        one_launch = (not self.dict_io and self.order is None and len(self.types) == 1 and self.state.n_pad % 256 == 0
                      and not self.ground_plane and self._type_id is None)
        rows_in = (one_launch and isinstance(action, torch.Tensor) and action.dtype == torch.float32 and action.is_contiguous()
                   and action.device == self.ctx.device and tuple(action.shape) == (self.NUM_DRONES, 4)
                   and action.data_ptr() % 16 == 0)
        if rows_in:
            args.options |= nat.OPT_ACTION_ROWS
            act_ptr = action.data_ptr()
        else:
            self._load_action(action)
            act_ptr = self._action_buf.data_ptr()
        obs = None
        if one_launch:
            obs = self._obs_tensor()
            args.obs_out, args.obs_width = obs.data_ptr(), 20
        nat.check(self.ctx.lib.dsim_step_adaptor(self.ctx.handle, self.ctx.stream_ptr(), self.NUM_DRONES,
                                                 self.state.view(), act_ptr, self._MODE,
                                                 self._last_action.data_ptr(), ctypes.byref(args)))
        self._use_last_action = True
        self.step_counter += self.AGGR_PHY_STEPS
        self._env_steps += 1
        out = self._computeObs(obs)
        self._step_plan = None
        if rows_in and out is obs:
            self._step_plan = _StepPlan(key=self._step_key(), action=action, action_ptr=action.data_ptr(), rows_in=True, args=args,
                                        state_view=self.state.view(), args_ref=ctypes.byref(args),
                                        echo_ptr=self._last_action.data_ptr(), out=out, info=self._computeInfo())
        return out, self._computeReward(), self._computeDone(), self._computeInfo()


class VelocityAviary(_AdaptorAviary):
    """dronesim/envs/VelocityAviary.py: action = (vx, vy, vz, speed fraction) per drone; the env runs
    the full INDI law with target_pos = current position, target yaw = current yaw and
    target_vel = SPEED_LIMIT |a3| unit(a0..2) (VelocityAviary.py:241-262), SPEED_LIMIT =
    MAX_SPEED_KMH / 3.6 (:92-94)."""

    _MODE = nat.ADAPT_VELOCITY

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.SPEED_LIMIT = [t.max_speed_kmh * (1000 / 3600) for t in (self.drones or self.types)]


class RPYTAviary(_AdaptorAviary):
    """dronesim/envs/RPYTAviary.py: action = (p, q, r body-rate set-points, thrust) per drone, fed to
    INDIControl._INDIRateControl only (RPYTAviary.py:181-193)."""

    _MODE = nat.ADAPT_RPYT
