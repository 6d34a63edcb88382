"""Fleet-sized stand-in for the reference's gym surface.

Mirrors ``BaseAviary`` / ``CtrlAviary`` for the hot path only
(dronesim/envs/BaseAviary.py:406-555 ``reset``/``step``; dronesim/envs/CtrlAviary.py
``_preprocessAction`` :258-263, ``_computeObs`` :225-232, dummy reward/done/info
:267-310): same constructor keywords, same ``reset()`` / ``step(action)`` / ``close()``
contract, same 20/22-wide state vector — but the whole fleet advances in one HIP
kernel launch and the state lives in HBM as a blocked-SoA tensor.

Rendering, video, GUI sliders, obstacles and the non-quad airframes are outside
the hot path ("out of scope") and raise if requested.
"""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass
from typing import Optional, Sequence, Union

import numpy as np
import torch

from .. import _native as nat
from ..fleet import Context, FleetState, Targets, WaypointTargets
from ..params import DroneType, builtin_type
from ..placement import PlacedFleetArrays
from .observation import FleetObs, FleetObservation  # noqa: F401
from .physics import Physics


# dict-of-ndarray I/O (the reference's format) is produced up to this many drones;
# beyond it step()/reset() return device tensors.
DICT_IO_MAX_DRONES = 64


@dataclass
class _StepPlan:
    """The prepared launch of a repeated ``step(action)`` call (the reference-shaped loop hands step() the tensor computeControl
    returned, every iteration the same object over the same buffers): launched again with the step counter moved on while
    nothing it was built from has changed."""

    key: tuple                  # what the argument block was built from: buffers, options, sub-steps, runs, seed (CtrlAviary._step_key)
    action: object              # the action tensor of the last call (held: its identity cannot be recycled)
    action_ptr: int             # ... and the device pointer the launch read
    rows_in: bool               # the launch takes [N, 4] rows as the caller holds them: ANY such tensor may come next
    args: object                # nat.StepArgs
    state_view: object
    args_ref: object            # ctypes.byref(args)
    echo_ptr: int               # the env's last_clipped_action
    out: object                 # the observation handed out
    info: dict

    def matches(self, key: tuple, action, is_rows) -> bool:
        if key != self.key:
            return False
        return (action is self.action and action.data_ptr() == self.action_ptr) or (self.rows_in and is_rows(action))


@dataclass
class _FusedPlan:
    """The prepared launch of a repeated ``step_fused(targets)`` call."""

    key: tuple                  # (control_timestep, n_steps, chained)
    args: object
    state_view: object
    targets_view: object
    args_ref: object
    targets: object             # the targets object itself (compared by identity while it is alive)
    targets_ptrs: tuple         # ... and the device pointers it was built from (a Targets whose tensor was swapped does not match)

    def matches(self, key: tuple, targets, ptrs: tuple) -> bool:
        return self.targets is targets and self.key == key and self.targets_ptrs == ptrs


class CtrlAviary(PlacedFleetArrays, FleetObservation):
    """PWM-action fleet environment (reference: ``CtrlAviary``)."""

    def __init__(
        self,
        drone_model: Union[Sequence[str], Sequence[DroneType]] = ("tello",),
        num_drones: int = 1,
        neighbourhood_radius: float = np.inf,
        initial_xyzs=None,
        initial_vels=None,
        initial_rpys=None,
        physics: Physics = Physics.PYB,
        freq: int = 240,
        aggregate_phy_steps: int = 1,
        gui=False,
        record=False,
        obstacles=False,
        user_debug_gui=True,
        *,
        device: int = 0,
        layout: Optional[str] = None,
        noise_seed: Optional[int] = None,
        dict_io: Optional[bool] = None,
        dist=None,
        chained: bool = False,
        downwash_exchange: str = "allgather",
        type_ids=None,
        neighbors_k: int = 0,
        options: int = 0,
        ground_plane: Optional[bool] = None,
        storage: str = "auto",
        downwash_split: Optional[bool] = None,
        defer_fallback: bool = False,
        placement: bool = False,
        dyn_ang_vel: str = "reference",
        noise: str = "auto",
        downwash_keep: Optional[int] = None,
        downwash_skin: float = 0.1,
    ):
        if gui or record or obstacles:
            raise NotImplementedError("gui/record/obstacles are rendering features outside the hot path")
        # the add-on terms of the PYB_* modes (dead code in the reference fork, intended formulas); Physics.DYN: the
        # reference's own explicit model, BaseAviary._dynamics (BaseAviary.py:1767-1828; DSIM_OPT_DYN)
        self._phys_options = {Physics.PYB: 0, Physics.PYB_DW: 0, Physics.PYB_GND: nat.OPT_GROUND,
                              Physics.PYB_DRAG: nat.OPT_DRAG, Physics.DYN: nat.OPT_DYN,
                              Physics.PYB_GND_DRAG_DW: nat.OPT_GROUND | nat.OPT_DRAG}[physics]
        if dyn_ang_vel not in ("reference", "body_rates"):
            raise ValueError(dyn_ang_vel)
        if physics == Physics.DYN:
            # What the state vector's ang_v holds under DYN.  "reference": the placeholder (-1, -1, -1) the reference stores
            # ("ang_vel not computed by DYN", BaseAviary.py:1821-1826) — what its INDI controller then reads, which is why that
            # loop tumbles.  "body_rates": R(quat) rpy_rates, a product-defined deviation with which the mode is flyable.
            if dyn_ang_vel == "body_rates":
                self._phys_options |= nat.OPT_DYN_BODY_RATES
            if ground_plane:
                raise ValueError("Physics.DYN sets the pose itself and skips p.stepSimulation (BaseAviary.py:541-543, "
                                 "1814-1819): the ground plane never acts")
            ground_plane = False
        # The reference always loads plane.urdf (BaseAviary.py:660).  ground_plane=True enforces z = 0 by the contact
        # model of oracle/dsim_oracle.c:orc_plane_contact (DSIM_OPT_PLANE, the general kernels); None = on for the
        # reference-sized fleets that get dict observations, off for the large in-flight fleets of the fast kernels,
        # where ground_contacts() reports how many drone-steps would have touched it.
        if ground_plane is None:
            ground_plane = num_drones <= DICT_IO_MAX_DRONES and (dict_io is None or dict_io)
        self.ground_plane = bool(ground_plane)
        if self.ground_plane:
            self._phys_options |= nat.OPT_PLANE
        # tuning bits handed to every call (nat.OPT_STREAM_ON/_OFF; the A/B knobs of a variants build); results do not depend on them
        self._tuning = int(options) & nat.TUNING_MASK
        # The rotor-noise stream is product-defined (the reference draws from numpy's unseeded global generator,
        # BaseAviary.py:1518-1525): Box-Muller pairs on a lattice (include/dronesim_amd.h: noise_seed).  "fine": 65 536 x 65 536
        # points (|n| <= 4.86 sigma), free on kernels of one physics sub-step; "coarse" (= "lattice256"): 256 x 256 (|n| <= 3.53
        # sigma), what the kernels that loop over several sub-steps can afford.  "auto": fine when Env.step is ONE sub-step
        # (aggregate_phy_steps == 1), coarse otherwise.  The env passes the lattice explicitly with every launch, so that an
        # Env.step it splits into single-sub-step launches (the per-sub-step neighbour downwash) draws the same stream.
        if noise not in ("auto", "fine", "coarse", "lattice256"):
            raise ValueError(noise)
        if noise == "auto":
            noise = "fine" if int(aggregate_phy_steps) == 1 else "coarse"
        self.noise = "coarse" if noise == "lattice256" else noise
        self._tuning |= nat.OPT_NOISE_FINE if self.noise == "fine" else nat.OPT_NOISE_COARSE
        self.neighbors_k = int(neighbors_k)
        if isinstance(drone_model, (str, DroneType)):
            drone_model = [drone_model]
        models = list(drone_model)
        self._type_names, types, tid = [], [], np.zeros(num_drones, dtype=np.uint8)
        if type_ids is not None:
            # large mixed fleets: drone_model is the table of distinct models, type_ids[i] indexes it
            tid = np.asarray(type_ids, dtype=np.uint8).reshape(num_drones)
            if int(tid.max(initial=0)) >= len(models):
                raise ValueError("type_ids index past drone_model")
            types = [builtin_type(m) if isinstance(m, str) else m for m in models]
            self._type_names = [t.name for t in types]
        else:
            if len(models) != 1 and len(models) != num_drones:
                raise ValueError("drone_model must name one model per drone (or one for all)")
            # type table: unique models in first-seen order
            for i, m in enumerate(models):
                key = m if isinstance(m, str) else m.name
                if key not in self._type_names:
                    self._type_names.append(key)
                    types.append(builtin_type(m) if isinstance(m, str) else m)
                tid[i] = self._type_names.index(key)
        self.drones = [types[k] for k in tid] if num_drones <= DICT_IO_MAX_DRONES else None
        self.types = types
        self.type_ids_caller = np.array(tid, dtype=np.uint8)      # per drone, in the caller's numbering
        self.G = 9.8                                    # BaseAviary.py:182
        self.SIM_FREQ = freq
        self.TIMESTEP = 1.0 / freq
        self.AGGR_PHY_STEPS = aggregate_phy_steps
        self.NUM_DRONES = num_drones
        self.NEIGHBOURHOOD_RADIUS = neighbourhood_radius
        self.PHYSICS = physics
        if initial_xyzs is None:
            # the reference crashes on None in this fork (BaseAviary.py:364-367)
            raise ValueError("initial_xyzs is required: (NUM_DRONES, 3)")
        self.INIT_XYZS = np.asarray(initial_xyzs, dtype=np.float64).reshape(num_drones, 3)
        self.INIT_RPYS = (np.zeros((num_drones, 3)) if initial_rpys is None
                          else np.asarray(initial_rpys, dtype=np.float64).reshape(num_drones, 3))
        self.INIT_VELS = None if initial_vels is None else np.asarray(initial_vels, np.float64).reshape(num_drones, 3)
        # the reference adds unseeded rotor noise every sub-step (BaseAviary.py:1518-1525):
        # None -> fresh seed per env, 0 -> noise off (parity runs), k -> reproducible
        self.noise_seed = int.from_bytes(os.urandom(7), "little") | 1 if noise_seed is None else int(noise_seed)
        self.dict_io = (num_drones <= DICT_IO_MAX_DRONES) if dict_io is None else dict_io

        self.ctx = Context(types, device)
        # Storage order.  Drones are independent on the path, so a heterogeneous fleet in ARBITRARY order (BASELINE
        # config 5: even index quad, odd index hexa) is stored type-major — a stable sort by type, no padding slots — and
        # every type becomes one run for the single-type kernel of its kind (dsim_step_args.runs).  Everything the caller
        # passes or reads keeps its own numbering: fleet.StorageOrder translates in the accessors.  storage="caller"
        # keeps the caller's order in HBM (the mixed-fleet kernel then partitions every tile by type).
        if storage not in ("auto", "caller"):
            raise ValueError(storage)
        self.order = None
        if len(types) > 1 and storage == "auto":
            from ..fleet import StorageOrder, type_runs as _tr
            if len(_tr(tid)) > len(types):             # not grouped by type already
                self.order = StorageOrder(tid, self.ctx.device)
                self.ctx.order = self.order
                tid = self.order.types_storage
        if layout is None:
            # plain SoA [F][n_pad] is the simplest view for small fleets; from a few hundred thousand drones on the
            # wave-tiled form [n/64][F][64] is 3-8 % faster (power-of-two field strides alias HBM channels)
            layout = "tile64" if num_drones >= 262144 else "soa"
        # placement=True (opt-in): where the observation rows of a large fleet lie relative to its state block is chosen by
        # timing the Env.step launch on a few candidates (placement.py; from ~1 M drones on).  Off by default since round 5:
        # on fresh boxes the search was worth between -10 % and +12 % of the two-call loop (BENCH_r04.json), a coin flip
        self.ctx.placement = bool(placement)
        self.state = FleetState(self.ctx, num_drones, layout)
        self._type_id = None
        if len(types) > 1:
            t = np.zeros(self.state.n_pad, dtype=np.uint8)
            t[:num_drones] = tid
            self._type_id = torch.from_numpy(t).to(self.ctx.device)
        # type-major storage: when the fleet is laid out as runs of one type starting at multiples of 256, each
        # run is stepped by the single-type kernel of its kind (fleet.type_major_order() builds such an order)
        self._runs = None
        if len(types) > 1:
            from ..fleet import type_runs
            runs = type_runs(tid)
            if len(runs) <= 64:                        # (a run may start anywhere: the launch begins at the tile that holds it)
                arr = (nat.TypeRun * len(runs))()
                for k, (f, c, ty) in enumerate(runs):
                    arr[k].first, arr[k].count, arr[k].type = f, c, ty
                self._runs = arr
        self.n_act = self.ctx.n_act
        # Physics.DYN: the model's own state beside pos / quat / vel, BaseAviary.rpy_rates (BaseAviary.py:670-671, 1785, 1828)
        self._dyn_rates = None
        if physics == Physics.DYN:
            if self.n_act != 4:
                raise NotImplementedError("Physics.DYN: both mixers of BaseAviary._dynamics read forces[0..3] "
                                          "(BaseAviary.py:1794-1803) — four-rotor types only")
            self._dyn_rates = torch.zeros((3, self.state.n_pad), dtype=torch.float32, device=self.ctx.device)
        # A fleet stored in another order than the caller's: Env.step and a bound computeControl take and return their
        # per-drone arrays (action, observation rows, command, errors) in the CALLER's numbering straight from the kernels
        # (DSIM_OPT_CALLER_IO: the run kernels gather / scatter by drone_id) — no second pass over them.  Where the run
        # kernels do not serve the fleet (more than 8 runs, drag / ground / plane options) the host translates instead.
        # (a fine-lattice launch with several sub-steps goes to the general kernels, which index by storage slot)
        self._caller_io = (self.order is not None and self._runs is not None and len(self._runs) <= 8
                           and self._phys_options == 0 and not (self.noise == "fine" and self.AGGR_PHY_STEPS > 1))
        self._action_buf = torch.zeros((self.n_act, self.state.n_pad), dtype=torch.float32, device=self.ctx.device)
        # the env's own last_clipped_action (BaseAviary.py:660-663, 545): separate from the
        # controller's cmd memory, exactly as env and controller are separate objects upstream
        self._last_action = torch.zeros_like(self._action_buf)
        self._use_last_action = True
        self._obs_buf = None      # [N, 16+n_act], allocated on the first observe()
        self._ground_trial = 0    # ground contacts counted by the placement trials of _obs_tensor (not Env.steps)
        self._graph_made = False  # a captured hipGraph holds the state block's address: it is not moved any more
        self._written_tail = None  # [n_act + 4, n_pad] behind the placed observation rows: a bound controller's outputs go there
        self._read_room = None     # room for two target blocks right behind the state block, in ITS allocation (_ensure_read_room)
        self._adjacency = None    # grid for neighbors(), built on first use
        self._action_keep = None  # keeps a zero-copy action tensor alive while the launch that reads it is queued
        self._action_ptr_last = None
        # chained fused stepping (DSIM_OPT_CHAINED): consecutive step_fused() calls skip the six
        # controller-memory fields that are functions of the stored rigid state (184 instead of 232
        # bytes per drone-step); anything else first calls materialize()
        self._chained_enabled = chained
        self._chain_live = False          # last_vel / last_rates in HBM are stale
        self._chain_ok = False            # the previous operation was a fused step (memory consistent with the state)
        self._fused_plan = None           # cached argument block of the repeated step_fused() call
        self._step_plan = None            # ... and of the repeated step(action) call of the reference-shaped loop
        self._fused_plan_dw = None        # ... of a downwash fleet (force, counter and the next grid are refreshed per call)
        # Physics.PYB_DW: neighbour downwash (BaseAviary.py:534-536, 1736-1763); `dist` = an initialised
        # torch.distributed module when the world's fleet is sharded over several ranks
        self._downwash = None
        self._fb_event = None             # a deferred WLS fallback pass is in flight on the side stream
        self._fb_stream = None
        if physics in (Physics.PYB_DW, Physics.PYB_GND_DRAG_DW):
            from ..downwash import Downwash, HaloPlan
            halo = None
            if downwash_exchange == "halo" and dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
                # spatially sharded fleet: positions travel between neighbouring slabs only
                halo = HaloPlan(self.ctx, self.state, dist, self.AGGR_PHY_STEPS * self.TIMESTEP,
                                max(t.max_coord_vel for t in self.types))
            elif downwash_exchange not in ("allgather", "halo"):
                raise ValueError(downwash_exchange)
            # downwash_keep = K: one neighbour query in K makes per-cell candidate lists that the other K - 1 re-use on refreshed
            # positions (downwash.Downwash, dsim_downwash_args.keep: exact for any motion; single-rank fleets at a density that takes the
            # banded query, otherwise ignored).  0 or 1: off.  None: the DSIM_DW_KEEP environment variable, else 32 for fleets of up to
            # 131 072 drones and off beyond: a query from lists that the whole fleet has left is a brute-force one, quadratic in the
            # fleet (7 ms at 65 536 drones), and although the lists pace their own BUILDs by what the device reports, the first few
            # queries of a fleet that breaks into motion can be of that kind — tolerable on a shard, not on a million drones.
            if downwash_keep is None:
                downwash_keep = int(os.environ.get("DSIM_DW_KEEP", "32" if self.NUM_DRONES <= (1 << 17) else "0"))
                downwash_skin = float(os.environ.get("DSIM_DW_SKIN", downwash_skin))
            self._downwash = Downwash(self.ctx, self.state, self._type_id, dist, halo=halo, split=downwash_split,
                                      keep_lists=downwash_keep if halo is None else 0, keep_skin=downwash_skin)
            if self.n_act == 6 and defer_fallback:
                # option (off by default): the WLS fallback pass of a step (normally an empty queue) runs on a side stream
                # beside the NEXT step's neighbour query instead of between the two on one stream
                # (DSIM_OPT_DEFER_FALLBACK); everything that reads the commands joins it first (_join_fallback).
                # Measured on one MI355X at BASELINE config 5's shard size: 56.6 us per step against 50.3 us with the pass
                # inline — the two cross-stream events cost more than the ~2 us launch they take off the chain.
                self._fb_stream = torch.cuda.Stream(device=self.ctx.device)
                self._fb_done, self._fb_go = torch.cuda.Event(), torch.cuda.Event()
        # the host accessors of the state block (state.pos, .mem_aos(), set_fields ...) see a CONSISTENT block: a deferred
        # fallback pass joined, and the six fields a chained sequence leaves stale written back first
        self.state.pre_access = self._before_host_access
        self.step_counter = 0
        self._env_steps = 0
        # what the ctx owns for this fleet size is allocated now, not inside the first step
        nat.check(self.ctx.lib.dsim_reserve(self.ctx.handle, self.ctx.stream_ptr(), self.state.n_pad))
        self._housekeeping()

    # ------------------------------------------------------------------ helpers
    def _soa3(self, a: np.ndarray) -> torch.Tensor:
        t = torch.zeros((3, self.state.n_pad), dtype=torch.float32)
        if self.order is not None:
            a = self.order.to_storage_np(a)
        t[:, : self.NUM_DRONES] = torch.from_numpy(np.ascontiguousarray(a.T)).float()
        return t.to(self.ctx.device)

    def _before_host_access(self) -> None:
        self._join_fallback()
        if self._chain_live:
            self.materialize()

    def _join_fallback(self) -> None:
        """Orders the current stream behind a deferred WLS fallback pass (see __init__)."""
        if self._fb_event is not None:
            torch.cuda.current_stream(self.ctx.device).wait_event(self._fb_event)
            self._fb_event = None

    def _defer_fallback(self) -> None:
        """Launches the fallback pass of the step just enqueued on the side stream."""
        main = torch.cuda.current_stream(self.ctx.device)
        self._fb_go.record(main)
        self._fb_stream.wait_event(self._fb_go)
        nat.check(self.ctx.lib.dsim_wls_fallback(self.ctx.handle, self._fb_stream.cuda_stream, self.NUM_DRONES,
                                                 self.state.view(), self._type_id.data_ptr() if self._type_id is not None else None,
                                                 None))
        self._fb_done.record(self._fb_stream)
        self._fb_event = self._fb_done

    def _dw_substepped(self) -> bool:
        """The neighbour-downwash term with several physics sub-steps per Env.step: evaluated per sub-step (see step())."""
        return self._downwash is not None and self.AGGR_PHY_STEPS > 1

    def _substep_args(self, args: nat.StepArgs, s_: int) -> None:
        """`args` (of this Env.step) for its sub-step s_ as a launch of its own: one sub-step, and the rotor-noise counter
        the multi-sub-step launch would have used there (step_index x AGGR_PHY_STEPS + s_)."""
        args.phys_substeps = 1
        args.step_index = self._env_steps * self.AGGR_PHY_STEPS + s_

    def step_args(self, dt_ctrl: Optional[float] = None, options: int = 0) -> nat.StepArgs:
        a = nat.StepArgs()
        a.phys_substeps = self.AGGR_PHY_STEPS
        a.dt_phys = self.TIMESTEP
        a.dt_ctrl = dt_ctrl if dt_ctrl is not None else self.TIMESTEP * self.AGGR_PHY_STEPS
        a.options = options | self._phys_options | self._tuning
        a.noise_seed = self.noise_seed
        a.step_index = self._env_steps
        a.noise_replay = None
        a.type_id = self._type_id.data_ptr() if self._type_id is not None else None
        a.action = None
        a.wp_table = a.wp_counter = a.wp_offset = None
        a.n_wp, a.n_steps = 0, 1
        a.ext_force = self._downwash.compute().data_ptr() if self._downwash is not None else None
        self._join_fallback()          # (behind the neighbour query, which does not read the commands)
        if self._runs is not None:
            a.runs, a.n_runs = ctypes.addressof(self._runs), len(self._runs)
        a.obs_out, a.obs_width, a.bin_next = None, 0, None
        a.drone_id = self.order.drone_id(self.state.n_pad).data_ptr() if self.order is not None else None
        a.dyn_rpy_rates = self._dyn_rates.data_ptr() if self._dyn_rates is not None else None
        return a

    # ------------------------------------------------------------------ gym surface
    def reset(self):
        """BaseAviary.reset (BaseAviary.py:406-424): housekeeping, then the initial observation."""
        self._housekeeping()
        return self._computeObs()

    def materialize(self):
        """Ends a chained sequence: last_vel / last_rates are written back into the state block."""
        self._join_fallback()
        self._fused_plan = None
        self._fused_plan_dw = None
        if self._chain_live:
            nat.check(self.ctx.lib.dsim_materialize(self.ctx.handle, self.ctx.stream_ptr(), self.NUM_DRONES,
                                                    self.state.view()))
            self._chain_live = False

    def _housekeeping(self):
        """BaseAviary._housekeeping (BaseAviary.py:640-714): zero counters, place every drone."""
        self._chain_live = self._chain_ok = False
        self._fused_plan = None
        self._fused_plan_dw = None
        self._step_plan = None
        self.step_counter = 0
        self._env_steps = 0
        if getattr(self, "_downwash", None) is not None:
            self._downwash.invalidate_prebin()
        pos, rpy = self._soa3(self.INIT_XYZS), self._soa3(self.INIT_RPYS)
        vel = self._soa3(self.INIT_VELS) if self.INIT_VELS is not None else None
        self._last_action.zero_()                      # BaseAviary.py:660-663
        if getattr(self, "_dyn_rates", None) is not None:
            self._dyn_rates.zero_()                    # BaseAviary.py:670-671
        self._use_last_action = True
        nat.check(self.ctx.lib.dsim_reset(
            self.ctx.handle, self.ctx.stream_ptr(), self.NUM_DRONES, self.state.view(), pos.data_ptr(),
            rpy.data_ptr(), vel.data_ptr() if vel is not None else None, None,
            self._type_id.data_ptr() if self._type_id is not None else None))
        torch.cuda.current_stream(self.ctx.device).synchronize()   # host buffers above go out of scope

    def step(self, action):
        """BaseAviary.step (BaseAviary.py:428-555): Physics.PYB and its add-on modes, or Physics.DYN."""
        self.materialize()
        self._chain_ok = False
        # The reference-shaped loop hands step() the tensor computeControl returned, every iteration the same object over
        # the same buffers: the prepared argument block is launched again with the step counter moved on (at the
        # reference's own fleet sizes the Python in front of the launch is most of what an iteration costs).
        plan = self._step_plan
        if plan is not None and self._downwash is None and plan.matches(self._step_key(), action, self._is_action_rows):
            # (the same tensor as last time — the command a bound controller returned — or, for a fleet that takes [N, 4] rows
            # as the caller holds them, any such tensor: a policy's fresh output every step)
            self._join_fallback()
            plan.args.step_index = self._env_steps
            if action is not plan.action:
                plan.args.action = plan.action_ptr = action.data_ptr()
                plan.action = self._action_keep = action
            nat.check(self.ctx.lib.dsim_physics(self.ctx.handle, self.ctx.stream_ptr(), self.NUM_DRONES, plan.state_view,
                                                plan.echo_ptr, plan.args_ref))
            self._use_last_action = True
            self.step_counter += self.AGGR_PHY_STEPS
            self._env_steps += 1
            return plan.out, self._computeReward(), self._computeDone(), plan.info
        # The neighbour-downwash term is evaluated per PHYSICS SUB-STEP, as the reference loops it (BaseAviary.py:510-536:
        # with AGGR_PHY_STEPS > 1 the positions are refreshed and _downwash applied inside the sub-step loop): one
        # [query -> one-sub-step physics] pair of launches per sub-step, the observation rows from the last one.  Without the
        # term (or with one sub-step) the whole Env.step is ONE launch.
        passes = self.AGGR_PHY_STEPS if self._dw_substepped() else 1
        obs = self._obs_tensor()
        for s_ in range(passes):
            args = self.step_args()
            if passes > 1:
                self._substep_args(args, s_)
            if self._caller_io:
                args.options |= nat.OPT_CALLER_IO
            if s_ == 0:
                # a homogeneous quad fleet on the fast kernel takes an [N, 4] device tensor as the caller holds it (no transpose)
                self._rows_in = (self.n_act == 4 and self._type_id is None and self._downwash is None and self._phys_options == 0
                                 and self.order is None and self.state.n_pad % 256 == 0 and self._is_action_rows(action)
                                 and not (getattr(self, "_cmd_token", None) is not None and action is self._cmd_token[0]))
            if self._rows_in:
                args.options |= nat.OPT_ACTION_ROWS
                self._action_keep = action
            args.action = (action.data_ptr() if self._rows_in else self._action_ptr(action, self._caller_io)) if s_ == 0 else self._action_ptr_last
            self._action_ptr_last = args.action
            if self._downwash is not None:
                # the physics launch fills the next neighbour grid from the new positions (the library keeps its own
                # record of what was binned when, and bins afresh when anything moved the drones in between)
                args.bin_next = self._downwash.bin_next_ptr()
            if s_ == passes - 1:
                # the observation rows are written by the physics launch itself (BaseAviary.py:547-555 returns them from step)
                args.obs_out, args.obs_width = obs.data_ptr(), 16 + self.n_act
            nat.check(self.ctx.lib.dsim_physics(self.ctx.handle, self.ctx.stream_ptr(), self.NUM_DRONES,
                                                self.state.view(), self._last_action.data_ptr(), ctypes.byref(args)))
        self._use_last_action = True
        self.step_counter += self.AGGR_PHY_STEPS
        self._env_steps += 1
        out = self._computeObs(obs if self._caller_io else self._rows_to_caller(obs))
        self._step_plan = None
        if (passes == 1 and self._downwash is None and out is obs and torch.is_tensor(action) and action.is_cuda
                and args.action in (action.data_ptr(), getattr(action, "T", action).data_ptr())):
            # (only when the launch read the caller's tensor itself — the controller's command array or [N, 4] rows — and
            # handed out the rows it wrote: nothing was copied on the way in or out)
            self._step_plan = _StepPlan(key=self._step_key(), action=action, action_ptr=action.data_ptr(), rows_in=bool(self._rows_in),
                                        args=args, state_view=self.state.view(), args_ref=ctypes.byref(args),
                                        echo_ptr=self._last_action.data_ptr(), out=out, info=self._computeInfo())
        return out, self._computeReward(), self._computeDone(), self._computeInfo()

    def _step_key(self) -> tuple:
        """What a prepared Env.step launch depends on besides the action (see _StepPlan)."""
        return (self.state.data.data_ptr(), self._obs_buf.data_ptr() if self._obs_buf is not None else 0, self._last_action.data_ptr(),
                self._phys_options, self._tuning, self.AGGR_PHY_STEPS, id(self._runs), self._caller_io, self.noise_seed)

    def _is_action_rows(self, action) -> bool:
        """An [N, 4] float32 device tensor, contiguous and 16-byte aligned: what DSIM_OPT_ACTION_ROWS takes as it is."""
        return (torch.is_tensor(action) and action.dtype == torch.float32 and action.is_contiguous()
                and action.device == self.ctx.device and tuple(action.shape) == (self.NUM_DRONES, 4)
                and action.data_ptr() % 16 == 0)

    def step_fused(self, targets, control_timestep: Optional[float] = None, action=None, n_steps: int = 1):
        """One launch = ``env.step(action)`` followed by ``computeControl`` for every drone:
        the body of the reference's example loop (examples/fly_INDI.py:223-239).  ``action``
        None = the controller's last command (every iteration after the first); the first
        iteration of the example passes its initial action 0.4 (fly_INDI.py:214).
        ``targets``: :class:`Targets` (per drone or broadcast) or :class:`WaypointTargets`.
        ``n_steps`` > 1 runs that many loop iterations inside the one launch."""
        # hot loop: the same targets object, no explicit action, no downwash -> reuse the prepared
        # argument block (the Python side of a launch drops from ~9 us to ~3 us, which is what bounds
        # small fleets)
        if n_steps > 1 and self._downwash is not None:
            raise ValueError("n_steps > 1 with the neighbour-downwash term: the force (and the position exchange behind "
                             "it) is evaluated once per Env.step")
        if not getattr(targets, "_placed", True):
            self._place_targets(targets, control_timestep)
        # (the plan holds the targets object itself: compared by identity while it is alive, so a new object that
        # happens to reuse the address of a dropped one can never match; and the device pointers it was built from
        # are re-checked, so a Targets whose tensor was swapped does not either)
        key = (control_timestep, n_steps, self._chained_enabled)
        plan = self._fused_plan
        if (action is None and plan is not None and self._downwash is None and self._chain_ok
                and (self._chain_live or not self._chained_enabled) and plan.matches(key, targets, self._targets_ptrs(targets))):
            plan.args.step_index = self._env_steps
            nat.check(self.ctx.lib.dsim_step(self.ctx.handle, self.ctx.stream_ptr(), self.NUM_DRONES, plan.state_view,
                                             plan.targets_view, plan.args_ref))
            self.step_counter += self.AGGR_PHY_STEPS * n_steps
            self._env_steps += n_steps
            return
        pd = self._fused_plan_dw
        if (action is None and pd is not None and self._downwash is not None and not self._dw_substepped()
                and self._chain_ok and self._fb_stream is None and pd.matches(key, targets, self._targets_ptrs(targets))):
            # the same call again on a downwash fleet: the prepared argument block, with this step's force, counter and
            # the grid the step kernel may fill (the Python side of a config-5 step is what paces a 65 536-drone shard)
            self._downwash.compute()
            pd.args.step_index = self._env_steps
            pd.args.bin_next = self._downwash.bin_next_ptr()
            nat.check(self.ctx.lib.dsim_step(self.ctx.handle, self.ctx.stream_ptr(), self.NUM_DRONES, pd.state_view,
                                             pd.targets_view, pd.args_ref))
            self.step_counter += self.AGGR_PHY_STEPS
            self._env_steps += 1
            return
        wp = isinstance(targets, WaypointTargets)
        if self._dw_substepped():
            # the downwash term per physics sub-step (see step()): all sub-steps but the last as Env.step-only launches
            # behind their own neighbour query, the last one fused with the control law
            self._use_last_action = False
            self.materialize()
            act_ptr = None
            for s_ in range(self.AGGR_PHY_STEPS - 1):
                pa = self.step_args(control_timestep)
                self._substep_args(pa, s_)
                if action is not None:
                    act_ptr = self._action_ptr(action) if s_ == 0 else act_ptr
                    pa.action = act_ptr
                pa.bin_next = self._downwash.bin_next_ptr()
                nat.check(self.ctx.lib.dsim_physics(self.ctx.handle, self.ctx.stream_ptr(), self.NUM_DRONES,
                                                    self.state.view(), None, ctypes.byref(pa)))
        args = self.step_args(control_timestep, nat.OPT_BCAST_TGT if targets.broadcast else 0)
        if self._dw_substepped():
            self._substep_args(args, self.AGGR_PHY_STEPS - 1)
        args.n_steps = n_steps
        if wp:
            targets.fill(args)
            tview = nat.View()
        else:
            tview = targets.view()
        if action is not None:
            args.action = self._action_ptr(action)
        defer = False
        if self._downwash is not None:
            args.bin_next = self._downwash.bin_next_ptr()      # the step kernel fills the next step's neighbour grid
            if self._fb_stream is not None:
                args.options |= nat.OPT_DEFER_FALLBACK
                defer = True
        self._use_last_action = False   # from here on the applied action IS the controller cmd
        chain = (self._chained_enabled and self._chain_ok and action is None and self._type_id is None
                 and self.n_act == 4 and args.ext_force is None and self._phys_options == 0)
        if chain:
            args.options |= nat.OPT_CHAINED
            self._chain_live = True
        else:
            self.materialize()
        sview = self.state.view()
        nat.check(self.ctx.lib.dsim_step(self.ctx.handle, self.ctx.stream_ptr(), self.NUM_DRONES,
                                         sview, tview, ctypes.byref(args)))
        if defer:
            self._defer_fallback()
        self._chain_ok = True
        self.step_counter += self.AGGR_PHY_STEPS * n_steps
        self._env_steps += n_steps
        # what the NEXT identical call would pass (a chained env switches to the chained form after this call)
        self._fused_plan = None
        self._fused_plan_dw = None
        if action is None and self._downwash is not None and not chain and not defer and not self._dw_substepped():
            self._fused_plan_dw = _FusedPlan(key, args, sview, tview, ctypes.byref(args), targets, self._targets_ptrs(targets))
        if action is None and self._downwash is None:
            nxt = nat.StepArgs.from_buffer_copy(args)
            if self._chained_enabled and chain:
                nxt.options |= nat.OPT_CHAINED
            if not self._chained_enabled or chain:
                self._fused_plan = _FusedPlan(key, nxt, sview, tview, ctypes.byref(nxt), targets, self._targets_ptrs(targets))

    def capture_fused(self, targets, steps: int, control_timestep: Optional[float] = None):
        """Captures ``steps`` consecutive :meth:`step_fused` launches into ONE hipGraph and returns a
        ``FusedGraph`` whose ``replay()`` runs them with a single host call.  For small fleets the loop is
        launch-latency bound (a few us per launch from Python; 10-16 us per graph replay whatever its
        length), which is what a graph removes.  Kernel arguments are frozen at capture time, so the
        env-step counter that seeds the rotor noise is read from device memory (``step_index_dev``: captured step
        i uses counter + i) and advanced by ``steps`` by a one-thread node at the end of the graph.  Any fleet
        composition, plain or waypoint targets.  With the neighbour-downwash term (single rank): every captured step is
        query -> step (which fills the next query's grid) -> fallback, and the grid's box is the one measured at capture
        time — drones that leave it are clamped to its border cells, which costs search efficiency,
        never exactness (``Downwash._grid_box``); capture again after the fleet has moved far."""
        dw = self._downwash
        if dw is not None:
            world = dw.dist.get_world_size() if (dw.dist is not None and dw.dist.is_initialized()) else 1
            if dw.halo is not None or world > 1:
                raise NotImplementedError("graph capture with the neighbour-downwash term: single-rank fleets only (the "
                                          "position exchange of a sharded fleet re-sizes its messages on the host)")
            if self._fb_stream is not None:
                raise ValueError("graph capture with the neighbour-downwash term: not with the deferred fallback pass")
            if self._dw_substepped():
                raise NotImplementedError("graph capture with the neighbour-downwash term and several physics sub-steps per "
                                          "Env.step (the term is evaluated per sub-step: step_fused() launches them one by one)")
            # kept candidate lists pace their BUILDs by what the device reports while it runs; a captured sequence is fixed:
            # the queries of a graph are made from scratch
            if dw.keep_lists > 1:
                dw.keep_lists = 0
                dw.invalidate_prebin()
        # nothing may allocate under capture: the fallback queue of hexa fleets is reserved up front
        nat.check(self.ctx.lib.dsim_reserve(self.ctx.handle, self.ctx.stream_ptr(), self.state.n_pad))
        self._graph_made = True
        from .fused_graph import FusedGraph
        return FusedGraph(self, targets, steps, control_timestep)

    @property
    def rpy_rates(self) -> torch.Tensor:
        """Physics.DYN: BaseAviary.rpy_rates (BaseAviary.py:670-671, 1828) as [3, N] in the caller's numbering."""
        if self._dyn_rates is None:
            raise AttributeError("rpy_rates exists with Physics.DYN only (BaseAviary.py:670-671)")
        r = self._dyn_rates[:, : self.NUM_DRONES]
        return r if self.order is None else self.order.to_caller(r, 1)

    def ground_contacts(self) -> int:
        """Drone x Env.steps so far that ended with the vehicle's collision cylinder at or below z = 0 (cumulative over
        this env's context; synchronises the stream).  The reference's PyBullet world has a ground plane with
        collisions on (BaseAviary.py:680).  With ``ground_plane=False`` (the default for large fleets: the flight
        kernels) contact is not modelled, and a non-zero count means part of the flight lies outside the domain in
        which trajectories are comparable with the reference; with ``ground_plane=True`` (the default for
        reference-sized fleets) the product-defined contact model of DSIM_OPT_PLANE acts there instead."""
        return self.ctx.query(nat.QUERY_GROUND_CONTACTS) - self._ground_trial

    def close(self):
        self.ctx.close()

    def getPyBulletClient(self):
        return -1   # there is no PyBullet client; kept so example scripts keep running

    # ------------------------------------------------------------------ pieces
    @staticmethod
    def _targets_ptrs(targets):
        if isinstance(targets, WaypointTargets):
            return (targets.table.data_ptr(), targets.counters.data_ptr(),
                    targets.offsets.data_ptr() if targets.offsets is not None else 0, targets.n_wp)
        return (targets.data.data_ptr(), tuple(targets.data.shape))

    def _action_ptr(self, action, caller_order: bool = False) -> int:
        """Device pointer of the action as SoA [n_act][n_pad].  A tensor that already IS such an array (the command
        a bound ``INDIControl`` returns is the transposed view of one) is passed through without a copy.
        ``caller_order``: the launch indexes the action by the caller's drone number itself (DSIM_OPT_CALLER_IO)."""
        tok = getattr(self, "_cmd_token", None)
        if not caller_order and tok is not None and action is tok[0] and action._version == tok[1]:
            return tok[2].data_ptr()     # the command a bound controller just returned: its storage-order array, no copy
        if (self.order is None or caller_order) and torch.is_tensor(action) and action.is_cuda and action.dtype == torch.float32:
            base = action.T if (action.ndim == 2 and action.shape[1] == self.n_act and action.shape[0] != self.n_act) else action
            if (base.ndim == 2 and base.shape[0] == self.n_act and base.stride() == (self.state.n_pad, 1)
                    and base.shape[1] <= self.state.n_pad and base.shape[1] >= self.NUM_DRONES):
                self._action_keep = action
                return base.data_ptr()
        self._load_action(action, caller_order)
        return self._action_buf.data_ptr()

    def _move_state(self, new_block: torch.Tensor) -> None:
        """The state block into another allocation (same contents).  Everything that holds its address is dropped: the
        prepared argument blocks of the fused step, the downwash and halo plans' cached views."""
        assert not self._chain_live and not self._graph_made, "the state block is pinned (chained sequence / captured graph)"
        new_block.copy_(self.state.data)
        self.state.data = new_block
        self._fused_plan = self._fused_plan_dw = self._step_plan = None
        dw = self._downwash
        if dw is not None:
            dw._single = dw._halo_args = None
            if dw.halo is not None:
                dw.halo._pack_call = None

    def _load_action(self, action, caller_order: bool = False) -> None:
        n = self.NUM_DRONES
        if isinstance(action, dict):                      # CtrlAviary.py:258-263 format
            a = np.zeros((self.n_act, n), dtype=np.float32)
            for k, v in action.items():
                v = np.asarray(v, dtype=np.float32)
                a[: v.shape[0], int(k)] = v
            t = torch.from_numpy(a).to(self.ctx.device)
        else:
            t = torch.as_tensor(action, dtype=torch.float32, device=self.ctx.device)
            if t.shape == (n, self.n_act):
                t = t.T
        self._action_buf[:, :n] = t if (self.order is None or caller_order) else self.order.to_storage(t, 1)

    @staticmethod
    def _computeReward():
        return -1                                   # CtrlAviary.py:267-279

    @staticmethod
    def _computeDone():
        return False                                # CtrlAviary.py:283-295

    @staticmethod
    def _computeInfo():
        return {"answer": 42}                       # CtrlAviary.py:299-310
