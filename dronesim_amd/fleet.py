"""Fleet state in HBM + the context that owns the per-type constant table.

Host-side plumbing only (PyTorch-ROCm for device memory and streams); all
arithmetic happens in the HIP kernels behind the C-ABI.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Sequence

import numpy as np
import torch

from . import _native as nat
from .params import DroneType, types_to_c_array

# field indices (include/dronesim_amd.h DSIM_F_*)
F_POS, F_QUAT, F_VEL, F_ANGVEL = 0, 3, 7, 10
F_LAST_VEL, F_LAST_RATES, F_LAST_THRUST, F_CMD = 13, 16, 19, 20


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)


READ_ROOM_MIN_DRONES = 1 << 20          # fleets from this size on keep room for one block of targets behind their state block (FleetState)


def pad_to(n: int, m: int = 256) -> int:
    """The C-ABI needs n_pad % 64 == 0; whole 256-drone tiles keep the entire fleet on the
    fused kernel's fast path (a ragged tail costs one extra small launch)."""
    return (n + m - 1) // m * m


class Context:
    """One per device: uploads the type table (dsim_create)."""

    def __init__(self, types: Sequence[DroneType], device: int = 0):
        self.lib = nat.load()
        if not torch.cuda.is_available():
            raise nat.DsimError("dronesim_amd needs a HIP device (MI355X); there is no CPU fallback")
        self.types = list(types)
        self.device = torch.device("cuda", device)
        self._h = ctypes.c_void_p()
        self._c_types = types_to_c_array(self.types)
        nat.check(self.lib.dsim_create(ctypes.byref(self._h), device, self._c_types, len(self.types)))
        self.n_act = max(t.n_act for t in self.types)
        self.n_fields = nat.NF_QUAD if self.n_act <= 4 else nat.NF_HEXA
        # StorageOrder of the fleet this context serves, or None: set by an env that stores an interleaved heterogeneous
        # fleet type-major; every per-drone block built on this context for that fleet (state, targets, waypoint
        # counters) then translates between the caller's numbering and the storage slots
        self.order = None
        # Placement (placement.py): where the observation rows of a large fleet lie relative to the state block is chosen
        # by timing the real launch.  `placement` False: allocated plainly.  `placement_log`: one dict per search made.
        self.placement = False                 # opt-in (CtrlAviary(placement=True) / a stand-alone controller sets it)
        self.placement_walk_bytes = None       # the transient budget of a search; None = placement.WALK_BYTES (4 GiB)
        self.placement_log = []
        self.query_offsets = {}                # counts that trial passes (placement.py) added to the device counters: subtracted by query()
        self.read_room = None                  # (flat tensor, n_pad, layout): room for one block of targets behind a large fleet's state block (FleetState)

    @property
    def handle(self):
        return self._h

    def stream_ptr(self) -> int:
        """The current stream of this device as a raw hipStream_t.  (torch.cuda.current_stream().cuda_stream builds a
        Stream object per call: 1.9 us, a quarter of the Python side of a launch; the raw accessor — the one PyTorch's own
        generated code uses — answers in 0.06 us.)"""
        if _RAW_STREAM is not None:
            return _RAW_STREAM(self.device.index)
        return torch.cuda.current_stream(self.device).cuda_stream

    def query(self, what: int) -> int:
        """dsim_query (synchronises the stream): nat.QUERY_WLS_FALLBACKS = drones that took the full active-set
        WLS loop so far, nat.QUERY_WLS_FAILURES = those where it did not converge (the reference would raise)."""
        v = ctypes.c_int64(0)
        nat.check(self.lib.dsim_query(self._h, self.stream_ptr(), int(what), ctypes.byref(v)))
        return int(v.value) - self.query_offsets.get(int(what), 0)     # (what placement trials counted is not the fleet's history)

    def close(self):
        self.read_room = None                  # (a view of a state allocation: dropped with the context, so that the block can go)
        if self._h:
            self.lib.dsim_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def type_runs(type_ids) -> list:
    """[(first, count, type), ...]: maximal runs of equal consecutive type ids."""
    t = np.asarray(type_ids).astype(np.int64).ravel()
    if t.size == 0:
        return []
    cut = np.flatnonzero(np.diff(t)) + 1
    first = np.concatenate([[0], cut])
    count = np.diff(np.concatenate([first, [t.size]]))
    return [(int(f), int(c), int(t[f])) for f, c in zip(first, count)]


class StorageOrder:
    """Drones are independent on the path, so the order in which a fleet is STORED is free.  A
    heterogeneous fleet in arbitrary order (BASELINE config 5: even index quad, odd index hexa) is stored type-major —
    a stable sort by type id, a pure permutation of [0, n) with no padding slots — so that every type is one run that
    the single-type kernel of its kind steps (dsim_step_args.runs: 0.78 of HBM instead of the mixed-fleet kernel's 0.64),
    while everything the caller sees (initial positions, targets, actions, observations, state accessors) keeps the
    caller's numbering.  slot[d] = storage slot of drone d; drone[s] = drone stored in slot s."""

    def __init__(self, type_ids, device):
        t = np.asarray(type_ids).astype(np.int64).ravel()
        self.n = int(t.size)
        self.drone_np = np.argsort(t, kind="stable")
        self.slot_np = np.empty_like(self.drone_np)
        self.slot_np[self.drone_np] = np.arange(self.n)
        self.types_storage = t[self.drone_np].astype(np.uint8)
        self.slot = torch.from_numpy(self.slot_np).to(device)            # int64: index_select indices
        self.drone = torch.from_numpy(self.drone_np).to(device)
        self._drone_id = {}

    def drone_id(self, n_pad: int) -> torch.Tensor:
        """int32 [n_pad] for dsim_step_args.drone_id (padding slots map to themselves)."""
        if n_pad not in self._drone_id:
            d = np.arange(n_pad, dtype=np.int32)
            d[: self.n] = self.drone_np
            self._drone_id[n_pad] = torch.from_numpy(d).to(self.slot.device)
        return self._drone_id[n_pad]

    def to_storage(self, values: torch.Tensor, dim: int = -1) -> torch.Tensor:
        """values indexed by drone along `dim` -> indexed by slot."""
        return values.index_select(dim, self.drone.to(values.device))

    def to_caller(self, values: torch.Tensor, dim: int = -1) -> torch.Tensor:
        return values.index_select(dim, self.slot.to(values.device))

    def to_storage_np(self, a: np.ndarray, axis: int = 0) -> np.ndarray:
        return np.take(a, self.drone_np, axis=axis)


class Frozen:
    """A per-drone device tensor the caller promises not to write while it keeps passing it (``frozen(t)``):
    ``Targets.set`` copies it into the target block the first time and skips the fleet-sized copy on every later
    call with the SAME object (the reference-shaped loop hands computeControl the same target_pos every iteration,
    examples/fly_INDI.py:229-239).  A plain tensor is always copied: torch's write counter does not see what this
    library's own kernels write through raw pointers (the state block, observation rows, a sampled trajectory), so an
    unchanged counter proves nothing."""

    def __init__(self, tensor: torch.Tensor):
        self.tensor = tensor


def frozen(tensor: torch.Tensor) -> Frozen:
    return Frozen(tensor)


def type_major_order(type_ids, align: int = 256):
    """Storage order for a heterogeneous fleet: drones grouped by type, each group starting at a multiple of
    `align` (the gaps are padding slots).  Returns (slot_of_drone [n] int64, n_slots, slot_types [n_slots] uint8):
    drone i of the caller's numbering lives in storage slot slot_of_drone[i]; padding slots carry the type of
    the group they follow.  A fleet built in this order (CtrlAviary(type_ids=slot_types, ...)) is stepped by one
    single-type kernel per group instead of the mixed-fleet kernel."""
    t = np.asarray(type_ids).astype(np.int64).ravel()
    slot = np.zeros(t.size, dtype=np.int64)
    slot_types, base = [], 0
    for ty in np.unique(t):
        idx = np.flatnonzero(t == ty)
        slot[idx] = base + np.arange(idx.size)
        size = -(-idx.size // align) * align
        slot_types.append(np.full(size, ty, dtype=np.uint8))
        base += size
    return slot, base, np.concatenate(slot_types) if slot_types else np.zeros(0, np.uint8)


class BlockedSoA:
    """fp32 device array addressed as the C-ABI's blocked SoA.

    layout "soa":    tensor [F, n_pad]            (block = n_pad)
    layout "tileB":  tensor [n_pad/B, F, B]       (B = 64, 256, 1024, 4096: B-drone blocks, each field a
                                                   contiguous 4B-byte row; n_pad is rounded up to a multiple of B)
    """

    def __init__(self, n: int, n_fields: int, device, layout: str = "soa", pad: int = 256, order=None, tail_fields: int = 0,
                 storage: Optional[torch.Tensor] = None):
        """tail_fields: that many more fields' worth of floats are allocated behind the block, in the SAME allocation
        (`self.tail`, flat).  storage: a flat zeroed float32 tensor of exactly the block's size to live in instead of a fresh
        allocation (a tail of another block)."""
        self.n, self.n_fields, self.layout = n, n_fields, layout
        self.version = 0            # bumped by every host-side write (set_fields): caches keyed on the contents check it
        self.order = order if (order is not None and order.n == n) else None    # StorageOrder: caller numbering <-> slots
        self.pre_access = None      # optional callable run before the block is read or written from the host side
        self.tail = None
        if layout == "soa":
            self.block = 0
            self.n_pad = pad_to(n, pad)
            shape = (n_fields, self.n_pad)
        elif layout.startswith("tile") and layout[4:].isdigit() and int(layout[4:]) in (64, 256, 1024, 4096):
            self.block = B = int(layout[4:])
            self.n_pad = pad_to(n, max(pad, B))
            shape = (self.n_pad // B, n_fields, B)
        else:
            raise ValueError(layout)
        numel = n_fields * self.n_pad
        if storage is not None and storage.numel() == numel and storage.is_contiguous() and storage.device == torch.device(device):
            self.data = storage.view(shape)
        elif tail_fields > 0:
            flat = torch.zeros(numel + tail_fields * self.n_pad, dtype=torch.float32, device=device)
            self.data, self.tail = flat[:numel].view(shape), flat[numel:]
        else:
            self.data = torch.zeros(shape, dtype=torch.float32, device=device)

    def view(self) -> nat.View:
        v = nat.View()
        v.base = self.data.data_ptr()
        v.n_pad = self.n_pad
        v.n_fields = self.n_fields
        if self.layout == "soa":
            v.block, v.field_stride, v.block_stride = self.n_pad, self.n_pad, self.n_pad * self.n_fields
        else:
            v.block, v.field_stride, v.block_stride = self.block, self.block, self.block * self.n_fields
        return v

    def raw_fields(self, f0: int, nf: int) -> torch.Tensor:
        """[nf, n] tensor of fields f0..f0+nf in STORAGE order (a view for "soa", a gather for the tiled layouts)."""
        if self.pre_access is not None:
            self.pre_access()
        if self.layout == "soa":
            return self.data[f0:f0 + nf, : self.n]
        return self.data[:, f0:f0 + nf, :].permute(1, 0, 2).reshape(nf, self.n_pad)[:, : self.n]

    def fields(self, f0: int, nf: int) -> torch.Tensor:
        """[nf, n] tensor of fields f0..f0+nf in the caller's numbering (with a storage order: a gathered copy)."""
        raw = self.raw_fields(f0, nf)
        return raw if self.order is None else self.order.to_caller(raw, 1)

    def set_fields(self, f0: int, values: torch.Tensor) -> None:
        """values: [nf, n] in the caller's numbering"""
        if self.pre_access is not None:
            self.pre_access()
        self.version += 1
        nf = values.shape[0]
        vals = values.to(self.data.device, torch.float32)
        if self.order is not None:
            vals = self.order.to_storage(vals, 1)
        if self.layout == "soa":
            self.data[f0:f0 + nf, : self.n] = vals
        elif self.n == self.n_pad:          # whole tiles: one strided copy, no staging buffer
            self.data[:, f0:f0 + nf, :] = vals.reshape(nf, self.n_pad // self.block, self.block).permute(1, 0, 2)
        else:
            full = torch.zeros((nf, self.n_pad), dtype=torch.float32, device=self.data.device)
            full[:, : self.n] = vals
            self.data[:, f0:f0 + nf, :] = full.reshape(nf, self.n_pad // self.block, self.block).permute(1, 0, 2)


class FleetState(BlockedSoA):
    """The 13 rigid-body floats Bullet holds per drone (BaseAviary.py:718-732) plus the
    controller memory of one INDIControl instance per drone (INDIControl.py:109-146)."""

    def __init__(self, ctx: Context, n: int, layout: str = "soa", pad: int = 256):
        # A large fleet's state block is allocated with room for ONE block of targets right behind it, in the same allocation:
        # an array a launch READS beside the state wants the state's own memory window (the fused step runs
        # 153-154 us with its targets there and 158 or 164 us with them in whatever block the allocator hands out — three
        # levels, by process), and the same allocation is the same window.  No search, no timing: the first per-drone Targets
        # of this layout and size made on the context lives there (40 B per drone, held with the state).
        big = pad_to(n, max(pad, 64)) >= READ_ROOM_MIN_DRONES and os.environ.get("DSIM_NO_READ_ROOM", "0") == "0"
        super().__init__(n, ctx.n_fields, ctx.device, layout, pad, order=ctx.order, tail_fields=nat.NT if big else 0)
        self.ctx = ctx
        if self.tail is not None:
            ctx.read_room = (self.tail, self.n_pad, layout)

    pos = property(lambda s: s.fields(F_POS, 3))
    quat = property(lambda s: s.fields(F_QUAT, 4))
    vel = property(lambda s: s.fields(F_VEL, 3))
    ang_vel = property(lambda s: s.fields(F_ANGVEL, 3))
    last_vel = property(lambda s: s.fields(F_LAST_VEL, 3))
    last_rates = property(lambda s: s.fields(F_LAST_RATES, 3))
    last_thrust = property(lambda s: s.fields(F_LAST_THRUST, 1))
    cmd = property(lambda s: s.fields(F_CMD, s.n_fields - F_CMD))

    def rigid_aos(self) -> np.ndarray:
        """[n,13] fp64 host copy (pos3 quat4 vel3 angvel3) — the oracle's layout."""
        return self.fields(0, 13).T.double().cpu().numpy().copy()

    def mem_aos(self) -> np.ndarray:
        """[n,13] fp64 host copy (last_vel3 last_rates3 last_thrust cmd6)."""
        m = np.zeros((self.n, 13))
        m[:, : self.n_fields - 13] = self.fields(13, self.n_fields - 13).T.double().cpu().numpy()
        return m

    def load_aos(self, rigid: np.ndarray, mem: np.ndarray) -> None:
        self.set_fields(0, torch.from_numpy(np.ascontiguousarray(rigid.T)))
        self.set_fields(13, torch.from_numpy(np.ascontiguousarray(mem[:, : self.n_fields - 13].T)))


class Targets(BlockedSoA):
    """Per-step targets: pos3 vel3 acc3 yaw (INDIControl.computeControl arguments)."""

    def __init__(self, ctx: Context, n: int, layout: str = "soa", broadcast: bool = False, pad: int = 256):
        self.broadcast = broadcast
        if broadcast:
            self.n, self.n_pad, self.n_fields, self.layout = 1, 64, nat.NT, "soa"
            self.data = torch.zeros((nat.NT, 1), dtype=torch.float32, device=ctx.device)
            self.order, self.pre_access, self.version = None, None, 0
        else:
            room, storage = getattr(ctx, "read_room", None), None
            if room is not None and room[2] == layout and room[1] == pad_to(n, max(pad, int(layout[4:]) if layout != "soa" else pad)):
                storage, ctx.read_room = room[0], None      # the room behind the state block (FleetState): taken once
            super().__init__(n, nat.NT, ctx.device, layout, pad, order=ctx.order, storage=storage)
            self._room_ptr = storage.data_ptr() if storage is not None else None
            self._placed = False      # CtrlAviary.step_fused may re-allocate `data` once, by trial (placement.py)

    @property
    def behind_the_state(self) -> bool:
        """Whether the targets lie in the room behind the fleet's state block (FleetState) — asked of the tensor itself, so that a
        Targets whose data was re-allocated since (placement by trial) says no."""
        return getattr(self, "_room_ptr", None) is not None and self.data.data_ptr() == self._room_ptr

    def view(self) -> nat.View:
        if not self.broadcast:
            return super().view()
        v = nat.View()
        v.base = self.data.data_ptr()
        v.n_pad, v.block, v.field_stride, v.block_stride, v.n_fields = 64, 64, 1, nat.NT, nat.NT
        return v

    def set(self, pos=None, vel=None, acc=None, yaw=None) -> None:
        """Each argument [3, n] / [n] (per drone) or length-3 / scalar (same for all)."""
        dev = self.data.device
        if not hasattr(self, "_const"):
            self._const = {}                  # field group -> the constant it was last filled with (None: per-drone data)
        for f0, val, nf in ((0, pos, 3), (3, vel, 3), (6, acc, 3), (9, yaw, 1)):
            if val is None:
                continue
            # the same broadcast constant as last time (e.g. the zero target_vel / target_acc of every
            # computeControl call): the fields already hold it, skip the fleet-sized fill
            key = None
            if isinstance(val, Frozen):
                # the caller's promise (see Frozen): the same OBJECT as last time means the fields already hold it — a
                # fleet-sized copy per computeControl call is 7 % of the reference-shaped loop.  The object is held, so
                # the identity cannot be recycled.
                old = self._const.get(f0)
                if isinstance(old, tuple) and len(old) == 2 and old[0] == "frozen" and old[1] is val:
                    continue
                key = ("frozen", val)
                val = val.tensor
                if val.ndim == 2 and val.shape[0] != nf and val.shape[1] == nf:
                    val = val.T
            elif not torch.is_tensor(val) and np.size(val) == nf:
                key = tuple(float(x) for x in np.asarray(val, dtype=np.float32).ravel())
                if self._const.get(f0) == key:
                    continue
            t = torch.as_tensor(val, dtype=torch.float32, device=dev).reshape(nf, -1)
            if self.broadcast:
                self.data[f0:f0 + nf, :] = t
            else:
                if t.shape[1] == 1:
                    t = t.expand(nf, self.n)
                BlockedSoA.set_fields(self, f0, t)
            self._const[f0] = key

    def set_fields(self, f0: int, values: torch.Tensor) -> None:
        self._const = {}                      # written behind set()'s back: nothing is known to be constant any more
        super().set_fields(f0, values)


class WaypointTargets:
    """Targets read from ONE waypoint table shared by the whole fleet, as
    examples/fly_INDI_TrajectoryTrack.py does per drone (:178-189 tables, :242-245 lookup,
    :253-256 counter advance with wrap).  The table lives in device memory once
    (1200 x 10 floats = 48 KB for the example); each drone keeps an int32 row counter and an
    optional position offset (so a fleet can fly the same figure side by side)."""

    def __init__(self, ctx: Context, n: int, target_pos, target_vel=None, target_acc=None, target_yaw=None,
                 wp_counters=None, offsets=None, pad: int = 256):
        dev = ctx.device
        pos = np.asarray(target_pos, dtype=np.float32)
        n_wp = pos.shape[0]
        tab = np.zeros((n_wp, 10), dtype=np.float32)
        tab[:, 0:3] = pos
        if target_vel is not None:
            tab[:, 3:6] = np.asarray(target_vel, dtype=np.float32)
        if target_acc is not None:
            tab[:, 6:9] = np.asarray(target_acc, dtype=np.float32)
        if target_yaw is not None:
            tab[:, 9] = np.asarray(target_yaw, dtype=np.float32)
        self.n, self.n_pad, self.n_wp = n, pad_to(n, pad), n_wp
        self.table = torch.from_numpy(tab).to(dev)
        order = ctx.order if (ctx.order is not None and ctx.order.n == n) else None    # per-drone arrays live in storage order
        c = np.zeros(self.n_pad, dtype=np.int32)
        if wp_counters is not None:
            w = np.asarray(wp_counters, dtype=np.int32)
            c[:n] = w if order is None else order.to_storage_np(w)
        self.counters = torch.from_numpy(c).to(dev)
        self.offsets = None
        if offsets is not None:
            o = np.zeros((3, self.n_pad), dtype=np.float32)
            off = np.asarray(offsets, dtype=np.float32)
            o[:, :n] = (off if order is None else order.to_storage_np(off)).T
            self.offsets = torch.from_numpy(o).to(dev)
        self.broadcast = False

    def fill(self, args: nat.StepArgs) -> None:
        args.wp_table = self.table.data_ptr()
        args.wp_counter = self.counters.data_ptr()
        args.wp_offset = self.offsets.data_ptr() if self.offsets is not None else None
        args.n_wp = self.n_wp


class TrajectoryTargets(Targets):
    """Targets sampled ON DEVICE from a min-snap trajectory (the reference pre-samples
    ``trajGenerator.get_des_state`` on the host into waypoint tables,
    examples/fly_INDI_TrajectoryTrack.py:133-160).  ``coeffs`` [n_seg*10, 3] and ``TS`` [n_seg+1]
    are ``trajGenerator.coeffs`` / ``.TS``; every drone has its own trajectory time ``t`` and its own
    yaw/heading memory (the reference's ``get_yaw`` is stateful, trajGen.py:128-143).  Call
    :meth:`sample` once per control step, then pass the object as ``targets``."""

    def __init__(self, ctx: Context, n: int, coeffs, TS, t0=None, offsets=None, layout: str = "soa", pad: int = 256):
        super().__init__(ctx, n, layout, pad=pad)
        self.ctx = ctx
        dev = ctx.device
        self.coeffs = torch.as_tensor(np.ascontiguousarray(coeffs), dtype=torch.float64).to(dev)
        self.TS = torch.as_tensor(np.ascontiguousarray(TS), dtype=torch.float64).to(dev)
        self.n_seg = int(self.TS.numel() - 1)
        self.t = torch.zeros((self.n_pad,), dtype=torch.float64, device=dev)      # (per-drone arrays: storage order)
        if t0 is not None:
            t0 = np.asarray(t0, dtype=np.float64)
            self.t[:n] = torch.from_numpy(t0 if self.order is None else self.order.to_storage_np(t0)).to(dev)
        self.yaw_state = torch.zeros((3, self.n_pad), dtype=torch.float64, device=dev)
        self.offsets = None
        if offsets is not None:
            o = np.zeros((3, self.n_pad), dtype=np.float32)
            off = np.asarray(offsets, dtype=np.float32)
            o[:, :n] = (off if self.order is None else self.order.to_storage_np(off)).T
            self.offsets = torch.from_numpy(o).to(dev)

    def sample(self, dt_advance: float) -> None:
        """targets <- get_des_state(t) per drone; t += dt_advance."""
        nat.check(self.ctx.lib.dsim_traj_sample(
            self.ctx.handle, self.ctx.stream_ptr(), self.n, self.coeffs.data_ptr(), self.TS.data_ptr(), self.n_seg,
            self.t.data_ptr(), float(dt_advance), self.yaw_state.data_ptr(),
            self.offsets.data_ptr() if self.offsets is not None else None, self.view()))
