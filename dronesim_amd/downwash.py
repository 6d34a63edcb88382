"""Neighbour downwash (formula P8, BaseAviary._downwash, dronesim/envs/BaseAviary.py:1736-1763) for a
fleet that may be sharded over several GPUs.

This is the ONLY inter-drone term on the path and therefore the only exchange step of the
multi-GPU layout.  Two forms: an all-gather of every shard's positions (12 B/drone; RCCL over xGMI;
6.3 MB in total for 524 288 drones; works for any index sharding), or — for a spatially sharded fleet —
a halo exchange between neighbouring slabs (HaloExchange: grouped send/recv of the boundary drones
only).  Either way each rank then evaluates the force on its OWN drones against the positions it holds
with a uniform-grid neighbour search (dsim_downwash).  Nothing else of the state ever leaves its GPU.

The reference's loop is O(N^2) over the whole world, per drone, per sub-step, and is dead code in
the fork; the intended semantics are kept: receivers use their own type's coefficients, the force
acts along the receiver's body z axis at the COM, and it is evaluated per physics SUB-STEP from the
positions at the start of that sub-step, as the reference's loop refreshes them (BaseAviary.py:510-536):
with AGGR_PHY_STEPS > 1 the env launches one [query -> one-sub-step physics] pair per sub-step
(envs/fleet_aviary.py); one evaluation serves a launch.
"""
from __future__ import annotations

import ctypes
import math
import os
from typing import Optional

import torch

from . import _native as nat

CUTOFF = 10.0     # BaseAviary.py:1752: "Ignore drones more than 10 meters away"
KEEP_RUN_AHEAD = 8    # kept candidate lists: list-served queries the host may be ahead of the device (Downwash._keep_next)


def shard_counts(n_local: int, dist=None):
    """Drones per rank, [world] python ints (one small all-gather; sharding.shard_range hands out shards whose sizes
    differ by one when the fleet does not divide evenly)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [int(n_local)]
    world = dist.get_world_size()
    wire = torch.device("cpu") if dist.get_backend() == "gloo" else torch.device("cuda", torch.cuda.current_device())
    mine = torch.tensor([int(n_local)], dtype=torch.int64, device=wire)
    out = torch.empty((world,), dtype=torch.int64, device=wire)
    dist.all_gather_into_tensor(out, mine)
    return [int(x) for x in out.cpu()]


def gather_positions(local_pos: torch.Tensor, dist=None, counts=None) -> torch.Tensor:
    """local_pos [3, n_local] -> world positions [3, sum(counts)] in global drone order (rank-major, i.e. the
    contiguous shards of sharding.shard_range).  ``counts`` = drones per rank (shard_counts); None = every rank
    holds n_local.  Unequal shards are padded to the largest for the collective and compacted afterwards."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local_pos.contiguous()
    world = dist.get_world_size()
    dev = local_pos.device
    n_max = local_pos.shape[1] if counts is None else max(counts)
    loc = local_pos.contiguous()
    if loc.shape[1] < n_max:
        loc = torch.cat([loc, loc.new_zeros((3, n_max - loc.shape[1]))], dim=1)
    loc = loc.to(_wire_device(dist, dev))
    out = torch.empty((world * 3, n_max), dtype=loc.dtype, device=loc.device)   # ranks stacked on dim 0
    dist.all_gather_into_tensor(out, loc)
    out = out.reshape(world, 3, n_max).permute(1, 0, 2)                          # [3, world, n_max]
    if counts is None or all(c == n_max for c in counts):
        return out.reshape(3, world * n_max).contiguous().to(dev)
    return torch.cat([out[:, r, :c] for r, c in enumerate(counts)], dim=1).contiguous().to(dev)


def _wire_device(dist, dev):
    """RCCL moves device memory directly; the gloo backend (CPU tests, and several rehearsal ranks sharing
    one GPU) has no device-memory point-to-point, so there the payload is staged through the host."""
    return torch.device("cpu") if dist.get_backend() == "gloo" else dev


class HaloWire:
    """Host logic of the halo exchange of a SPATIALLY sharded fleet (BASELINE config 5: slabs along x, one rank per GPU),
    on whatever device its buffers live.  The device operations are hooks — HaloPlan implements them with the library's
    kernels; the CPU rehearsal of tests/test_sharding_cpu.py with torch on the host.

    A message to a peer is a HEADER (nat.HALO_HDR floats: how many positions follow, and the sender's xy box at the time
    of sending) + xyz triples, in a persistent per-peer buffer.  Every Env.step `_pack` selects, per peer, the own drones
    inside that peer's box — as its LAST message reported it — grown by reach = cut-off + v_clamp x dt_env: the box is one
    step old, and no coordinate can move further than that in one step (Bullet clamps every coordinate velocity to
    max_coord_vel, P4), so the selection is exact for ANY motion — no measured-speed assumption (BASELINE config 5's
    random world holds near-vertical pairs whose P8 term is singular: drones leave at tens of m/s).  Then ONE grouped
    batch of isend / irecv moves the buffers.  Message sizes are fixed on the host (capacities with headroom); the
    count travels in the header.  Only `resize` synchronises the host, every `resize_every` steps: it all-gathers the
    boxes (seeding the headers), lets `_pack` count what would travel, all-gathers the counts and re-makes the capacities;
    a pair of ranks whose boxes are further apart than cut-off + 2 x v_clamp x dt_env x resize_every (both may move, towards
    each other) cannot meet before the next resize and exchanges nothing.  A selection that outgrows its capacity in between is dropped AND counted
    (`_overflow`): 0 certifies that nothing was missed; a non-zero count raises at the next resize."""

    def __init__(self, dist, device, cap: int, dt_env: float, v_clamp: float, cutoff: float = CUTOFF, resize_every: int = 128,
                 headroom: float = 1.25, slack: int = 512):
        self.dist = dist
        self.cutoff, self.resize_every, self.dt_env, self.v_clamp = float(cutoff), int(resize_every), float(dt_env), float(v_clamp)
        self.headroom, self.slack = float(headroom), int(slack)
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        if self.world > nat.MAX_PEERS:
            raise ValueError(f"halo plan: at most {nat.MAX_PEERS} ranks (one node)")
        dev = torch.device(device)
        self.device, self.cap = dev, int(cap)
        self.stride = nat.HALO_HDR + 3 * self.cap
        self.gloo = dist.get_backend() == "gloo"
        self.send = torch.zeros((self.world, self.stride), dtype=torch.float32, device=dev)
        self.recv = torch.zeros((self.world, self.stride), dtype=torch.float32, device=dev)
        self.bounds_dev = torch.zeros((5,), dtype=torch.float32, device=dev)
        self.bounds_all = torch.zeros((self.world, 5), dtype=torch.float32, device=dev)
        self.count_dev = torch.zeros((self.world,), dtype=torch.int32, device=dev)
        self.table_dev = torch.zeros((self.world, self.world), dtype=torch.int32, device=dev)
        self.send_cap = [0] * nat.MAX_PEERS
        self.recv_cap = [0] * nat.MAX_PEERS
        self.reach = [0.0] * nat.MAX_PEERS
        self.step_reach = self.cutoff + self.v_clamp * self.dt_env
        self._age = None
        self._ops = []
        self.bounds_host = None                   # [world, 5] of the last resize
        self.sent_per_step = self.recv_per_step = 0          # positions selected at the last resize (what a step ships, +- the flux)
        self._overflow_seen = 0
        self._staged = self.gloo and dev.type == "cuda"      # gloo has no device-memory point-to-point: through pinned host memory
        if self._staged:
            self._h_send = torch.zeros((self.world, self.stride), dtype=torch.float32).pin_memory()
            self._h_recv = torch.zeros((self.world, self.stride), dtype=torch.float32).pin_memory()

    # ---- the device operations ----
    def _bounds(self) -> None:
        raise NotImplementedError                 # -> self.bounds_dev: xmin, ymin, xmax, ymax, max |coordinate velocity|

    def _pack(self, stream=None) -> None:
        raise NotImplementedError                 # self.send[p] = header + positions, for every p with send_cap[p] > 0

    def _overflow(self) -> int:
        return 0                                  # positions dropped for lack of capacity so far

    def _sync(self) -> None:
        pass

    def _caps_changed(self) -> None:
        pass

    # ---- collectives on small tensors (gloo with device tensors: through the host) ----
    def _all_gather(self, out: torch.Tensor, mine: torch.Tensor) -> None:
        if self.gloo and out.device.type == "cuda":
            o = torch.empty(out.shape, dtype=out.dtype)
            self.dist.all_gather_into_tensor(o, mine.cpu().reshape(1, -1).contiguous())
            out.copy_(o)
        else:
            self.dist.all_gather_into_tensor(out, mine.reshape(1, -1).contiguous())

    def halo_total(self) -> int:
        """Capacity of what the peers send (the upper bound of the halo's size that the grid is laid out for)."""
        return int(sum(self.recv_cap[p] for p in range(self.world) if p != self.rank))

    def messages(self):
        return [p for p in range(self.world) if self.send_cap[p] > 0]

    def resize(self) -> None:
        """Re-makes which pairs of ranks talk and how large their messages are (class docstring): two small collectives,
        one dry pack, ONE read-back."""
        ov = self._overflow()
        if ov != self._overflow_seen:
            self._overflow_seen = ov
            raise RuntimeError("halo plan: a selection outgrew its message since the last resize; the force of those steps may "
                               "have missed pairs inside the cut-off (raise headroom / slack or resize more often)")
        self._bounds()
        self._all_gather(self.bounds_all, self.bounds_dev)
        b = self.bounds_all.cpu().numpy()
        self.bounds_host = b
        # (BOTH boxes may move towards each other at the clamp: the gap closes by up to 2 v_clamp dt_env per step)
        far = self.cutoff + 2.0 * self.v_clamp * self.dt_env * self.resize_every
        talk = [False] * nat.MAX_PEERS
        for p in range(self.world):
            if p != self.rank:
                gap = max(b[p, 0] - b[self.rank, 2], b[self.rank, 0] - b[p, 2], b[p, 1] - b[self.rank, 3], b[self.rank, 1] - b[p, 3], 0.0)
                talk[p] = bool(gap <= far)
        # the peers' boxes as of now go where their messages' headers will keep them fresh
        self.recv[:, 1:5] = self.bounds_all[:, 0:4]
        for p in range(nat.MAX_PEERS):
            self.send_cap[p] = self.cap if talk[p] else 0          # a dry pack counts what each talking peer would get
            self.reach[p] = self.step_reach
        self._caps_changed()
        self._pack()
        self._sync()
        self.count_dev.copy_(self.send[:, 0].view(torch.int32))
        self._all_gather(self.table_dev, self.count_dev)
        table = self.table_dev.cpu()                                   # the read-back of the window

        def capacity(count):
            return int(min(self.cap, -(-int(count * self.headroom + self.slack) // 256) * 256))
        for p in range(nat.MAX_PEERS):
            self.send_cap[p] = capacity(table[self.rank, p]) if talk[p] else 0
            self.recv_cap[p] = capacity(table[p, self.rank]) if talk[p] else 0
        self.sent_per_step = int(sum(int(table[self.rank, p]) for p in range(self.world) if talk[p]))
        self.recv_per_step = int(sum(int(table[p, self.rank]) for p in range(self.world) if talk[p]))
        self._caps_changed()
        # the grouped batch of the window: persistent buffers, fixed sizes
        src, dst = (self._h_send, self._h_recv) if self._staged else (self.send, self.recv)
        n_s = lambda p: nat.HALO_HDR + 3 * self.send_cap[p]
        n_r = lambda p: nat.HALO_HDR + 3 * self.recv_cap[p]
        self._ops = [self.dist.P2POp(self.dist.isend, src[p, : n_s(p)], p) for p in range(self.world) if self.send_cap[p]]
        self._ops += [self.dist.P2POp(self.dist.irecv, dst[p, : n_r(p)], p) for p in range(self.world) if self.recv_cap[p]]
        self._age = 0

    def due(self) -> bool:
        return self._age is None or self._age >= self.resize_every

    def _wire(self, sync=None) -> None:
        """The packed buffers travel: one grouped batch (ncclGroupStart/End over xGMI point-to-point on RCCL)."""
        if not self._ops:
            return
        if self._staged:
            for p in range(self.world):
                if self.send_cap[p]:
                    k = nat.HALO_HDR + 3 * self.send_cap[p]
                    self._h_send[p, :k].copy_(self.send[p, :k], non_blocking=True)
            sync()                                                     # the rehearsal path synchronises the host here
        for w in self.dist.batch_isend_irecv(self._ops):
            w.wait()                                                   # RCCL: a stream-level wait, the host runs on
        if self._staged:
            for p in range(self.world):
                if self.recv_cap[p]:
                    k = nat.HALO_HDR + 3 * self.recv_cap[p]
                    self.recv[p, :k].copy_(self._h_recv[p, :k], non_blocking=True)

    def exchange(self) -> None:
        if self.due():
            self.resize()
        self._age += 1
        self._pack()
        self._wire()

    def received_positions(self) -> torch.Tensor:
        """[3, halo] of what the peers sent last (host-synchronous: reads the headers' counts; tests and the rare
        counting-sort form of the grid)."""
        parts = []
        for p in range(self.world):
            if self.recv_cap[p]:
                k = min(int(self.recv[p, 0:1].view(torch.int32)[0]), self.recv_cap[p])
                parts.append(self.recv[p, nat.HALO_HDR: nat.HALO_HDR + 3 * k].reshape(k, 3))
        if not parts:
            return torch.zeros((3, 0), dtype=torch.float32, device=self.device)
        return torch.cat(parts, dim=0).T.contiguous()


class HaloPlan(HaloWire):
    """HaloWire on the device, device-paced.  Everything per-step is preallocated and enqueued, nothing synchronises
    the host:

        main stream:  dsim_halo_pack (selection against the peers' boxes + packing + headers, one launch)
        wire:         ONE grouped batch of isend/irecv on persistent per-peer buffers, behind the pack (RCCL's own stream)
        main stream:  dsim_downwash(DSIM_DW_LOCAL) — the local part of the query runs while the positions are on the
                      wire — wait(wire) -> dsim_downwash(DSIM_DW_HALO_BIN) -> dsim_downwash(DSIM_DW_HALO_QUERY) -> dsim_step

    DSIM_Q_HALO_OVERFLOW counts what a message could not hold.  The gloo backend (rehearsal ranks sharing one GPU) stages
    the packed buffers through pinned host memory, which synchronises the host: a rehearsal of the logic, not of the
    pacing; there and with bench.MirrorDist the wire runs on the main stream and the query is one pass over one grid
    (Downwash.split = None: two passes only where a wire with real latency runs beside the first)."""

    def __init__(self, ctx, state, dist, dt_env: float, v_clamp: float, cutoff: float = CUTOFF, resize_every: int = 128,
                 headroom: float = 1.25, slack: int = 512):
        super().__init__(dist, ctx.device, state.n_pad, dt_env, v_clamp, cutoff, resize_every, headroom, slack)
        self.ctx, self.state = ctx, state
        self.scratch = torch.zeros((32,), dtype=torch.int32, device=ctx.device)
        self.plan = nat.HaloPlan()
        self.plan.world, self.plan.rank, self.plan.cap = self.world, self.rank, self.cap
        self.plan.send, self.plan.recv, self.plan.scratch = self.send.data_ptr(), self.recv.data_ptr(), self.scratch.data_ptr()
        self.timing = None                        # a list: every step appends (start of the pack, end of the halo binning) events
        self._pack_call = None
        self._works = None
        self._started = False
        self.on_rccl = dist.get_backend() == "nccl"

    def _bounds(self) -> None:
        st = self.state
        nat.check(self.ctx.lib.dsim_fleet_bounds(self.ctx.handle, self.ctx.stream_ptr(), st.n, st.view(), self.bounds_dev.data_ptr()))

    def _pack(self, stream=None) -> None:
        st = self.state
        if self._pack_call is None:
            self._pack_call = (st.view(), ctypes.byref(self.plan))
        nat.check(self.ctx.lib.dsim_halo_pack(self.ctx.handle, stream if stream is not None else self.ctx.stream_ptr(), st.n,
                                              *self._pack_call))

    def _sync(self) -> None:
        torch.cuda.current_stream(self.ctx.device).synchronize()

    def _overflow(self) -> int:
        return self.overflow()

    def _caps_changed(self) -> None:
        for p in range(nat.MAX_PEERS):
            self.plan.send_cap[p], self.plan.recv_cap[p], self.plan.reach[p] = self.send_cap[p], self.recv_cap[p], self.reach[p]

    def overflow(self) -> int:
        """Positions that a message could not hold so far (synchronises the stream); 0 = nothing was ever missed."""
        return self.ctx.query(nat.QUERY_HALO_OVERFLOW)

    def start(self) -> None:
        """Select + pack on the CURRENT stream, then the wire behind it — without waiting for it: the caller enqueues the
        local pass of the query next, then `finish()`.  On RCCL the grouped batch runs on the collective library's own
        stream, ordered behind the pack, and nobody waits for it until `finish()`.  The stand-in transports (gloo staging,
        bench.MirrorDist) run on the current stream itself.

        What the kernel traces of `bench.py --workload config5 --mirror-peer` taught about streams on this stack (one
        MI355X, 65 536 drones, a wire of two small device ops): a stream that sits in a cross-stream
        wait resumes 13-19 us after the signal — whatever the event's release scope, whatever the stream's priority — and a
        pack that shares the CUs with the local pass takes 32 us instead of 16.  Pack + wire + halo binning on a side
        stream beside the local pass: 101 us per step; everything on ONE stream with the two-pass query: 92 us; ONE
        stream and ONE grid (no second pass): 76-80 us.  A second stream of our own never paid; only the collective
        library's, which is not ours to remove, is used."""
        if self.due():
            self.resize()                              # (host-synchronous, every resize_every steps)
        self._age += 1
        self._works = None
        self._started = bool(self._ops)
        if not self._ops:
            return
        if self.timing is not None:
            self._t0 = torch.cuda.Event(enable_timing=True)
            self._t0.record(torch.cuda.current_stream(self.ctx.device))
        self._pack(self.ctx.stream_ptr())
        if self.on_rccl:
            self._works = self.dist.batch_isend_irecv(self._ops)       # RCCL's stream waits for the pack; nobody waits for RCCL yet
        else:
            self._wire(torch.cuda.current_stream(self.ctx.device).synchronize)

    def finish(self) -> None:
        """Orders the current stream behind the wire (what arrived is in `recv`)."""
        if self._works:
            for w in self._works:
                w.wait()                                               # stream-level: the host runs on

    def exchange(self) -> None:
        self.start()
        self.finish()


class Downwash:
    """Evaluates formula P8 for the drones of one env/state block against world positions."""

    def __init__(self, ctx, state, type_id: Optional[torch.Tensor] = None, dist=None, cell: Optional[float] = None,
                 box_refresh: int = 256, halo: Optional[HaloPlan] = None, split: Optional[bool] = None,
                 keep_lists: int = 0, keep_skin: float = 0.1, keep_movers: int = 48):
        # cell = None: 5 m cells (half the cut-off, 5 x 5 cells scanned per drone: 30 % fewer candidate pairs than
        # 3 x 3 cells of 10 m) whenever the world's shape takes the bucket form of the grid, 10 m cells otherwise
        self.ctx, self.state, self.type_id, self.dist = ctx, state, type_id, dist
        self._auto_cell = cell is None
        self.cell = 0.5 * CUTOFF if cell is None else float(cell)
        self.halo = halo                 # None: all-gather of the world's positions (any index sharding)
        # halo form: a local pass beside the exchange, then a halo pass — or one grid, one pass behind the wire.  None: two
        # passes on RCCL, whose wire has latency to hide (its own stream, its events), one pass for the stand-in transports
        # (measured with bench.MirrorDist: 76-80 us per step against 92 with two passes on the same stream)
        self.split = (halo is not None and getattr(halo, "on_rccl", False)) if split is None else bool(split)
        self.force = torch.zeros((3, state.n_pad), dtype=torch.float32, device=ctx.device)
        self._counts = None              # drones per rank (all-gather form; fetched once)
        self._last = None                # DownwashArgs of the last compute(): the grid a step kernel may fill for the next
        self._prebin_version = None      # state.version at the time a step kernel was handed that grid
        self._ws = None
        self._box = None                 # (xmin, ymin, nx, ny): a search-efficiency hint, never a correctness input
        self._box_age, self._box_refresh = 0, box_refresh
        self._gather_out = None          # all-gather form: the preallocated [world * 3, n_max] receive buffer
        self._single = None              # single-rank form: (args, state view, byref) reused while the grid stands
        self._halo_args = None           # halo form: the three phases' argument blocks, rebuilt at a resize
        self._last_ok = {}               # id(args) -> dsim_downwash_prebin_ok of its shape
        self.pair_counter = None         # diagnostics (count_pairs): int64[1] the query adds its evaluated pairs to
        # kept candidate lists (dsim_downwash_args.keep; single-rank fleets at a density that takes the banded query): one
        # BUILD query serves keep_lists Env.steps, the others re-use its lists on refreshed positions.  Exact for any motion
        # (a drone that leaves the skin is handed to the overflow list); how often to BUILD only decides how long that list gets.
        self.keep_lists, self.keep_skin = max(int(keep_lists), 0), float(keep_skin)
        if self.keep_lists > 1 and self._auto_cell:
            self.cell = 0.5 * CUTOFF + self.keep_skin          # two rings of cells cover the reach widened by twice the skin
        self._keep_ws = None
        self._keep_age = 0               # queries since the last BUILD
        # Pacing.  The lists serve a fleet that HOVERS for keep_lists queries; one in coordinated motion leaves any skin within a few
        # steps, every drone joins the overflow list and a REUSE costs many times a BUILD.  The device reports how long that list
        # was and how many drones were half way out (dsim_downwash_keep_stats: host memory it writes, nothing synchronises); above
        # keep_movers (or with a sixteenth of the fleet half way out) the next query BUILDs and the period shrinks to what the fleet's
        # motion allows, growing back a quarter at a time; a period below 3 suspends the lists.
        self.keep_movers = int(keep_movers)
        self._keep_period = self.keep_lists
        self._keep_seq0 = 0              # REUSE queries enqueued before the last BUILD: later readings are about ITS lists
        self._keep_off = 0               # > 0: plain queries for that many calls, then another try
        self._keep_stats = (ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64())

    def _grid_box(self, wp, lo_hi=None):
        """Bounding box of the world in xy -> grid.  Drones that later leave the box are clamped to
        its border cells by the kernels (pairs within 10 m stay in adjacent cells), so a stale box
        only costs search efficiency; it is re-measured every `box_refresh` calls (one host sync), or, with a halo
        plan, taken from the bounds its refresh has already read back (lo_hi)."""
        if self._box is None or self._box_age >= self._box_refresh or lo_hi is not None:
            if lo_hi is not None:
                lo, hi = lo_hi
            else:
                assert wp is not None
                lo = wp[:2].min(dim=1).values.cpu()
                hi = wp[:2].max(dim=1).values.cpu()
            xmin, ymin = float(lo[0]) - self.cell, float(lo[1]) - self.cell
            nx = max(1, int(math.floor((float(hi[0]) + self.cell - xmin) / self.cell)) + 1)
            ny = max(1, int(math.floor((float(hi[1]) + self.cell - ymin) / self.cell)) + 1)
            while nx * ny > (1 << 22):      # a pathological spread: coarsen the grid, the search stays exact
                self.cell *= 2.0
                nx, ny = (nx + 1) // 2, (ny + 1) // 2
            self._box, self._box_age = (xmin, ymin, nx, ny), 0
        self._box_age += 1
        return self._box

    def _fill(self, wp, m, local_offset, box) -> nat.DownwashArgs:
        xmin, ymin, nx, ny = box
        a = nat.DownwashArgs()
        a.pos_all, a.m, a.m_pad = (wp.data_ptr() if wp is not None else None), m, m
        a.xmin, a.ymin, a.cell, a.nx, a.ny = xmin, ymin, self.cell, nx, ny
        a.type_id = self.type_id.data_ptr() if self.type_id is not None else None
        a.local_offset = int(local_offset or 0)
        a.prebinned, a.phase, a.halo = 0, nat.DW_ALL, None
        a.pairs_evaluated = self.pair_counter.data_ptr() if self.pair_counter is not None else None
        return a

    def _workspace(self, a, need) -> None:
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty((int(need),), dtype=torch.int32, device=self.ctx.device)
        a.workspace, a.workspace_len = self._ws.data_ptr(), self._ws.numel()

    def _grid_args(self, world_pos, local_offset) -> nat.DownwashArgs:
        st = self.state
        single = (world_pos is None and self.halo is None
                  and (self.dist is None or not self.dist.is_initialized() or self.dist.get_world_size() == 1))
        if single:
            # the world is this fleet: the kernels read positions straight from the state block (no gathered copy)
            wp, m = None, st.n
            local_offset = 0
            box_src = st.raw_fields(0, 2) if (self._box is None or self._box_age >= self._box_refresh) else None
        elif world_pos is None:
            if self._counts is None:
                self._counts = shard_counts(st.n, self.dist)
            world_pos = self._gather(st.raw_fields(0, 3))
            rank = self.dist.get_rank() if (self.dist is not None and self.dist.is_initialized()) else 0
            local_offset = sum(self._counts[:rank])
        if not single:
            wp = world_pos.to(torch.float32).contiguous()
            m = wp.shape[1]
            box_src = wp
        box = self._grid_box(box_src)
        if self._auto_cell and self.cell < CUTOFF and (m < 4 * box[2] * box[3] or not self.ctx.lib.dsim_downwash_prebin_ok(m, box[2], box[3])):
            # sparse world (fewer than 4 drones per 5 m cell: four times fewer, fuller cells serve it better), or too
            # many cells / too dense for the bucket form (the counting-sort form wants cells of the full cut-off)
            self.cell, self._box = CUTOFF, None
            if box_src is None:
                box_src = st.raw_fields(0, 2) if single else wp
            box = self._grid_box(box_src)
        a = self._fill(wp, m, local_offset, box)
        self._workspace(a, self.ctx.lib.dsim_downwash_workspace(m, box[2], box[3]))
        self._keep = wp                     # the kernels read it asynchronously on the stream
        if single and self.keep_lists > 1 and self.ctx.lib.dsim_downwash_keep_ok(m, box[2], box[3], self.cell, self.keep_skin):
            need = int(self.ctx.lib.dsim_downwash_keep_workspace(st.n_pad, box[2], box[3]))
            if self._keep_ws is None or self._keep_ws.numel() < need:
                self._keep_ws = torch.empty((need,), dtype=torch.int32, device=self.ctx.device)
            a.keep, a.keep_skin = nat.DW_KEEP_BUILD, self.keep_skin
            a.keep_ws, a.keep_ws_len = self._keep_ws.data_ptr(), self._keep_ws.numel()
            self._keep_age = 0
        return a

    def _gather(self, local_pos: torch.Tensor) -> torch.Tensor:
        """All-gather form with equal shards on RCCL: one collective into a preallocated buffer, one strided copy into
        global drone order; anything else (unequal shards, gloo) goes through gather_positions."""
        dist, counts = self.dist, self._counts
        if dist.get_backend() == "gloo" or any(c != counts[0] for c in counts) or local_pos.shape[1] != counts[0]:
            return gather_positions(local_pos, dist, counts)
        world, nmax = len(counts), counts[0]
        if self._gather_out is None or self._gather_out.shape != (world * 3, nmax):
            self._gather_out = torch.empty((world * 3, nmax), dtype=torch.float32, device=local_pos.device)
            self._gather_wp = torch.empty((3, world * nmax), dtype=torch.float32, device=local_pos.device)
        dist.all_gather_into_tensor(self._gather_out, local_pos.contiguous())
        self._gather_wp.view(3, world, nmax).copy_(self._gather_out.view(world, 3, nmax).permute(1, 0, 2))
        return self._gather_wp

    # ---- the step kernel fills the next step's grid (dsim_step_args.bin_next) ------------------------------------
    def bin_next_ptr(self):
        """Pointer to the grid description of the NEXT compute() for ``dsim_step_args.bin_next``, or None when that
        grid cannot be filled ahead (no compute() yet, or a shape that takes the counting-sort form).  The library
        re-checks everything; a grid that changes after all (box refresh, halo resize) just means a full binning pass."""
        a = self._last
        if a is None:
            return None
        key = (id(a), a.m, a.nx, a.ny)
        ok = self._last_ok.get(key)
        if ok is None:
            self._last_ok = {key: bool(self.ctx.lib.dsim_downwash_prebin_ok(a.m, a.nx, a.ny))}
            ok = self._last_ok[key]
        if not ok:
            return None
        self._prebin_version = self.state.version
        return ctypes.addressof(a)

    def count_pairs(self, on: bool = True) -> None:
        """Diagnostics: have the query count the (receiver, candidate) pairs it evaluates into ``pair_counter`` (int64[1],
        cumulative; dsim_downwash_args.pairs_evaluated).  Results do not change; bench.py's vector-pipe roofline reads it."""
        self.pair_counter = torch.zeros((1,), dtype=torch.int64, device=self.ctx.device) if on else None
        self._single = self._halo_args = None          # (the prepared argument blocks carry the pointer)

    def invalidate_prebin(self) -> None:
        """The state was changed by something other than the fused step that pre-binned it."""
        self._prebin_version = None
        self._single = None

    def compute(self, world_pos: Optional[torch.Tensor] = None, local_offset: Optional[int] = None) -> torch.Tensor:
        """Returns the SoA [3, n_pad] body-frame force (x = y = 0) to pass as ``ext_force``.
        ``world_pos`` [3, m]: positions of every drone of the world with this block's drones at
        ``local_offset`` (default: gathered from the ranks' states / the block alone)."""
        if self.halo is not None and world_pos is None:
            return self._compute_halo()
        single = (world_pos is None and (self.dist is None or not self.dist.is_initialized() or self.dist.get_world_size() == 1))
        if single and self._single is not None and self._box_age < self._box_refresh:
            # the world is this fleet and the grid has not changed: the argument block of the last call is this call's
            # (the Python side of a config-5 step is what paces small fleets: every struct that is not rebuilt counts)
            a, view, ref = self._single
            self._box_age += 1
        else:
            a = self._grid_args(world_pos, local_offset)
            view, ref = self.state.view(), ctypes.byref(a)
            self._single = (a, view, ref) if single else None
        a.prebinned = int(self._prebin_version is not None and self._prebin_version == self.state.version)
        self._prebin_version = None
        nat.check(self.ctx.lib.dsim_downwash(self.ctx.handle, self.ctx.stream_ptr(), self.state.n, view, ref,
                                             self.force.data_ptr()))
        if a.keep_ws:
            self._keep_next(a)
            a.keep_age = self._keep_age + 1 if a.keep == nat.DW_KEEP_REUSE else 0      # (which REUSE of these lists the next query is)
        self._last = a
        return self.force

    def _keep_next(self, a) -> None:
        """What the NEXT query of this grid will be — decided now, because the step in between has to be told (bin_next: it refreshes
        the lists' positions in front of a REUSE and bins in front of a BUILD or a plain query).

        The device's reports are as old as the host is ahead of it, and a fleet that starts to march leaves the skin within a handful
        of steps: the host therefore stays at most KEEP_RUN_AHEAD list-served queries ahead of the device while lists are in use (it
        polls the report's sequence number; nothing is synchronised, and 8 queued steps are a third of a millisecond of device work), a
        period that proved too long is remembered, and it grows back only while the reports are fresh."""
        if self._keep_off > 0:                           # suspended: plain queries, then another try with a short period
            self._keep_off -= 1
            a.keep = nat.DW_KEEP_OFF if self._keep_off > 0 else nat.DW_KEEP_BUILD
            return
        out, half, ofq, q = self._keep_stats
        stats = self.ctx.lib.dsim_downwash_keep_stats
        refs = (ctypes.byref(out), ctypes.byref(half), ctypes.byref(ofq), ctypes.byref(q))
        stats(self.ctx.handle, *refs)
        spins = 0
        while out.value >= 0 and q.value - ofq.value > KEEP_RUN_AHEAD:
            spins += 1
            if spins == 64 and self.ctx.device.type == "cuda" and torch.cuda.is_current_stream_capturing():
                break                                    # (a capture: nothing runs, nothing will be reported)
            stats(self.ctx.handle, *refs)
        if a.keep == nat.DW_KEEP_BUILD:
            self._keep_age, self._keep_seq0 = 0, q.value
        else:
            self._keep_age += 1
        # (a reading of THESE lists: the overflow list too long already, or a sixteenth of the fleet half way out — at the age the
        # lists had when it was taken, which is what the period learns)
        crowded = ofq.value > self._keep_seq0 and (out.value > self.keep_movers or
                                                  half.value > max(4 * self.keep_movers, self.state.n // 16))
        if crowded:
            self._keep_period = max(ofq.value - self._keep_seq0, 1)
            if self._keep_period < 3:                   # (a BUILD every other query costs more than plain queries do)
                self._keep_off, self._keep_period = 256, 4
                a.keep = nat.DW_KEEP_OFF
                return
            a.keep = nat.DW_KEEP_BUILD
        elif self._keep_age + 1 >= self._keep_period:
            a.keep = nat.DW_KEEP_BUILD
            if q.value - ofq.value <= 2:                 # (fresh reports: probe a longer period)
                self._keep_period = min(self.keep_lists, self._keep_period + max(1, self._keep_period // 4))
        else:
            a.keep = nat.DW_KEEP_REUSE

    def _compute_halo(self) -> torch.Tensor:
        """Spatially sharded fleet: the rest of the world is what the halo plan's peers send (class HaloPlan)."""
        lib, h, st, hp = self.ctx.lib, self.ctx.handle, self.state, self.halo
        resized = hp.due()
        hp.start()                                       # (resize when due: host-synchronous, rare) pack; the wire behind it
        if resized:
            # the grid covers this rank's box grown by what can happen before the next resize on every side (what lies
            # beyond is clamped to the border cells: exact, see _grid_box); re-made from bounds that are on the host anyway
            b = hp.bounds_host[hp.rank]
            grow = hp.cutoff + min(hp.v_clamp * hp.dt_env * hp.resize_every, 2.0 * hp.cutoff)
            self._halo_box = self._grid_box(None, ((b[0] - grow, b[1] - grow), (b[2] + grow, b[3] + grow)))
            m = st.n + hp.halo_total()
            if self._auto_cell and self.cell < CUTOFF and m < 4 * self._halo_box[2] * self._halo_box[3]:
                self.cell = CUTOFF
                self._halo_box = self._grid_box(None, ((b[0] - grow, b[1] - grow), (b[2] + grow, b[3] + grow)))
        box = self._halo_box
        if resized or self._halo_args is None:
            self._halo_m = st.n + hp.halo_total()
            self._halo_ok = bool(lib.dsim_downwash_prebin_ok(self._halo_m, box[2], box[3]))
        m = self._halo_m
        if not self._halo_ok:
            # a world too dense or too vast for the bucket form: one array, the counting-sort form (no overlap)
            hp.finish()
            wp = torch.cat([st.raw_fields(0, 3), hp.received_positions()], dim=1).contiguous()
            m = wp.shape[1]
            a = self._fill(wp, m, 0, box)
            self._workspace(a, lib.dsim_downwash_workspace(m, box[2], box[3]))
            self._keep = wp
            self._prebin_version = None
            nat.check(lib.dsim_downwash(h, self.ctx.stream_ptr(), st.n, st.view(), ctypes.byref(a), self.force.data_ptr()))
            self._last = None
            return self.force
        if resized or self._halo_args is None:
            blocks = []
            for ph in (nat.DW_HALO_BIN, nat.DW_LOCAL, nat.DW_HALO_QUERY, nat.DW_ALL):
                a = self._fill(None, m, 0, box)
                a.halo = ctypes.addressof(hp.plan)
                self._workspace(a, lib.dsim_downwash_workspace_halo(st.n, hp.halo_total(), box[2], box[3]))
                a.phase = ph
                blocks.append((a, ctypes.byref(a)))
            self._halo_args = (blocks, st.view(), self.force.data_ptr())
        (a_bin, r_bin), (a_loc, r_loc), (a_qry, r_qry), (a_all, r_all) = self._halo_args[0]
        view, fptr = self._halo_args[1], self._halo_args[2]
        pre = int(self._prebin_version is not None and self._prebin_version == self.state.version)
        self._prebin_version = None
        sp = self.ctx.stream_ptr()
        if not self.split:
            hp.finish()
            a_all.prebinned = pre
            nat.check(lib.dsim_downwash(h, sp, st.n, view, r_all, fptr))
            if hp.timing is not None and hp._started:
                t1 = torch.cuda.Event(enable_timing=True)
                t1.record(torch.cuda.current_stream(self.ctx.device))
                hp.timing.append((hp._t0, t1))
            self._last = a_all
            return self.force
        a_loc.prebinned = pre
        nat.check(lib.dsim_downwash(h, sp, st.n, view, r_loc, fptr))           # the local pass: runs while the positions travel
        hp.finish()
        nat.check(lib.dsim_downwash(h, sp, st.n, view, r_bin, None))           # what arrived -> the halo grid
        if hp.timing is not None and hp._started:
            t1 = torch.cuda.Event(enable_timing=True)
            t1.record(torch.cuda.current_stream(self.ctx.device))
            hp.timing.append((hp._t0, t1))
        nat.check(lib.dsim_downwash(h, sp, st.n, view, r_qry, fptr))           # the halo pass: force +=
        self._last = a_loc                               # what a step kernel that bins ahead is told about the next grid
        return self.force

    def adjacency(self, radius: float, max_k: int = 0, world_pos: Optional[torch.Tensor] = None,
                  local_offset: Optional[int] = None):
        """Fleet-scale form of BaseAviary._getAdjacencyMatrix (BaseAviary.py:901-921): per local drone
        the number of drones within ``radius`` and, if ``max_k`` > 0, up to max_k of their world
        indices ([max_k, n], -1 padded).  The dense O(N^2) matrix of the reference is produced only by
        the dict-mode observations of small fleets."""
        if self.cell < float(radius):          # the adjacency pass (counting-sort form) wants cells of the radius or more
            self.cell, self._box = float(radius), None
        self._auto_cell = False
        self._prebin_version = None
        a = self._grid_args(world_pos, local_offset)
        st = self.state
        count = torch.zeros((st.n_pad,), dtype=torch.int32, device=self.ctx.device)
        lst = torch.full((max_k, st.n_pad), -1, dtype=torch.int32, device=self.ctx.device) if max_k > 0 else None
        nat.check(self.ctx.lib.dsim_adjacency(self.ctx.handle, self.ctx.stream_ptr(), st.n, st.view(), ctypes.byref(a),
                                              float(radius), count.data_ptr(), lst.data_ptr() if lst is not None else None,
                                              max_k))
        return count[: st.n], (lst[:, : st.n] if lst is not None else None)
