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
acts along the receiver's body z axis at the COM.  Deviation (documented in DESIGN.md): the force is
evaluated once per Env.step from the positions at the start of the step and held over its
sub-steps (identical for phys_substeps = 1).
"""
from __future__ import annotations

import ctypes
import math
from typing import Optional

import torch

from . import _native as nat

CUTOFF = 10.0     # BaseAviary.py:1752: "Ignore drones more than 10 meters away"


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


class HaloExchange:
    """Halo exchange of positions between the ranks of a SPATIALLY sharded fleet (BASELINE config 5: slab
    decomposition, RCCL send/recv between neighbouring slabs only) — the alternative to the all-gather.

    Every `refresh` Env.steps the ranks all-gather their xy bounding boxes (4 floats each, one host sync)
    and each rank fixes, per peer, the index list of its own drones that can come within the 10 m cut-off
    of ANY drone of that peer before the next refresh: those inside the peer's box grown by
    `cutoff + 2 * margin`, where `margin = v_axis_max * dt_env * refresh` bounds how far a drone moves
    along one axis in that time (Bullet clamps every coordinate velocity to `max_coord_vel`, P4) — once for
    the sender's own motion, once for the growth of the peer's box.  Between refreshes the message sizes
    are therefore known on the host, and one step's exchange is: gather the listed positions, one grouped
    batch of isend/irecv (ncclGroupStart/End over xGMI point-to-point links), concatenate.  Nothing is
    approximated: a drone that is not in the list cannot reach the cut-off, and the force kernel re-tests
    every candidate pair with the current positions.  Ranks whose boxes are far apart exchange nothing."""

    def __init__(self, dist, v_axis_max: float, dt_env: float, cutoff: float = CUTOFF, refresh: int = 16):
        self.dist, self.cutoff, self.refresh = dist, float(cutoff), int(refresh)
        self.margin = float(v_axis_max) * float(dt_env) * self.refresh
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self._age = None
        self._send_idx, self._recv_buf = {}, {}
        self.sent_per_step = 0            # drones this rank sends per step (diagnostic)

    def _refresh(self, pos: torch.Tensor) -> None:
        dist = self.dist
        box = torch.cat([pos[:2].min(dim=1).values, pos[:2].max(dim=1).values]).to(torch.float32)
        wire = _wire_device(dist, pos.device)
        boxes = torch.empty((self.world, 4), dtype=torch.float32, device=wire)
        dist.all_gather_into_tensor(boxes, box.reshape(1, 4).contiguous().to(wire))
        boxes = boxes.cpu()
        reach = self.cutoff + 2.0 * self.margin
        mine = boxes[self.rank]
        self._send_idx = {}
        counts = torch.zeros((self.world,), dtype=torch.int64)
        for p in range(self.world):
            if p == self.rank:
                continue
            lo, hi = boxes[p, :2] - reach, boxes[p, 2:] + reach
            if bool((mine[2:] < lo).any() or (mine[:2] > hi).any()):      # my whole box is out of that peer's reach
                continue
            m = (pos[0] >= lo[0]) & (pos[0] <= hi[0]) & (pos[1] >= lo[1]) & (pos[1] <= hi[1])
            idx = torch.nonzero(m).squeeze(1)
            if idx.numel():
                self._send_idx[p] = idx
                counts[p] = idx.numel()
        table = torch.empty((self.world, self.world), dtype=torch.int64, device=wire)
        dist.all_gather_into_tensor(table, counts.reshape(1, -1).to(wire))
        table = table.cpu()
        self._recv_buf = {p: torch.empty((3, int(table[p, self.rank])), dtype=pos.dtype, device=wire)
                          for p in range(self.world) if p != self.rank and int(table[p, self.rank]) > 0}
        self.sent_per_step = int(counts.sum())
        self._age = 0

    def exchange(self, local_pos: torch.Tensor) -> torch.Tensor:
        """local_pos [3, n] -> [3, n + halo]: this rank's drones first (local_offset = 0), then the halo
        drones received from the peers."""
        pos = local_pos.contiguous()
        if self._age is None or self._age >= self.refresh:
            self._refresh(pos)
        self._age += 1
        dist = self.dist
        ops, keep = [], []
        for p, idx in self._send_idx.items():
            buf = pos.index_select(1, idx).contiguous().to(self._recv_wire(pos))
            keep.append(buf)
            ops.append(dist.P2POp(dist.isend, buf, p))
        for p, buf in self._recv_buf.items():
            ops.append(dist.P2POp(dist.irecv, buf, p))
        if ops:
            for r in dist.batch_isend_irecv(ops):
                r.wait()
        return torch.cat([pos] + [self._recv_buf[p].to(pos.device) for p in sorted(self._recv_buf)], dim=1)

    def _recv_wire(self, pos):
        return _wire_device(self.dist, pos.device)


class Downwash:
    """Evaluates formula P8 for the drones of one env/state block against world positions."""

    def __init__(self, ctx, state, type_id: Optional[torch.Tensor] = None, dist=None, cell: Optional[float] = None,
                 box_refresh: int = 256, halo: Optional[HaloExchange] = None):
        # cell = None: 5 m cells (half the cut-off, 5 x 5 cells scanned per drone: 30 % fewer candidate pairs than
        # 3 x 3 cells of 10 m) whenever the world's shape takes the bucket form of the grid, 10 m cells otherwise
        self.ctx, self.state, self.type_id, self.dist = ctx, state, type_id, dist
        self._auto_cell = cell is None
        self.cell = 0.5 * CUTOFF if cell is None else float(cell)
        self.halo = halo                 # None: all-gather of the world's positions (any index sharding)
        self.force = torch.zeros((3, state.n_pad), dtype=torch.float32, device=ctx.device)
        self._counts = None              # drones per rank (all-gather form; fetched once)
        self._last = None                # DownwashArgs of the last compute(): the grid a step kernel may fill for the next
        self._prebin_version = None      # state.version at the time a step kernel was handed that grid
        self._ws = None
        self._box = None                 # (xmin, ymin, nx, ny): a search-efficiency hint, never a correctness input
        self._box_age, self._box_refresh = 0, box_refresh

    def _grid_box(self, wp: torch.Tensor):
        """Bounding box of the world in xy -> grid.  Drones that later leave the box are clamped to
        its border cells by the kernels (pairs within 10 m stay in adjacent cells), so a stale box
        only costs search efficiency; it is re-measured every `box_refresh` calls (one host sync)."""
        if self._box is None or self._box_age >= self._box_refresh:
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

    def _grid_args(self, world_pos, local_offset) -> nat.DownwashArgs:
        st = self.state
        single = (world_pos is None and self.halo is None
                  and (self.dist is None or not self.dist.is_initialized() or self.dist.get_world_size() == 1))
        if single:
            # the world is this fleet: the kernels read positions straight from the state block (no gathered copy)
            wp, m = None, st.n
            local_offset = 0
            box_src = st.fields(0, 2) if (self._box is None or self._box_age >= self._box_refresh) else None
        elif world_pos is None and self.halo is not None:
            world_pos, local_offset = self.halo.exchange(st.fields(0, 3)[:, : st.n]), 0
        elif world_pos is None:
            if self._counts is None:
                self._counts = shard_counts(st.n, self.dist)
            world_pos = gather_positions(st.fields(0, 3), self.dist, self._counts)
            rank = self.dist.get_rank() if (self.dist is not None and self.dist.is_initialized()) else 0
            local_offset = sum(self._counts[:rank])
        if not single:
            wp = world_pos.to(torch.float32).contiguous()
            m = wp.shape[1]
            box_src = wp
        xmin, ymin, nx, ny = self._grid_box(box_src)
        if self._auto_cell and self.cell < CUTOFF and (m < 4 * nx * ny or not self.ctx.lib.dsim_downwash_prebin_ok(m, nx, ny)):
            # sparse world (fewer than 4 drones per 5 m cell: four times fewer, fuller cells serve it better), or too
            # many cells / too dense for the bucket form (the counting-sort form wants cells of the full cut-off)
            self.cell, self._box = CUTOFF, None
            if box_src is None:
                box_src = st.fields(0, 2) if single else wp
            xmin, ymin, nx, ny = self._grid_box(box_src)
        need = self.ctx.lib.dsim_downwash_workspace(m, nx, ny)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty((need,), dtype=torch.int32, device=self.ctx.device)
        a = nat.DownwashArgs()
        a.pos_all, a.m, a.m_pad = (wp.data_ptr() if wp is not None else None), m, m
        a.xmin, a.ymin, a.cell, a.nx, a.ny = xmin, ymin, self.cell, nx, ny
        a.workspace, a.workspace_len = self._ws.data_ptr(), self._ws.numel()
        a.type_id = self.type_id.data_ptr() if self.type_id is not None else None
        a.local_offset = int(local_offset or 0)
        a.prebinned = 0
        self._keep = wp                     # the kernels read it asynchronously on the stream
        return a

    # ---- the step kernel fills the next step's grid (dsim_step_args.bin_next) ------------------------------------
    def bin_next_ptr(self):
        """Pointer to the grid description of the NEXT compute() for ``dsim_step_args.bin_next``, or None when that
        grid cannot be filled ahead (no compute() yet, or a shape that takes the counting-sort form).  The library
        re-checks everything; a grid that changes after all (box refresh, halo resize) just means a full binning pass."""
        a = self._last
        if a is None or not self.ctx.lib.dsim_downwash_prebin_ok(a.m, a.nx, a.ny):
            return None
        self._prebin_version = self.state.version
        return ctypes.addressof(a)

    def invalidate_prebin(self) -> None:
        """The state was changed by something other than the fused step that pre-binned it."""
        self._prebin_version = None

    def compute(self, world_pos: Optional[torch.Tensor] = None, local_offset: Optional[int] = None) -> torch.Tensor:
        """Returns the SoA [3, n_pad] body-frame force (x = y = 0) to pass as ``ext_force``.
        ``world_pos`` [3, m]: positions of every drone of the world with this block's drones at
        ``local_offset`` (default: gathered from the ranks' states / the block alone)."""
        a = self._grid_args(world_pos, local_offset)
        a.prebinned = int(self._prebin_version is not None and self._prebin_version == self.state.version)
        self._prebin_version = None
        nat.check(self.ctx.lib.dsim_downwash(self.ctx.handle, self.ctx.stream_ptr(), self.state.n, self.state.view(),
                                             ctypes.byref(a), self.force.data_ptr()))
        self._last = a
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
