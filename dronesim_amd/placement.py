"""Where the arrays a large fleet streams beside its state block lie in HBM.

Measured on MI355X (tools/membench.hip --bigsweep / --pairs / --regions, the placement / arena probes of rounds 3-5, in the git history up to 48ba88d;
profiles/r03_placement_*.txt): the Env.step launch of a 4 194 304-drone fleet (k_physics_fast: state
updated in place, 80-byte observation rows written beside it) takes 144-150 us or 166-169 us — and up to 204 us —
depending on nothing but WHERE the rows' allocation lies relative to the state's.  Inside one 24 GB allocation two
arrays streamed this way behave one way while they are less than 16 GiB apart and the other way from exactly 16 GiB on,
so device memory is organised in regions and what matters is whether two arrays share one; but which physical blocks
an allocation is made of is the memory manager's business (virtual addresses say nothing, one allocation need not be
contiguous), and a plain two-stream copy probe does NOT predict the real kernel.  This explains what earlier rounds
recorded as box-to-box spread of the reference-shaped loop (291-333 us in one and the same box).

So the placement is chosen by timing the real launch: `dsim_physics` with ZERO physics sub-steps is the same kernel with the
same memory streams — the state is read and written back bit for bit, the action is clipped and echoed, the rows are
written — and changes nothing.  `place_rows` allocates candidates for the rows one after the other, holding them all so that
the walk moves through device memory (up to 16 GiB, transient), times three such passes on each, stops at the first that
is clearly faster than the first one, keeps it (or the fastest of the whole walk) and releases the rest.
When every candidate times alike the state block itself may lie across two regions (one process in ten): CtrlAviary then
moves it to a fresh allocation (same contents) and walks once more, keeping the better pair.  Only for fleets whose rows
are at least `MIN_BYTES`: smaller fleets are bound by launch latency, not by HBM.

The same holds for what computeControl WRITES beside the state block whose controller memory it updates (command, position
error, yaw error: 32 bytes per drone; round 3's control-output probe: 144.7 us as allocated, 136 us with the outputs
elsewhere).  A controller bound to an env takes the 8 x n_pad floats the env left behind its placed rows (one allocation,
one search); one without an env searches for itself: that launch has no neutral form, so it takes a snapshot of the state
block, times real passes on the candidates and puts the snapshot back.

The per-drone targets of the fused step are READ beside the state, and there the SAME region is the good case: 154, 158 or
164 us per launch by where they lie (round 3's state probe), and the first candidate — right behind the state — may be the
middle one.  CtrlAviary.step_fused places them at its first call: real passes behind a snapshot of the state block on every
candidate of a 4 GiB walk (25 of them, no early end), the fastest kept.  One box of the round gave 155 / 158 / 164 us from
process to process without it and 154.1-154.8 us in six of six with it (profiles/r03_repeat_headline_8_processes.txt).

The search can only FIND a place: on one box of round 3 no candidate for the rows within 16 GiB was a good one.

Round 4 — bounded and self-contained.  Candidates come straight from the driver (`dsim_dev_alloc`: hipMalloc on the ctx's
device) and the ones not kept go straight back (`dsim_dev_free`): PyTorch's caching allocator is not involved, nothing
calls `torch.cuda.empty_cache()`, a co-resident policy network keeps its cached blocks.  The walk holds at most
min(WALK_BYTES, WALK_FRACTION of the device memory that is free when it starts) and never less than two candidates' worth
(else there is nothing to choose from: the array is allocated plainly); an allocation failure ends the walk with what it has;
a walk that could time nothing falls back to a plain zeroed array.  Every search reports what it cost: `seconds`,
`peak_bytes`.  The rows' walk strides (1 GiB of untimed ballast behind every candidate, `STRIDE_BYTES`) and may hold 34 GiB
for the fraction of a second it takes: back-to-back candidates within 16 GiB — round 3's walk — all timed alike (159-165 us)
on this round's boxes, where a striding walk finds 143 us from its third candidate on (round 4's region probe,
profiles/r04_region_probe.txt): the walk has to LEAVE the 16 GiB region the state block lies in.  (One arena — state block and written
arrays one 16 GiB window apart inside ONE allocation, the layout the probes show to be the good one in a fresh process — was
tried as a fallback for a walk that finds every candidate alike and never beat the walk's best inside the product: removed.)
Arrays that are READ beside the state (the targets of the fused step and of computeControl) want the state's OWN window:
CtrlAviary moves a large fleet's state block once into a driver allocation with room for two target blocks right behind it
(`_ensure_read_room`), and the targets' searches compare that block with a 4 GiB walk.
"""
from __future__ import annotations

import ctypes
import time
from typing import Callable, Optional

import torch

from . import _native as nat
from .fleet import Targets

MIN_BYTES = 64 << 20                          # arrays at least this large, or they are allocated plainly
WALK_BYTES = 4 << 30                          # held at once while searching (transient).  Round 4 walked 34 GiB (two 16 GiB regions and a candidate): on
                                              # fresh boxes that bought nothing reliable (BENCH_r04: -10.7 % to +3 %), so the opt-in default is small;
                                              # Context.placement_walk_bytes overrides it
WALK_FRACTION = 0.125                         # ... and never more than this share of the free device memory
STRIDE_BYTES = 1 << 30                        # untimed ballast between two candidates of the rows' walk: fewer, further apart
CLEARLY = 0.93                                # one candidate this much faster than another: the two cases are apart, stop


class _DriverBlock:
    """A device allocation made through the library (hipMalloc) and handed to PyTorch as a tensor by the CUDA array
    interface: the tensor keeps this object alive, and the block goes back to the driver when the last tensor on it dies."""

    def __init__(self, ctx, shape):
        self.ctx, self.lib, self.shape = ctx, ctx.lib, tuple(int(d) for d in shape)
        n = 4
        for d in self.shape:
            n *= d
        self.nbytes = n
        ptr = ctypes.c_void_p()
        rc = ctx.lib.dsim_dev_alloc(ctx.handle, n, ctypes.byref(ptr))
        if rc != 0 or not ptr.value:
            raise MemoryError(f"dsim_dev_alloc({n}) -> {rc}")
        self.ptr = int(ptr.value)
        self.__cuda_array_interface__ = {"shape": self.shape, "typestr": "<f4", "data": (self.ptr, False), "version": 2,
                                         "strides": None}

    def tensor(self) -> torch.Tensor:
        return torch.as_tensor(self, device=self.ctx.device)

    def __del__(self):
        # (the block goes back whether or not its Context is still open: env.close() destroys the ctx, the tensors on the
        # block are usually dropped afterwards — dsim_dev_free does not need the ctx)
        try:
            if self.ptr:
                self.lib.dsim_dev_free(None, ctypes.c_void_p(self.ptr))
        except Exception:
            pass
        self.ptr = 0


def _event_timer(trial, c, passes: int) -> float:
    """Microseconds per pass: `passes` passes behind one untimed pass, between two events on the current stream."""
    trial(c)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(passes):
        trial(c)
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / passes


def _free_bytes(device) -> int:
    if torch.device(device).type != "cuda":
        return 1 << 62
    return int(torch.cuda.mem_get_info(device)[0])


def place_rows(device, shape, trial: Callable[[torch.Tensor], None], passes: int = 3, report: Optional[list] = None,
               label: str = "observation rows", walk_bytes: Optional[int] = None, timer=None,
               clearly: float = CLEARLY, ctx=None, free_bytes: Optional[int] = None, stride_bytes: int = 0) -> torch.Tensor:
    """A zeroed fp32 array of `shape`.  `trial(array)` enqueues ONE pass of the real kernel writing its output to `array`
    (the caller restores whatever the passes change).  Candidates are allocated one after the other and all held, so that
    the walk moves through device memory; each is timed over `passes` passes behind one untimed pass; the walk ends as soon
    as one candidate is clearly faster than the first (it is kept), when the budget is held (the fastest is kept), or when
    the driver has no more to give.  (A candidate clearly SLOWER than the first does not end it: there are more than two
    levels — 154 / 158 / 164 us for the fused step by where its targets lie — and the first may be the middle one.)
    `ctx`: the fleet's Context — candidates come from the driver through it (dsim_dev_alloc), not from PyTorch's caching
    allocator; None (CPU tests): torch.empty.  `timer(trial, candidate, passes)`: the clock (tests).  `clearly`: the ratio
    that counts as clear; 0: no early end, the whole walk.  `free_bytes`: the free device memory to budget against (tests).
    `stride_bytes`: an untimed ballast block of that size is allocated (and held, inside the budget) behind every candidate:
    the walk then covers the budget with fewer candidates.  Round 4, round 4's region probe: 36 back-to-back 470 MB candidates
    — 16 GiB, the old budget — all timed 159-165 us on boxes where a walk with 1 GiB strides finds 143 us from its third
    candidate on: back-to-back allocations may never leave the 16 GiB region the state block lies in."""
    t_start = time.perf_counter()
    nbytes = 4
    for d in shape:
        nbytes *= int(d)
    free = _free_bytes(device) if free_bytes is None else int(free_bytes)
    if walk_bytes is None:
        walk_bytes = getattr(ctx, "placement_walk_bytes", None) or WALK_BYTES
    budget = min(int(walk_bytes), int(WALK_FRACTION * free))
    cands, blocks, times = [], [], []
    chosen, decided = None, ""
    held = 0
    if budget >= 2 * nbytes:
        while chosen is None and held + nbytes <= budget:
            try:
                if ctx is not None:
                    blk = _DriverBlock(ctx, shape)
                    c = blk.tensor()
                else:
                    blk, c = None, torch.empty(tuple(shape), dtype=torch.float32, device=device)
            except (MemoryError, RuntimeError):             # (torch.cuda.OutOfMemoryError is a RuntimeError)
                break
            cands.append(c); blocks.append(blk)
            held += nbytes
            times.append((timer or _event_timer)(trial, c, passes))
            if clearly > 0.0 and times[-1] < clearly * times[0]:
                chosen, decided = len(times) - 1, "a candidate clearly faster than the first"
            elif stride_bytes > 0:
                if held + stride_bytes + nbytes > budget:       # no room for another stride and a candidate behind it
                    break
                try:
                    blocks.append(_DriverBlock(ctx, (stride_bytes // 4,)) if ctx is not None else
                                  torch.empty((stride_bytes // 4,), dtype=torch.float32, device=device))
                    held += stride_bytes
                except (MemoryError, RuntimeError):
                    break
    if not times:
        # nothing to choose from (too little free memory for two candidates, or the driver refused the first): plainly
        keep = torch.zeros(tuple(shape), dtype=torch.float32, device=device)
        if report is not None:
            report.append({"array": label, "bytes": nbytes, "candidates": 0, "chosen": None, "decided_by": "no walk: allocated plainly",
                           "free_bytes": free, "budget_bytes": budget, "peak_bytes": 0,
                           "seconds": round(time.perf_counter() - t_start, 4)})
        return keep
    if chosen is None:
        chosen = min(range(len(times)), key=times.__getitem__)
        decided = "the fastest of the walk" if times[chosen] < CLEARLY * max(times) else "all alike"
    keep = cands[chosen]
    peak = held
    n_c = len(cands)
    c = blk = None
    del cands, blocks                         # what was not kept goes back to the driver (the tensors were the only references)
    keep.zero_()
    if report is not None:
        shown = times if len(times) <= 12 else times[:4] + times[-4:]
        report.append({"array": label, "bytes": nbytes, "candidates": n_c, "chosen": chosen,
                       "decided_by": decided, "chosen_pass_us": round(times[chosen], 1), "first_pass_us": round(times[0], 1),
                       "pass_us" if len(times) <= 12 else "pass_us_first4_last4": [round(t, 1) for t in shown],
                       "budget_bytes": budget, "peak_bytes": peak, "stride_bytes": int(stride_bytes),
                       "seconds": round(time.perf_counter() - t_start, 4), "memory": "driver (dsim_dev_alloc)" if ctx is not None else "torch"})
    return keep



class PlacedFleetArrays:
    """What ``CtrlAviary`` does with the above (a mix-in: the env's own attributes — ctx, state, _last_action, _obs_buf, _runs,
    _downwash, _phys_options ... — are used as they are).  Opt-in since round 5 (``CtrlAviary(placement=True)``): on fresh boxes
    the searches were worth between -10 % and +12 % of the two-call loop (BENCH_r04.json)."""

    def _placement_applies(self, nbytes: int) -> bool:
        """Arrays written beside the state block are placed by trial (placement.py) for fleets that are bound by HBM and that
        the fast kernels serve: large ones, without the downwash chain, without the drag / ground / plane options."""
        served = self._type_id is None or (self._runs is not None and len(self._runs) <= 8)
        return bool(self.ctx.placement and nbytes >= MIN_BYTES and served and self._downwash is None
                    and self._phys_options == 0)

    def _ensure_read_room(self) -> None:
        """Arrays a launch READS beside the state block it updates — the targets of the fused step, of computeControl — want
        the state's own 16 GiB window of device memory (placement.py; round 4's region probe, arena mode, G: computeControl 136 us
        with its targets there, 143 us with them one window on), and where a separate allocation falls is the memory
        manager's business.  So a large fleet's state block moves ONCE into a driver allocation with room for two target
        blocks right behind it: the same allocation is the same window (but for the 1-in-20 case that a window boundary runs
        through it, which the trials would show)."""
        if self._read_room is not None or self._chain_live or self._graph_made or not self.ctx.placement:
            return
        self._read_room = []
        nst, ntg = self.state.data.numel(), nat.NT * self.state.n_pad
        try:
            blk = _DriverBlock(self.ctx, (nst + 2 * ntg,)).tensor()
        except (MemoryError, RuntimeError):
            return
        self._move_state(blk[:nst].view(self.state.data.shape))
        blk[nst:].zero_()
        self._read_room = [blk[nst:nst + ntg], blk[nst + ntg:]]
        self.ctx.placement_log.append({"array": "state block + room for two target blocks", "bytes": 4 * (nst + 2 * ntg),
                                       "held_bytes": 4 * 2 * ntg, "placed": "one driver allocation"})

    def _take_read_room(self, numel: int):
        """One of the two target-sized blocks behind the state block (a flat fp32 tensor), or None."""
        self._ensure_read_room()
        if self._read_room and self._read_room[0].numel() == numel:
            return self._read_room.pop(0)
        return None

    def _place_obs_rows(self, shape) -> None:
        """A large fleet on the fast kernels (the Env.step launch writes the rows beside the state it updates in place): WHERE
        the rows lie is worth 10-15 % of that launch and is chosen by timing it."""
        self.materialize()
        self._ensure_read_room()                  # (the state block in its final place before anything is timed against it)
        before = self.ctx.query(nat.QUERY_GROUND_CONTACTS)
        echo = self._last_action.clone()          # the passes echo the (clipped) action buffer: put back below
        # Zero-sub-step passes change nothing on a quad fleet (the state is read and written back bit for bit); on
        # the morphing hexa the base-link / composite offset makes the round trip of the velocity round: a snapshot
        # of the state block is put back behind the passes.
        snap = None if (self.n_act == 4 and self._type_id is None) else self.state.data.clone()
        log = self.ctx.placement_log
        # (one allocation for everything that is written beside the state block: the rows, and behind them the
        # (n_act + 4) x n_pad floats a bound INDIControl writes — command, position error, yaw error; what suits the
        # one suits the other, and the controller need not search)
        n_rows, n_tail = shape[0] * shape[1], (self.n_act + 4) * self.state.n_pad
        flat = (n_rows + n_tail,)

        def split(block):
            self._obs_buf = block[:n_rows].view(shape)
            self._written_tail = block[n_rows:].view(self.n_act + 4, self.state.n_pad)
        # (never worse than no search: the plain allocation is timed too and kept when the walk's best is not faster —
        # on some boxes no block the driver hands out lies well, and PyTorch's may lie better)
        plain = torch.zeros(flat, dtype=torch.float32, device=self.ctx.device)
        t_plain = _event_timer(self._rows_trial, plain, 3)
        block = place_rows(self.ctx.device, flat, self._rows_trial, report=log, ctx=self.ctx,
                                     stride_bytes=STRIDE_BYTES)
        if log:
            log[-1]["plain_pass_us"] = round(t_plain, 1)
            if t_plain <= 1.005 * log[-1].get("chosen_pass_us", 0.0):
                plain.zero_()
                block = plain
                log[-1]["decided_by"] = "the plain allocation is as fast as the walk's best: kept"
        split(block)
        del plain, block
        # (When every candidate times alike there is nothing more to try.  Round 3 moved the state block to a fresh
        # allocation and walked again; round 4 tried one arena — state block and written arrays one 16 GiB window apart
        # in a single allocation, the layout round 4's region probe, arena mode shows to be the good one in a fresh process —
        # and measured it in the product: it never beat the walk's best in any of two dozen processes (142-179 us against
        # 140-158), so it is gone.  What did help is the state block's own move into a fresh driver allocation before
        # the walk: _ensure_read_room.)
        if snap is not None:
            self.state.data.copy_(snap)
        self._last_action.copy_(echo)
        # (a drone that sits on the ground is counted by every pass, also by these: not Env.steps)
        self._ground_trial += self.ctx.query(nat.QUERY_GROUND_CONTACTS) - before

    def _place_targets(self, targets, control_timestep) -> None:
        """Large homogeneous quad fleets, per-drone targets (READ beside the state block the fused step updates in place:
        the SAME region of device memory is the good case, placement.py).  Allocated right behind the state they usually land
        well — but not in every process: the same box gives the fused step at 154.6 or at 159 us by that alone.  The
        launch has no neutral form: snapshot of the state block, real passes on candidates holding a copy of the targets,
        snapshot back."""
        targets._placed = True
        old = targets.data
        if not (self.ctx.placement and isinstance(targets, Targets) and not targets.broadcast and targets.order is None
                and 4 * old.numel() >= MIN_BYTES and self._type_id is None and self.n_act == 4
                and self._downwash is None and self._phys_options == 0 and not self._chained_enabled
                and not self._graph_made):
            return
        self.materialize()
        snap, echo = self.state.data.clone(), self._last_action.clone()
        before = self.ctx.query(nat.QUERY_GROUND_CONTACTS)
        args = self.step_args(control_timestep)
        sview, tview, ref = self.state.view(), targets.view(), ctypes.byref(args)
        filled = set()

        def trial(c):
            if c.data_ptr() not in filled:           # (the first pass on a candidate is the untimed one)
                filled.add(c.data_ptr())
                c.copy_(old)
            tview.base = c.data_ptr()
            nat.check(self.ctx.lib.dsim_step(self.ctx.handle, self.ctx.stream_ptr(), self.NUM_DRONES, sview, tview, ref))
        room = self._take_read_room(old.numel())
        t_room = None
        if room is not None:                          # the block right behind the state, in the state's own allocation
            room = room.view(old.shape)
            sview = self.state.view()                 # (the state block may just have moved there)
            t_room = _event_timer(trial, room, 3)
        keep = place_rows(self.ctx.device, tuple(old.shape), trial, report=self.ctx.placement_log,
                                    label="per-drone targets", clearly=0.0, walk_bytes=4 << 30, ctx=self.ctx)
        if t_room is not None:
            rep = self.ctx.placement_log[-1]
            rep["behind_the_state_pass_us"] = round(t_room, 1)
            if t_room <= 1.01 * rep.get("chosen_pass_us", 0.0):
                keep = room
                rep["decided_by"] = "the block behind the state, in its allocation, is as fast as the walk's best: kept"
        keep.copy_(old)
        targets.data = keep
        self.state.data.copy_(snap)
        self._last_action.copy_(echo)
        self._ground_trial += self.ctx.query(nat.QUERY_GROUND_CONTACTS) - before
        self._fused_plan = self._fused_plan_dw = None

    def _rows_trial(self, rows: torch.Tensor) -> None:
        """One pass of the Env.step launch with ZERO physics sub-steps writing its rows to `rows`: the same kernel and
        memory streams, the state read and written back bit for bit (place_rows times it)."""
        args = self.step_args()
        args.phys_substeps = 0
        if self._caller_io:
            args.options |= nat.OPT_CALLER_IO
        args.action = self._action_buf.data_ptr()
        args.obs_out, args.obs_width = rows.data_ptr(), 16 + self.n_act
        nat.check(self.ctx.lib.dsim_physics(self.ctx.handle, self.ctx.stream_ptr(), self.NUM_DRONES,
                                            self.state.view(), self._last_action.data_ptr(), ctypes.byref(args)))
