"""Where the arrays a large fleet streams beside its state block lie in HBM.

Measured on MI355X (tools/membench.hip --bigsweep / --pairs / --regions, tools/placement_probe.py, tools/arena_probe.py;
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
error, yaw error: 32 bytes per drone; `tools/placement_probe_ctrl.py`: 144.7 us as allocated, 136 us with the outputs
elsewhere).  A controller bound to an env takes the 8 x n_pad floats the env left behind its placed rows (one allocation,
one search); one without an env searches for itself: that launch has no neutral form, so it takes a snapshot of the state
block, times real passes on the candidates and puts the snapshot back.

The per-drone targets of the fused step are READ beside the state, and there the SAME region is the good case: 154, 158 or
164 us per launch by where they lie (tools/state_probe.py), and the first candidate — right behind the state — may be the
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
on this round's boxes, where a striding walk finds 143 us from its third candidate on (tools/region_probe.py,
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
    the walk then covers the budget with fewer candidates.  Round 4, tools/region_probe.py: 36 back-to-back 470 MB candidates
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

