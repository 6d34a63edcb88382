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

The search can only FIND a place: on one box of the round no candidate for the rows within 16 GiB was a good one.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch

MIN_BYTES = 64 << 20                          # arrays at least this large, or they are allocated plainly
WALK_BYTES = 16 << 30                         # candidates held at once while searching (transient); good places mostly turn up within 2-3 GiB
CLEARLY = 0.93                                # one candidate this much faster than another: the two cases are apart, stop


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


def place_rows(device, shape, trial: Callable[[torch.Tensor], None], passes: int = 3, report: Optional[list] = None,
               label: str = "observation rows", walk_bytes: int = WALK_BYTES, timer=None,
               clearly: float = CLEARLY) -> torch.Tensor:
    """A zeroed fp32 array of `shape`.  `trial(array)` enqueues ONE pass of the real kernel writing its output to `array`
    (Env.step: a zero-sub-step pass; computeControl: a real pass, the caller restores the state afterwards).  Candidates
    are allocated one after the other and all held, so that the walk moves through device memory; each is timed over
    `passes` passes behind one untimed pass; the walk ends as soon as one candidate is clearly faster than the first (it
    is kept) or when `walk_bytes` are held (the fastest is kept).  (A candidate clearly SLOWER than the first does not end it:
    there are more than two levels — 154 / 158 / 164 us for the fused step by where its targets lie — and the first may be
    the middle one.)  `timer(trial, candidate, passes)`: the clock (tests).
    `clearly`: the ratio that counts as clear; 0: no early end, the whole walk."""
    nbytes = 4
    for d in shape:
        nbytes *= int(d)
    cands, times = [], []
    chosen, decided = None, ""
    torch.cuda.empty_cache()                  # candidates from whole device blocks, not from pieces the allocator has cached
    while chosen is None and (len(cands) + 1) * nbytes <= max(walk_bytes, 2 * nbytes):
        try:
            c = torch.empty(tuple(shape), dtype=torch.float32, device=device)
        except torch.cuda.OutOfMemoryError:
            break
        cands.append(c)
        times.append((timer or _event_timer)(trial, c, passes))
        if clearly > 0.0 and times[-1] < clearly * times[0]:
            chosen, decided = len(times) - 1, "a candidate clearly faster than the first"
    if chosen is None:
        chosen = min(range(len(times)), key=times.__getitem__)
        decided = "the fastest of the walk" if times[chosen] < CLEARLY * max(times) else "all alike"
    keep = cands[chosen]
    if report is not None:
        shown = times if len(times) <= 12 else times[:4] + times[-4:]
        report.append({"array": label, "bytes": nbytes, "candidates": len(cands), "chosen": chosen,
                       "decided_by": decided, "chosen_pass_us": round(times[chosen], 1), "first_pass_us": round(times[0], 1),
                       "pass_us" if len(times) <= 12 else "pass_us_first4_last4": [round(t, 1) for t in shown]})
    c = None
    del cands
    torch.cuda.empty_cache()                  # what was not kept goes back to the device
    keep.zero_()
    return keep
