#!/usr/bin/env python3
"""Timeline of config 5's neighbour query from in-kernel stamps (dev helper; needs the stamped copy of the library:
tools/variants/make_stamped_query.py).  Prints when the workgroups (one per 5 m cell) start, how long their first round
trip, their set-up (second round trip, ordering, banding) and their pair loops take, when they end, and how the work
spreads over the compute units.  usage: python tools/c5_query_timeline.py build/libdsim_stamp.so"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dronesim_amd import _native as nat  # noqa: E402

nat.load(sys.argv[1])
from dronesim_amd import fleet, params  # noqa: E402
from dronesim_amd.downwash import Downwash  # noqa: E402


def main():
    n = 65536
    rng = np.random.default_rng(1234)
    xyz = np.stack([rng.uniform(0, 128, n), rng.uniform(0, 512, n), rng.uniform(0.5, 20.5, n)], 1)
    types = [params.builtin_type("robobee"), params.builtin_type("hexa_6DOF")]
    ctx = fleet.Context(types)
    st = fleet.FleetState(ctx, n, "tile64")
    rigid = np.zeros((n, 13)); rigid[:, :3] = xyz; rigid[:, 6] = 1.0
    st.load_aos(rigid, np.zeros((n, 13)))
    tid = torch.zeros(st.n_pad, dtype=torch.uint8, device=ctx.device)
    tid[:n] = torch.from_numpy((np.arange(n) % 2).astype(np.uint8))
    dw = Downwash(ctx, st, tid)
    dw.compute()
    g = dw._last
    grid = g.nx * g.ny + 16
    dw.pair_counter = torch.zeros((8 + 6 * grid,), dtype=torch.int64, device=ctx.device)
    dw._single = None
    for _ in range(5):
        dw.compute()
    torch.cuda.synchronize()
    dw.pair_counter.zero_()
    dw.compute()
    torch.cuda.synchronize()
    s = dw.pair_counter[8:].cpu().numpy().reshape(grid, 6).astype(np.int64)
    tick = 1e-2        # wall_clock64: 100 MHz -> us
    live = s[:, 2] > 0
    t0 = s[s[:, 0] > 0, 0].min()
    end = np.maximum(s[:, 3], s[:, 4])
    print(f"grid {g.nx} x {g.ny} = {g.nx * g.ny} cells, {int(live.sum())} with receivers on the banded path, {int((s[:, 0] > 0).sum())} workgroups stamped")
    q = lambda a: "min %6.2f  p10 %6.2f  median %6.2f  p90 %6.2f  max %6.2f" % tuple(np.percentile(a, [0, 10, 50, 90, 100]))
    print("start (us after the first)     ", q((s[live, 0] - t0) * tick))
    print("first round trip (counts)      ", q((s[live, 1] - s[live, 0]) * tick))
    print("set-up (tile, order, bands)    ", q((s[live, 2] - s[live, 1]) * tick))
    print("pair loops (to the later wave) ", q((end[live] - s[live, 2]) * tick))
    print("life of a workgroup            ", q((end[live] - s[live, 0]) * tick))
    print("end (us after the first start) ", q((end[live] - t0) * tick))
    print("span of the launch              %.2f us" % ((end[live].max() - t0) * tick))
    hw = s[live, 5]
    cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7            # gfx9 HW_ID: CU_ID [11:8], SH_ID [12], SE_ID [15:13]
    key = se * 32 + sh * 16 + cu
    # (the XCC is not in HW_ID on this part: workgroups go round-robin over the eight XCDs by index)
    xcd = np.flatnonzero(live) % 8
    per = {}
    for k, x, e, b in zip(key, xcd, end[live], s[live, 0]):
        per.setdefault((int(x), int(k)), []).append(((b - t0) * tick, (e - t0) * tick))
    ends = np.array([max(e for _, e in v) for v in per.values()])
    cnts = np.array([len(v) for v in per.values()])
    print(f"{len(per)} (XCD, SE/SH/CU) slots seen; cells per slot min {cnts.min()} median {int(np.median(cnts))} max {cnts.max()}; "
          f"last end per slot: {q(ends)}")
    late = (s[live, 0] - t0) * tick > 3.0
    print(f"workgroups that start more than 3 us after the first: {int(late.sum())} of {int(live.sum())}; their start: "
          + (q((s[live, 0][late] - t0) * tick) if late.any() else "-"))
    idx = np.flatnonzero(live)[late]
    print("late workgroups' block indices:", idx[:40].tolist())
    # how many workgroups of the same slot were alive when a late one started
    alive = []
    for i in idx:
        kx = (int(i % 8), int(((s[i, 5] >> 13) & 7) * 32 + ((s[i, 5] >> 12) & 1) * 16 + ((s[i, 5] >> 8) & 0xF)))
        b0 = (s[i, 0] - t0) * tick
        alive.append(sum(1 for (b, e) in per[kx] if b < b0 < e))
    print("workgroups alive on the same slot when a late one starts:", alive[:40])
    # all stamped workgroups (also the empty cells): how long the ones without receivers live
    empty = (s[:, 0] > 0) & ~live
    print(f"{int(empty.sum())} workgroups without a banded pass (empty cells, overflow groups): they leave after",
          q((s[empty, 1] - s[empty, 0]) * tick) if (s[empty, 1] > 0).any() else "(no second stamp)")
    print("raw HW_ID of the first eight workgroups:", [hex(int(x)) for x in s[:8, 5]])
    ctx.close()


if __name__ == "__main__":
    main()
