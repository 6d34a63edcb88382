#!/usr/bin/env python3
"""Is config 5's neighbour query paced by the MOST LOADED compute unit?  (dev helper)  The 65 536-drone shard's grid is
2 621 cells = one workgroup each, all resident at once (about ten per CU): the launch ends when the CU with the largest sum
of work ends.  Same number of drones, same density, the cell populations (a) Poisson as in the config's uniform random world,
(b) exactly 25 per 5 m cell (a 1 m lattice, heights random as in (a)).  Run under a kernel trace: tools/kt_py.sh."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dronesim_amd import fleet, params  # noqa: E402
from dronesim_amd.downwash import Downwash  # noqa: E402


def run(label, xyz, reps=200):
    n = xyz.shape[0]
    types = [params.builtin_type("robobee"), params.builtin_type("hexa_6DOF")]
    ctx = fleet.Context(types)
    st = fleet.FleetState(ctx, n, "tile64")
    rigid = np.zeros((n, 13)); rigid[:, :3] = xyz; rigid[:, 6] = 1.0
    st.load_aos(rigid, np.zeros((n, 13)))
    tid = torch.zeros(st.n_pad, dtype=torch.uint8, device=ctx.device)
    tid[:n] = torch.from_numpy((np.arange(n) % 2).astype(np.uint8))
    dw = Downwash(ctx, st, tid)
    for _ in range(10):
        dw.compute()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        dw.compute()
    e1.record(); e1.synchronize()
    print(f"{label:<28s} bin + query {e0.elapsed_time(e1) * 1e3 / reps:7.2f} us   cell {dw._last.cell} grid {dw._last.nx} x {dw._last.ny}", flush=True)
    ctx.close()


def main():
    n = 65536
    rng = np.random.default_rng(1234)
    z = rng.uniform(0.5, 20.5, n)
    run("uniform random (the config)", np.stack([rng.uniform(0, 128, n), rng.uniform(0, 512, n), z], 1))
    ij = np.arange(n)
    lat = np.stack([(ij % 128) + 0.5 + rng.uniform(-0.3, 0.3, n), (ij // 128) + 0.5 + rng.uniform(-0.3, 0.3, n), z], 1)
    run("25 per cell exactly", lat)


if __name__ == "__main__":
    main()
