#!/usr/bin/env python3
"""Eager stepping against hipGraph replay for single-rank fleets WITH the neighbour-downwash term (capture_fused):
what the graph removes is the Python side of three dependent launches per step.
usage: python tools/graph_dw_probe.py [--out FILE]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dronesim_amd.envs import CtrlAviary, Physics  # noqa: E402
from dronesim_amd.fleet import Targets  # noqa: E402


def fleet(n, density, mixed):
    rng = np.random.default_rng(5)
    side = (n / density) ** 0.5
    xyz = np.stack([rng.uniform(0, side, n), rng.uniform(0, side, n), rng.uniform(1, 9, n)], 1)
    kw = dict(type_ids=(np.arange(n) % 2).astype(np.uint8)) if mixed else {}
    env = CtrlAviary(["robobee", "hexa_6DOF"] if mixed else ["robobee"], n, initial_xyzs=xyz, physics=Physics.PYB_DW,
                     noise_seed=1, dict_io=False, aggregate_phy_steps=1, **kw)
    tg = Targets(env.ctx, n)
    tg.set(pos=(xyz + 0.2).T.astype(np.float32), yaw=0.1)
    return env, tg


def timed(fn, steps_per_call, calls):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(calls):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (calls * steps_per_call) * 1e6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    rows = []
    for n, density, mixed in ((4096, 1.0, False), (4096, 1.0, True), (16384, 1.0, True), (65536, 1.0, True)):
        env, tg = fleet(n, density, mixed)
        for _ in range(20):
            env.step_fused(tg)
        eager = timed(lambda: env.step_fused(tg), 1, 400)
        g = env.capture_fused(tg, steps=32)
        graph = timed(g.replay, 32, 40)
        rows.append({"drones": n, "mixed": mixed, "eager_us_per_step": round(eager, 2), "graph32_us_per_step": round(graph, 2),
                     "wls_failures": env.ctx.query(1) if mixed else 0})
        print(rows[-1], flush=True)
        env.close()
    if a.out:
        with open(a.out, "w") as fh:
            json.dump(rows, fh, indent=1)


if __name__ == "__main__":
    main()
