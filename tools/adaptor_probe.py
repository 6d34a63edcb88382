#!/usr/bin/env python3
"""Dev probe: Env.step of the action-adaptor envs (VelocityAviary / RPYTAviary) on 4 194 304 quads, microseconds per step.
usage: python tools/adaptor_probe.py [steps]"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dronesim_amd.envs import RPYTAviary, VelocityAviary  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    n = 4096 * 1024
    ij = np.arange(n) % 4096
    xyz = np.stack([(ij % 64) * 1.0, (ij // 64) * 1.0, np.full(n, 0.5)], 1)
    out = {}
    for cls in (VelocityAviary, RPYTAviary):
        env = cls(["robobee"], n, initial_xyzs=xyz, aggregate_phy_steps=1, noise_seed=1, dict_io=False, layout="tile64")
        if cls is VelocityAviary:
            act = torch.tensor([1.0, 0.0, 0.2, 0.5], device=env.ctx.device).repeat(n, 1)
        else:
            act = torch.tensor([0.0, 0.0, 0.0, 0.0], device=env.ctx.device).repeat(n, 1)
            act[:, 3] = 9.81 * env.types[0].mass
        for _ in range(10):
            env.step(act)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            env.step(act)
        e1.record()
        e1.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / steps
        out[cls.__name__] = {"us_per_env_step": round(us, 2), "drones": n, "drone_steps_per_s": round(n / us * 1e6, -6)}
        print(cls.__name__, out[cls.__name__], flush=True)
        env.close()
        del env
        torch.cuda.empty_cache()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
