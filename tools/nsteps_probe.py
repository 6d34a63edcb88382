"""Runs ON the GPU box: BASELINE's literal sizes with 32 Env.steps per launch (dsim_step_args.n_steps; the EXT instances of
k_step_fast), us per launch and per Env.step.  usage: python tools/nsteps_probe.py [lib]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
import bench  # noqa: E402

if len(sys.argv) > 1:
    from dronesim_amd import _native
    _native.load(sys.argv[1])
for name, kw in {"configs[1] 4096 quads x 5 sub-steps": dict(n=4096, sub=5, wp=False),
                 "configs[2] 65536 quads on the waypoint table x 2 sub-steps": dict(n=65536, sub=2, wp=True)}.items():
    for ns in (1, 32):
        fl = bench.Fleet(kw["n"], 1, 0, kw["sub"], "tile64", 1, waypoints=kw["wp"], n_steps=ns)
        for _ in range(20):
            fl.step()
        torch.cuda.synchronize()
        k = 2000 // ns + 20
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(k):
            fl.step()
        e1.record()
        e1.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / k
        print("%-62s n_steps %2d: %8.2f us per launch, %6.2f us per Env.step, %.3e drone-steps/s" % (name, ns, us, us / ns, kw["n"] * ns / us * 1e6))
        fl.env.close()
