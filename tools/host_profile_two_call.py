#!/usr/bin/env python3
"""Dev probe (GPU box): where the HOST time of one iteration of the reference-shaped loop goes at BASELINE's literal size
(4 096 quads x 5 sub-steps: env.step(cmd) then ctrl.computeControlFromState).  usage: python tools/host_profile_two_call.py [hexa]"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402

hexa = len(sys.argv) > 1 and sys.argv[1] == "hexa"
fl = bench.Fleet(4096, 1, 0, 5, "tile64", 1, hexa=hexa)
fl.make_two_call_loop()
for _ in range(200):
    fl.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2000):
    fl.step()
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"host enqueue {t_host / 2000 * 1e6:.1f} us/iteration, with the device drained {t_all / 2000 * 1e6:.1f} us/iteration")
pr = cProfile.Profile()
pr.enable()
for _ in range(2000):
    fl.step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
