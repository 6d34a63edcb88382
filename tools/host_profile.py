#!/usr/bin/env python3
"""Dev probe (GPU box): where the HOST time of one config-5 step goes (cProfile over 400 steps of bench.Fleet.step with
the mirrored synthetic neighbour, or alone).  usage: python tools/host_profile.py [alone|mirror]"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "mirror"
dist = bench.MirrorDist(128.0) if mode == "mirror" else None
fl = bench.Fleet(65536, 1, 0, 1, "tile64", 1, config5=True, dist=dist, rank=0)
for _ in range(50):
    fl.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(24):                       # (an empty queue: nothing pushes back on the host for the first steps)
    fl.step()
t_pure = time.perf_counter() - t0
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(400):
    fl.step()
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"{mode}: host alone {t_pure / 24 * 1e6:.1f} us/step (first 24 steps on an empty queue); host enqueue {t_host / 400 * 1e6:.1f} "
      f"us/step, with the device drained {t_all / 400 * 1e6:.1f} us/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(400):
    fl.step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(22)
