import time, numpy as np, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
for name, spec in {"hover65536": (4096, 16, 1, False, 1), "wp65536": (65536, 1, 2, True, 1), "hover4096": (4096, 1, 5, False, 1)}.items():
    f = bench.Fleet(spec[0], spec[1], 0, spec[2], "tile64", 1, waypoints=spec[3], n_steps=spec[4])
    env = f.env
    for _ in range(50): f.step()
    torch.cuda.synchronize()
    hits = 0
    t0 = time.perf_counter()
    for _ in range(2000):
        p = env._fused_plan
        hits += int(p is not None and p[5] is f.tgt and p[6] == env._targets_ptrs(f.tgt))
        f.step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(name, "us/step", round(el / 2000 * 1e6, 2), "plan hits", hits, "chain_ok", env._chain_ok, type(f.tgt).__name__)
