"""Debug aid: per-region host and device times of one bench variant (GPU)."""
import sys, time
sys.path.insert(0, ".")
import torch
import bench
name = sys.argv[1] if len(sys.argv) > 1 else "hexa"
a = bench.parse(["--workload", name, "--no-also", "--no-cpu-baseline"])
from dronesim_amd import sharding
for trial in range(3):
    fl = bench.Fleet(4096, 1024, 0, 1, "tile64", 1, hexa=name == "hexa", mixed=name == "mixed")
    for _ in range(10):
        fl.step()
    for reg in range(6):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); e0.record()
        ts = []
        for _ in range(100):
            t1 = time.perf_counter(); fl.step(); ts.append(time.perf_counter() - t1)
        e1.record(); torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        print(f"trial {trial} region {reg}: wall {wall*1e3:7.2f} ms  device {e0.elapsed_time(e1):7.2f} ms  slowest call {max(ts)*1e3:7.3f} ms at {ts.index(max(ts))}")
    fl.env.close(); del fl
