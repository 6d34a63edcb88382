"""Runs ON the GPU box: how long does a launch take as a function of how long the device has been under load?

The vector-heavy kernels (five sub-steps per Env.step) start at one duration, pass through a throttled episode a few
milliseconds into a busy stretch and settle at a third (profiles/r05_clock_*.txt): the device's power management, not the
kernel.  This probe launches one workload back to back for `seconds` and times batches of 10 launches with events on the launch
stream.  (The device's own clock / power readings are not used: on the pool's boxes the cards an ordinary user can read under
/sys are not the leased device.)

usage: python tools/clock_probe.py [--workload config2x1024|hexa|mixed|dyn] [--substeps 5] [--seconds 3] [--idle-ms 0]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="config2x1024")
    ap.add_argument("--substeps", type=int, default=5)
    ap.add_argument("--seconds", type=float, default=3.0)
    ap.add_argument("--batch", type=int, default=10)
    ap.add_argument("--idle-ms", type=float, default=0.0, help="sleep this long (device idle) every 200 batches")
    ap.add_argument("--lib", default=None)
    a = ap.parse_args()
    import torch
    import bench
    if a.lib:
        from dronesim_amd import _native
        _native.load(a.lib)                 # an A/B build of the same ABI, bound first
    dev = torch.device("cuda:0")
    kw = dict(hexa=a.workload == "hexa", mixed=a.workload == "mixed", dyn=a.workload == "dyn")
    fl = bench.Fleet(4096, 1024, 0, a.substeps, "tile64", 1, **kw)
    torch.cuda.synchronize()
    time.sleep(0.5)                         # the device idle for half a second in front of the run
    series = []
    t_begin = time.perf_counter()
    ev = []
    n_batches = 0
    while time.perf_counter() - t_begin < a.seconds:
        # 50 batches are enqueued at a time so that the host never lets the queue run dry
        for _ in range(50):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.batch):
                fl.step()
            e1.record()
            ev.append((e0, e1))
            n_batches += 1
        torch.cuda.synchronize()
        now = time.perf_counter() - t_begin
        for e0, e1 in ev:
            series.append(e0.elapsed_time(e1) * 1e3 / a.batch)
        ev = []
        if a.idle_ms > 0 and n_batches % 200 == 0:
            time.sleep(a.idle_ms * 1e-3)
    print("workload %s substeps %d: %d batches of %d launches in %.2f s" % (a.workload, a.substeps, len(series), a.batch, time.perf_counter() - t_begin))
    print("first 40 batches (us per launch): " + " ".join("%.0f" % s for s in series[:40]))
    step = max(1, len(series) // 60)
    print("every %d-th batch: " % step + " ".join("%.0f" % s for s in series[::step]))
    import numpy as np
    s = np.array(series)
    q = len(s) // 4
    print("mean of quarters: " + " ".join("%.1f" % s[i * q:(i + 1) * q].mean() for i in range(4)) + "   min %.1f  median %.1f  max %.1f" % (s.min(), np.median(s), s.max()))
    fl.env.close()


if __name__ == "__main__":
    main()
