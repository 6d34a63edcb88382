#!/usr/bin/env python3
"""Long flights on one GPU: the headline fleet (4 194 304 quads hovering, noise on), the same with five sub-steps, the
interleaved quad/hexa fleet and a config-5 shard, N steps each; everything must stay finite, hold its hover point and (hexas)
report no WLS failure.  usage: python tools/soak.py [--steps 20000] [--out FILE]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from dronesim_amd import _native as nat  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20000)
    ap.add_argument("--out", default=None)
    ap.add_argument("--only", default=None, help="comma-separated workload names")
    a = ap.parse_args()
    rows = []
    only = [w for w in (a.only or "").split(",") if w]
    for name, kw, steps in (("config2x1024", dict(n_fleet=4096, replicas=1024, substeps=1), a.steps),
                            ("config2x1024_sub5", dict(n_fleet=4096, replicas=1024, substeps=5), a.steps // 2),
                            ("mixed_quad_hexa", dict(n_fleet=4096, replicas=1024, substeps=1, mixed=True), a.steps // 2),
                            ("config5_shard", dict(n_fleet=65536, replicas=1, substeps=1, config5=True), a.steps),
                            # round 4: the reference-shaped two-call loop on the run kernels (hexa; interleaved, caller-order I/O;
                            # five sub-steps: the looped instances with the noise tables)
                            ("two_call_hexa", dict(n_fleet=4096, replicas=1024, substeps=1, hexa=True, two_call=True), a.steps // 4),
                            ("two_call_mixed_interleaved", dict(n_fleet=4096, replicas=1024, substeps=1, mixed=True, two_call=True), a.steps // 4),
                            ("two_call_mixed_interleaved_sub5", dict(n_fleet=4096, replicas=1024, substeps=5, mixed=True, two_call=True), a.steps // 8),
                            ("two_call_config5_shard", dict(n_fleet=65536, replicas=1, substeps=1, config5=True, two_call=True), a.steps // 2)):
        if only and name not in only:
            continue
        two_call = kw.pop("two_call", False)
        fl = bench.Fleet(kw.pop("n_fleet"), kw.pop("replicas"), 0, kw.pop("substeps"), "tile64", 1, **kw)
        if two_call:
            fl.make_two_call_loop()
        p0 = fl.env.state.raw_fields(0, 3).clone()
        t0 = time.perf_counter()
        for _ in range(steps):
            fl.step()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        st = fl.env.state.raw_fields(0, fl.env.state.n_fields)
        drift = (fl.env.state.raw_fields(0, 3) - p0).norm(dim=0)
        rows.append({"workload": name, "steps": steps, "seconds": round(el, 2), "us_per_step": round(el / steps * 1e6, 1),
                     "finite": bool(torch.isfinite(st).all()), "drift_from_start_median_m": round(float(drift.median()), 4),
                     "drift_from_start_max_m": round(float(drift.max()), 3), "speed_max_m_s": round(float(st[7:10].abs().max()), 3),
                     "ground_contacts": fl.env.ground_contacts(),
                     "wls_failures": fl.env.ctx.query(nat.QUERY_WLS_FAILURES), "wls_fallbacks": fl.env.ctx.query(nat.QUERY_WLS_FALLBACKS)})
        print(rows[-1], flush=True)
        fl.env.close()
        del fl
        torch.cuda.empty_cache()
    if a.out:
        with open(a.out, "w") as fh:
            json.dump(rows, fh, indent=1)


if __name__ == "__main__":
    main()
