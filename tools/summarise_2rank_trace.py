#!/usr/bin/env python3
"""profiles/r03_c5_2rank_gloo_* from gpurun_out/r03_final/kt_c5_2rank (tools/gpu_evidence_r03.sh c): per rank (= per
traced process) the step period, the kernel-busy time per step and the largest gap between two kernels of a step; a step
begins at a k_halo_pack launch.  Only the NEWEST two traces are used (gpurun merges, it never deletes)."""
import csv
import glob
import json
import os
import shutil
import statistics

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "r03_final", "kt_c5_2rank")
traces = sorted(glob.glob(os.path.join(src, "*", "*_kernel_trace.csv")), key=os.path.getmtime)
ranks = {}
picked = []
for f in reversed(traces):
    rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith(("k_", "void k_"))]
    if sum(1 for r in rows if r["Kernel_Name"].startswith("k_halo_pack")) < 50:
        continue                                      # the launcher / a child without a fleet
    picked.append((f, rows))
    if len(picked) == 2:
        break
for idx, (f, rows) in enumerate(sorted(picked, key=lambda p: p[0])):
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_halo_pack")]
    per, busy, gap, nk = [], [], [], []
    for a, b in zip(starts[20:-1], starts[21:]):          # (behind the warm-up and the first resizes)
        seg = rows[a:b]
        t0 = int(seg[0]["Start_Timestamp"])
        per.append((int(rows[b]["Start_Timestamp"]) - t0) / 1e3)
        busy.append(sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg) / 1e3)
        ends = [int(r["End_Timestamp"]) for r in seg]
        nxt = [int(r["Start_Timestamp"]) for r in seg[1:]] + [int(rows[b]["Start_Timestamp"])]
        gap.append(max(n - e for n, e in zip(nxt, ends)) / 1e3)
        nk.append(len(seg))
    ranks[f"rank{idx}"] = {"steps": len(per), "us_per_step_median": statistics.median(per),
                           "kernel_busy_us_per_step_median": statistics.median(busy),
                           "largest_gap_us_median_per_step": statistics.median(gap), "kernels_per_step": statistics.mean(nk)}
    ks = f.replace("_kernel_trace.csv", "_kernel_stats.csv")
    if os.path.exists(ks):
        shutil.copy(ks, os.path.join(ROOT, "profiles", f"r03_c5_2rank_gloo_rank{idx}_kernel_stats.csv"))
old = json.load(open(os.path.join(ROOT, "profiles", "r03_c5_2rank_gloo_summary.json")))
json.dump({"note": old["note"], "ranks": ranks}, open(os.path.join(ROOT, "profiles", "r03_c5_2rank_gloo_summary.json"), "w"), indent=1)
print(json.dumps(ranks, indent=1))
