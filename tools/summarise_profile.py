#!/usr/bin/env python3
"""Summarise gpurun_out/prof_<tag>/ (rocprofv3 CSVs from tools/profile_gpu.sh) into
profiles/<tag>_kernel_stats.csv, profiles/<tag>_summary.json and profiles/traffic.json.

HBM bytes follow MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are in KiB and come from
separate --pmc passes; on gfx950 FETCH_SIZE reports exactly half the bytes of a coalesced
streaming read, so hbm_read = 2 * FETCH_SIZE * 1024.  Calibration on a known byte count in this
code's own access pattern (one dword per lane, SoA): k_reset reads exactly 24 B/drone
(init_pos + init_rpy), and its corrected FETCH_SIZE is checked against that below.
usage: python tools/summarise_profile.py <tag> <workload> <layout> <drones_per_launch> [bytes_per_drone_step=232]
(traffic.json, which bench.py reads, is only rewritten for the default workload config2x1024)"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag, workload, layout, n = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
BPD = int(sys.argv[5]) if len(sys.argv) > 5 else 232
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
ks = glob.glob(os.path.join(src, "kt", "*", "*_kernel_stats.csv"))[0]
shutil.copy(ks, os.path.join(dst, f"{tag}_kernel_stats.csv"))
stats = {r["Name"]: r for r in csv.DictReader(open(ks))}


def counter(sub, name):
    f = glob.glob(os.path.join(src, sub, "*", "*_counter_collection.csv"))[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == name:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


fetch, write = counter("fetch", "FETCH_SIZE"), counter("write", "WRITE_SIZE")
out = {"tag": tag, "workload": workload, "layout": layout, "drones_per_launch": n, "kernels": {}}
for k, r in stats.items():
    if not (k.startswith("void k_") or k.startswith("k_")):
        continue
    e = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "min_us": float(r["MinNs"]) / 1e3,
         "max_us": float(r["MaxNs"]) / 1e3}
    if k in fetch:
        e["FETCH_SIZE_KiB_avg"] = fetch[k]
        e["hbm_read_bytes"] = 2 * fetch[k] * 1024
    if k in write:
        e["WRITE_SIZE_KiB_avg"] = write[k]
        e["hbm_write_bytes"] = write[k] * 1024
    out["kernels"][k] = e
step = max((k for k in out["kernels"] if "k_step" in k), key=lambda k: out["kernels"][k]["calls"], default=None)
if step:
    e = out["kernels"][step]
    tot = e.get("hbm_read_bytes", 0) + e.get("hbm_write_bytes", 0)
    out["dominant_kernel"] = step
    out["hbm_bytes_per_launch"] = tot
    out["algorithmic_bytes_per_launch"] = BPD * n
    out["traffic_over_algorithmic"] = tot / (BPD * n)
    out["achieved_GBps_profiled"] = BPD * n / (e["avg_us"] * 1e-6) / 1e9
rs = out["kernels"].get("k_reset(ResetK)")
if rs and "hbm_read_bytes" in rs:
    out["calibration"] = {"kernel": "k_reset", "known_read_bytes": 24 * n,
                          "corrected_FETCH_bytes": rs["hbm_read_bytes"],
                          "ratio": rs["hbm_read_bytes"] / (24 * n)}
json.dump(out, open(os.path.join(dst, f"{tag}_summary.json"), "w"), indent=1)
if workload == "config2x1024":
    json.dump({"workload": workload, "layout": layout, "hbm_bytes_per_launch": out.get("hbm_bytes_per_launch"),
               "source": f"profiles/{tag}_summary.json"}, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
