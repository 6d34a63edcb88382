#!/usr/bin/env python3
"""Summarise gpurun_out/<tag>/ (tools/profile_sq.sh) into profiles/<tag>_summary.json: per kernel the average duration,
the SQ counters as fractions of SQ_WAVE_CYCLES and the HBM bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE, KiB, separate
passes: MI355X_MICROARCH.md "HBM"), against the algorithmic bytes given on the command line.
usage: python tools/summarise_sq.py <tag> <drones_per_launch> <bytes_per_drone_step> [kernel-substring]"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag, n, bpd = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
want = sys.argv[4] if len(sys.argv) > 4 else "k_"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles")
ks = glob.glob(os.path.join(src, "kt", "*", "*_kernel_stats.csv"))[0]
shutil.copy(ks, os.path.join(dst, f"{tag}_kernel_stats.csv"))
stats = {r["Name"]: r for r in csv.DictReader(open(ks))}


def counters(sub):
    fs = glob.glob(os.path.join(src, sub, "*", "*_counter_collection.csv"))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in fs:
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}


sq = {}
for sub in ("sq1", "sq2"):
    for k, d in counters(sub).items():
        sq.setdefault(k, {}).update({c: v for c, v in d.items() if c != "SQ_WAVE_CYCLES" or "SQ_WAVE_CYCLES" not in sq.get(k, {})})
fetch, write = counters("fetch"), counters("write")
out = {"tag": tag, "drones_per_launch": n, "bytes_per_drone_step": bpd, "kernels": {}}
for k, r in stats.items():
    if want not in k or not (k.startswith("void k_") or k.startswith("k_")):
        continue
    e = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "min_us": float(r["MinNs"]) / 1e3}
    if k in sq and sq[k].get("SQ_WAVE_CYCLES"):
        wc = sq[k]["SQ_WAVE_CYCLES"]
        e["SQ_WAVE_CYCLES"] = wc
        e["fraction_of_wave_cycles"] = {c: round(v / wc, 4) for c, v in sorted(sq[k].items()) if c not in ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES")}
        e["per_wave_instructions"] = {c: round(v, 1) for c, v in sorted(sq[k].items()) if c.startswith("SQ_INSTS")}
    if k in fetch and "FETCH_SIZE" in fetch[k]:
        e["hbm_read_bytes"] = 2 * fetch[k]["FETCH_SIZE"] * 1024
    if k in write and "WRITE_SIZE" in write[k]:
        e["hbm_write_bytes"] = write[k]["WRITE_SIZE"] * 1024
    if "hbm_read_bytes" in e and "hbm_write_bytes" in e:
        e["hbm_bytes_per_launch"] = e["hbm_read_bytes"] + e["hbm_write_bytes"]
        e["traffic_over_algorithmic"] = e["hbm_bytes_per_launch"] / (bpd * n)
        e["achieved_GBps_profiled"] = bpd * n / (e["avg_us"] * 1e-6) / 1e9
    out["kernels"][k] = e
json.dump(out, open(os.path.join(dst, f"{tag}_summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
