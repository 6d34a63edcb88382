#!/usr/bin/env python3
"""Summarise gpurun_out/<tag>/ (tools/profile_sq.sh) into profiles/<tag>_summary.json: per kernel the average duration,
the SQ counters as fractions of SQ_WAVE_CYCLES and the HBM bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE, KiB, separate
passes: MI355X_MICROARCH.md "HBM"), against the ALGORITHMIC bytes of THAT kernel.
usage: python tools/summarise_sq.py <tag> <drones_per_launch> <bytes_per_drone_step> [kernel-substring] [kernel=bytes ...]
<bytes_per_drone_step> is the budget of the workload's step kernel (k_step_*); the kernels of the two-call loop have their own
(KERNEL_BYTES below, or kernel=bytes on the command line) - a kernel without a budget gets its measured traffic and no rate: up
to round 5 every kernel of a loop was divided into the LOOP's bytes and the summaries printed 10-12 TB/s for single kernels.
       python tools/summarise_sq.py --fix profiles/<tag>_summary.json      (recompute the rates of a stored summary in place)"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

# algorithmic bytes per drone of the kernels that are NOT the workload's step kernel (floats x 4)
KERNEL_BYTES = [
    ("k_physics_fast<", lambda k: 216 if _targ(k, 2) else 136),    # 13 + 4 read, 13 + 4 echo (+ 20 row floats) written
    ("k_control_fast<", lambda k: 212),                            # 13 + 11 + 10 read, 11 + 4 + 3 + 1 written
    ("k_adaptor_fast<", lambda k: 304),                            # 24 + 4 read, 24 + 4 + 20 written
    ("k_dyn<true", lambda k: 256), ("k_dyn<false", lambda k: 240 if _targ(k, 2) else 160),
    ("k_wls_fallback", lambda k: None), ("k_dw_", lambda k: None), ("k_reset", lambda k: None), ("k_observe", lambda k: None),
]


def _targ(kernel, i):
    """i-th template argument of a kernel name as a bool"""
    try:
        return kernel[kernel.index("<") + 1:kernel.index(">")].split(",")[i].strip() == "true"
    except (ValueError, IndexError):
        return False


def kernel_budget(kernel, step_bpd, extra):
    name = kernel[5:] if kernel.startswith("void ") else kernel
    for sub, b in extra.items():
        if sub in name:
            return b
    for prefix, f in KERNEL_BYTES:
        if name.startswith(prefix):
            return f(name)
    return step_bpd if name.startswith("k_step") else None


def rates(e, bytes_per_drone, n):
    e.pop("achieved_GBps_profiled", None)
    e.pop("traffic_over_algorithmic", None)
    e["algorithmic_bytes_per_drone"] = bytes_per_drone
    if bytes_per_drone is None:
        return
    if "hbm_bytes_per_launch" in e:
        e["traffic_over_algorithmic"] = e["hbm_bytes_per_launch"] / (bytes_per_drone * n)
    e["achieved_GBps_profiled"] = bytes_per_drone * n / (e["avg_us"] * 1e-6) / 1e9
    e["hbm_frac_profiled"] = e["achieved_GBps_profiled"] / 8000.0


if len(sys.argv) > 1 and sys.argv[1] == "--fix":
    for path in sys.argv[2:]:
        d = json.load(open(path))
        for k, e in d["kernels"].items():
            # (a stored two-call summary carries the LOOP's bytes: they are no kernel's budget, the step kernel's included)
            rates(e, kernel_budget(k, None if "two_call" in str(d.get("tag", path)) else d.get("bytes_per_drone_step"), {}), d["drones_per_launch"])
        d["note"] = ("rates recomputed per kernel from the stored durations (tools/summarise_sq.py --fix): bytes_per_drone_step is the "
                     "LOOP's budget, not one kernel's; kernels without a budget of their own carry their measured traffic only")
        json.dump(d, open(path, "w"), indent=1)
        print(path, {k[:40]: round(e.get("achieved_GBps_profiled") or 0) for k, e in d["kernels"].items()})
    sys.exit(0)

tag, n, bpd = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
rest = sys.argv[4:]
extra = {a.split("=")[0]: float(a.split("=")[1]) for a in rest if "=" in a}
rest = [a for a in rest if "=" not in a]
want = rest[0] if rest else "k_"
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
    rates(e, kernel_budget(k, bpd, extra), n)
    out["kernels"][k] = e
meta = os.path.join(src, "meta.json")          # (tools/profile_sq.sh: the profiled command and its timing protocol)
if os.path.exists(meta):
    out.update(json.load(open(meta)))
json.dump(out, open(os.path.join(dst, f"{tag}_summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
