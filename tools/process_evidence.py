#!/usr/bin/env python3
"""Turns what tools/gpu_evidence.sh left under gpurun_out/ into the files under profiles/ (run in the build container
after the gpurun calls): per tag the newest trace of every pass is kept, tools/summarise_sq.py writes
profiles/<tag>_summary.json + <tag>_kernel_stats.csv, traffic.json follows r06_main."""
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAGS = {"r06_main": (4194304, 232), "r06_mixed": (4194304, 241), "r06_hexa": (4194304, 248), "r06_sub5": (4194304, 232), "r06_hexa_sub5": (4194304, 248),
        "r06_c5": (65536, 241), "r06_dyn": (4194304, 256), "r06_dyn_sub5": (4194304, 256), "r06_two_call_quad": (4194304, 428), "r06_two_call_hexa": (4194304, 476),
        "r06_two_call_mixed": (4194304, 468), "r06_two_call_config5": (65536, 480)}
for tag, (n, b) in TAGS.items():
    src = os.path.join(ROOT, "gpurun_out", tag)
    if not os.path.isdir(os.path.join(src, "kt")):
        continue
    for sub in ("kt", "sq1", "sq2", "fetch", "write"):
        infos = sorted(glob.glob(os.path.join(src, sub, "*", "*_agent_info.csv")), key=os.path.getmtime)
        for old in infos[:-1]:                       # gpurun merges, it never deletes: older runs' files stay behind
            for f in glob.glob(old.replace("_agent_info.csv", "_*")):
                os.remove(f)
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "summarise_sq.py"), tag, str(n), str(b)], check=True,
                   stdout=subprocess.DEVNULL)
    d = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_summary.json")))
    for k, v in d["kernels"].items():
        if v["calls"] > 10:
            print(f"{tag:24s} {k[:58]:58s} {v['calls']:5d} {v['avg_us']:8.1f} us  wait {v['fraction_of_wave_cycles'].get('SQ_WAIT_ANY')}"
                  f"  traffic x{v.get('traffic_over_algorithmic', 0):.3f}")
    if tag == "r06_main":
        k = [v for v in d["kernels"].values() if v["calls"] > 10][0]
        json.dump({"workload": "config2x1024", "layout": "tile64", "hbm_bytes_per_launch": k["hbm_bytes_per_launch"],
                   "source": "profiles/r06_main_summary.json", "hbm_read_bytes": k["hbm_read_bytes"],
                   "hbm_write_bytes": k["hbm_write_bytes"]}, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
