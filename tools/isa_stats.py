#!/usr/bin/env python3
"""Static instruction mix per kernel from the gfx950 assembly of dsim_api.hip (dev helper).
usage: python tools/isa_stats.py [kernel-name-substring]"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "dronesim_amd", "csrc", "dsim_api.hip")
out = os.path.join(tempfile.gettempdir(), "dsim_isa.s")
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fno-slp-vectorize", "-S", "-o", out,
                       "--cuda-device-only", src], stderr=subprocess.DEVNULL)
lines = open(out).read().split("\n")
want = sys.argv[1] if len(sys.argv) > 1 else ""
cur, stats, meta = None, {}, {}
for ln in lines:
    m = re.match(r"^(_Z\w+):", ln)
    if m:
        cur = m.group(1)
        stats[cur] = collections.Counter()
        continue
    if ln.startswith("\t.end_amdhsa_kernel") or ln.startswith(".Lfunc_end"):
        cur = None
    if cur and ln.startswith("\t") and not ln.strip().startswith((".", ";")):
        op = ln.strip().split()[0]
        c = stats[cur]
        c["total"] += 1
        if re.match(r"v_(exp|log|rcp|rsq|sqrt|sin|cos)_", op):
            c["trans"] += 1
        if op.startswith("v_"):
            c["valu"] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            c["vmem"] += 1
        elif op.startswith("ds_"):
            c["lds"] += 1
    m = re.match(r"\s*;\s*(NumVgprs|NumSgprs|Occupancy|ScratchSize): (\d+)", ln)
    if m and stats:
        meta.setdefault(list(stats)[-1], {})[m.group(1)] = int(m.group(2))
for k, c in stats.items():
    if want in k:
        print(f"{k[:44]:44s} total {c['total']:5d} valu {c['valu']:5d} trans {c['trans']:3d} salu {c['salu']:4d} "
              f"vmem {c['vmem']:3d} lds {c['lds']:3d}  {meta.get(k, {})}")
