#!/usr/bin/env python3
"""Static instruction mix per kernel from the gfx950 assembly of the library's translation units (dev helper).
usage: python tools/isa_stats.py [kernel-name-substring]"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

flags = [f for f in ge.HIPCC_FLAGS if f not in ("-shared", "-fPIC")]
procs = []
for src in ge.HIP_SRCS:            # every translation unit of the library, side by side
    out = os.path.join(tempfile.gettempdir(), os.path.basename(src) + ".s")
    procs.append((out, subprocess.Popen(["/opt/rocm/bin/hipcc", *flags, "-S", "-o", out, "--cuda-device-only", src], stderr=subprocess.DEVNULL)))
lines = []
for out, p in procs:
    assert p.wait() == 0, out
    lines += open(out).read().split("\n")
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
