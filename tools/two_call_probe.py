#!/usr/bin/env python3
"""Dev probe: the reference-shaped loop obs = env.step(cmd); cmd = ctrl.computeControlFromState(obs) on every fleet kind
(bench.Fleet + make_two_call_loop), microseconds per loop iteration and the HBM fraction on the loop's algorithmic bytes.
usage: python tools/two_call_probe.py [kinds...]   kinds: quad hexa mixed mixed_caller config5 (default: all but mixed_caller)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

# bytes per drone and loop iteration: physics (13 + na) r + (13 + na + 16 + na_table) w; control (13 + 7 + na + 10) r + (7 + na + na_table... see DESIGN.md
BYTES = {"quad": 428, "hexa": 476, "mixed": 460, "mixed_caller": 460, "config5": 484}


def main():
    import torch
    kinds = sys.argv[1:] or ["quad", "hexa", "mixed", "config5"]
    steps = int(os.environ.get("PROBE_STEPS", "100"))
    out = {}
    for kind in kinds:
        big = kind != "config5"
        fl = bench.Fleet(4096 if big else 65536, 1024 if big else 1, 0, 1, "tile64", 1, hexa=kind == "hexa",
                         mixed=kind.startswith("mixed"), config5=kind == "config5",
                         storage="caller" if kind == "mixed_caller" else None)
        fused_w, fused_d, reg = fl.timed(steps, 10, min_s=0.05)
        fused_us = fused_d / (steps * reg) * 1e6
        fl.make_two_call_loop()
        w, d, reg = fl.timed(steps, 10, min_s=0.05)
        us = d / (steps * reg) * 1e6
        out[kind] = {"drones": fl.n, "two_call_us": round(us, 2), "two_call_host_us": round(w / (steps * reg) * 1e6, 2),
                     "fused_us": round(fused_us, 2), "bytes": BYTES[kind],
                     "hbm_frac": round(fl.n * BYTES[kind] / (us * 1e-6) / 8e12, 4), "ratio_to_fused": round(us / fused_us, 3)}
        print(kind, json.dumps(out[kind]), flush=True)
        fl.env.close()
        del fl
        torch.cuda.empty_cache()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
