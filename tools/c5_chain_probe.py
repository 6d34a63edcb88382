#!/usr/bin/env python3
"""What each launch of config 5's chain costs IN the chain (dev helper): the 65 536-drone shard stepped (a) as the product does
(query, step + grid binning, WLS fallback pass), (b) with the fallback pass left out (DSIM_OPT_DEFER_FALLBACK and nobody
launching it: timing only — the queue is empty in this hover workload, so results do not change), (c) the fallback pass alone,
back to back, (d) with Env.step and computeControl as separate launches — in front of them the chain with kept candidate lists (the default).  usage: python tools/c5_chain_probe.py [steps [lib]]"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from dronesim_amd import _native as nat  # noqa: E402


def timed(f, k):
    for _ in range(20):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k):
        f()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / k


def main():
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    if len(sys.argv) > 2:
        nat.load(sys.argv[2])                      # an A/B build of the same ABI
    fk = bench.Fleet(65536, 1, 0, 1, "tile64", 1, config5=True)                   # the default: kept candidate lists
    print("chain, as the product steps it          %.2f us   (kept lists: one query in %d makes them, skin %.2f m)"
          % (timed(fk.step, k), fk.env._downwash.keep_lists, fk.env._downwash.keep_skin))
    r = fk.env.ctx.query(nat.QUERY_DW_REUSES)
    print("   queries answered from the lists %d, mean drones outside the skin %.1f" % (r, fk.env.ctx.query(nat.QUERY_DW_MOVERS) / max(r, 1)))
    fk.env.close()
    del fk
    # everything below: the neighbour query made from scratch every step (downwash_keep = 0), as rounds 1-5 stepped it
    fl = bench.Fleet(65536, 1, 0, 1, "tile64", 1, config5=True, dw_keep=0)
    env = fl.env
    print("chain with a plain query every step     %.2f us" % timed(fl.step, k))
    env._tuning |= nat.OPT_DEFER_FALLBACK          # the step no longer launches the pass (and this probe does not either)
    env._fused_plan_dw = None
    print("chain without the fallback launch       %.2f us" % timed(fl.step, k))
    env._tuning &= ~nat.OPT_DEFER_FALLBACK
    env._fused_plan_dw = None
    tid = env._type_id.data_ptr()

    def fb():
        nat.check(env.ctx.lib.dsim_wls_fallback(env.ctx.handle, env.ctx.stream_ptr(), env.NUM_DRONES, env.state.view(), tid, None))
    print("fallback pass alone, back to back       %.2f us" % timed(fb, k))
    print("chain again                             %.2f us" % timed(fl.step, k))
    # (d) the chain with Env.step and computeControl as separate launches: query -> physics (force in, next grid out) -> control ->
    # fallback.  Under rocprofv3 --kernel-trace (round 6): k_step_runs 13.24 us against k_physics_runs 11.97 + k_control_runs 5.37 —
    # a control law riding in the next query's shadow would take 1.3 us off the chain, not the 5 us round 5 hoped for
    tg = fl.tgt
    cmd_out = torch.zeros((env.n_act, env.state.n_pad), device=env.ctx.device)
    pe = torch.zeros((3, env.state.n_pad), device=env.ctx.device)
    ye = torch.zeros(env.state.n_pad, device=env.ctx.device)

    def split():
        a = env.step_args(None)
        a.bin_next = env._downwash.bin_next_ptr()
        nat.check(env.ctx.lib.dsim_physics(env.ctx.handle, env.ctx.stream_ptr(), env.NUM_DRONES, env.state.view(), None, ctypes.byref(a)))
        b = nat.StepArgs.from_buffer_copy(a)
        b.ext_force = None
        b.bin_next = None
        nat.check(env.ctx.lib.dsim_control2(env.ctx.handle, env.ctx.stream_ptr(), env.NUM_DRONES, env.state.view(), tg.view(), ctypes.byref(b),
                                            pe.data_ptr(), ye.data_ptr(), cmd_out.data_ptr()))
        env._env_steps += 1
    env._fused_plan_dw = None
    print("query, physics, control, fallback apart %.2f us" % timed(split, k))


if __name__ == "__main__":
    main()
