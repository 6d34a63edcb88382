#!/usr/bin/env python3
"""Dev probe (round 4): why does every candidate of the rows' walk time alike on some boxes?  One 4 194 304-quad env,
zero-sub-step passes of the Env.step launch (what placement.place_rows times) with
  A  the rows as torch allocates them,
  B  rows from the driver, 1 GiB of ballast between candidates, 28 GiB walked,
  C  the STATE moved to a driver allocation, the same walk,
  D  the action / echo arrays from the driver too.
usage: python tools/region_probe.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dronesim_amd import placement  # noqa: E402
from dronesim_amd.envs import CtrlAviary  # noqa: E402


def main():
    n = 4096 * 1024
    ij = np.arange(n) % 4096
    xyz = np.stack([(ij % 64) * 1.0, (ij // 64) * 1.0, np.full(n, 0.5)], 1)
    env = CtrlAviary(["robobee"], n, initial_xyzs=xyz, aggregate_phy_steps=1, noise_seed=1, dict_io=False, layout="tile64",
                     placement=False)
    ctx = env.ctx
    gib = 1 << 30

    def t(rows):
        return placement._event_timer(env._rows_trial, rows, 5)

    def walk(tag, span_gib=28, ballast_gib=1.0):
        keep, out = [], []
        held = 0
        while held < span_gib * gib:
            c = placement._DriverBlock(ctx, (n, 28))
            rows = c.tensor()
            rows.zero_()
            out.append(round(t(rows), 1))
            keep.append(rows)
            held += c.nbytes
            b = placement._DriverBlock(ctx, (int(ballast_gib * gib) // 4,))
            keep.append(b)
            held += b.nbytes
        print(tag, "state", hex(env.state.data.data_ptr()), out, flush=True)
        del keep
        torch.cuda.synchronize()

    if "--arena" in sys.argv:
        # E: ONE driver allocation, the state at its start, the rows D GiB behind it (membench --bigsweep: good from 16 GiB on)
        a0 = torch.zeros((n, 28), dtype=torch.float32, device=ctx.device)
        print("A torch rows", round(t(a0), 1), flush=True)
        nst = env.state.data.numel()
        total = 24 * gib // 4
        blk = placement._DriverBlock(ctx, (total,)).tensor()
        old_state = env.state.data
        env._move_state(blk[:nst].view(old_state.shape))
        out = []
        for d_gib in (1, 4, 8, 12, 15, 16, 17, 20, 22):
            off = d_gib * gib // 4
            rows = blk[off: off + n * 28].view(n, 28)
            out.append((d_gib, round(t(rows), 1)))
        print("E arena: state at 0, rows at D GiB:", out, flush=True)
        # the state in the MIDDLE of the arena: rows 16 GiB in front of it
        env._move_state(blk[20 * gib // 4: 20 * gib // 4 + nst].view(old_state.shape))
        out = []
        for d_gib in (0, 2, 3, 4, 8, 12, 18):
            off = d_gib * gib // 4
            rows = blk[off: off + n * 28].view(n, 28)
            out.append((d_gib, round(t(rows), 1)))
        print("E arena: state at 20 GiB, rows at D GiB:", out, flush=True)
        # F: the OTHER arrays of the launch — the echoed action (written, 16 B per drone) and the action (read): state at the
        # start of the arena, rows one window on; echo / action in the state's window or in the rows'
        env._move_state(blk[:nst].view(old_state.shape))
        rows = blk[17 * gib // 4: 17 * gib // 4 + n * 28].view(n, 28)
        keep_la, keep_ab = env._last_action, env._action_buf
        out = []
        for name, e_gib, a_gib in (("echo+action as torch put them", None, None), ("echo in the state's window", 4, None),
                                   ("echo in the rows' window", 19, None), ("echo and action in the state's window", 4, 5),
                                   ("echo and action in the rows' window", 19, 20), ("echo rows' window, action state's", 19, 5)):
            env._last_action = keep_la if e_gib is None else blk[e_gib * gib // 4: e_gib * gib // 4 + keep_la.numel()].view(keep_la.shape)
            env._action_buf = keep_ab if a_gib is None else blk[a_gib * gib // 4: a_gib * gib // 4 + keep_ab.numel()].view(keep_ab.shape)
            env._last_action.zero_(); env._action_buf.fill_(0.4)
            out.append((name, round(t(rows), 1)))
        print("F arena, state at 0, rows at 17 GiB:", out, flush=True)
        env._last_action, env._action_buf = keep_la, keep_ab
        # G: computeControl (k_control_fast): state at 0; its targets (read) and its outputs (written) in either window
        from dronesim_amd.control import INDIControl
        from dronesim_amd.fleet import frozen
        ctrl = INDIControl("robobee", env=env)
        tp = frozen(torch.from_numpy(xyz.astype(np.float32)).to(ctx.device))
        ctrl._outputs_placed = True

        def tc():
            for _ in range(3):
                ctrl.computeControlFromState(1 / 240, None, target_pos=tp)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ctrl.computeControlFromState(1 / 240, None, target_pos=tp)
            e1.record(); e1.synchronize()
            return round(e0.elapsed_time(e1) * 100, 1)
        out = [("as torch put them", tc())]
        tdata, n_pad = ctrl._targets.data, env.state.n_pad
        for name, t_gib, o_gib in (("targets state's window, outputs rows' window", 4, 19), ("targets rows' window, outputs rows' window", 21, 19),
                                   ("targets state's window, outputs state's window", 4, 6), ("targets rows' window, outputs state's window", 21, 6)):
            nt = blk[t_gib * gib // 4: t_gib * gib // 4 + tdata.numel()].view(tdata.shape)
            nt.copy_(tdata)
            ctrl._targets.data = nt
            ob = blk[o_gib * gib // 4: o_gib * gib // 4 + 8 * n_pad].view(8, n_pad)
            ctrl._cmd, ctrl._pos_e, ctrl._yaw_e = ob[0:4], ob[4:7], ob[7]
            ctrl._plan = None
            out.append((name, tc()))
        print("G computeControl, state at 0:", out, flush=True)
        return
    a = torch.zeros((n, 28), dtype=torch.float32, device=ctx.device)
    print("A torch rows", hex(a.data_ptr()), round(t(a), 1), "state", hex(env.state.data.data_ptr()), flush=True)
    walk("B driver rows")
    blk = placement._DriverBlock(ctx, tuple(env.state.data.shape))
    env._move_state(blk.tensor())
    print("C state moved to the driver: torch rows", round(t(a), 1), flush=True)
    walk("C driver rows")
    for name in ("_action_buf", "_last_action"):
        old = getattr(env, name)
        nb = placement._DriverBlock(ctx, tuple(old.shape)).tensor()
        nb.copy_(old)
        setattr(env, name, nb)
    print("D action + echo from the driver: torch rows", round(t(a), 1), flush=True)
    walk("D driver rows", span_gib=12)


if __name__ == "__main__":
    main()
