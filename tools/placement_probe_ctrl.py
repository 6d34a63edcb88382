#!/usr/bin/env python3
"""How much does the computeControl launch (k_control_fast, 4 194 304 drones) depend on where its targets (read) and its
outputs (cmd, pos_e, yaw_e: written) lie?  Candidates for either, 1 GiB of ballast between them.
usage: python tools/placement_probe_ctrl.py"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dronesim_amd import _native as nat  # noqa: E402
from dronesim_amd.control import INDIControl  # noqa: E402
from dronesim_amd.envs import CtrlAviary  # noqa: E402


def main():
    n = 4096 * 1024
    ij = np.arange(n) % 4096
    xyz = np.stack([(ij % 64) * 1.0, (ij // 64) * 1.0, np.full(n, 0.5)], 1)
    env = CtrlAviary(["robobee"], n, initial_xyzs=xyz, aggregate_phy_steps=1, noise_seed=1, dict_io=False, layout="tile64")
    ctrl = INDIControl("robobee", env=env)
    dev = env.ctx.device
    tpos = torch.from_numpy(np.ascontiguousarray(xyz.T.astype(np.float32))).to(dev)
    cmd = torch.full((n, 4), 0.4, device=dev)
    env.step(cmd)
    ctrl.computeControlFromState(1 / 240, None, target_pos=tpos, target_rpy=np.array([0, 0, 0.4]))
    a = nat.StepArgs()
    a.phys_substeps, a.dt_phys, a.dt_ctrl = 0, 1 / 240, 1 / 240
    lib, h = env.ctx.lib, env.ctx.handle

    def timed(tg_view, pos_e, yaw_e, cmd_out):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for it in range(23):
            if it == 3:
                e0.record()
            nat.check(lib.dsim_control2(h, env.ctx.stream_ptr(), n, env.state.view(), tg_view, ctypes.byref(a), pos_e.data_ptr(),
                                        yaw_e.data_ptr(), cmd_out.data_ptr()))
        e1.record()
        torch.cuda.synchronize()
        return round(e0.elapsed_time(e1) * 1e3 / 20, 1)

    keep = []
    base = timed(ctrl._targets.view(), ctrl._pos_e, ctrl._yaw_e, ctrl._cmd)
    print("as allocated:", base, flush=True)
    tg0 = ctrl._targets.data
    for k in range(8):
        keep.append(torch.empty((1 << 30,), dtype=torch.uint8, device=dev))
        cand = tg0.clone()
        keep.append(cand)
        ctrl._targets.data = cand
        t_tg = timed(ctrl._targets.view(), ctrl._pos_e, ctrl._yaw_e, ctrl._cmd)
        ctrl._targets.data = tg0
        outs = [torch.zeros_like(ctrl._pos_e), torch.zeros_like(ctrl._yaw_e), torch.zeros_like(ctrl._cmd)]
        keep += outs
        t_out = timed(ctrl._targets.view(), *outs)
        print({"candidate": k, "targets_elsewhere_us": t_tg, "outputs_elsewhere_us": t_out}, flush=True)


if __name__ == "__main__":
    main()
