"""A captured sequence of fused steps (``CtrlAviary.capture_fused``): one hipGraph, one host call per replay."""
from __future__ import annotations

import ctypes

import torch

from .. import _native as nat
from ..fleet import WaypointTargets


class FusedGraph:
    """A captured sequence of fused steps (see :meth:`CtrlAviary.capture_fused`)."""

    def __init__(self, env, targets, steps: int, control_timestep):
        self.env, self.steps = env, steps
        env.materialize()
        dev = env.ctx.device
        self._counter = torch.zeros((1,), dtype=torch.int64, device=dev)
        wp = isinstance(targets, WaypointTargets)
        if env._downwash is not None:
            env._downwash._box = None           # the eager query inside step_args measures the fleet's box afresh
            env._downwash.invalidate_prebin()
        self._args = env.step_args(control_timestep, nat.OPT_BCAST_TGT if targets.broadcast else 0)
        self._args.step_index = 0
        self._args.step_index_dev = self._counter.data_ptr()
        if wp:
            targets.fill(self._args)
            self._tview = nat.View()
        else:
            self._tview = targets.view()
        self._sview = env.state.view()
        self._targets = targets
        lib, h, n = env.ctx.lib, env.ctx.handle, env.NUM_DRONES
        dw = env._downwash
        self._graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            # one untimed eager pass on the side stream (lazy module loading must not happen under capture)
            nat.check(lib.dsim_counter_add(h, side.cuda_stream, self._counter.data_ptr(), 0))
            with torch.cuda.graph(self._graph, stream=side):
                sp = torch.cuda.current_stream(dev).cuda_stream
                refresh = None
                if dw is not None:
                    # The grid stands for the whole graph (no box re-measurement: that is a host read-back).  Captured
                    # behind dsim_downwash_reset the first query clears both count buffers and bins the fleet itself, so a
                    # replay assumes nothing about what ran before it; the last step fills no grid ahead.
                    refresh, dw._box_refresh = dw._box_refresh, 1 << 62
                    dw._prebin_version = None
                    nat.check(lib.dsim_downwash_reset(h))
                try:
                    for i in range(steps):
                        self._args.step_index = i          # frozen offset; the base is read from the device counter
                        if dw is not None:
                            self._args.ext_force = dw.compute().data_ptr()
                            self._args.bin_next = dw.bin_next_ptr() if i + 1 < steps else None
                        nat.check(lib.dsim_step(h, sp, n, self._sview, self._tview, ctypes.byref(self._args)))
                finally:
                    if dw is not None:
                        dw._box_refresh = refresh
                        dw._prebin_version = None
                        nat.check(lib.dsim_downwash_reset(h))
                        # the captured launches hold these addresses: an eager step that later re-measures the box and
                        # outgrows the workspace allocates a new one — this one must outlive the graph
                        self._keepalive = (dw._ws, dw.force, dw.type_id)
                nat.check(lib.dsim_counter_add(h, sp, self._counter.data_ptr(), steps))
        torch.cuda.current_stream(dev).wait_stream(side)
        self._counter_host = 0

    def replay(self) -> None:
        env = self.env
        env.materialize()
        if self._counter_host != env._env_steps:  # eager steps in between: realign the noise stream
            self._counter.fill_(env._env_steps)
        self._graph.replay()
        if env._downwash is not None:      # the buffers are as the graph left them, not as the ctx last saw them
            env._downwash._prebin_version = None
            nat.check(env.ctx.lib.dsim_downwash_reset(env.ctx.handle))
        self._counter_host = env._env_steps + self.steps
        env._use_last_action = False
        env._chain_ok = True
        env._fused_plan = None
        env.step_counter += env.AGGR_PHY_STEPS * self.steps
        env._env_steps += self.steps
