#!/usr/bin/env python3
"""Dev probe (GPU box, ONE process): device time of the kernels the halo exchange adds to a config-5 step — dsim_halo_pack
(select + pack + headers), halo binning, the halo pass of the query — on a 65 536-drone slab with ONE synthetic peer whose
"message" is this rank's own packed strip mirrored across the slab edge (no wire: a device copy stands in for it)."""
import ctypes
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

graft.build()
from dronesim_amd import _native as nat, fleet, params  # noqa: E402

n, slab = 65536, 128.0
rng = np.random.default_rng(1)
types = [params.builtin_type("robobee"), params.builtin_type("hexa_6DOF")]
ctx = fleet.Context(types)
st = fleet.FleetState(ctx, n, "tile64")
pos = np.stack([rng.uniform(0, slab, n), rng.uniform(0, 512, n), rng.uniform(0.5, 20.5, n)], 0).astype(np.float32)
st.set_fields(0, torch.from_numpy(pos))
st.set_fields(6, torch.ones((1, n)))
tid = torch.from_numpy((np.arange(st.n_pad) % 2).astype(np.uint8)).to(ctx.device)
cap, HDR = st.n_pad, nat.HALO_HDR
stride = HDR + 3 * cap
send = torch.zeros((2, stride), device=ctx.device)
recv = torch.zeros((2, stride), device=ctx.device)
scratch = torch.zeros(32, dtype=torch.int32, device=ctx.device)
recv[1, 1:5] = torch.tensor([slab, 0.0, 2 * slab, 512.0])             # the peer: the next slab
pl = nat.HaloPlan()
pl.world, pl.rank, pl.cap = 2, 0, cap
pl.send, pl.recv, pl.scratch = send.data_ptr(), recv.data_ptr(), scratch.data_ptr()
pl.send_cap[1], pl.reach[1] = cap, 10.0 + 100.0 / 240.0
lib, h, sp = ctx.lib, ctx.handle, ctx.stream_ptr()


def timed(fn, iters=200):
    for _ in range(10):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


nat.check(lib.dsim_halo_pack(h, sp, n, st.view(), ctypes.byref(pl)))
torch.cuda.synchronize()
cnt = int(send[1, 0:1].view(torch.int32)[0])
C = -(-int(cnt * 1.25 + 512) // 256) * 256
pl.send_cap[1] = pl.recv_cap[1] = C
t_pack = timed(lambda: nat.check(lib.dsim_halo_pack(h, sp, n, st.view(), ctypes.byref(pl))))
# the "message": the strip mirrored across the edge x = slab
msg = send[1, : HDR + 3 * C].clone()
xyz = msg[HDR:].view(-1, 3)
xyz[:, 0] = 2 * slab - xyz[:, 0]
recv[1, : HDR + 3 * C] = msg
recv[1, 1:5] = torch.tensor([slab, 0.0, 2 * slab, 512.0])
cell, reach = 5.0, 12.5
xmin, ymin = -reach - cell, -reach - cell
nx, ny = int((slab + reach + cell - xmin) // cell) + 1, int((512 + reach + cell - ymin) // cell) + 1
g = nat.DownwashArgs()
g.pos_all, g.m, g.m_pad = None, n + C, n + C
g.xmin, g.ymin, g.cell, g.nx, g.ny = xmin, ymin, cell, nx, ny
ws = torch.empty((lib.dsim_downwash_workspace_halo(n, C, nx, ny),), dtype=torch.int32, device=ctx.device)
g.workspace, g.workspace_len, g.type_id, g.local_offset = ws.data_ptr(), ws.numel(), tid.data_ptr(), 0
g.halo = ctypes.addressof(pl)
force = torch.zeros((3, st.n_pad), device=ctx.device)


def phase(ph):
    g.phase = ph
    nat.check(lib.dsim_downwash(h, sp, n, st.view(), ctypes.byref(g), force.data_ptr()))


def split():
    phase(nat.DW_HALO_BIN); phase(nat.DW_LOCAL); phase(nat.DW_HALO_QUERY)


def local_only():
    phase(nat.DW_HALO_BIN); phase(nat.DW_LOCAL)
    g.phase = nat.DW_HALO_QUERY      # (parity bookkeeping of the halo grid: flip without launching is not exposed; run it)
    nat.check(lib.dsim_downwash(h, sp, n, st.view(), ctypes.byref(g), force.data_ptr()))


t_split = timed(split)
t_all = timed(lambda: phase(nat.DW_ALL))
t_bin = timed(lambda: (phase(nat.DW_HALO_BIN), phase(nat.DW_HALO_QUERY)))
print(json.dumps({"drones": n, "selected": cnt, "message_capacity": C, "halo_pack_us": round(t_pack, 2),
                  "split_bin_local_haloquery_us": round(t_split, 2), "one_grid_bin_query_us": round(t_all, 2),
                  "halo_bin_plus_halo_query_us": round(t_bin, 2), "overflow": ctx.query(nat.QUERY_HALO_OVERFLOW)}))
