"""Device-side flight log with the reference Logger's layout (dronesim/utils/Logger.py:12-157).

The reference appends one 20-vector per drone per step to ``states[N, 20, T]`` (and ``controls
[N, 12, T]``, ``timestamps[N, T]``) on the host, growing the arrays by ``np.concatenate`` when they
are not pre-allocated (O(T^2)).  Here one kernel launch appends the whole fleet's rows for a step
into a pre-allocated device ring ``[T, F, n_pad]`` (field-major slabs: coalesced writes); ``save()``
and the array properties hand the data back in the reference's shapes and ``np.savez`` keys
(``timestamps``, ``states``, ``controls``; Logger.py:152-157) so the reference's plotting code works
on a fleet slice.  Deviation: when more than ``duration_sec * logging_freq_hz`` steps are logged
the ring keeps the most recent ones instead of growing.
"""
from __future__ import annotations

import os
from datetime import datetime
from typing import Optional

import numpy as np
import torch

from .. import _native as nat


class Logger:
    def __init__(self, logging_freq_hz: int, env, duration_sec: int = 10, control_length: int = 12):
        self.env = env
        self.LOGGING_FREQ_HZ = logging_freq_hz
        self.NUM_DRONES = env.NUM_DRONES
        self.state_length = 16 + env.n_act            # Logger.py:20 (20 for quads)
        self.control_length = control_length
        self.capacity = max(1, int(duration_sec * logging_freq_hz))
        dev, n_pad = env.ctx.device, env.state.n_pad
        self._states = torch.zeros((self.capacity, self.state_length, n_pad), dtype=torch.float32, device=dev)
        self._controls = torch.zeros((self.capacity, control_length, n_pad), dtype=torch.float32, device=dev)
        self._times = np.zeros(self.capacity)
        self.count = 0                                # rows logged so far (Logger.counters, same for all drones)

    def log(self, timestamp: float, control: Optional[torch.Tensor] = None) -> None:
        """Append the whole fleet's current state vectors (and optional control targets [12, N])."""
        env, k = self.env, self.count % self.capacity
        la = env._last_action.data_ptr() if env._use_last_action else None
        nat.check(env.ctx.lib.dsim_observe_soa(env.ctx.handle, env.ctx.stream_ptr(), env.NUM_DRONES,
                                               env.state.view(), la, self._states[k].data_ptr(),
                                               self.state_length))
        if control is not None:
            c = torch.as_tensor(control, dtype=torch.float32, device=self._controls.device)
            c = c.reshape(self.control_length, -1)
            if getattr(env, "order", None) is not None:
                c = env.order.to_storage(c, 1)
            self._controls[k, :, : self.NUM_DRONES] = c
        self._times[k] = timestamp
        self.count += 1

    def _order(self):
        n = min(self.count, self.capacity)
        start = self.count % self.capacity if self.count > self.capacity else 0
        return [(start + j) % self.capacity for j in range(n)]

    def arrays(self, drones=slice(None)):
        """(timestamps [n, T], states [n, state_length, T], controls [n, 12, T]) for a fleet slice,
        as numpy arrays in the reference's shapes."""
        idx = self._order()
        sel = torch.arange(self.NUM_DRONES)[drones]
        if getattr(self.env, "order", None) is not None:      # the slabs are written per storage slot
            sel = self.env.order.slot.cpu()[sel]
        st = self._states[idx][:, :, sel].permute(2, 1, 0).double().cpu().numpy()
        ct = self._controls[idx][:, :, sel].permute(2, 1, 0).double().cpu().numpy()
        ts = np.tile(self._times[idx], (st.shape[0], 1))
        return ts, st, ct

    timestamps = property(lambda s: s.arrays()[0])
    states = property(lambda s: s.arrays()[1])
    controls = property(lambda s: s.arrays()[2])

    def save(self, file_path: Optional[str] = None, file_name: Optional[str] = None, drones=slice(None)) -> str:
        """Logger.save (Logger.py:143-157): one ``.npy``-named npz with the three arrays."""
        file_path = file_path or os.path.join(os.getcwd(), "files", "logs", "")
        os.makedirs(file_path, exist_ok=True)
        file_name = file_name or "save-flight-" + datetime.now().strftime("%m.%d.%Y_%H.%M.%S")
        ts, st, ct = self.arrays(drones)
        out = os.path.join(file_path, file_name + ".npy")
        with open(out, "wb") as fh:
            np.savez(fh, timestamps=ts, states=st, controls=ct)
        return out
