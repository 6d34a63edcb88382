"""The observation side of the fleet env: the 20/22-wide state vectors of ``_getDroneStateVector`` (BaseAviary.py:764-790),
``_computeObs`` in the reference's dict form for small fleets and as device tensors for large ones (CtrlAviary.py:225-232),
and the neighbour lists that stand in for the reference's dense adjacency rows (BaseAviary.py:901-921).  A mix-in of
``CtrlAviary``: it uses the env's attributes as they are."""
from __future__ import annotations

from typing import NamedTuple, Optional

import numpy as np
import torch

from .. import _native as nat


class FleetObs(NamedTuple):
    """Observation of a fleet in tensor mode when neighbour lists are requested (``neighbors_k`` > 0): what the
    reference returns per drone as ``{"state": ..., "neighbors": ...}`` (CtrlAviary.py:225-232), for the whole fleet
    on the device.  ``neighbor_count[i]`` = drones within NEIGHBOURHOOD_RADIUS of drone i (the row sum of the
    reference's adjacency matrix minus the diagonal), ``neighbor_list[:, i]`` = up to ``neighbors_k`` of their indices
    (-1 padded): the sparse form of the reference's dense O(N^2) row."""

    state: torch.Tensor            # [N, 16 + n_act]
    neighbor_count: torch.Tensor   # [N] int32
    neighbor_list: torch.Tensor    # [neighbors_k, N] int32


class FleetObservation:
    def _obs_tensor(self) -> torch.Tensor:
        """The [N, 16 + n_act] rows Env.step hands out (allocated on first use; with placement=True by trial: placement.py)."""
        if self._obs_buf is None:
            shape = (self.NUM_DRONES, 16 + self.n_act)
            if self._placement_applies(4 * shape[0] * shape[1]):
                self._place_obs_rows(shape)
            else:
                self._obs_buf = torch.zeros(shape, dtype=torch.float32, device=self.ctx.device)
        return self._obs_buf

    def _rows_to_caller(self, rows: torch.Tensor) -> torch.Tensor:
        """Observation rows as the kernels write them (one per storage slot) -> the caller's numbering."""
        return rows if self.order is None else self.order.to_caller(rows, 0)

    def observe(self) -> torch.Tensor:
        """[N, 16+n_act] rows of _getDroneStateVector (BaseAviary.py:780-790), on device."""
        buf = self._obs_tensor()
        if self.order is not None:
            # (rows per storage slot first, then gathered into the caller's numbering: not through the buffer Env.step
            # hands out, which holds the caller's numbering already when the run kernels wrote it)
            if getattr(self, "_obs_slots", None) is None:
                self._obs_slots = torch.zeros_like(buf)
            buf = self._obs_slots
        self._join_fallback()
        la = self._last_action.data_ptr() if self._use_last_action else None
        nat.check(self.ctx.lib.dsim_observe(self.ctx.handle, self.ctx.stream_ptr(), self.NUM_DRONES,
                                            self.state.view(), la, buf.data_ptr(), 16 + self.n_act))
        return self._rows_to_caller(buf)

    def neighbors(self, max_k: Optional[int] = None):
        """Fleet-scale form of the observation's ``neighbors`` entry (BaseAviary._getAdjacencyMatrix,
        BaseAviary.py:901-921; CtrlAviary.py:225-231): (count [N] int32, list [max_k, N] int32, -1 padded) of the drones
        within NEIGHBOURHOOD_RADIUS of each drone, by the uniform-grid query of dsim_adjacency."""
        from ..downwash import Downwash
        if not np.isfinite(self.NEIGHBOURHOOD_RADIUS):
            raise ValueError("neighbourhood_radius is infinite: every drone neighbours every other (the reference's "
                             "default); pass a finite radius for neighbour lists")
        if self._adjacency is None:
            self._adjacency = Downwash(self.ctx, self.state, self._type_id, None)
        if self._downwash is not None:
            self._downwash.invalidate_prebin()            # the adjacency pass re-uses the ctx's grid bookkeeping
        k = self.neighbors_k if max_k is None else int(max_k)
        cnt, lst = self._adjacency.adjacency(float(self.NEIGHBOURHOOD_RADIUS), max_k=k)
        if self.order is not None:              # per-slot results of slot indices -> per-drone results of drone indices
            cnt = self.order.to_caller(cnt, 0)
            if lst is not None:
                lst = self.order.to_caller(lst, 1).long()
                lst = torch.where(lst >= 0, self.order.drone[lst.clamp(min=0)], lst).to(torch.int32)
        return cnt, lst

    def _getAdjacencyMatrix(self, pos: np.ndarray) -> np.ndarray:
        """BaseAviary.py:901-921 — O(N^2), only produced in dict mode (small fleets)."""
        d = np.linalg.norm(pos[:, None, :] - pos[None, :, :], axis=2)
        adj = (d < self.NEIGHBOURHOOD_RADIUS).astype(np.float64)
        np.fill_diagonal(adj, 1.0)
        return adj

    def _computeObs(self, obs=None):
        obs = self.observe() if obs is None else obs
        if not self.dict_io:
            if self.neighbors_k > 0:
                cnt, lst = self.neighbors()
                return FleetObs(obs, cnt, lst)
            return obs
        o = obs.double().cpu().numpy()
        self.pos, self.quat, self.rpy = o[:, 0:3], o[:, 3:7], o[:, 7:10]
        self.vel, self.ang_v = o[:, 10:13], o[:, 13:16]
        adj = self._getAdjacencyMatrix(self.pos)
        out = {}
        for i in range(self.NUM_DRONES):
            na = self.drones[i].n_act
            out[str(i)] = {"state": o[i, : 16 + na].copy(), "neighbors": adj[i, :]}
        return out

    def _getDroneStateVector(self, nth_drone: int) -> np.ndarray:
        return self.observe()[nth_drone].double().cpu().numpy()
