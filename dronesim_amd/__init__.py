"""dronesim_amd — MI355X-native fleet dynamics + INDI control step.

A from-scratch gfx950 implementation of ONE hot path of enac-drones/dronesim: the
per-timestep multi-drone physics (``BaseAviary.step``) + INDI control law
(``INDIControl.computeControl``), behind the reference's own call surfaces.
Python here is host plumbing (torch-ROCm tensors, streams); the arithmetic is a
hand-written HIP kernel behind the C-ABI in ``include/dronesim_amd.h``.

There is no CPU fallback: importing the submodules that touch the device fails
loudly when ``libdronesim_amd.so`` is missing (build with ``__graft_entry__.build()``).
"""
from . import params  # noqa: F401
from .params import DroneType, builtin_type, parse_urdf  # noqa: F401

__all__ = ["params", "DroneType", "builtin_type", "parse_urdf"]
