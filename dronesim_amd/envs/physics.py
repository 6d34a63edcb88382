"""The reference's physics-mode enumeration."""
from enum import Enum


class Physics(Enum):
    """dronesim/envs/BaseAviary.py:41-49.  Only PYB works in the reference fork
    (every other branch is dead code there); the add-on terms are
    exposed here as their intended formulas."""

    PYB = "pyb"
    DYN = "dyn"
    PYB_GND = "pyb_gnd"
    PYB_DRAG = "pyb_drag"
    PYB_DW = "pyb_dw"
    PYB_GND_DRAG_DW = "pyb_gnd_drag_dw"
