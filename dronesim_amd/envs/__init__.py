from .fleet_aviary import CtrlAviary, Physics  # noqa: F401
