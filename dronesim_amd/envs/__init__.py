from .fleet_aviary import CtrlAviary, Physics, RPYTAviary, VelocityAviary  # noqa: F401
