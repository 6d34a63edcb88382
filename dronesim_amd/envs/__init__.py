from .fleet_aviary import CtrlAviary, FleetObs, Physics, RPYTAviary, VelocityAviary  # noqa: F401
