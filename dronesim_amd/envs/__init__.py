from .adaptor_aviary import RPYTAviary, VelocityAviary  # noqa: F401
from .fleet_aviary import CtrlAviary, FleetObs, Physics  # noqa: F401
from .fused_graph import FusedGraph  # noqa: F401
