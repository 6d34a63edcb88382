"""6-DOF INDI controller surface — dronesim/control/INDIControl_6DOF.py:110-634 (the reference names
this class ``INDIControl`` too, in its own module; examples/fly_hexa_6DOF.py:18 imports it from there).

Same batched implementation as :mod:`dronesim_amd.control.indi_fleet`; what differs, as in the
reference, is the positional order of ``computeControl`` (``target_rpy`` before ``target_vel``,
``target_rpy_rates`` before ``target_acc``; INDIControl_6DOF.py:259-270) and the law itself
(``target_acc`` unused, attitude target forced level, WLS allocation; selected by the vehicle type).
"""
from __future__ import annotations

import numpy as np

from .indi_fleet import INDIControl as _QuadSurface


class INDIControl(_QuadSurface):
    def __init__(self, drone_model="hexa_6DOF", g: float = 9.8, **kw):
        super().__init__(drone_model, g, **kw)

    def computeControl(self, control_timestep, cur_pos, cur_quat, cur_vel, cur_ang_vel, target_pos,
                       target_rpy=np.zeros(3), target_vel=np.zeros(3), target_rpy_rates=np.zeros(3),
                       target_acc=np.zeros(3)):
        return super().computeControl(control_timestep, cur_pos, cur_quat, cur_vel, cur_ang_vel, target_pos,
                                      target_vel=target_vel, target_acc=target_acc, target_rpy=target_rpy,
                                      target_rpy_rates=target_rpy_rates)
