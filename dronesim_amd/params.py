"""Per-type vehicle constants (host side) and their C-ABI image.

Replaces, for the hot path only, what the reference parses per drone instance:
``BaseAviary._parseURDFParameters`` (dronesim/envs/BaseAviary.py:2041-2140, the
``Drone`` dataclass :69-95) and ``INDIControl._parseURDFControlParameters``
(dronesim/control/INDIControl.py:55-106).  The build keeps ONE constant record
per vehicle *type*; the fleet indexes it through a ``type_id`` byte.

The shipped vehicle types are built in (values cited to the reference URDFs) so
the package works where the reference tree is absent (the GPU box);
:func:`parse_urdf` reads any URDF written in the reference's dialect.

Deviation from the reference, documented: the reference's
``Gains.att/rate`` are class attributes (dronesim/utils/utils.py:21-24), so in a
fleet mixing two quad types the last-constructed type's gains leak to all quads.
Here gains are strictly per type.
"""
from __future__ import annotations

import ctypes
import math
import os
import xml.etree.ElementTree as ET
from dataclasses import dataclass, field
from typing import Dict, List, Sequence

import numpy as np

MAX_ACT = 6
KIND_QUAD = 0
KIND_HEXA6DOF = 1
KIND_HEXA_QUADLAW = 2     # morphing-hexa physics, quad INDI law on six actuators (hexa_6DOF_simple.urdf)
DYN_MIXER_X, DYN_MIXER_PLUS = 0, 1     # Physics.DYN: DroneModel.CF2X / CF2P-HB mixers (BaseAviary.py:1794-1803)

# Bullet's btMultiBody defaults are float literals (0.04f) held in double
# precision builds; keeping the exact fp32 value makes fp32 (GPU) and fp64
# (oracle) use the same constant.
BULLET_DAMPING = float(np.float32(0.04))
BULLET_MAX_COORD_VEL = 100.0


class TypeParamsC(ctypes.Structure):
    """ctypes mirror of ``dsim_type_params`` (include/dronesim_amd.h)."""

    _fields_ = [
        ("kind", ctypes.c_int32),
        ("n_act", ctypes.c_int32),
        ("mass", ctypes.c_double),
        ("inertia", ctypes.c_double * 3),
        ("kf", ctypes.c_double),
        ("km", ctypes.c_double),
        ("pwm2rpm_scale", ctypes.c_double * MAX_ACT),
        ("pwm2rpm_const", ctypes.c_double * MAX_ACT),
        ("pwm_min", ctypes.c_double * MAX_ACT),
        ("pwm_max", ctypes.c_double * MAX_ACT),
        ("rotor_pos", (ctypes.c_double * 3) * MAX_ACT),
        ("rotor_axis", (ctypes.c_double * 3) * MAX_ACT),
        ("rotor_spin", ctypes.c_double * MAX_ACT),
        ("G1", (ctypes.c_double * MAX_ACT) * MAX_ACT),
        ("alloc", (ctypes.c_double * MAX_ACT) * MAX_ACT),
        ("alloc2", (ctypes.c_double * MAX_ACT) * MAX_ACT),
        ("kp_pos", ctypes.c_double),
        ("kd_pos", ctypes.c_double),
        ("att_gain", ctypes.c_double * 3),
        ("rate_gain", ctypes.c_double * 3),
        ("gravity", ctypes.c_double),
        ("lin_damping", ctypes.c_double),
        ("ang_damping", ctypes.c_double),
        ("max_coord_vel", ctypes.c_double),
        ("drag_coeff", ctypes.c_double * 3),
        ("gnd_eff_coeff", ctypes.c_double),
        ("prop_radius", ctypes.c_double),
        ("gnd_eff_h_clip", ctypes.c_double),
        ("dw_coeff", ctypes.c_double * 3),
        ("max_speed_kmh", ctypes.c_double),
        ("collision_radius", ctypes.c_double),
        ("collision_below", ctypes.c_double),
        ("contact_friction", ctypes.c_double),
        ("base_offset", ctypes.c_double * 3),
        ("arm", ctypes.c_double),
        ("dyn_mixer", ctypes.c_int32),
        ("_pad_dyn", ctypes.c_int32),
    ]


@dataclass
class DroneType:
    """Constants of one vehicle type (the reference's ``Drone`` dataclass plus the
    controller constants of ``INDIControl``), fp64."""

    name: str
    kind: int
    n_act: int
    mass: float                      # rigid-body mass used by the integrator
    inertia: Sequence[float]         # principal moments
    kf: float
    km: float
    pwm2rpm_scale: Sequence[float]
    pwm2rpm_const: Sequence[float]
    pwm_min: Sequence[float]
    pwm_max: Sequence[float]
    rotor_pos: Sequence[Sequence[float]]
    rotor_axis: Sequence[Sequence[float]]
    rotor_spin: Sequence[float]
    G1: np.ndarray                   # [n_out][n_act]
    kp_pos: float
    kd_pos: float
    att_gain: Sequence[float]
    rate_gain: Sequence[float]
    ctrl_mass: float = 0.0           # INDIControl.m = first link's mass (INDIControl.py:66-67)
    gravity: float = 9.8             # BaseAviary.py:182
    lin_damping: float = BULLET_DAMPING
    ang_damping: float = BULLET_DAMPING
    max_coord_vel: float = BULLET_MAX_COORD_VEL
    drag_coeff: Sequence[float] = (0.0, 0.0, 0.0)
    gnd_eff_coeff: float = 0.0
    prop_radius: float = 0.0
    gnd_eff_h_clip: float = 0.0
    dw_coeff: Sequence[float] = (0.0, 0.0, 0.0)
    max_speed_kmh: float = 30.0      # URDF properties max_speed_kmh (all shipped types: 30)
    collision_radius: float = 0.0    # bounding cylinder of the <collision> shapes about body z: radius ...
    collision_below: float = 0.0     # ... and extent below the COM (ground-plane watch; 0 = none)
    contact_friction: float = 0.5    # DSIM_OPT_PLANE: plane.urdf lateral_friction 1.0 x PyBullet's default 0.5 for the vehicle
    base_offset: Sequence[float] = (0.0, 0.0, 0.0)   # integrated COM -> the point PyBullet reports (base link COM), body frame
    arm: float = 0.0                 # URDF <properties arm=...> (BaseAviary.py:2058): the lever L of Physics.DYN's mixer
    dyn_mixer: int = DYN_MIXER_X     # Physics.DYN: which mixer of BaseAviary.py:1794-1803 (the examples' help text: "default: CF2X")
    reset_thrust: float = 0.0        # INDIControl.reset (INDIControl.py:127); 6DOF 0.3 (:232)
    reset_cmd: float = 0.0           # INDIControl.py:129; 6DOF 0.5 (:234)
    alloc: np.ndarray = field(default=None)  # type: ignore[assignment]

    def __post_init__(self):
        self.G1 = np.asarray(self.G1, dtype=np.float64)
        if self.alloc is None:
            self.alloc = self.default_alloc()
        if self.gnd_eff_h_clip == 0.0 and self.gnd_eff_coeff > 0.0:
            # BaseAviary.py:235 (commented out in the fork): height below which the
            # ground-effect boost would exceed the maximum thrust
            max_rpm = max(s * hi + c for s, hi, c in
                          zip(self.pwm2rpm_scale, self.pwm_max, self.pwm2rpm_const))
            max_thrust = self.n_act * self.kf * max_rpm ** 2
            self.gnd_eff_h_clip = 0.25 * self.prop_radius * math.sqrt(
                (15 * max_rpm ** 2 * self.kf * self.gnd_eff_coeff) / max_thrust)

    # ---- controller allocation matrix ------------------------------------
    def default_alloc(self) -> np.ndarray:
        """quad (or any type flown by the quad law): ``pinv(G1/0.05)`` exactly as
        INDIControl.py:459 computes it every call (a per-type constant).
        hexa: M1 of :meth:`wls_first_iteration`."""
        B = self.G1 / 0.05
        if self.kind != KIND_HEXA6DOF:          # the quad law, on four or (hexa_6DOF_simple) six actuators: [n_act][4]
            return np.linalg.pinv(B)
        return self.wls_first_iteration()[0]

    def wls_first_iteration(self):
        """First iteration of ``wls_alloc`` as INDIControl_6DOF calls it (free set = all actuators,
        ``u0 = (umin+umax)/2``, ``up = None``; wls_alloc.py:164-252, INDIControl_6DOF.py:607-628):
        ``p = lstsq(A, d)`` with ``A = [gamma Wv B; Wu]``, ``d = [gamma Wv (v - B u0); -Wu u0]``, so
        ``u_opt = u0 + p = M1 v + M4 u0`` with ``M1 = P1 diag(gamma Wv)``, ``M4 = I - M1 B - P2 diag(Wu)``
        and ``[P1 | P2] = pinv(A)``.  Evaluated in fp64 here so the kernel never forms the 1e8-scaled
        rows in fp32.  When ``u_opt`` is inside the +-1.0-slackened box (wls_alloc.py:255-259) the
        reference returns it after this one iteration."""
        B = self.G1 / 0.05
        Wv = np.array([1000, 1000, 0.1, 10, 10, 100.0])   # INDIControl_6DOF.py:614
        Wu = np.ones(self.n_act)                           # :615
        gam = 100000.0                                     # wls_alloc.py:125 (gamma_sq)
        A = np.vstack([gam * Wv[:, None] * B, np.diag(Wu)])
        P = np.linalg.pinv(A)
        M1 = P[:, :6] * (gam * Wv)[None, :]
        M4 = np.eye(self.n_act) - M1 @ B - P[:, 6:] * Wu[None, :]
        return M1, M4

    @property
    def rest_height(self) -> float:
        """z of the reported point (base link COM) when the vehicle stands level on the ground plane."""
        return float(self.collision_below) + float(self.base_offset[2])

    def to_c(self) -> TypeParamsC:
        c = TypeParamsC()
        c.kind, c.n_act = self.kind, self.n_act
        c.mass, c.kf, c.km = self.mass, self.kf, self.km
        for k in range(3):
            c.inertia[k] = self.inertia[k]
            c.att_gain[k] = self.att_gain[k]
            c.rate_gain[k] = self.rate_gain[k]
            c.drag_coeff[k] = self.drag_coeff[k]
            c.dw_coeff[k] = self.dw_coeff[k]
        for j in range(self.n_act):
            c.pwm2rpm_scale[j] = self.pwm2rpm_scale[j]
            c.pwm2rpm_const[j] = self.pwm2rpm_const[j]
            c.pwm_min[j], c.pwm_max[j] = self.pwm_min[j], self.pwm_max[j]
            c.rotor_spin[j] = self.rotor_spin[j]
            for k in range(3):
                c.rotor_pos[j][k] = self.rotor_pos[j][k]
                c.rotor_axis[j][k] = self.rotor_axis[j][k]
        n_out = self.G1.shape[0]
        for i in range(n_out):
            for j in range(self.n_act):
                c.G1[i][j] = self.G1[i, j]
        a = np.asarray(self.alloc)
        for j in range(self.n_act):
            for i in range(min(a.shape[1], MAX_ACT)):
                c.alloc[j][i] = a[j, i]
        if self.kind == KIND_HEXA6DOF:
            m4 = self.wls_first_iteration()[1]
            for j in range(self.n_act):
                for i in range(self.n_act):
                    c.alloc2[j][i] = m4[j, i]
        c.kp_pos, c.kd_pos = self.kp_pos, self.kd_pos
        c.gravity = self.gravity
        c.lin_damping, c.ang_damping = self.lin_damping, self.ang_damping
        c.max_coord_vel = self.max_coord_vel
        c.gnd_eff_coeff, c.prop_radius = self.gnd_eff_coeff, self.prop_radius
        c.gnd_eff_h_clip = self.gnd_eff_h_clip
        c.max_speed_kmh = self.max_speed_kmh
        c.collision_radius, c.collision_below = self.collision_radius, self.collision_below
        c.contact_friction = self.contact_friction
        for j in range(3):
            c.base_offset[j] = float(self.base_offset[j])
        c.arm, c.dyn_mixer = float(self.arm), int(self.dyn_mixer)
        return c

    @property
    def hover_pwm(self) -> float:
        """PWM at which the vertical thrust of a level vehicle equals its weight (noise-free)."""
        az = sum(self.rotor_axis[j][2] for j in range(self.n_act))
        rpm = math.sqrt(self.mass * self.gravity / (az * self.kf))
        return (rpm - self.pwm2rpm_const[0]) / self.pwm2rpm_scale[0]


def types_to_c_array(types: Sequence[DroneType]):
    arr = (TypeParamsC * len(types))()
    for i, t in enumerate(types):
        arr[i] = t.to_c()
    return arr


# ---------------------------------------------------------------------------
# built-in types (values cited to the reference's URDF assets)
# ---------------------------------------------------------------------------
_QUAD_SPIN = (-1.0, 1.0, -1.0, 1.0)          # BaseAviary.py:1527: -t0 + t1 - t2 + t3
_Z = (0.0, 0.0, 1.0)
_AERO = dict(                                 # robobee.urdf:31 / tello.urdf:27 (identical)
    drag_coeff=(9.1785e-7, 9.1785e-7, 10.311e-7),
    gnd_eff_coeff=11.36859,
    dw_coeff=(2267.18, 0.16, -0.11),
)


def _robobee() -> DroneType:
    # dronesim/assets/robobee.urdf: properties :31, control :33-53, base inertial :58-62,
    # prop link inertial origins :83,102,121,140
    return DroneType(
        name="robobee", kind=KIND_QUAD, n_act=4, mass=0.75, ctrl_mass=0.75,
        inertia=(6.2e-4, 6.2e-4, 1.1e-3), kf=2.0e-8, km=2.74e-10,
        pwm2rpm_scale=(20000.0,) * 4, pwm2rpm_const=(0.0,) * 4,
        pwm_min=(0.0,) * 4, pwm_max=(1.0,) * 4,
        rotor_pos=((0.11, 0.11, 0.0), (-0.11, 0.11, 0.0), (-0.11, -0.11, 0.03), (0.11, -0.11, 0.04)),
        rotor_axis=(_Z,) * 4, rotor_spin=_QUAD_SPIN,
        G1=np.array([[50.0, 50.0, -50.0, -50.0], [-50.0, 50.0, 50.0, -50.0],
                     [-7.0, 7.0, -7.0, 7.0], [1.7, 1.7, 1.7, 1.7]]),
        kp_pos=1.0, kd_pos=2.2, att_gain=(7.0, 7.0, 5.0), rate_gain=(18.0, 18.0, 10.0),
        prop_radius=3.31348e-2, collision_radius=0.15, collision_below=0.05, arm=0.0635, **_AERO,      # robobee.urdf:72-77, 31
    )


def _tello() -> DroneType:
    # dronesim/assets/tello.urdf: properties :27, control :29-49, base inertial :54-58,
    # prop link inertial origins :79,98,117,136
    a = 0.0475
    return DroneType(
        name="tello", kind=KIND_QUAD, n_act=4, mass=0.08, ctrl_mass=0.08,
        inertia=(4.28e-5, 4.28e-5, 8.36e-5), kf=2.0e-9, km=4.74e-12,
        pwm2rpm_scale=(20000.0,) * 4, pwm2rpm_const=(0.0,) * 4,
        pwm_min=(0.0,) * 4, pwm_max=(1.0,) * 4,
        rotor_pos=((a, a, 0.0), (-a, a, 0.0), (-a, -a, 0.0), (a, -a, 0.0)),
        rotor_axis=(_Z,) * 4, rotor_spin=_QUAD_SPIN,
        G1=np.array([[30.0, 30.0, -30.0, -30.0], [-30.0, 30.0, 30.0, -30.0],
                     [-5.0, 5.0, -5.0, 5.0], [1.7, 1.7, 1.7, 1.7]]),
        kp_pos=1.7, kd_pos=2.5, att_gain=(10.0, 10.0, 4.0), rate_gain=(12.0, 12.0, 7.0),
        prop_radius=3.31348e-2, collision_radius=0.0475, collision_below=0.0205, arm=0.0635, **_AERO,   # tello.urdf:68-73, 27
    )


def _hexa_6dof() -> DroneType:
    # dronesim/assets/hexa_6DOF.urdf: properties :27, control :29-53, link inertials :78-81, 94-98,
    # 238-241, 254-258, joints :382-476.  PyBullet flies it as an articulated body (six revolute arm
    # joints held by default motors); here it is ONE rigid body: total mass, and inertia / rotor lever
    # arms about the composite COM, computed by parse_urdf() from those lines (the state block holds
    # the base link's COM as PyBullet reports it; the composite COM, base_offset away, is what is integrated).  Rotor j: thrust along the prop link's z
    # (tilted +-0.3 rad about the arm), applied at the prop link's inertial origin.
    return DroneType(
        name="hexa_6DOF", kind=KIND_HEXA6DOF, n_act=6, mass=0.8600000000000003, ctrl_mass=0.2,
        inertia=(0.005362949147591898, 0.005381363215733879, 0.009256930068007832),
        kf=1.9e-8, km=1.9e-9,
        pwm2rpm_scale=(20000.0,) * 6, pwm2rpm_const=(0.0,) * 6, pwm_min=(0.0,) * 6, pwm_max=(1.0,) * 6,
        rotor_pos=((0.1128207649635189, -0.06522600911031864, 0.031918173037913264),
                   (0.0003321933335286087, -0.13000085698891597, 0.031918173037913264),
                   (-0.11309080300220549, -0.06479695890916311, 0.031918173037913264),
                   (-0.11309080300220549, 0.0647947362218831, 0.031918173037913264),
                   (0.0003321933335286087, 0.12999863430163597, 0.031918173037913264),
                   (0.1128207649635189, 0.06522378642303864, 0.031918173037913264)),
        rotor_axis=((-0.14760683340635503, -0.2560164355601206, 0.955336489125606),
                    (0.29552011296128977, 0.00023533063412583874, 0.955336489125606),
                    (-0.14801439152343665, 0.2557810244078079, 0.955336489125606),
                    (-0.14801439152343665, -0.2557810244078079, 0.955336489125606),
                    (0.29552011296128977, -0.00023533063412583874, 0.955336489125606),
                    (-0.14760683340635503, 0.2560164355601206, 0.955336489125606)),
        rotor_spin=(-1.0, 1.0, -1.0, 1.0, -1.0, 1.0),     # BaseAviary.py:1439-1440
        G1=np.array([[-7.5, -15.0, -7.5, 7.5, 15.0, 7.5], [-13.0, 0.0, 13.0, 13.0, 0.0, -13.0],
                     [-5.0, 5.0, -5.0, 5.0, -5.0, 5.0], [-2.0, 4.0, -2.0, -2.0, 4.0, -2.0],
                     [-3.0, 0.0, 3.0, -3.0, 0.0, 3.0], [1.5, 1.5, 1.5, 1.5, 1.5, 1.5]]),
        kp_pos=1.7, kd_pos=2.5, att_gain=(10.0, 10.0, 5.0), rate_gain=(18.0, 18.0, 12.0),
        prop_radius=6.7e-2, reset_thrust=0.3, reset_cmd=0.5, arm=1.0635,
        collision_radius=0.18986507827381124, collision_below=0.06903716345121234,   # all links' <collision> shapes
        base_offset=(-1.1106382076676865e-05, -1.111343640005967e-06, 0.010962836548787658),   # mainbody COM - composite COM
        **_AERO,
    )


def _hexa_6dof_simple() -> DroneType:
    # dronesim/assets/hexa_6DOF_simple.urdf: the same airframe, link for link, as hexa_6DOF.urdf; its <control> block
    # (:28-33) declares output_nr = 4 — roll, pitch, yaw, thrust rows of G1 over the six rotors — and
    # examples/fly_hexa_6DOF_simple.py:18 flies it with the QUAD controller class (INDIControl.py, actuator_nr = 6):
    # pinv(G1 / 0.05) is 6 x 4, the controller memory starts from zero like a quad's (INDIControl.py:127-129).
    t = _hexa_6dof()
    return DroneType(
        name="hexa_6DOF_simple", kind=KIND_HEXA_QUADLAW, n_act=6, mass=t.mass, ctrl_mass=t.ctrl_mass, inertia=t.inertia,
        kf=t.kf, km=t.km, pwm2rpm_scale=t.pwm2rpm_scale, pwm2rpm_const=t.pwm2rpm_const, pwm_min=t.pwm_min, pwm_max=t.pwm_max,
        rotor_pos=t.rotor_pos, rotor_axis=t.rotor_axis, rotor_spin=t.rotor_spin,
        G1=np.array([[-7.5, -15.0, -7.5, 7.5, 15.0, 7.5], [-13.0, 0.0, 13.0, 13.0, 0.0, -13.0],
                     [-5.0, 5.0, -5.0, 5.0, -5.0, 5.0], [1.7, 1.7, 1.7, 1.7, 1.7, 1.7]]),
        kp_pos=t.kp_pos, kd_pos=t.kd_pos, att_gain=t.att_gain, rate_gain=t.rate_gain,
        prop_radius=t.prop_radius, reset_thrust=0.0, reset_cmd=0.0, arm=t.arm,
        collision_radius=t.collision_radius, collision_below=t.collision_below, base_offset=t.base_offset, **_AERO,
    )


_BUILTIN_FACTORIES = {"robobee": _robobee, "tello": _tello, "hexa_6DOF": _hexa_6dof, "hexa_6DOF_simple": _hexa_6dof_simple}


def builtin_type(name: str) -> DroneType:
    """Constants of a shipped vehicle type by the name the reference uses for its
    URDF (``drone_model=["robobee"]``, examples/fly_INDI.py:33-38)."""
    try:
        return _BUILTIN_FACTORIES[name]()
    except KeyError:
        raise KeyError(
            f"unknown drone model {name!r}; built in: {sorted(_BUILTIN_FACTORIES)} "
            f"(use parse_urdf() for a custom URDF)") from None


def register_builtin(name: str, factory) -> None:
    _BUILTIN_FACTORIES[name] = factory


# ---------------------------------------------------------------------------
# URDF reader (the reference's dialect)
# ---------------------------------------------------------------------------
def _floats(s: str) -> List[float]:
    return [float(t) for t in s.split(" ") if t != ""]


def _rpy_matrix(rpy: Sequence[float]) -> np.ndarray:
    r, p, y = rpy
    cr, sr, cp, sp, cy, sy = math.cos(r), math.sin(r), math.cos(p), math.sin(p), math.cos(y), math.sin(y)
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def _origin(elem) -> tuple:
    if elem is None:
        return np.zeros(3), np.eye(3)
    xyz = np.array(_floats(elem.attrib.get("xyz", "0 0 0")))
    rpy = _floats(elem.attrib.get("rpy", "0 0 0"))
    return xyz, _rpy_matrix(rpy)


def parse_urdf(path: str) -> DroneType:
    """Read a vehicle URDF in the reference's dialect.

    Reads the same attributes as BaseAviary._parseURDFParameters
    (BaseAviary.py:2041-2140) and INDIControl._parseURDFControlParameters
    (INDIControl.py:55-106), and additionally walks the link/joint tree (which the
    reference leaves to PyBullet's loader, BaseAviary.py:681-694) to obtain the
    points and directions at which the rotor forces act: PyBullet numbers links
    depth-first in joint order, forces are applied in LINK_FRAME at the link's
    inertial origin (BaseAviary.py:1528-1536, 1442-1457).
    """
    root = ET.parse(path).getroot()
    name = os.path.splitext(os.path.basename(path))[0]
    conf = root.find("configuration").attrib["type"]
    prop = root.find("properties").attrib
    links = {l.attrib["name"]: l for l in root.findall("link")}
    joints = root.findall("joint")
    children: Dict[str, list] = {}
    child_names = set()
    for j in joints:
        children.setdefault(j.find("parent").attrib["link"], []).append(j)
        child_names.add(j.find("child").attrib["link"])
    base = next(n for n in links if n not in child_names)

    # depth-first link numbering + transform of each link frame in the base frame
    order: List[str] = []
    frames = {base: (np.zeros(3), np.eye(3))}

    def walk(parent):
        for j in children.get(parent, []):
            c = j.find("child").attrib["link"]
            jx, jR = _origin(j.find("origin"))
            px, pR = frames[parent]
            frames[c] = (px + pR @ jx, pR @ jR)
            order.append(c)
            walk(c)

    walk(base)

    def inertial(lname):
        ine = links[lname].find("inertial")
        if ine is None:
            return 0.0, np.zeros(3), np.eye(3), np.zeros(3)
        m = float(ine.find("mass").attrib["value"])
        ox, oR = _origin(ine.find("origin"))
        I = ine.find("inertia").attrib
        diag = np.array([float(I["ixx"]), float(I["iyy"]), float(I["izz"])])
        lx, lR = frames[lname]
        return m, lx + lR @ ox, lR @ oR, diag

    ctrl = root.find("control")
    indi = ctrl.find("indi").attrib
    n_act, n_out = int(indi["actuator_nr"]), int(indi["output_nr"])
    G1 = np.zeros((n_out, n_act))
    for i in range(n_out):  # children 1..n_out of <control> (BaseAviary.py:2097-2099)
        G1[i] = _floats(list(ctrl[i + 1].attrib.values())[0])
    gg = ctrl.find("indi_guidance_gains/pos").attrib
    att = ctrl.find("indi_att_gains/att").attrib
    rate = ctrl.find("indi_att_gains/rate").attrib
    p2r = list(ctrl.find("pwm/pwm2rpm").attrib.values())
    lim = list(ctrl.find("pwm/limit").attrib.values())

    is_hexa = "morphing_hexa" in conf
    if is_hexa and n_out not in (4, 6):
        raise ValueError(f"morphing_hexa with output_nr = {n_out}: neither the 6-DOF law (6) nor the quad law (4)")
    if not is_hexa and "quad" not in conf:
        raise ValueError(f"vehicle configuration {conf!r} is outside the hot path (quad / morphing_hexa only)")
    rotor_links = [order[i] for i in (range(1, 2 * n_act, 2) if is_hexa else range(n_act))]
    rotor_pos, rotor_axis = [], []
    for ln in rotor_links:
        _, com, R, _ = inertial(ln)
        rotor_pos.append(tuple(com))
        rotor_axis.append(tuple(R @ np.array([0.0, 0.0, 1.0])))

    m0, com0, _, diag0 = inertial(base)
    if is_hexa:
        # rigid composite of all links (arm joints treated as locked): total mass, COM, inertia
        ms, coms, Is = [], [], []
        for ln in [base] + order:
            m, com, R, d = inertial(ln)
            if m > 0:
                ms.append(m); coms.append(com); Is.append(R @ np.diag(d) @ R.T)
        M = float(sum(ms))
        C = sum(m * c for m, c in zip(ms, coms)) / M
        J = np.zeros((3, 3))
        for m, c, I in zip(ms, coms, Is):
            r = c - C
            J += I + m * (r @ r * np.eye(3) - np.outer(r, r))
        mass, inertia = M, tuple(np.diag(J))
        rotor_pos = [tuple(np.array(p) - C) for p in rotor_pos]
        spin = tuple(-1.0 if j % 2 == 0 else 1.0 for j in range(n_act))  # BaseAviary.py:1439-1440
        # output_nr = 6: the 6-DOF law (INDIControl_6DOF.py, reset 0.3 / 0.5 :232-234); output_nr = 4: the quad law on six
        # actuators (INDIControl.py, reset 0 / 0 :127-129) — hexa_6DOF_simple.urdf
        kind, rt, rc = (KIND_HEXA6DOF, 0.3, 0.5) if n_out == 6 else (KIND_HEXA_QUADLAW, 0.0, 0.0)
    else:
        mass, inertia = m0, tuple(diag0)
        spin = _QUAD_SPIN
        kind, rt, rc = KIND_QUAD, 0.0, 0.0

    # bounding cylinder (about the body z axis through the COM) of every <collision> shape of every link, with the
    # links in their URDF rest pose: radius and extent below the COM
    com_ref = C if is_hexa else com0
    coll_r = coll_below = 0.0
    for ln in [base] + order:
        for col in links[ln].findall("collision"):
            geo = col.find("geometry")
            ox, oR = _origin(col.find("origin"))
            lx, lR = frames[ln]
            ctr, Rs = lx + lR @ ox - com_ref, lR @ oR
            if geo.find("cylinder") is not None:
                r, h = float(geo.find("cylinder").attrib["radius"]), 0.5 * float(geo.find("cylinder").attrib["length"])
                az = abs(Rs[2, 2])
                down, out = h * az + r * math.sqrt(max(0.0, 1 - az * az)), r * az + h * math.sqrt(max(0.0, 1 - az * az))
            elif geo.find("sphere") is not None:
                down = out = float(geo.find("sphere").attrib["radius"])
            elif geo.find("box") is not None:
                half = 0.5 * np.array(_floats(geo.find("box").attrib["size"]))
                down, out = float(np.abs(Rs[2]) @ half), float(np.linalg.norm((np.abs(Rs[:2]) @ half)))
            else:
                continue        # meshes: not used by the shipped vehicles' collision shapes
            coll_below = max(coll_below, down - ctr[2])
            coll_r = max(coll_r, float(np.linalg.norm(ctr[:2])) + out)

    return DroneType(
        name=name, kind=kind, n_act=n_act, mass=mass, ctrl_mass=m0, inertia=inertia,
        kf=float(prop["kf"]), km=float(prop["km"]),
        pwm2rpm_scale=_floats(p2r[0]), pwm2rpm_const=_floats(p2r[1]),
        pwm_min=_floats(lim[0]), pwm_max=_floats(lim[1]),
        rotor_pos=rotor_pos, rotor_axis=rotor_axis, rotor_spin=spin, G1=G1,
        kp_pos=float(gg["kp"]), kd_pos=float(gg["kd"]),
        att_gain=(float(att["p"]), float(att["q"]), float(att["r"])),
        rate_gain=(float(rate["p"]), float(rate["q"]), float(rate["r"])),
        drag_coeff=(float(prop["drag_coeff_xy"]), float(prop["drag_coeff_xy"]), float(prop["drag_coeff_z"])),
        gnd_eff_coeff=float(prop["gnd_eff_coeff"]), prop_radius=float(prop["prop_radius"]),
        dw_coeff=(float(prop["dw_coeff_1"]), float(prop["dw_coeff_2"]), float(prop["dw_coeff_3"])),
        max_speed_kmh=float(prop["max_speed_kmh"]), arm=float(prop["arm"]),
        reset_thrust=rt, reset_cmd=rc,
        collision_radius=coll_r, collision_below=coll_below,
        base_offset=tuple(float(x) for x in (com0 - C)) if is_hexa else (0.0, 0.0, 0.0),
    )
