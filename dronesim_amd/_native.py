"""ctypes binding of libdronesim_amd.so (the C-ABI in include/dronesim_amd.h).

There is no CPU fallback: if the shared library is missing or does not load the
import fails loudly, and ``Context`` fails when no HIP device is present.
"""
from __future__ import annotations

import ctypes
import os

from .params import TypeParamsC

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdronesim_amd.so")

# every symbol include/dronesim_amd.h declares
EXPORTS = (
    "dsim_abi_version", "dsim_strerror", "dsim_create", "dsim_destroy", "dsim_reset", "dsim_step",
    "dsim_physics", "dsim_control", "dsim_control2", "dsim_step_adaptor", "dsim_traj_sample", "dsim_materialize",
    "dsim_counter_add", "dsim_reserve", "dsim_observe", "dsim_observe_soa", "dsim_query", "dsim_downwash",
    "dsim_downwash_workspace", "dsim_downwash_prebin_ok", "dsim_downwash_reset", "dsim_adjacency", "dsim_wls_fallback", "dsim_fleet_bounds",
    "dsim_halo_pack", "dsim_downwash_workspace_halo", "dsim_dev_alloc", "dsim_dev_free", "dsim_noise_draw",
    "dsim_downwash_keep_workspace", "dsim_downwash_keep_ok", "dsim_downwash_keep_stats",
)

ABI_VERSION = 9
MAX_PEERS = 8
HALO_HDR = 8           # header floats of a halo message (DSIM_HALO_HDR)
DW_ALL, DW_LOCAL, DW_HALO_BIN, DW_HALO_QUERY = 0, 1, 2, 3
DW_KEEP_OFF, DW_KEEP_BUILD, DW_KEEP_REUSE = 0, 1, 2      # dsim_downwash_args.keep
NF_QUAD, NF_HEXA, NT = 24, 26, 10
OPT_DRAG, OPT_GROUND, OPT_BCAST_TGT, OPT_CHAINED = 1, 2, 4, 8
OPT_STREAM_ON, OPT_STREAM_OFF = 16, 32      # tuning knobs (results do not depend on them)
# option bits of measured-and-rejected kernel forms (tools/variants/ up to round 5, in the git history): reserved, ignored by the library
VAR_GENERIC, VAR_MIXED_V1, VAR_MIXED_RING, VAR_MIXED_V3, VAR_RUNS_SEPARATE = 64, 128, 256, 512, 1 << 13
TUNING_MASK = OPT_STREAM_ON | OPT_STREAM_OFF | VAR_GENERIC | VAR_MIXED_V1 | VAR_MIXED_RING | VAR_MIXED_V3 | VAR_RUNS_SEPARATE
OPT_PLANE = 1 << 10        # ground-plane contact (product-defined model, oracle/dsim_oracle.c:orc_plane_contact)
OPT_DEFER_FALLBACK = 1 << 11   # the caller launches dsim_wls_fallback itself (scheduling only)
OPT_CALLER_IO = 1 << 14    # dsim_physics / dsim_control2: action, rows, command, errors indexed by drone_id[i] (the caller's numbering)
OPT_ACTION_ROWS = 1 << 15  # dsim_step_adaptor: the action row-major [n][4]
OPT_DYN = 1 << 16          # Physics.DYN: BaseAviary._dynamics instead of the Bullet step (StepArgs.dyn_rpy_rates required)
OPT_DYN_BODY_RATES = 1 << 17   # ... with ang_v = R(quat) rpy_rates instead of the reference's placeholder (-1, -1, -1)
# rotor noise: which Box-Muller lattice a launch draws on (changes results).  Neither bit: the fine one (16 + 16 bits per pair) for a launch
# of ONE physics sub-step, the coarse one (8 + 8 bits) for a launch of several; the bits force one at any count (include/dronesim_amd.h)
OPT_NOISE_FINE = 1 << 18
OPT_NOISE_COARSE = 1 << 19
ADAPT_VELOCITY, ADAPT_RPYT = 0, 1
QUERY_WLS_FALLBACKS, QUERY_WLS_FAILURES, QUERY_GROUND_CONTACTS, QUERY_HALO_OVERFLOW = 0, 1, 2, 3
QUERY_DW_REUSES, QUERY_DW_MOVERS = 4, 5


class View(ctypes.Structure):
    _fields_ = [
        ("base", ctypes.c_void_p),
        ("n_pad", ctypes.c_int64),
        ("block", ctypes.c_int64),
        ("field_stride", ctypes.c_int64),
        ("block_stride", ctypes.c_int64),
        ("n_fields", ctypes.c_int32),
        ("_pad", ctypes.c_int32),
    ]


class StepArgs(ctypes.Structure):
    _fields_ = [
        ("phys_substeps", ctypes.c_int32),
        ("dt_phys", ctypes.c_float),
        ("dt_ctrl", ctypes.c_float),
        ("options", ctypes.c_uint32),
        ("noise_seed", ctypes.c_uint64),
        ("step_index", ctypes.c_uint64),
        ("noise_replay", ctypes.c_void_p),
        ("type_id", ctypes.c_void_p),
        ("action", ctypes.c_void_p),
        ("wp_table", ctypes.c_void_p),
        ("wp_counter", ctypes.c_void_p),
        ("wp_offset", ctypes.c_void_p),
        ("n_wp", ctypes.c_int32),
        ("n_steps", ctypes.c_int32),
        ("ext_force", ctypes.c_void_p),
        ("step_index_dev", ctypes.c_void_p),
        ("runs", ctypes.c_void_p),
        ("n_runs", ctypes.c_int32),
        ("obs_width", ctypes.c_int32),
        ("obs_out", ctypes.c_void_p),
        ("bin_next", ctypes.c_void_p),
        ("drone_id", ctypes.c_void_p),
        ("dyn_rpy_rates", ctypes.c_void_p),
    ]


class TypeRun(ctypes.Structure):
    _fields_ = [("first", ctypes.c_int64), ("count", ctypes.c_int64), ("type", ctypes.c_int32), ("_pad", ctypes.c_int32)]


class DownwashArgs(ctypes.Structure):
    _fields_ = [
        ("pos_all", ctypes.c_void_p),
        ("m", ctypes.c_int64),
        ("m_pad", ctypes.c_int64),
        ("xmin", ctypes.c_float),
        ("ymin", ctypes.c_float),
        ("cell", ctypes.c_float),
        ("nx", ctypes.c_int32),
        ("ny", ctypes.c_int32),
        ("workspace", ctypes.c_void_p),
        ("workspace_len", ctypes.c_int64),
        ("type_id", ctypes.c_void_p),
        ("local_offset", ctypes.c_int64),
        ("prebinned", ctypes.c_int32),
        ("phase", ctypes.c_int32),
        ("halo", ctypes.c_void_p),
        ("pairs_evaluated", ctypes.c_void_p),
        ("keep", ctypes.c_int32),
        ("keep_age", ctypes.c_int32),
        ("keep_skin", ctypes.c_float),
        ("keep_ws", ctypes.c_void_p),
        ("keep_ws_len", ctypes.c_int64),
    ]


class HaloPlan(ctypes.Structure):
    _fields_ = [
        ("world", ctypes.c_int32),
        ("rank", ctypes.c_int32),
        ("cap", ctypes.c_int64),
        ("send", ctypes.c_void_p),
        ("recv", ctypes.c_void_p),
        ("scratch", ctypes.c_void_p),
        ("send_cap", ctypes.c_int32 * MAX_PEERS),
        ("recv_cap", ctypes.c_int32 * MAX_PEERS),
        ("reach", ctypes.c_float * MAX_PEERS),
    ]


class DsimError(RuntimeError):
    pass


_lib = None


def load(path: str = None) -> ctypes.CDLL:
    """Load the HIP extension; raises if it has not been built (``__graft_entry__.build()``).  ``path``: a
    differently-tuned build of the same ABI (A/B runs); must be given on the first call."""
    global _lib
    if _lib is not None:
        return _lib
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()'). dronesim_amd has no CPU fallback.")
    # torch first: its wheel bundles the HIP runtime; loading ours afterwards binds to that same
    # libamdhip64 instead of pulling a second runtime into the process
    import torch  # noqa: F401
    lib = ctypes.CDLL(path)
    vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
    lib.dsim_abi_version.restype = ctypes.c_int
    lib.dsim_strerror.restype = ctypes.c_char_p
    lib.dsim_strerror.argtypes = [ctypes.c_int]
    lib.dsim_create.argtypes = [ctypes.POINTER(vp), ctypes.c_int, ctypes.POINTER(TypeParamsC), ctypes.c_int]
    lib.dsim_destroy.argtypes = [vp]
    lib.dsim_reset.argtypes = [vp, vp, i64, View, vp, vp, vp, vp, vp]
    lib.dsim_step.argtypes = [vp, vp, i64, View, View, ctypes.POINTER(StepArgs)]
    lib.dsim_physics.argtypes = [vp, vp, i64, View, vp, ctypes.POINTER(StepArgs)]  # (.., last_action_out, args)
    lib.dsim_control.argtypes = [vp, vp, i64, View, View, ctypes.POINTER(StepArgs), vp, vp]
    lib.dsim_control2.argtypes = [vp, vp, i64, View, View, ctypes.POINTER(StepArgs), vp, vp, vp]
    lib.dsim_downwash_prebin_ok.argtypes = [i64, i32, i32]
    lib.dsim_downwash_reset.argtypes = [ctypes.c_void_p]
    lib.dsim_step_adaptor.argtypes = [vp, vp, i64, View, vp, i32, vp, ctypes.POINTER(StepArgs)]
    lib.dsim_traj_sample.argtypes = [vp, vp, i64, vp, vp, i32, vp, ctypes.c_double, vp, vp, View]
    lib.dsim_materialize.argtypes = [vp, vp, i64, View]
    lib.dsim_counter_add.argtypes = [vp, vp, vp, ctypes.c_uint64]
    lib.dsim_reserve.argtypes = [vp, vp, i64]
    lib.dsim_observe.argtypes = [vp, vp, i64, View, vp, vp, i32]
    lib.dsim_observe_soa.argtypes = [vp, vp, i64, View, vp, vp, i32]
    lib.dsim_query.argtypes = [vp, vp, i32, ctypes.POINTER(ctypes.c_int64)]
    lib.dsim_downwash_workspace.restype = ctypes.c_int64
    lib.dsim_downwash_workspace.argtypes = [i64, i32, i32]
    lib.dsim_downwash.argtypes = [vp, vp, i64, View, ctypes.POINTER(DownwashArgs), vp]
    lib.dsim_adjacency.argtypes = [vp, vp, i64, View, ctypes.POINTER(DownwashArgs), ctypes.c_float, vp, vp, i32]
    lib.dsim_wls_fallback.argtypes = [vp, vp, i64, View, vp, vp]
    lib.dsim_fleet_bounds.argtypes = [vp, vp, i64, View, vp]
    lib.dsim_halo_pack.argtypes = [vp, vp, i64, View, ctypes.POINTER(HaloPlan)]
    lib.dsim_dev_alloc.argtypes = [vp, i64, ctypes.POINTER(vp)]
    lib.dsim_dev_free.argtypes = [vp, vp]
    lib.dsim_noise_draw.argtypes = [vp, vp, i64, i64, i32, ctypes.c_uint64, ctypes.c_uint64, i32, ctypes.c_uint32, vp, vp]
    lib.dsim_downwash_workspace_halo.restype = ctypes.c_int64
    lib.dsim_downwash_workspace_halo.argtypes = [i64, i64, i32, i32]
    lib.dsim_downwash_keep_workspace.restype = ctypes.c_int64
    lib.dsim_downwash_keep_workspace.argtypes = [i64, i32, i32]
    lib.dsim_downwash_keep_ok.argtypes = [i64, i32, i32, ctypes.c_float, ctypes.c_float]
    lib.dsim_downwash_keep_stats.argtypes = [vp] + [ctypes.POINTER(ctypes.c_int64)] * 4
    if lib.dsim_abi_version() != ABI_VERSION:
        raise ImportError(f"libdronesim_amd.so ABI {lib.dsim_abi_version()} != binding {ABI_VERSION}")
    _lib = lib
    return lib


def check(code: int) -> None:
    if code != 0:
        raise DsimError(f"dsim error {code}: {load().dsim_strerror(code).decode()}")
