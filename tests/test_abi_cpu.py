"""CPU-side checks of the C-ABI library: it loads, exports every symbol the header declares,
agrees on struct sizes, and fails loudly (no fallback) without a device."""
import ctypes
import os
import re

import pytest

import __graft_entry__ as graft
from dronesim_amd import params


@pytest.fixture(scope="module")
def nat():
    graft.build()
    from dronesim_amd import _native
    return _native


def test_header_symbols_exported(nat):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "dronesim_amd.h")).read()
    declared = set(re.findall(r"^\s*(?:int|int64_t|const char\*)\s+(dsim_\w+)\s*\(", hdr, flags=re.M))
    assert declared == set(nat.EXPORTS), declared ^ set(nat.EXPORTS)
    lib = nat.load()
    for sym in declared:
        assert getattr(lib, sym) is not None
    assert lib.dsim_abi_version() == nat.ABI_VERSION
    assert lib.dsim_strerror(0) == b"ok"
    assert b"no HIP device" in lib.dsim_strerror(-3)


def test_struct_sizes_match_c(nat):
    from oracle import oracle as orc
    assert orc.lib().orc_sizeof_params() == ctypes.sizeof(params.TypeParamsC)
    assert ctypes.sizeof(nat.View) == 48
    assert ctypes.sizeof(nat.StepArgs) == 120


def test_no_cpu_fallback(nat):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a device is present")
    from dronesim_amd import fleet
    with pytest.raises(nat.DsimError):
        fleet.Context([params.builtin_type("robobee")])
    # and the raw ABI refuses too
    h = ctypes.c_void_p()
    arr = params.types_to_c_array([params.builtin_type("robobee")])
    assert nat.load().dsim_create(ctypes.byref(h), 0, arr, 1) == -3


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under dronesim_amd/ may reference it."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for dp, _, fs in os.walk(os.path.join(root, "dronesim_amd")):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "liboracle" not in src, f


@pytest.mark.parametrize("model", ["robobee", "tello"])
def test_builtin_types_match_reference_urdf(model):
    urdf = f"/root/reference/dronesim/assets/{model}.urdf"
    if not os.path.exists(urdf):
        pytest.skip("reference tree not present (GPU box)")
    a, b = params.builtin_type(model), params.parse_urdf(urdf)
    for f in ("kind", "n_act", "mass", "kf", "km", "kp_pos", "kd_pos", "prop_radius", "gnd_eff_coeff"):
        assert getattr(a, f) == getattr(b, f), f
    import numpy as np
    for f in ("inertia", "pwm2rpm_scale", "pwm2rpm_const", "pwm_min", "pwm_max", "rotor_pos", "rotor_axis",
              "rotor_spin", "G1", "att_gain", "rate_gain", "drag_coeff", "dw_coeff", "alloc"):
        np.testing.assert_allclose(np.asarray(getattr(a, f), dtype=float), np.asarray(getattr(b, f), dtype=float),
                                   rtol=0, atol=1e-15, err_msg=f)
