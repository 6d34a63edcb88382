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


def test_struct_sizes_match_c(nat, tmp_path):
    """Every struct of the header: the ctypes mirror has the C compiler's size and field offsets (a plain-C
    translation unit that includes include/dronesim_amd.h, compiled here with gcc)."""
    import subprocess
    from oracle import oracle as orc
    assert orc.lib().orc_sizeof_params() == ctypes.sizeof(params.TypeParamsC)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    mirrors = {"dsim_view": nat.View, "dsim_step_args": nat.StepArgs, "dsim_type_run": nat.TypeRun,
               "dsim_downwash_args": nat.DownwashArgs, "dsim_type_params": params.TypeParamsC}
    lines = []
    for cname, cls in mirrors.items():
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "dronesim_amd.h"\nint main(void) {\n'
                   + "\n".join(lines) + "\nreturn 0; }\n")
    exe = tmp_path / "sizes"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-I", os.path.join(root, "include"), "-o", str(exe), str(src)])
    got = dict(ln.split() for ln in subprocess.check_output([str(exe)], text=True).splitlines())
    for cname, cls in mirrors.items():
        assert int(got[cname]) == ctypes.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(cls, fname).offset, (cname, fname)


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
    for f in ("kind", "n_act", "mass", "kf", "km", "kp_pos", "kd_pos", "prop_radius", "gnd_eff_coeff", "collision_radius",
              "collision_below"):
        assert getattr(a, f) == getattr(b, f), f
    import numpy as np
    for f in ("inertia", "pwm2rpm_scale", "pwm2rpm_const", "pwm_min", "pwm_max", "rotor_pos", "rotor_axis",
              "rotor_spin", "G1", "att_gain", "rate_gain", "drag_coeff", "dw_coeff", "alloc"):
        np.testing.assert_allclose(np.asarray(getattr(a, f), dtype=float), np.asarray(getattr(b, f), dtype=float),
                                   rtol=0, atol=1e-15, err_msg=f)
