"""Host-side check of dronesim_amd/csrc/dsim_math.h (the kernel's short elementary functions)
against libm on dense grids.  The header is compiled for the host with g++; on the device the
only difference is v_rcp/v_rsq/v_sqrt (1 ulp) in place of the IEEE division."""
import ctypes
import os
import subprocess
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r'''
#include "dsim_math.h"
extern "C" {
void t_sincos(const float* x, float* s, float* c, int n) { for (int i = 0; i < n; ++i) dsim_sincos(x[i], s + i, c + i); }
void t_atan2(const float* y, const float* x, float* o, int n) { for (int i = 0; i < n; ++i) o[i] = dsim_atan2(y[i], x[i]); }
void t_asin(const float* x, float* o, int n) { for (int i = 0; i < n; ++i) o[i] = dsim_asin(x[i]); }
}
'''


@pytest.fixture(scope="module")
def lib():
    d = tempfile.mkdtemp()
    cpp, so = os.path.join(d, "t.cpp"), os.path.join(d, "t.so")
    open(cpp, "w").write(SRC)
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-I",
                           os.path.join(ROOT, "dronesim_amd", "csrc"), cpp, "-o", so])
    return ctypes.CDLL(so)


def _p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def test_sincos(lib):
    x = np.concatenate([np.linspace(-40, 40, 400001), np.linspace(-3000, 3000, 100001),
                        [0.0, np.pi, -np.pi, np.pi / 2, 1e-20]]).astype(np.float32)
    s, c = np.zeros_like(x), np.zeros_like(x)
    lib.t_sincos(_p(x), _p(s), _p(c), len(x))
    xd = x.astype(np.float64)
    small = np.abs(xd) <= 40
    assert np.abs(s[small] - np.sin(xd[small])).max() < 2e-7
    assert np.abs(c[small] - np.cos(xd[small])).max() < 2e-7
    assert np.abs(s - np.sin(xd)).max() < 2e-4 and np.abs(c - np.cos(xd)).max() < 2e-4   # |x| to 3000: reduction error grows ~ |x| eps


def test_atan2(lib):
    rng = np.random.default_rng(0)
    y = np.concatenate([rng.normal(size=300000), [0, 0, 1, -1, 0.0, -0.0, 1e-30]]).astype(np.float32)
    x = np.concatenate([rng.normal(size=300000), [1, -1, 0, 0, 0.0, -1.0, 1e-30]]).astype(np.float32)
    o = np.zeros_like(x)
    lib.t_atan2(_p(y), _p(x), _p(o), len(x))
    ref = np.arctan2(y.astype(np.float64), x.astype(np.float64))
    assert np.abs(o - ref).max() < 4e-7
    # scaled arguments (the kernel passes 2(yz+wx) etc. of any magnitude)
    for sc in (1e-6, 1e6):
        ys, xs = (y * sc).astype(np.float32), (x * sc).astype(np.float32)
        lib.t_atan2(_p(ys), _p(xs), _p(o), len(x))
        assert np.abs(o - np.arctan2(ys.astype(np.float64), xs.astype(np.float64))).max() < 4e-7


def test_asin(lib):
    x = np.linspace(-1, 1, 400001).astype(np.float32)
    o = np.zeros_like(x)
    lib.t_asin(_p(x), _p(o), len(x))
    assert np.abs(o - np.arcsin(x.astype(np.float64))).max() < 3e-7
