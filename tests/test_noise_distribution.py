"""The rotor-noise stream against what it stands for.  The reference draws np.random.normal(0, sigma) per rotor and sub-step
from numpy's unseeded global generator (BaseAviary.py:1518-1525, 1429-1432); the product's stream is counter-based and
LATTICE-valued (include/dronesim_amd.h: dsim_step_args.noise_seed): Box-Muller pairs on 65 536 x 65 536 points for launches
of one sub-step (and DSIM_OPT_NOISE_FINE), on 256 x 256 for launches of several (and DSIM_OPT_NOISE_COARSE).  Measured here on 1e7 draws of the definition (oracle/dsim_oracle.c:orc_noise_normals,
orc_noise_normals_fine — the kernels draw these very numbers: tests/test_gpu_noise.py): Kolmogorov distance from N(0, 1),
moments, mass beyond 3 sigma, independence.  CPU only."""
import math

import numpy as np
from scipy import stats

from dronesim_amd import params
from oracle import oracle as orc

N_DRAWS = 10_000_000
TAIL3 = 2 * stats.norm.sf(3.0)            # 2.6998e-3: mass of N(0, 1) beyond 3 sigma


def _draws(fine, n_act=4):
    O = orc.Oracle([params.builtin_type("robobee" if n_act == 4 else "hexa_6DOF")])
    per = 2 * n_act
    n_drones = 5000
    n_sub = -(-N_DRAWS // (per * n_drones))
    return O.noise_batch(0x5EED1234ABCD, 17, n_drones, 3, n_sub, n_act, fine=fine)       # [drones, subs, 2 n_act]


def _kolmogorov(z):
    """sup |F_n - Phi| of a sample, exact (both one-sided suprema)."""
    z = np.sort(z)
    n = z.size
    cdf = stats.norm.cdf(z)
    return max(float((np.arange(1, n + 1) / n - cdf).max()), float((cdf - np.arange(0, n) / n).max()))


def test_coarse_lattice_256x256_against_the_normal_distribution():
    """8 + 8 bits per pair, the lattice of launches of SEVERAL sub-steps.  What the coarse lattice costs, in numbers: the radius
    takes 256 values and the direction 256 (cell centres, (k + 1/2) / 256: no point is 0, so a single normal has NO atom — up to
    round 5 the points sat at cell corners and 3 / 256 of the mass was exactly 0), so a normal is a discrete distribution at
    Kolmogorov distance 1.4e-3 from N(0, 1) (a sample of 1e7 true normals would sit at ~3e-4); kurtosis 2.977, nothing beyond
    3.535 sigma, 2.75e-3 of the mass beyond 3 sigma (normal: 2.70e-3).  Mean, variance and the independence of the normals of a
    sub-step are exact properties of the construction."""
    z = _draws(False)
    flat = z.ravel()
    assert flat.size >= N_DRAWS
    assert abs(flat.mean()) < 1e-3 and abs(flat.var() - 1.0) < 2e-3
    assert abs((flat ** 3).mean()) < 5e-3 and abs((flat ** 4).mean() - 2.9767) < 0.01
    assert np.abs(flat).max() <= math.sqrt(2 * 1.0013550008475642 * math.log(512.0)) + 1e-9          # 3.5346
    ks = _kolmogorov(flat)
    assert ks < 3.0e-3, ks                               # <= 3e-3 (the corner lattice sat at 6.0e-3)
    assert 1.0e-3 < ks, ks                               # ... and measurably a lattice: population value 1.43e-3
    assert float((flat == 0.0).mean()) == 0.0 and np.abs(flat).min() > 7e-4      # no exact zeros: the smallest |n| is 7.7e-4
    tail = float((np.abs(flat) > 3.0).mean())
    assert 2.5e-3 < tail < 3.0e-3, tail
    c = np.corrcoef(z.reshape(-1, 8).T)
    assert np.abs(c - np.eye(8)).max() < 4e-3           # 1.25e6 sub-steps: sampling sigma 9e-4
    # ... and of consecutive sub-steps of one drone (an even sub-step and the odd one behind it share a Threefry block)
    lag = np.corrcoef(z[:, :-1, :].reshape(-1, 8)[:, 0], z[:, 1:, :].reshape(-1, 8)[:, 0])[0, 1]
    assert abs(lag) < 4e-3


def test_fine_lattice_65536x65536_against_the_normal_distribution():
    """16 + 16 bits per pair — the lattice of every launch of ONE sub-step (BASELINE's metric) and of DSIM_OPT_NOISE_FINE:
    indistinguishable from N(0, 1) at 1e7 draws by the Kolmogorov distance (the critical value at the 1 % level is
    1.63 / sqrt(n) = 5.2e-4), kurtosis 2.9998, support to 4.855 sigma, no exact zeros, the mass beyond 3 sigma that of the normal
    distribution within sampling error."""
    z = _draws(True)
    flat = z.ravel()
    assert abs(flat.mean()) < 1e-3 and abs(flat.var() - 1.0) < 2e-3
    assert abs((flat ** 3).mean()) < 5e-3 and abs((flat ** 4).mean() - 2.9998) < 0.012
    assert 4.0 < np.abs(flat).max() <= math.sqrt(2 * 1.0000052883115735 * math.log(131072.0)) + 1e-9      # 4.8546
    assert float((flat == 0.0).mean()) == 0.0
    ks = _kolmogorov(flat)
    assert ks < 1.63 / math.sqrt(flat.size), ks
    tail = float((np.abs(flat) > 3.0).mean())
    assert abs(tail - TAIL3) < 5 * math.sqrt(TAIL3 / flat.size), tail
    c = np.corrcoef(z.reshape(-1, 8).T)
    assert np.abs(c - np.eye(8)).max() < 4e-3


def test_hexa_streams_and_the_two_lattices_are_distinct_streams():
    """Six-actuator types draw SIX normals per sub-step on either lattice — those of the body wrench the twelve per-rotor normals of
    BaseAviary.py:1429-1430 add up to (dsim_device.h:noise_normals); rows 6 .. 11 of what the oracle hands out are zero.  The fine
    stream's blocks live in a counter domain of their own: no draw of it repeats a draw of the coarse stream."""
    O = orc.Oracle([params.builtin_type("hexa_6DOF")])
    a = O.noise_batch(77, 0, 2000, 0, 4, 6, fine=False)
    b = O.noise_batch(77, 0, 2000, 0, 4, 6, fine=True)
    assert a.shape == b.shape == (2000, 4, 12)
    for z in (a, b):
        assert np.all(z[:, :, 6:] == 0.0)
        z = z[:, :, :6]
        assert abs(z.mean()) < 0.01 and abs(z.var() - 1.0) < 0.02
        assert np.abs(np.corrcoef(z.reshape(-1, 6).T) - np.eye(6)).max() < 0.05
    assert abs(np.corrcoef(a[:, :, :6].ravel(), b[:, :, :6].ravel())[0, 1]) < 0.01
    assert np.abs(a).max() < 3.54 and np.abs(b).max() > 3.54
    # consecutive sub-steps share a block of the coarse stream (even: words 0, 1; odd: words 2, 3) and are independent draws
    lag = np.corrcoef(a[:, 0, :6].ravel(), a[:, 1, :6].ravel())[0, 1]
    assert abs(lag) < 0.03
    # the single-draw entry point agrees with the batch one
    np.testing.assert_array_equal(O.noise_normals(77, 5, 2, 6, fine=True), b[5, 2])
    np.testing.assert_array_equal(O.noise_normals(77, 5, 2, 6), a[5, 2])


def test_hexa_wrench_noise_has_the_covariance_of_the_per_rotor_noise():
    """W = L z (the device's factor, tests/util.py:hexa_noise_maps follows dsim_api.hip:to_dev) has the covariance of the wrench that twelve
    independent per-rotor normals — N(0, 0.01) on every rotor force, N(0, 0.001) on every rotor moment, BaseAviary.py:1429-1457 — add
    up to under the oracle's per-rotor map; and the per-rotor values the tests hand to the oracle reproduce W exactly."""
    from tests.util import hexa_noise_maps, rotor_noise
    for name in ("hexa_6DOF", "hexa_6DOF_simple"):
        t = params.builtin_type(name)
        L, P = hexa_noise_maps(t)
        r, a, sp = np.asarray(t.rotor_pos)[:6], np.asarray(t.rotor_axis)[:6], np.asarray(t.rotor_spin)[:6]
        M = np.zeros((6, 12))
        M[0:3, 0:6], M[3:6, 0:6], M[3:6, 6:12] = a.T, np.cross(r, a).T, (sp[:, None] * a).T
        sig = np.array([0.01] * 6 + [0.001] * 6)
        cov = (M * sig) @ (M * sig).T
        assert np.abs(L @ L.T - cov).max() < 2e-6 * np.abs(cov).max()             # (fp32 image of the geometry and of the factor)
        assert np.allclose(np.triu(L, 1), 0.0) and np.all(np.diag(L) > 0.0)
        z = np.random.default_rng(0).normal(size=6)
        f, m = rotor_noise(t, np.concatenate([z, np.zeros(6)]))
        assert np.abs(M @ np.concatenate([f, m]) - L @ z).max() < 1e-12 * np.abs(L).max()
        # sampled: the wrench of per-rotor draws against L z
        rng = np.random.default_rng(1)
        n12 = rng.normal(size=(200000, 12)) * sig
        Wa = n12 @ M.T
        Wb = rng.normal(size=(200000, 6)) @ L.T
        ca, cb = np.cov(Wa.T), np.cov(Wb.T)
        assert np.abs(ca - cb).max() < 0.02 * np.sqrt(np.outer(np.diag(cov), np.diag(cov))).max()


def test_population_moments_of_the_fine_lattice_are_exact():
    """Every (radius word, direction word) pair is equally likely: the population variance is 1 by the choice of
    ORC_BM16_CORR, the kurtosis 2.99982."""
    k = (np.arange(65536) + 0.5) / 65536.0
    r2 = -2.0 * 1.0000052883115735 * np.log(k)
    th = 2 * np.pi * (np.arange(65536) + 0.5) / 65536.0
    c2, c4 = (np.cos(th) ** 2).mean(), (np.cos(th) ** 4).mean()
    assert abs(r2.mean() * c2 - 1.0) < 1e-12
    assert abs((r2 ** 2).mean() * c4 / (r2.mean() * c2) ** 2 - 2.9998211245) < 1e-6
