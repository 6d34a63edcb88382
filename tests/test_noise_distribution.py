"""The rotor-noise stream against what it stands for.  The reference draws np.random.normal(0, sigma) per rotor and sub-step
from numpy's unseeded global generator (BaseAviary.py:1518-1525, 1429-1432); the product's stream is counter-based and
LATTICE-valued (include/dronesim_amd.h: dsim_step_args.noise_seed): Box-Muller pairs on 256 x 256 points by default,
65 536 x 65 536 with DSIM_OPT_NOISE_FINE.  Measured here on 1e7 draws of the definition (oracle/dsim_oracle.c:orc_noise_normals,
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


def test_default_lattice_256x256_against_the_normal_distribution():
    """8 + 8 bits per pair.  What the coarse lattice costs, in numbers: the radius takes 256 values and the direction 256, so
    a single normal is an atomic distribution — 3 / 256 of its mass sits at 0 (the two directions with cos = 0, resp.
    sin = 0, and the radius 0 of u1 = 1) — at Kolmogorov distance 6.0e-3 from N(0, 1) (a sample of 1e7 true normals would sit at ~3e-4); kurtosis 2.922,
    nothing beyond 3.354 sigma, 2.06e-3 of the mass beyond 3 sigma instead of 2.70e-3.  Mean, variance and the independence
    of the normals of a sub-step are exact properties of the construction."""
    z = _draws(False)
    flat = z.ravel()
    assert flat.size >= N_DRAWS
    assert abs(flat.mean()) < 1e-3 and abs(flat.var() - 1.0) < 2e-3
    assert abs((flat ** 3).mean()) < 5e-3 and abs((flat ** 4).mean() - 2.9221) < 0.01
    assert np.abs(flat).max() <= math.sqrt(2 * 1.0146323169375748 * math.log(256.0)) + 1e-9          # 3.3545
    ks = _kolmogorov(flat)
    assert 5.0e-3 < ks < 7.0e-3, ks                      # measurably not normal, and this close
    assert abs(float((np.abs(flat) < 1e-9).mean()) - (1 / 128 + 1 / 256 - 1 / 32768)) < 2e-4      # the atom at zero: a quarter turn, or u1 = 1
    tail = float((np.abs(flat) > 3.0).mean())
    assert 1.9e-3 < tail < 2.2e-3 and tail < TAIL3, tail
    c = np.corrcoef(z.reshape(-1, 8).T)
    assert np.abs(c - np.eye(8)).max() < 4e-3           # 1.25e6 sub-steps: sampling sigma 9e-4
    # ... and of consecutive sub-steps of one drone (an even sub-step and the odd one behind it share a Threefry block)
    lag = np.corrcoef(z[:, :-1, :].reshape(-1, 8)[:, 0], z[:, 1:, :].reshape(-1, 8)[:, 0])[0, 1]
    assert abs(lag) < 4e-3


def test_fine_lattice_65536x65536_against_the_normal_distribution():
    """16 + 16 bits per pair (DSIM_OPT_NOISE_FINE): indistinguishable from N(0, 1) at 1e7 draws by the Kolmogorov distance
    (the critical value at the 1 % level is 1.63 / sqrt(n) = 5.2e-4), kurtosis 2.9987, support to 4.71 sigma, the mass
    beyond 3 sigma that of the normal distribution within sampling error."""
    z = _draws(True)
    flat = z.ravel()
    assert abs(flat.mean()) < 1e-3 and abs(flat.var() - 1.0) < 2e-3
    assert abs((flat ** 3).mean()) < 5e-3 and abs((flat ** 4).mean() - 2.9987) < 0.012
    assert 4.0 < np.abs(flat).max() <= math.sqrt(2 * 1.000098644331326 * math.log(65536.0)) + 1e-9      # 4.7099
    ks = _kolmogorov(flat)
    assert ks < 1.63 / math.sqrt(flat.size), ks
    tail = float((np.abs(flat) > 3.0).mean())
    assert abs(tail - TAIL3) < 5 * math.sqrt(TAIL3 / flat.size), tail
    c = np.corrcoef(z.reshape(-1, 8).T)
    assert np.abs(c - np.eye(8)).max() < 4e-3


def test_hexa_streams_and_the_two_lattices_are_distinct_streams():
    """Twelve normals per hexa sub-step on either lattice (one block, resp. two); the fine stream's blocks live in a counter
    domain of their own: no draw of it repeats a draw of the default stream."""
    O = orc.Oracle([params.builtin_type("hexa_6DOF")])
    a = O.noise_batch(77, 0, 2000, 0, 4, 6, fine=False)
    b = O.noise_batch(77, 0, 2000, 0, 4, 6, fine=True)
    assert a.shape == b.shape == (2000, 4, 12)
    for z in (a, b):
        assert abs(z.mean()) < 0.01 and abs(z.var() - 1.0) < 0.02
        assert np.abs(np.corrcoef(z.reshape(-1, 12).T) - np.eye(12)).max() < 0.05
    assert abs(np.corrcoef(a.ravel(), b.ravel())[0, 1]) < 0.01
    assert np.abs(a).max() < 3.36 and np.abs(b).max() > 3.4
    # the single-draw entry point agrees with the batch one
    np.testing.assert_array_equal(O.noise_normals(77, 5, 2, 6, fine=True), b[5, 2])
    np.testing.assert_array_equal(O.noise_normals(77, 5, 2, 6), a[5, 2])


def test_population_moments_of_the_fine_lattice_are_exact():
    """Every (radius word, direction word) pair is equally likely: the population variance is 1 by the choice of
    ORC_BM16_CORR, the kurtosis 2.99867."""
    k = (np.arange(65536) + 1) / 65536.0
    r2 = -2.0 * 1.000098644331326 * np.log(k)
    th = 2 * np.pi * np.arange(65536) / 65536.0
    c2, c4 = (np.cos(th) ** 2).mean(), (np.cos(th) ** 4).mean()
    assert abs(r2.mean() * c2 - 1.0) < 1e-12
    assert abs((r2 ** 2).mean() * c4 / (r2.mean() * c2) ** 2 - 2.99867149) < 1e-6
