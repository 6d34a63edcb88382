"""Shared helpers for the parity tests (test infrastructure)."""
import math

import numpy as np

from oracle import oracle as orc

# natural scale of each quantity: |gpu - oracle| / (|oracle| + scale) is the relative
# error the 1e-4 bar of BASELINE.json is applied to (so that exact zeros do not divide)
RIGID_SCALE = np.array([1.0] * 3 + [1.0] * 4 + [1.0] * 3 + [1.0] * 3)      # m, -, m/s, rad/s
MEM_SCALE = np.array([1.0] * 3 + [1.0] * 3 + [1.0] + [1.0] * 6)


def rel_err(got, ref, scale):
    return np.abs(got - ref) / (np.abs(ref) + scale[: ref.shape[1]])


def f32(a):
    """Round to fp32-representable values so GPU (fp32) and oracle (fp64) see identical inputs."""
    return np.asarray(a, dtype=np.float32).astype(np.float64)


def random_fleet(rng, n, n_act=4, tilt=0.5, speed=2.0, rate=1.5, spread=50.0):
    """Seeded in-flight fleet state: rigid [n,13], mem [n,13], targets [n,10] (fp32-representable)."""
    rpy = np.stack([rng.uniform(-tilt, tilt, n), rng.uniform(-tilt, tilt, n), rng.uniform(-math.pi, math.pi, n)], 1)
    quat = np.stack([orc.quat_from_euler(r) for r in rpy]) if n <= 20000 else _quat_from_euler_np(rpy)
    pos = np.concatenate([rng.uniform(-spread, spread, (n, 2)), rng.uniform(0.5, 20.0, (n, 1))], 1)
    vel = rng.uniform(-speed, speed, (n, 3))
    om = rng.uniform(-rate, rate, (n, 3))
    rigid = f32(np.concatenate([pos, quat, vel, om], 1))
    mem = np.zeros((n, 13))
    mem[:, 0:3] = rigid[:, 7:10] + rng.uniform(-0.02, 0.02, (n, 3))
    mem[:, 3:6] = rng.uniform(-rate, rate, (n, 3))
    mem[:, 6] = rng.uniform(0.0, 1.0, n)
    mem[:, 7:7 + n_act] = rng.uniform(0.3, 0.7, (n, n_act))
    mem = f32(mem)
    tgt = np.concatenate([rigid[:, 0:3] + rng.uniform(-1, 1, (n, 3)), rng.uniform(-0.5, 0.5, (n, 3)),
                          rng.uniform(-0.5, 0.5, (n, 3)), rng.uniform(-4, 4, (n, 1))], 1)
    return rigid, mem, f32(tgt)


def _quat_from_euler_np(rpy):
    h = rpy / 2.0
    s, c = np.sin(h), np.cos(h)
    q = np.stack([
        s[:, 0] * c[:, 1] * c[:, 2] - c[:, 0] * s[:, 1] * s[:, 2],
        c[:, 0] * s[:, 1] * c[:, 2] + s[:, 0] * c[:, 1] * s[:, 2],
        c[:, 0] * c[:, 1] * s[:, 2] - s[:, 0] * s[:, 1] * c[:, 2],
        c[:, 0] * c[:, 1] * c[:, 2] + s[:, 0] * s[:, 1] * s[:, 2]], 1)
    return q / np.linalg.norm(q, axis=1, keepdims=True)
