#!/usr/bin/env python3
"""A mixed fleet sharded over several GPUs with the neighbour-downwash term on (BASELINE configs[4]): every rank owns a
slab of the world along x with its own quads and hexas (handed over interleaved; the env stores them type-major by itself)
and exchanges only the positions of the drones next to a slab edge with its neighbours (the reference's `_downwash`,
BaseAviary.py:1736-1763, loops over the whole world in one process).  One process per GPU:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
        examples/fly_sharded_downwash_fleet.py --drones_per_rank 65536 --steps 480

(`--backend gloo` lets several ranks share one GPU: a rehearsal, positions then travel through the host.)
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from dronesim_amd import _native as nat  # noqa: E402
from dronesim_amd.envs import CtrlAviary, Physics  # noqa: E402
from dronesim_amd.fleet import Targets  # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--drones_per_rank", type=int, default=65536)
    ap.add_argument("--steps", type=int, default=240)
    ap.add_argument("--slab_m", type=float, default=128.0)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    A = ap.parse_args(argv)
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    if world > 1:
        if A.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo")
    n = A.drones_per_rank
    rng = np.random.default_rng(1234 + rank)
    xyz = np.stack([rng.uniform(rank * A.slab_m, (rank + 1) * A.slab_m, n), rng.uniform(0, 4 * A.slab_m, n),
                    rng.uniform(2.0, 22.0, n)], 1)
    tid = (np.arange(n) % 2).astype(np.uint8)                             # even index quad, odd index hexa
    env = CtrlAviary(["robobee", "hexa_6DOF"], n, initial_xyzs=xyz, physics=Physics.PYB_DW, dict_io=False, device=local,
                     type_ids=tid, layout="tile64", dist=dist if world > 1 else None, downwash_exchange="halo")
    tgt = Targets(env.ctx, n, "tile64")
    tgt.set(pos=xyz.T.astype(np.float32), yaw=0.0)                        # hold the start position against the others' wash
    env.step_fused(tgt, action=np.full((n, 6), 0.45, dtype=np.float32))
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(A.steps):
        env.step_fused(tgt)
    torch.cuda.synchronize()
    el = time.time() - t0
    r = env.state.rigid_aos()
    drift = np.linalg.norm(r[:, 0:3] - xyz, axis=1)
    halo = env._downwash.halo
    sys.stdout.write(f"rank {rank}/{world}: {n} drones x {A.steps} env steps in {el:.2f} s ({n * A.steps / el:.3g} drone-steps/s); "
          f"median drift from the hold point {np.median(drift):.3f} m; ground contacts {env.ground_contacts()}; WLS failures "
          f"{env.ctx.query(nat.QUERY_WLS_FAILURES)}"
          + (f"; ships {halo.sent_per_step} positions per step to {len(halo.messages())} neighbour(s), overflow {halo.overflow()}"
             if halo is not None else "") + "\n")
    sys.stdout.flush()
    env.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
