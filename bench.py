#!/usr/bin/env python3
"""bench.py — drone-steps/s of the fused Env.step()+INDI kernel on synthetic fleets.

    python bench.py --gpus N --steps K --warmup W [--workload NAME]
    (N>1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one Env.step() (phys_substeps physics sub-steps) + one INDI evaluation for
every drone of the rank's fleet = ONE launch of k_step_fast.  Inputs are resident in HBM
before the timed region.  Drones are independent, so ranks shard the fleet with no
data-path collective (weak scaling: per-GPU fleet fixed).

Workloads:
  config2x1024 (default)  BASELINE.json configs[1] — 4 096 robobee quads, INDI hover at own start,
                          initial action 0.4 — as 1 024 vectorised env replicas per GPU
                          (4 194 304 drones, 0.97 GB of state: exceeds the 256 MB Infinity
                          Cache, so the HBM fraction means something).  phys_substeps=1.
  config2                 the single 4 096-drone fleet (launch-latency bound; reported under
                          "also" in every run)
  config3                 65 536 robobee, per-drone targets (also under "also")
  config4                 configs[3]: 65 536 robobee per GPU (x8 = 524 288), hover, no coupling
  config5                 configs[4]: 65 536 per GPU, even index robobee / odd index hexa_6DOF, neighbour downwash,
                          slab shards with halo exchange of positions
  hexa, mixed             4 194 304 hexa_6DOF / interleaved quad+hexa drones (roofline-size variants)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
BYTES_PER_DRONE_STEP = 232      # read 13+11+10 floats, write 13+11 floats (quad, per-drone targets)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=200)
    p.add_argument("--warmup", type=int, default=20)
    p.add_argument("--workload", default="config2x1024", choices=["config2x1024", "config2", "config3", "config4", "config5", "hexa", "mixed"])
    p.add_argument("--substeps", type=int, default=1)
    p.add_argument("--layout", default="tile64", choices=["soa", "tile64", "tile256", "tile1024", "tile4096"])
    p.add_argument("--noise-seed", type=int, default=1)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-also", action="store_true")
    return p.parse_args()


def grid_fleet(n_fleet, replicas, pitch=1.0, z=0.5):
    """config-2 set-up: sqrt(n) x sqrt(n) grid, 1 m pitch, z = 0.5, level, at rest; replicated."""
    side = int(round(n_fleet ** 0.5))
    ij = np.arange(n_fleet)
    xyz = np.stack([(ij % side) * pitch, (ij // side) * pitch, np.full(n_fleet, z)], 1)
    return np.tile(xyz, (replicas, 1))


class Fleet:
    """A resident fleet + per-drone hover targets at the start position."""

    def __init__(self, n_fleet, replicas, device, substeps, layout, noise_seed, waypoints=False, n_steps=1,
                 config5=False, dist=None, rank=0, chained=False, hexa=False, mixed=False):
        import torch
        from dronesim_amd.envs import CtrlAviary, Physics
        from dronesim_amd.fleet import Targets, WaypointTargets
        self.torch = torch
        self.n_steps = n_steps
        self.graph = None
        xyz = grid_fleet(n_fleet, replicas)
        self.n = xyz.shape[0]
        models, physics, type_ids = ["robobee"], Physics.PYB, None
        if hexa:
            models = ["hexa_6DOF"]
        if mixed:       # config 5's fleet composition (even index quad, odd index hexa) without the downwash term
            models, type_ids = ["robobee", "hexa_6DOF"], (np.arange(self.n) % 2).astype(np.uint8)
            if mixed == "type_major":                  # the same fleet stored type-major (fleet.type_major_order)
                type_ids = np.sort(type_ids)
        if config5:
            # BASELINE configs[4]: even index robobee (quad INDI), odd index
            # hexa_6DOF (6DOF INDI + WLS); positions uniform in a 1024 x 512 x [0.5, 20.5] m box so that
            # downwash pairs exist; neighbour downwash on; this rank's shard of the world fleet
            # sharded runs: slab decomposition along x (rank r owns x in [r, r+1) * 1024/world), positions exchanged
            # between neighbouring slabs only (downwash.HaloExchange: grouped RCCL send/recv)
            rng = np.random.default_rng(1234 + rank)
            world = dist.get_world_size() if dist is not None else 1
            xyz = np.stack([rng.uniform(rank * 1024 / world, (rank + 1) * 1024 / world, self.n),
                            rng.uniform(0, 512, self.n), rng.uniform(0.5, 20.5, self.n)], 1)
            models, type_ids = ["robobee", "hexa_6DOF"], (np.arange(self.n) % 2).astype(np.uint8)
            physics = Physics.PYB_DW
        if waypoints:
            # config 3 (examples/fly_INDI_TrajectoryTrack.py): the reference's own 1200-row waypoint
            # table (fixture captured from its trajGen), gate 0 + grid offset, phase i*NUM_WP/6
            g = np.load(os.path.join(ROOT, "tests", "golden", "traj_track_waypoints.npz"))
            n_wp = g["target_pos"].shape[0]
            off = xyz.copy(); off[:, 2] = 0.0
            xyz = g["gates"][0][None, :] + off
            wp0 = (np.arange(self.n) * n_wp // 6) % n_wp
        self.env = CtrlAviary(models, self.n, initial_xyzs=xyz, aggregate_phy_steps=substeps, physics=physics,
                              device=device, layout=layout, noise_seed=noise_seed, dict_io=False, dist=dist,
                              chained=chained, downwash_exchange=os.environ.get("DSIM_DW_EXCHANGE", "halo"),
                              type_ids=type_ids)
        if waypoints:
            self.tgt = WaypointTargets(self.env.ctx, self.n, g["target_pos"], g["target_vel"], g["target_acc"],
                                       g["target_yaw"], wp_counters=wp0, offsets=off)
        else:
            self.tgt = Targets(self.env.ctx, self.n, layout)
            self.tgt.set(pos=xyz.T.astype(np.float32), yaw=0.4)
        # fly_INDI.py:214: the loop starts from action 0.4; afterwards the action is the controller's cmd
        self.env.step_fused(self.tgt, action=np.full((self.n, self.env.n_act), 0.4, dtype=np.float32))

    def step(self):
        if self.graph is not None:
            self.graph.replay()
        else:
            self.env.step_fused(self.tgt, n_steps=self.n_steps)

    def use_graph(self, steps):
        """Replay `steps` fused launches per step() call from one captured hipGraph."""
        self.graph = self.env.capture_fused(self.tgt, steps)
        self.n_steps = steps

    def timed(self, steps, warmup, barrier=None):
        torch = self.torch
        for _ in range(warmup):
            self.step()
        if barrier:
            barrier()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()                      # on torch's current stream == the stream the kernel is launched on
        for _ in range(steps):
            self.step()
        e1.record()
        if barrier:
            barrier()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        return wall, e0.elapsed_time(e1) * 1e-3


def host_threads():
    """Threads this process may really use: affinity mask, cgroup CPU quota, and the GPU
    box's per-GPU CPU share (16) — not the host's raw core count."""
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return max(1, min(n, 16))


def cpu_baseline(substeps, seconds=4.0):
    """The fp64 oracle (a scalar C port of the same step) timed on the host cores on bounded samples of the same
    workload: fleets of 4 096 and 65 536 drones, one thread and all usable threads each."""
    from dronesim_amd.params import builtin_type
    from oracle import oracle as orc
    t = builtin_type("robobee")
    O = orc.Oracle([t])
    dt = 1.0 / 240.0
    runs, fleets = {}, {}
    for n in (4096, 65536):
        xyz = grid_fleet(4096, n // 4096)
        rigid = np.concatenate([xyz, np.tile([0, 0, 0, 1.0], (n, 1)), np.zeros((n, 6))], 1)
        mem = O.reset_mem(n); mem[:, 7:11] = 0.4
        fleets[n] = (rigid, mem, np.concatenate([xyz, np.zeros((n, 6)), np.full((n, 1), 0.4)], 1))
    for nth in (1, host_threads()):                # one switch of the OpenMP team size, not one per sample
        for n in (4096, 65536):
            r, m, tg = (x.copy() for x in fleets[n])
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.5:                            # thread team start-up, untimed
                O.step(r, m, tg, substeps, dt, substeps * dt, nthreads=nth)
            k, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < seconds:                        # bounded sample
                O.step(r, m, tg, substeps, dt, substeps * dt, nthreads=nth)
                k += 1
            el = time.perf_counter() - t0
            runs[(n, nth)] = (n * k / el, k, el)
    nth = host_threads()
    v_all, k, el = runs[(65536, nth)]
    return {
        "value": v_all, "unit": "drone-steps/s", "cores": nth, "kind": "port",
        "sample": f"fp64 C oracle (oracle/dsim_oracle.c), 65536 robobee x {k} steps, phys_substeps={substeps}, "
                  f"{el:.1f} s on {nth} OpenMP threads",
        "single_thread_value": runs[(65536, 1)][0],
        "by_fleet_and_threads": {f"{n}_drones_{th}_threads": round(v[0]) for (n, th), v in runs.items()},
        "reference_python_note": "reference INDIControl.computeControl alone: 8.2e3 calls/s/core (SURVEY.md 6, survey "
                                 "container); PyBullet Env.step not measurable (engine absent)",
    }


def main():
    a = parse()
    import torch
    import __graft_entry__ as graft
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if rank == 0:
        graft.build()
    dist = None
    # rehearsal knob: DSIM_BENCH_BACKEND=gloo lets several ranks share one GPU (RCCL refuses that)
    backend = os.environ.get("DSIM_BENCH_BACKEND", "nccl")
    local = local % max(1, torch.cuda.device_count())
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
        dist.barrier()
    if rank != 0:
        graft.build()       # no-op once rank 0 has built
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    torch.cuda.set_device(local)
    barrier = (lambda: dist.barrier()) if dist else None

    n_fleet, replicas = {"config2x1024": (4096, 1024), "config2": (4096, 1), "config3": (65536, 1), "config4": (4096, 16),
                         "config5": (65536, 1), "hexa": (4096, 1024), "mixed": (4096, 1024)}[a.workload]
    from dronesim_amd import sharding
    # weak scaling: every rank owns a same-sized contiguous shard of the N-GPU fleet; no data-path collective
    # (config5 only: one halo exchange of positions per step for the neighbour-downwash term)
    fl = Fleet(n_fleet, replicas, local, a.substeps, a.layout, sharding.rank_seed(a.noise_seed, rank),
               waypoints=a.workload == "config3", config5=a.workload == "config5",
               dist=dist if a.workload == "config5" else None, rank=rank, hexa=a.workload == "hexa", mixed=a.workload == "mixed")
    wall, dev_s = fl.timed(a.steps, a.warmup, barrier)
    wall, dev_s = sharding.reduce_step_times(dist, "cuda" if backend == "nccl" else "cpu", wall, dev_s)  # MAX over ranks
    value = sharding.aggregate_throughput([fl.n] * world, a.steps, wall)
    launch_s = dev_s / a.steps
    # config 5: half quads (232 B) half hexas (248 B) + 1 B type id + the 12 B downwash force the step kernel reads;
    # its step is a chain of kernels (grid build, neighbour query, step, WLS fallback), timed as a whole
    bytes_per = {"config5": 253, "hexa": 248, "mixed": 241}.get(a.workload, BYTES_PER_DRONE_STEP)
    kernel = {"config5": "k_dw_count+scan+scatter+query, k_step_lean, k_wls_fallback",
              "hexa": "k_step_hexa (+ k_wls_fallback)",
              "mixed": "k_step_mixed (+ k_wls_fallback)"}.get(a.workload, "k_step_fast")
    achieved = fl.n * bytes_per / launch_s / 1e9

    if rank == 0:
        traffic = None
        tp = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tp):
            tj = json.load(open(tp))
            if tj.get("workload") == a.workload and tj.get("layout", "soa") == a.layout:
                traffic = tj.get("hbm_bytes_per_launch")
        out = {
            "metric": "drone-steps/sec (num_drones x env steps/s)", "value": value, "unit": "drone-steps/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": wall / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": {"config2x1024": "configs[1] 4096 robobee INDI hover x 1024 vectorised envs/GPU",
                                    "config2": "configs[1] 4096 robobee INDI hover (single fleet)",
                                    "config3": "65536 robobee, waypoint-table tracking (fly_INDI_TrajectoryTrack)",
                                    "config4": "configs[3] shard: 65536 robobee INDI hover per GPU (524288 over 8 GPUs), no coupling",
                                    "mixed": "even index robobee, odd index hexa_6DOF, 4096 x 1024 envs/GPU, no downwash",
                                    "hexa": "4096 hexa_6DOF (6-DOF INDI + WLS) hover x 1024 vectorised envs/GPU",
                                    "config5": "65536/GPU slab shard of 50% robobee + 50% hexa_6DOF, neighbour downwash on, "
                                               "halo exchange of positions between neighbouring slabs"}[a.workload],
                       "drones_per_gpu": fl.n, "phys_substeps": a.substeps, "layout": a.layout,
                       "noise_seed": a.noise_seed, "launches_per_step": 1, "parallelism": f"shard{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "kernel": kernel, "bytes_per_drone_step": bytes_per,
                         "launch_us": launch_s * 1e6},
        }
        if world == 1 and not a.no_also:
            also = {}
            # yardstick: achievable copy bandwidth on this device (read + write bytes / time)
            src = torch.empty(1 << 28, dtype=torch.float32, device="cuda"); dst = torch.empty_like(src)
            for _ in range(3):
                dst.copy_(src)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                dst.copy_(src)
            e1.record(); torch.cuda.synchronize()
            also["device_copy_GBps"] = 2 * src.numel() * 4 * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9
            del src, dst
            # (fleet, replicas, phys_substeps, waypoint table, Env.steps per launch)
            for name, (nf, rep, sub, wp, ns) in {
                    "config2_single_fleet_4096_sub5": (4096, 1, 5, False, 1),
                    "config2_single_fleet_4096_sub5_32steps_per_launch": (4096, 1, 5, False, 32),
                    "config2_single_fleet_4096_sub5_hipgraph_of_32_launches": (4096, 1, 5, False, 32),
                    "config3_65536_waypoints_sub2": (65536, 1, 2, True, 1),
                    "config3_65536_waypoints_sub2_hipgraph_of_32_launches": (65536, 1, 2, True, 32),
                    "config3_65536_waypoints_sub2_32steps_per_launch": (65536, 1, 2, True, 32),
                    # configs[3]: one GPU's 65 536-drone shard of the 524 288-drone fleet (no coupling between shards)
                    "config4_shard_65536_hover_sub1": (4096, 16, 1, False, 1),
                    "config2x1024_sub5": (4096, 1024, 5, False, 1),
                    "config5_shard_65536_mixed_downwash": (65536, 1, 1, False, 1),
                    # DSIM_OPT_CHAINED: the six controller-memory fields that are functions of the stored
                    # rigid state are neither read nor written: 184 B of real traffic per drone-step
                    "config2x1024_chained_184B": (4096, 1024, 1, False, 1),
                    # homogeneous morphing-hexa fleet: 6-DOF INDI + WLS allocation, 248 B/drone-step
                    "hexa_6DOF_4194304_indi6dof_wls": (4096, 1024, 1, False, 1),
                    # config 5's composition at roofline size, no downwash: 240 B average + 1 B type id
                    "mixed_quad_hexa_4194304": (4096, 1024, 1, False, 1),
                    # the same fleet in type-major storage: one single-type launch per type
                    "mixed_quad_hexa_4194304_type_major": (4096, 1024, 1, False, 1)}.items():
                f2 = Fleet(nf, rep, local, sub, a.layout, a.noise_seed, waypoints=wp, n_steps=ns,
                           config5=name.startswith("config5"), chained="chained" in name,
                           hexa=name.startswith("hexa"),
                           mixed=("type_major" if "type_major" in name else name.startswith("mixed")))
                if "hipgraph" in name:
                    f2.n_steps = 1
                    f2.use_graph(ns)
                k2 = max(20, a.steps // 2)
                w2, d2 = min(f2.timed(k2, 10), f2.timed(k2, 0))       # these timed regions are ~10 ms: best of two
                also[name] = {"drone_steps_per_s": f2.n * k2 * ns / w2, "launch_us": d2 / k2 * 1e6,
                              "env_steps_per_launch": ns}
                if ns == 1:
                    bts = 184 if "chained" in name else (248 if name.startswith("hexa") else
                                                         (241 if name.startswith("mixed") else BYTES_PER_DRONE_STEP))
                    also[name]["hbm_frac"] = f2.n * bts / (d2 / k2) / 1e9 / HBM_PEAK_GBPS
                    also[name]["bytes_per_drone_step"] = bts
                    if name.startswith("hexa"):
                        also[name]["note"] = ("248 B is the budgeted figure (SURVEY 8d); the 6-DOF law never reads the target "
                                              "acceleration and yaw, measured HBM traffic is 232 B per drone-step")
                f2.env.close(); del f2
            # the reference-shaped loop at the same size: obs = env.step(cmd); cmd = ctrl.computeControlFromState(...)
            # (three kernels + the [N,20] observation instead of the fused one: 480+ B per drone-step)
            from dronesim_amd.control import INDIControl
            from dronesim_amd.envs import CtrlAviary
            xyz = grid_fleet(4096, 1024)
            env = CtrlAviary(["robobee"], xyz.shape[0], initial_xyzs=xyz, aggregate_phy_steps=1, noise_seed=a.noise_seed,
                             dict_io=False, layout=a.layout, device=local)
            ctrl = INDIControl("robobee", env=env)
            tp = torch.from_numpy(np.ascontiguousarray(xyz.T.astype(np.float32))).to(env.ctx.device)
            cmd = torch.full((xyz.shape[0], 4), 0.4, device=env.ctx.device)
            for timed_pass in (False, True):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(20):
                    obs, _, _, _ = env.step(cmd)
                    cmd, _, _ = ctrl.computeControlFromState(1 / 240, None, target_pos=tp, target_rpy=np.array([0, 0, 0.4]))
                torch.cuda.synchronize(); el = time.perf_counter() - t0
            also["config2x1024_env_step_then_computeControl"] = {"drone_steps_per_s": xyz.shape[0] * 20 / el,
                                                                 "loop_us": el / 20 * 1e6}
            env.close(); del env, ctrl
            out["also"] = also
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.substeps)
        print(json.dumps(out))
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
