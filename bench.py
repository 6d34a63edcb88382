#!/usr/bin/env python3
"""bench.py — drone-steps/s of the fused Env.step()+INDI kernel on synthetic fleets.

    python bench.py --gpus N --steps K --warmup W [--workload NAME]

N > 1: when the process was not started by a launcher (WORLD_SIZE unset) bench.py starts N ranks itself —
``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 bench.py ...`` as a CHILD
process, before anything in this process touches the GPU — one rank per device, RCCL (backend "nccl") over xGMI.
Started by such a launcher (RANK / LOCAL_RANK / WORLD_SIZE in the environment) it is one of the ranks.

One "step" = one Env.step() (phys_substeps physics sub-steps) + one INDI evaluation for
every drone of the rank's fleet = ONE launch of k_step_fast.  Inputs are resident in HBM
before the timed region.  Drones are independent, so ranks shard the fleet with no
data-path collective (weak scaling: per-GPU fleet fixed; N = 1 is exactly the BENCH workload).

Workloads:
  config2x1024 (default)  BASELINE.json configs[1] — 4 096 robobee quads, INDI hover at own start,
                          initial action 0.4 — as 1 024 vectorised env replicas per GPU
                          (4 194 304 drones, 0.97 GB of state: exceeds the 256 MB Infinity
                          Cache, so the HBM fraction means something).  phys_substeps=1.
  config2                 the single 4 096-drone fleet (launch-latency bound)
  config3                 65 536 robobee, waypoint-table tracking
  config4                 configs[3]: 65 536 robobee per GPU (x8 = 524 288), hover, no coupling
  config5                 configs[4]: 65 536 per GPU, even index robobee / odd index hexa_6DOF, neighbour downwash at the
                          config's density (524 288 drones in 1024 m x 512 m = one per m^2: every rank owns a
                          128 m x 512 m slab), halo exchange of positions between neighbouring slabs
  hexa, mixed             4 194 304 hexa_6DOF / interleaved quad+hexa drones (roofline-size variants)
The N = 1 line also carries the literal BASELINE configs 1-3 at their own sizes ("baseline_configs") and further
variants ("also").
"""
import argparse
import gc
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_COPY_YARDSTICK_GBPS = 6290.0    # MI355X_MICROARCH.md: the float4-copy rate that guide measured — a YARDSTICK, not a ceiling
                                    # (a kernel whose access shape suits the memory system better than a copy exceeds it)
BYTES_PER_DRONE_STEP = 232      # read 13+11+10 floats, write 13+11 floats (quad, per-drone targets)
# The neighbour query of config 5 is bound by vector issue, not by HBM.  Its unit of work is one (receiver, candidate) pair of
# formula P8 (BaseAviary.py:1752-1755): 20 vector instructions in the kernel's loop, 18 full-rate and 2 transcendental
# (v_rcp_f32, v_exp_f32).  Issue cost per wave64 instruction with 8 waves per SIMD, measured on MI355X with tools/valubench.hip
# (profiles/r05_valubench.json; nominal 2.4 GHz cycles): v_fma_f32 2.82, v_exp_f32 / v_rcp_f32 9.4 -> 69.6 cycles per 64 pairs
# per SIMD, 1 024 SIMDs: 2.26e12 pair evaluations per second if the vector pipes did nothing else.
VALU_PAIR_PEAK = 256 * 4 * 2.4e9 * 64 / (18 * 2.82 + 2 * 9.4)
WATCHDOG_RC = 3                 # exit code of every rank when a collective section hangs (Watchdog)
MIN_TIMED_S = 0.05              # the timed region is repeated (whole K-step regions) until it covers this much
SETTLE_S = 0.3                  # "also" entries of chip-filling fleets: under load for this long before the timed regions (Fleet.timed)
WORKLOADS = ["config2x1024", "config2", "config3", "config4", "config5", "hexa", "mixed", "mixed_type_major", "two_call_loop", "dyn"]


def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=200)
    p.add_argument("--warmup", type=int, default=20)
    p.add_argument("--workload", default="config2x1024", choices=WORKLOADS)
    p.add_argument("--substeps", type=int, default=1)
    p.add_argument("--layout", default="tile64", choices=["soa", "tile64", "tile256", "tile1024", "tile4096"])
    p.add_argument("--noise-seed", type=int, default=1)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-also", action="store_true")
    p.add_argument("--stream", choices=["auto", "on", "off"], default="auto",
                   help="nontemporal state accesses: DSIM_OPT_STREAM_ON/_OFF (A/B knob; default: the library's size rule)")
    p.add_argument("--slab-m", type=float, default=128.0,
                   help="config5: width of a rank's slab; 128 = the config's density, 1024 = round 1's definition of the line")
    p.add_argument("--lib", default=None, help="a differently-tuned build of libdronesim_amd.so (A/B runs)")
    p.add_argument("--mirror-peer", action="store_true",
                   help="config5 on ONE rank with a synthetic mirrored neighbour (MirrorDist): traces the device-paced exchange path")
    p.add_argument("--replicas", type=int, default=0, help="override the number of vectorised env replicas (A/B runs)")
    p.add_argument("--two-call-kind", default="quad", choices=["quad", "hexa", "mixed", "config5", "dyn"],
                   help="--workload two_call_loop: the fleet the reference-shaped loop runs on (4 194 304 quads / morphing hexas / "
                        "interleaved quads + hexas; or the 65 536-drone config-5 shard with the downwash term)")
    p.add_argument("--settle-seconds", type=float, default=0.0,
                   help="keep stepping, untimed, for this long behind the --warmup steps (Fleet.timed: the device's clocks "
                        "settle ~0.1 s into a vector-heavy stretch); 0 = exactly --warmup steps, the driver's contract")
    p.add_argument("--dry-run", action="store_true",
                   help="host logic only (launcher, rendezvous, reductions, the JSON line); no device work — CPU tests")
    return p.parse_args(argv)


# ----------------------------------------------------------------------------------------------------------------
# launcher: N ranks as child processes, started before this process touches the GPU
# ----------------------------------------------------------------------------------------------------------------
_REAL_STDOUT = None


def protect_stdout():
    """stdout carries the JSON line and nothing else: whatever a library prints on file descriptor 1 from here on (RCCL's version
    banner at communicator creation) goes to stderr."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def emit(line):
    if _REAL_STDOUT is None:
        print(line, flush=True)
    else:
        os.write(_REAL_STDOUT, (line + "\n").encode())


def launch_ranks(a, argv):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL needs it on this stack
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def grid_fleet(n_fleet, replicas, pitch=1.0, z=0.5):
    """config-2 set-up: sqrt(n) x sqrt(n) grid, 1 m pitch, z = 0.5, level, at rest; replicated."""
    side = int(round(n_fleet ** 0.5))
    ij = np.arange(n_fleet)
    xyz = np.stack([(ij % side) * pitch, (ij // side) * pitch, np.full(n_fleet, z)], 1)
    return np.tile(xyz, (replicas, 1))


class Fleet:
    """A resident fleet + per-drone hover targets at the start position."""

    def __init__(self, n_fleet, replicas, device, substeps, layout, noise_seed, waypoints=False, n_steps=1,
                 config5=False, dist=None, rank=0, chained=False, hexa=False, mixed=False, options=0, slab_m=128.0, storage=None,
                 dyn=False, dw_keep=None):
        import torch
        from dronesim_amd.envs import CtrlAviary, Physics
        from dronesim_amd.fleet import Targets, WaypointTargets
        self.torch = torch
        self.n_steps = n_steps
        self.graph = None
        self.loop = None
        self.ctrl = None
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
            # BASELINE configs[4]: even index robobee (quad INDI), odd index hexa_6DOF (6DOF INDI +
            # WLS); 524 288 drones uniform in a 1024 x 512 x [0.5, 20.5] m box = one drone per m^2 of ground, neighbour
            # downwash on.  Slab decomposition along x: every rank owns slab_m = 1024/8 = 128 m of it with its 65 536
            # drones, whatever the number of ranks in this run (weak scaling at the config's own density: per-GPU work —
            # about 900 candidate pairs per drone — does not change with N); positions are exchanged between
            # neighbouring slabs only (downwash.HaloExchange: grouped RCCL send/recv)
            rng = np.random.default_rng(1234 + rank)
            xyz = np.stack([rng.uniform(rank * slab_m, (rank + 1) * slab_m, self.n),
                            rng.uniform(0, 512, self.n), rng.uniform(0.5, 20.5, self.n)], 1)
            models, type_ids = ["robobee", "hexa_6DOF"], (np.arange(self.n) % 2).astype(np.uint8)
            physics = Physics.PYB_DW
        if dyn:         # Physics.DYN: the reference's own explicit model (BaseAviary._dynamics), flyable form of ang_v
            physics = Physics.DYN
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
                              type_ids=type_ids, options=options, storage=storage or os.environ.get("DSIM_STORAGE", "auto"),
                              downwash_split={"": None, "0": False, "1": True}[os.environ.get("DSIM_DW_SPLIT", "")],
                              defer_fallback=os.environ.get("DSIM_DEFER_FB", "0") != "0",
                              placement=os.environ.get("DSIM_PLACEMENT", "0") != "0",
                              dyn_ang_vel="body_rates" if dyn else "reference", downwash_keep=dw_keep)
        if waypoints:
            self.tgt = WaypointTargets(self.env.ctx, self.n, g["target_pos"], g["target_vel"], g["target_acc"],
                                       g["target_yaw"], wp_counters=wp0, offsets=off)
        else:
            self.tgt = Targets(self.env.ctx, self.n, layout)
            self.tgt.set(pos=xyz.T.astype(np.float32), yaw=0.4)
        # fly_INDI.py:214: the loop starts from action 0.4; afterwards the action is the controller's cmd
        self.env.step_fused(self.tgt, action=np.full((self.n, self.env.n_act), 0.4, dtype=np.float32))

    def step(self):
        if self.loop is not None:
            self.loop()
        elif self.graph is not None:
            self.graph.replay()
        else:
            self.env.step_fused(self.tgt, n_steps=self.n_steps)

    def make_two_call_loop(self):
        """step() = one iteration of the reference's example loop (examples/fly_INDI.py:223-239) through the two
        reference-shaped surfaces: Env.step(action) -> obs, then computeControlFromState -> action."""
        from dronesim_amd.control import INDIControl
        from dronesim_amd.fleet import frozen
        torch = self.torch
        tpos = frozen(self.tgt.fields(0, 3).contiguous())      # the same hover target every iteration: copied once
        if getattr(self.tgt, "behind_the_state", False):
            # this loop reads the CONTROLLER's targets beside the state, not the fused step's: the room behind the state block
            # (fleet.FleetState) goes back to the context for them, as in a program that never made a fused-step Targets
            room = self.tgt.data.view(-1)
            room.zero_()
            self.env.ctx.read_room = (room, self.tgt.n_pad, self.tgt.layout)
            self.tgt = None
        ctrl = self.ctrl = INDIControl(self.env.types[-1].name, env=self.env)     # one controller object for the whole fleet, whatever its types
        state = {"cmd": torch.full((self.n, self.env.n_act), 0.4, device=self.env.ctx.device)}
        yaw = np.array([0, 0, 0.4])

        def one():
            obs, _, _, _ = self.env.step(state["cmd"])
            state["cmd"], _, _ = ctrl.computeControlFromState(self.env.TIMESTEP * self.env.AGGR_PHY_STEPS, None, target_pos=tpos,
                                                               target_rpy=yaw)
        self.loop = one

    def use_graph(self, steps):
        """Replay `steps` fused launches per step() call from one captured hipGraph."""
        self.graph = self.env.capture_fused(self.tgt, steps)
        self.n_steps = steps

    def timed(self, steps, warmup, barrier=None, min_s=0.0, repeat_rule=None, settle_s=0.0):
        """W untimed warm-up steps, then regions of EXACTLY `steps` steps each, bracketed by barrier + device
        synchronisation on both sides, repeated until they cover `min_s` seconds.  Returns (wall seconds, device
        seconds, regions): sums over the regions; the device seconds come from events on the launch stream.
        repeat_rule(region_wall_s) -> total number of regions (so that every rank runs the same number).
        settle_s: keep stepping, untimed, for this long behind the warm-up steps.  A device that has been idle runs its first
        launches at its top clock, throttles hard a few milliseconds into a vector-heavy stretch and settles ~100 ms later
        (tools/clock_probe.py, profiles/r05_clock_*.txt: five sub-steps of 4 194 304 quads 154 us -> up to 238 us -> 169.7 us
        for as long as the load lasts); a region timed right behind a short warm-up measures that episode, not the kernel."""
        torch = self.torch
        for _ in range(warmup):
            self.step()
        if settle_s > 0.0:
            return self._timed_settled(steps, barrier, min_s, repeat_rule, settle_s)
        # the collector stays off inside the timed regions (as timeit does): a collection that frees another variant's
        # graphs or fleet in the middle of a region stalls the host for tens of milliseconds (seen: 3x wall vs device)
        gc.collect()
        torch.cuda.synchronize()
        gc_was_on = gc.isenabled()
        gc.disable()
        wall_sum = dev_sum = 0.0
        regions, want = 0, 1
        try:
            while regions < want:
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
                wall_sum += wall
                dev_sum += e0.elapsed_time(e1) * 1e-3
                regions += 1
                if regions == 1 and min_s > 0.0:
                    want = repeat_rule(wall) if repeat_rule else max(1, int(np.ceil(min_s / max(wall, 1e-9))))
        finally:
            if gc_was_on:
                gc.enable()
        return wall_sum, dev_sum, regions


    def _timed_settled(self, steps, barrier, min_s, repeat_rule, settle_s):
        """The settled form of timed(): the device is kept under THIS load for settle_s and the timed launches follow without
        a gap — the events that bracket them are recorded in the stream, the host synchronises only behind the last one.
        (A synchronisation in front of every region lets the device idle for a moment, and a vector-heavy kernel then pays a
        piece of the throttling episode again at every region's start: regions of 100 launches of the five-sub-step kernel
        179 us per launch, the same launches back to back 168 us — 71 000 of them in 12.04 s of wall clock,
        profiles/r05_clock_*.txt.)  Returns (device seconds, device seconds, regions): there is no host clock around a
        region that begins in the middle of a busy stream."""
        torch = self.torch
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            self.step()
        e1.record()
        torch.cuda.synchronize()
        t_launch = max(e0.elapsed_time(e1) * 1e-3 / 20, 1e-7)
        regions = repeat_rule(t_launch * steps) if repeat_rule else max(1, int(np.ceil(min_s / (t_launch * steps))))
        n_settle = int(np.ceil(settle_s / t_launch))
        gc.collect()
        gc_was_on = gc.isenabled()
        gc.disable()
        try:
            if barrier:
                barrier()
            for _ in range(n_settle):
                self.step()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(steps * regions):
                self.step()
            e1.record()
            if barrier:
                barrier()
            torch.cuda.synchronize()
        finally:
            if gc_was_on:
                gc.enable()
        dev = e0.elapsed_time(e1) * 1e-3
        return dev, dev, regions

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


class MirrorDist:
    """A synthetic neighbour for ONE rank (``--mirror-peer``): the world continues mirrored beyond the slab edge x = edge, so
    "the other rank" is this rank's own reflection — its box, its counts and its messages are this rank's, mirrored.  It
    stands in for torch.distributed in downwash.HaloPlan with a wire that never leaves the device (a copy and three small
    element-wise kernels on the side stream), so that the device-paced exchange path — select + pack, wire, halo binning
    beside the local pass of the query, halo pass — can be traced and timed on one GPU without gloo's host staging.  Not a
    multi-GPU measurement: there is no second device and no RCCL in it."""

    class _Done:
        def wait(self):
            return None

    class P2POp:
        def __init__(self, op, tensor, peer):
            self.op, self.tensor, self.peer = op, tensor, peer

    isend, irecv = "isend", "irecv"

    def __init__(self, edge):
        self.edge = float(edge)

    def is_initialized(self):
        return True

    def get_rank(self):
        return 0

    def get_world_size(self):
        return 2

    def get_backend(self):
        return "mirror"

    def all_gather_into_tensor(self, out, mine):
        out[0] = mine.reshape(-1)
        if out.shape[1] == 5:                       # xmin ymin xmax ymax vmax -> the reflection's box
            out[1, 0], out[1, 2] = 2 * self.edge - mine.reshape(-1)[2], 2 * self.edge - mine.reshape(-1)[0]
            out[1, 1], out[1, 3], out[1, 4] = mine.reshape(-1)[1], mine.reshape(-1)[3], mine.reshape(-1)[4]
        else:                                       # counts [to rank 0, to rank 1] -> the reflection ships what I ship
            out[1] = mine.reshape(-1).flip(0)

    def batch_isend_irecv(self, ops):
        """The wire: TWO device ops per step (anything more and the Python side of this stand-in, not the device, would pace
        the loop): the count word, and the payload with x -> 2 edge - x.  The box words of the header keep what the last
        resize seeded (the reflection's box as of then) — a synthetic neighbour does not need more."""
        import torch
        send = [o.tensor for o in ops if o.op == "isend"]
        recv = [o.tensor for o in ops if o.op == "irecv"]
        for s_, r_ in zip(send, recv):
            k = s_.numel() - 8
            if getattr(self, "_k", None) != k:
                self._scale = torch.ones(k, device=s_.device); self._scale[0::3] = -1.0
                self._off = torch.zeros(k, device=s_.device); self._off[0::3] = 2 * self.edge
                self._k = k
            r_[0:1].view(torch.int32).copy_(s_[0:1].view(torch.int32))
            torch.addcmul(self._off, s_[8:], self._scale, out=r_[8:])
        return [MirrorDist._Done()]


def memory_yardstick(n_drones):
    """tools/membench --json as a child process (built by __graft_entry__.build()): {} when it cannot run."""
    exe = os.path.join(ROOT, "tools", "membench")
    try:
        out = subprocess.run([exe, "--json", str(int(n_drones))], capture_output=True, text=True, timeout=120)
        d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
        return {"device_copy_GBps": d["float4_copy_GBps"], "access_shape_floor_us": d["access_shape_floor_us"],
                "access_shape_GBps": d["access_shape_GBps"], "yardstick_source": "tools/membench.hip (float4 copy; the "
                "headline kernel's 58 dword accesses per lane, wave-tiled layout, streaming, no arithmetic)"}
    except Exception as e:      # a missing probe must not cost the headline
        return {"device_copy_GBps": None, "yardstick_error": repr(e)[:200]}


def child_line(a, extra, placement=False, timeout=300):
    """`bench.py <extra>` as a child process (this one keeps its fleets but is idle meanwhile), placement by trial on or off
    (DSIM_PLACEMENT): its JSON line."""
    cmd = [sys.executable, os.path.abspath(__file__), "--warmup", str(a.warmup), "--no-also", "--no-cpu-baseline", "--layout", a.layout,
           "--noise-seed", str(a.noise_seed), "--stream", a.stream] + (["--lib", a.lib] if a.lib else []) + list(extra)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                            "DSIM_BENCH_FORCE_DIST")}
    env["DSIM_PLACEMENT"] = "1" if placement else "0"
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
    return json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])


def placement_cost(log):
    """What the searches of a run cost: seconds spent, the most device memory held at once (transient), and what stays
    held for the life of the fleet (the room for two target blocks behind a placed fleet's state block)."""
    rows = [r for r in (log or []) if "seconds" in r]
    return {"placement_s": round(sum(r["seconds"] for r in rows), 3),
            "placement_peak_bytes": max([max(r.get("peak_bytes", 0), r.get("bytes", 0) if "held_bytes" in r else 0) for r in rows] + [0]),
            "placement_held_bytes": sum(r.get("held_bytes", 0) for r in (log or [])), "searches": len(rows)}


def two_call_child(a, kind="quad", placement=False):
    """`bench.py --workload two_call_loop --two-call-kind KIND` as a child process (this one keeps its fleets but is idle
    meanwhile): its line, reduced to the entry the default line carries."""
    cmd = [sys.executable, os.path.abspath(__file__), "--workload", "two_call_loop", "--two-call-kind", kind,
           "--steps", str(max(50, a.steps // 2)),
           "--warmup", str(a.warmup), "--no-also", "--no-cpu-baseline", "--layout", a.layout, "--noise-seed", str(a.noise_seed),
           "--stream", a.stream, "--settle-seconds", str(SETTLE_S)] + (["--lib", a.lib] if a.lib else [])
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                            "DSIM_BENCH_FORCE_DIST")}
    env["DSIM_PLACEMENT"] = "1" if placement else "0"
    try:
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
        d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
        e = {"drone_steps_per_s": d["value"], "loop_us": d["ms_per_step"] * 1e3, "steps_timed": d["steps_timed"],
             "drones": d["config"]["drones_per_gpu"], "loop_us_device": d["roofline"]["launch_us"],
             "bytes_per_drone_step": TWO_CALL_BYTES[kind][0], "hbm_frac": d["roofline"]["frac"],
             "kernel": d["roofline"]["kernel"], "placement_by_trial": bool(placement), "settled_for_s": d.get("settle_seconds"),
             "protocol": "settled" if d.get("settle_seconds") else "from_idle",
             "measured_in": f"a child process running `bench.py --workload two_call_loop --two-call-kind {kind}` alone"}
        if placement:
            e.update(note=TWO_CALL_BYTES[kind][1], placement=d.get("placement"), placement_cost=placement_cost(d.get("placement")))
        return e
    except Exception as e:          # an extra must not cost the headline
        return {"error": repr(e)[:300]}


def self_check(world, gpus, backend_seen, backend_wanted, devices, rows_launch_us, extra=None, distinct_devices=True):
    """What a first run on N devices must show for its line to mean what it says; every entry True or the line says which
    is not (and the run leaves with SELF_CHECK_RC).  `devices`: the device ordinal every rank used."""
    checks = {
        "world_size_equals_gpus_flag": world == gpus,
        "backend_is_" + str(backend_wanted): (backend_seen == backend_wanted) if world > 1 else True,
        "every_rank_reported": len(devices) == world and len(rows_launch_us) == world,
        # (RCCL wants one device per rank; a gloo rehearsal may share one GPU between its ranks: distinct_devices = False)
        "one_device_per_rank": (len(set(int(d) for d in devices)) == len(devices)) if distinct_devices else True,
        "every_rank_timed_something": all(u > 0.0 for u in rows_launch_us),
    }
    checks.update(extra or {})
    failed = sorted(k for k, v in checks.items() if not v)
    return {"ok": not failed, "failed": failed, "checks": checks}


SELF_CHECK_RC = 4               # exit code when a multi-rank run's self-check fails (the line is still printed)


def gather_ranks(dist, red_dev, values):
    """[world][len(values)] of every rank's numbers (a small all-gather; world 1: the values themselves)."""
    import torch
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=red_dev)
    if dist is None:
        return [t.tolist()]
    out = torch.zeros((dist.get_world_size(), len(values)), dtype=torch.float64, device=red_dev)
    dist.all_gather_into_tensor(out, t.reshape(1, -1))
    return out.tolist()


def exchange_report(fl, dist, red_dev, steps=20):
    """config 5 on several ranks: what one step's exchange moves, and how long its side-stream part (select + pack,
    send/recv, halo binning) takes — timed with events on the side stream over `steps` extra steps after the timed regions."""
    import torch
    dwn = fl.env._downwash
    hp = dwn.halo if dwn is not None else None
    if hp is None:
        n_world = fl.n * (dist.get_world_size() if dist else 1)
        return {"form": "allgather", "bytes_per_step_max": 12 * n_world, "sent_per_step_max": fl.n}
    hp.timing = []
    for _ in range(steps):
        fl.step()
    torch.cuda.synchronize()
    us = [a.elapsed_time(b) * 1e3 for a, b in hp.timing]
    hp.timing = None
    msg_bytes = sum(4 * (8 + 3 * hp.send_cap[p]) for p in hp.messages())
    rows = gather_ranks(dist, red_dev, [hp.sent_per_step, hp.recv_per_step, float(np.mean(us)) if us else 0.0,
                                        hp.overflow(), msg_bytes, len(hp.messages())])
    sent = [int(r[0]) for r in rows]
    return {"form": "halo", "split_phase_query": bool(dwn.split), "resize_every_steps": hp.resize_every,
            "selection_margin_m": hp.step_reach - hp.cutoff,
            "sent_per_step_max": max(sent), "sent_per_step_min": min(sent), "bytes_per_step_max": 12 * max(sent),
            "message_bytes_per_step_max": int(max(r[4] for r in rows)), "peers_max": int(max(r[5] for r in rows)),
            "recv_per_step_max": int(max(r[1] for r in rows)),
            "exchange_span_us_max": max(r[2] for r in rows), "exchange_span_us_min": min(r[2] for r in rows),
            "exchange_span_note": "from the start of select + pack to the end of the call that bins what arrived (two-pass query: the "
                                  "wire and the local pass run side by side inside this span; one-pass query: pack, wire, binning and "
                                  "the whole query)",
            "overflow": int(sum(r[3] for r in rows))}


def config5_all_ranks(a, local, rank, world, dist, red_dev, barrier, options):
    """N > 1, default workload: BASELINE configs[4] — the one workload with an exchange step — measured on the ranks of
    THIS run as well (65 536 mixed drones per rank in 128 m slabs, halo exchange over the run's backend), so that a
    multi-GPU run of the default line also says what the RCCL path does.  Every rank takes part (the exchange is
    collective); rank 0 gets the entry."""
    from dronesim_amd import sharding
    steps = max(20, a.steps // 2)

    def rule(first_wall):
        w, _ = sharding.reduce_step_times(dist, red_dev, first_wall, 0.0)
        return max(1, int(np.ceil(MIN_TIMED_S / max(w, 1e-9))))

    def one(split):
        # DSIM_DW_SPLIT as the Fleet reads it: "1" the two-pass query around the wire (the default on RCCL), "0" one grid behind it
        os.environ["DSIM_DW_SPLIT"] = "1" if split else "0"
        try:
            fl5 = Fleet(65536, 1, local, 1, a.layout, sharding.rank_seed(a.noise_seed, rank), config5=True, dist=dist, rank=rank,
                        options=options)
        finally:
            os.environ.pop("DSIM_DW_SPLIT", None)
        wall_l, dev_l, regions = fl5.timed(steps, 10, barrier, min_s=MIN_TIMED_S, repeat_rule=rule)
        wall, dev = sharding.reduce_step_times(dist, red_dev, wall_l, dev_l)
        k = steps * regions
        rows = gather_ranks(dist, red_dev, [dev_l / k * 1e6, wall_l / k * 1e6, fl5.env.ground_contacts()])
        ex = exchange_report(fl5, dist, red_dev)
        e = {"drones_per_gpu": fl5.n, "drone_steps_per_s": sharding.aggregate_throughput([fl5.n] * world, k, wall),
             "ms_per_step": wall / k * 1e3, "steps_timed": k, "step_chain_us_per_rank": [round(r[0], 2) for r in rows],
             "step_chain_us_min": min(r[0] for r in rows), "step_chain_us_max": max(r[0] for r in rows),
             "host_us_per_step_max": max(r[1] for r in rows), "ground_contacts": int(sum(r[2] for r in rows)), "exchange": ex}
        fl5.env.close()
        return e
    # both forms of the query in the one job, so that a single run on N devices answers which one the wire wants
    two_pass, one_grid = one(True), one(False)
    best = two_pass if two_pass["drone_steps_per_s"] >= one_grid["drone_steps_per_s"] else one_grid
    e = {"workload": WORKLOAD_TEXT["config5"], "n_gpus": world, "drones_per_gpu": best["drones_per_gpu"],
         "drone_steps_per_s": best["drone_steps_per_s"], "ms_per_step": best["ms_per_step"],
         "faster_form": "two-pass query around the wire" if best is two_pass else "one grid behind the wire",
         "two_pass_query_around_the_wire": two_pass, "one_grid_behind_the_wire": one_grid,
         "overflow_is_zero": two_pass["exchange"].get("overflow", 0) == 0 and one_grid["exchange"].get("overflow", 0) == 0,
         "kernel": "dsim_halo_pack + send/recv + halo binning (side stream) | k_dw_query_cell local pass, halo pass, "
                   "k_step_runs + fused grid binning, k_wls_fallback"}
    return e


def rccl_selftest(dist, fl, red_dev):
    """World size 1 on the real backend: the device-side collectives the N-rank path uses, each checked — the all-gather of
    positions of the downwash term's all-gather form (downwash.Downwash._gather, its RCCL branch), and ONE grouped batch
    of isend / irecv on slices of persistent halo buffers (the wire of downwash.HaloWire), here from the rank to itself."""
    import torch
    from dronesim_amd import _native as nat
    from dronesim_amd.downwash import Downwash
    env = fl.env
    dev = env.ctx.device
    out = {"backend": dist.get_backend(), "world_size": dist.get_world_size()}
    dw = Downwash(env.ctx, env.state, env._type_id, dist)
    dw._counts = [env.state.n]
    pos = env.state.raw_fields(0, 3).contiguous()
    got = dw._gather(pos)
    out["all_gather_positions_ok"] = bool(torch.equal(got, pos)) and dist.get_backend() == "nccl"
    cap = 4096
    send = torch.arange(nat.HALO_HDR + 3 * cap, dtype=torch.float32, device=dev)
    recv = torch.zeros_like(send)
    k = nat.HALO_HDR + 3 * 1024                                  # a message shorter than its buffer, as a halo message is
    ops = [dist.P2POp(dist.isend, send[:k], 0), dist.P2POp(dist.irecv, recv[:k], 0)]
    for w in dist.batch_isend_irecv(ops):
        w.wait()
    torch.cuda.synchronize()
    out["grouped_isend_irecv_ok"] = bool(torch.equal(recv[:k], send[:k])) and float(recv[k:].abs().sum()) == 0.0
    t = torch.tensor([3.0, 1.0], dtype=torch.float64, device=red_dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    out["all_reduce_on"] = str(t.device)
    out["ipc_mode_legacy_env"] = os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")
    return out


class Watchdog:
    """A section that every rank must finish together (a collective workload behind the headline): if it does not within
    `seconds`, rank 0 prints the line it already has — the headline measurement must not be lost to a hang in an extra —
    and every rank leaves with a NON-ZERO exit code (WATCHDOG_RC): a hung collective is a failed run, whatever was printed."""

    def __init__(self, seconds, rank, line):
        import threading
        self.rank, self.line = rank, line
        self.t = threading.Timer(seconds, self.fire)
        self.t.daemon = True
        self.seconds = seconds

    def fire(self):
        if self.rank == 0:
            self.line["config5_all_ranks"] = {"error": f"did not finish within {self.seconds} s; abandoned"}
            emit(json.dumps(self.line))
        os._exit(WATCHDOG_RC)

    def __enter__(self):
        self.t.start()
        return self

    def __exit__(self, *exc):
        self.t.cancel()
        return False


# the reference-shaped loop: algorithmic bytes per drone and loop iteration, and what they are
#   Env.step      reads 13 rigid + n_act action floats; writes 13 rigid + the echoed action + the (16 + table n_act)-wide row
#   computeControl reads 13 rigid + (7 + n_act) controller memory + 10 targets; writes the memory, the command, pos_e, yaw_e
TWO_CALL_BYTES = {
    "quad": (428, "physics 68 r + 148 w (13 rigid, 4 action | 13 rigid, 4 echo, 20-wide row); control 136 r + 76 w "
                  "(13 + 11 + 10 | 11 memory, 4 cmd, 3 pos_e, yaw_e)"),
    "hexa": (476, "physics 76 r + 164 w (13 rigid, 6 action | 13 rigid, 6 echo, 22-wide row); control 144 r + 92 w "
                  "(13 + 13 + 10 | 13 memory, 6 cmd, 3 pos_e, yaw_e)"),
    "mixed": (468, "average of a quad (68 r + 164 w; 136 r + 84 w: rows and command arrays are those of a six-actuator "
                   "table) and a hexa (476), + 4 B for the caller's drone number"),
    "config5": (480, "as mixed, + 12 B for the downwash force the Env.step launch reads"),
    "dyn": (452, "Physics.DYN: physics 80 r + 160 w (13 rigid, 3 rpy rates, 4 action | 13 rigid, 3 rates, 4 echo, 20-wide row); "
                 "control as a quad's"),
}
WORKLOAD_TEXT = {
    "config2x1024": "configs[1] 4096 robobee INDI hover x 1024 vectorised envs/GPU",
    "config2": "configs[1] 4096 robobee INDI hover (single fleet)",
    "config3": "configs[2] 65536 robobee, waypoint-table tracking (fly_INDI_TrajectoryTrack)",
    "config4": "configs[3] shard: 65536 robobee INDI hover per GPU (524288 over 8 GPUs), no coupling",
    "mixed": "even index robobee, odd index hexa_6DOF, 4096 x 1024 envs/GPU, no downwash",
    "mixed_type_major": "the same 50/50 robobee + hexa_6DOF fleet stored type-major (one run per type)",
    "two_call_loop": "configs[1] x 1024 envs/GPU through the reference-shaped loop: obs = env.step(cmd); "
                     "cmd = ctrl.computeControlFromState(obs)",
    "hexa": "4096 hexa_6DOF (6-DOF INDI + WLS) hover x 1024 vectorised envs/GPU",
    "dyn": "configs[1] x 1024 envs/GPU on Physics.DYN (the reference's explicit model, BaseAviary._dynamics) + INDI",
    "config5": "configs[4] shard: 65536/GPU, 50% robobee + 50% hexa_6DOF interleaved, neighbour downwash at the config's "
               "density (one drone per m^2: a 128 m x 512 m slab per GPU), halo exchange between neighbouring slabs",
}


def useful_pairs(torch, pos):
    """Pairs (i, j) of formula P8 that contribute a term: j above i and within the 10 m cut-off in xy (BaseAviary.py:1752).
    Brute force on the device, in slabs of receivers; pos [3, n]."""
    n = pos.shape[1]
    x, y, z = pos[0], pos[1], pos[2]
    total = 0
    for s_ in range(0, n, 2048):
        e_ = min(n, s_ + 2048)
        dz = z[None, :] - z[s_:e_, None]
        dd = (x[None, :] - x[s_:e_, None]) ** 2 + (y[None, :] - y[s_:e_, None]) ** 2
        total += int(((dz > 0) & (dd < 100.0)).sum())
    return total


def valu_roofline(torch, fl, chain_s):
    """config 5: the vector-pipe roofline of the neighbour query (VALU_PAIR_PEAK above).  `achieved` = USEFUL pair
    evaluations per second of the whole step chain — pairs that contribute a term of formula P8, counted by brute force on
    this fleet's positions — the analogue of algorithmic bytes; `evaluated` = the pairs the query's loops really run (its
    diagnostics counter, one extra untimed step), the analogue of measured traffic."""
    dwn = fl.env._downwash
    pos = fl.env.state.raw_fields(0, 3).contiguous()
    useful = useful_pairs(torch, pos)
    dwn.count_pairs(True)
    fl.env._fused_plan_dw = None
    fl.step()                              # (with kept lists: the query that makes them — the ones that follow are the common case)
    torch.cuda.synchronize()
    first = int(dwn.pair_counter.item())
    fl.step()
    torch.cuda.synchronize()
    evaluated = int(dwn.pair_counter.item()) - first
    dwn.count_pairs(False)
    fl.env._fused_plan_dw = None
    achieved = useful / chain_s
    return {"bound": "valu", "achieved": achieved, "peak": VALU_PAIR_PEAK, "unit": "pair evaluations/s", "frac": achieved / VALU_PAIR_PEAK,
            "useful_pairs_per_step": useful, "evaluated_pairs_per_step": evaluated,
            "evaluated_over_useful": evaluated / max(1, useful),
            "evaluated_pairs_of_a_query_that_makes_lists": first if dwn.keep_lists > 1 else None,
            "frac_counting_evaluated_pairs": evaluated / chain_s / VALU_PAIR_PEAK,
            "note": "per step CHAIN (neighbour query, step + grid binning or list refresh, WLS fallback pass), not per query launch: the "
                    "query is ~28 of its ~42 us with kept lists, ~30 of ~46 without (profiles/r06_c5_*); peak = 1 024 SIMDs x 2.4 GHz x 64 lanes / 69.6 cycles per 64 pairs "
                    "(18 full-rate + 2 transcendental instructions per pair; issue costs measured by tools/valubench.hip)"}


def measured_traffic_ratio(name):
    """profiles/traffic_by_workload.json: counter traffic over budgeted bytes of the workloads whose kernels move fewer bytes than
    SURVEY 8d budgets (FETCH_SIZE x 2 + WRITE_SIZE from separate rocprofv3 --pmc passes, tools/profile_sq.sh), or None."""
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic_by_workload.json")))
    except (OSError, ValueError):
        return None
    for key, v in tj.items():
        if name.startswith(key) and "caller_order" not in name and "type_major" not in name and isinstance(v, dict):
            return v
    return None


def measure_variant(torch, local, layout, seed, steps, name, nf, rep, sub, wp, ns, options=0):
    """One entry of "baseline_configs" / "also": a fresh fleet, one warm region, then timed regions of k steps until
    they cover MIN_TIMED_S (sums reported, nothing picked)."""
    f2 = Fleet(nf, rep, local, sub, layout, seed, waypoints=wp, n_steps=ns,
               config5=name.startswith("config5"), dist=(MirrorDist(128.0) if "mirrored_neighbour" in name else None),
               chained="chained" in name, hexa=name.startswith("hexa"),
               mixed=("type_major" if "type_major" in name else name.startswith("mixed")), options=options,
               slab_m=(1024.0 if "lowdensity" in name else 128.0), storage=("caller" if "caller_order" in name else None),
               dyn=name.startswith("physics_dyn"), dw_keep=(0 if "plain_query" in name else None))
    if "hipgraph" in name:
        f2.n_steps = 1
        f2.use_graph(ns)
    k2 = max(20, steps // 2)
    w2, d2, reg = f2.timed(k2, 10, min_s=MIN_TIMED_S)
    from_idle_us = d2 / (k2 * reg) * 1e6
    if f2.n >= (1 << 20):
        # fleets that fill the chip: the regions above start a few launches after an idle device woke up (what rounds 1-4
        # reported; kept as launch_us_from_idle); the entry's figures are those of a device that has been under this load
        # for SETTLE_S (Fleet.timed: settle_s) — what a simulation that runs for longer than 0.1 s sees
        w2, d2, reg = f2.timed(k2, 0, min_s=MIN_TIMED_S, settle_s=SETTLE_S)
    # drone_steps_per_s: from the host's clock around the regions; launch_us (and device_drone_steps_per_s): from the
    # events on the launch stream — they differ when the host, not the device, paces the loop (or hiccups)
    e = {"drone_steps_per_s": f2.n * k2 * reg * ns / w2, "launch_us": d2 / (k2 * reg) * 1e6, "env_steps_per_launch": ns,
         "drones": f2.n, "phys_substeps": sub, "steps_timed": k2 * reg,
         "device_drone_steps_per_s": f2.n * k2 * reg * ns / d2}
    if f2.n >= (1 << 20):
        e["launch_us_from_idle"] = from_idle_us
        e["settled_for_s"] = SETTLE_S
    e["protocol"] = "settled" if f2.n >= (1 << 20) else "from_idle"
    if ns == 1:
        bts = 184 if "chained" in name else (256 if name.startswith("physics_dyn") else 248 if name.startswith("hexa") else
                                             (253 if name.startswith("config5") else
                                              (241 if name.startswith("mixed") else BYTES_PER_DRONE_STEP)))
        e["hbm_frac"] = f2.n * bts / (d2 / (k2 * reg)) / 1e9 / HBM_PEAK_GBPS
        e["bytes_per_drone_step"] = bts
        # where counters say the kernel moves FEWER bytes than the budget (the 6-DOF law never reads target acceleration and yaw:
        # 232 of the budgeted 248 B), the fraction of peak is quoted on the MEASURED bytes and the budgeted one becomes secondary
        mt = measured_traffic_ratio(name)
        if mt is not None:
            e["hbm_frac_budgeted"] = e["hbm_frac"]
            e["hbm_frac"] = e["hbm_frac_measured"] = e["hbm_frac_budgeted"] * mt["traffic_over_budget"]
            e["hbm_frac_note"] = (f"hbm_frac is on the MEASURED traffic ({mt['traffic_over_budget']:.4f} x the budgeted {bts} B, "
                                  f"{mt['source']}); hbm_frac_budgeted on the budget")
    if "mirrored_neighbour" in name:
        e["exchange"] = exchange_report(f2, None, "cpu", steps=20)
    if name.startswith("config5") and f2.env._downwash is not None:
        dwn = f2.env._downwash
        q = f2.env.ctx.query
        from dronesim_amd import _native as nat
        reuses = q(nat.QUERY_DW_REUSES)
        e["kept_lists"] = {"one_query_in": dwn.keep_lists, "skin_m": dwn.keep_skin, "queries_answered_from_lists": reuses,
                           "mean_drones_outside_the_skin": (q(nat.QUERY_DW_MOVERS) / reuses) if reuses else None} if dwn.keep_lists > 1 else None
    if name.startswith("config5") and "mirrored" not in name:
        try:
            e["roofline"] = valu_roofline(torch, f2, d2 / (k2 * reg))
        except Exception as ex:           # an extra must not cost the line
            e["roofline"] = {"error": repr(ex)[:200]}
    f2.env.close()
    del f2
    return e


def measure_adaptor_env(torch, local, layout, seed, steps, cls_name):
    """Env.step of the reference's alternate action adaptors (VelocityAviary.py:221-264 / RPYTAviary.py:181-193) on the
    headline fleet: the (part of the) INDI law inside step() on the current state, the physics, the observation rows —
    one launch (k_adaptor_fast), the [N, 4] action taken as the caller holds it.  304 B per drone-step: 24 state + 4 action
    floats in, 24 state + 4 echoed command + 20 row floats out."""
    import numpy as np
    from dronesim_amd import envs
    try:
        n = 4096 * 1024
        ij = np.arange(n) % 4096
        xyz = np.stack([(ij % 64) * 1.0, (ij // 64) * 1.0, np.full(n, 0.5)], 1)
        env = getattr(envs, cls_name)(["robobee"], n, initial_xyzs=xyz, aggregate_phy_steps=1, noise_seed=seed, dict_io=False,
                                      layout=layout, device=local)
        if cls_name == "VelocityAviary":
            act = torch.tensor([1.0, 0.0, 0.2, 0.5], device=env.ctx.device).repeat(n, 1)
        else:
            act = torch.tensor([0.0, 0.0, 0.0, 9.81 * env.types[0].mass], device=env.ctx.device).repeat(n, 1)
        k = max(300, steps // 2)
        for _ in range(10):
            env.step(act)
        torch.cuda.synchronize()
        for _ in range(int(SETTLE_S / 200e-6)):         # (under load for ~SETTLE_S, the timed launches right behind: Fleet.timed)
            env.step(act)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(k):
            env.step(act)
        e1.record()
        e1.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / k
        out = {"drone_steps_per_s": n / us * 1e6, "env_step_us": us, "drones": n, "phys_substeps": 1, "steps_timed": k,
               "protocol": "settled", "settled_for_s": SETTLE_S,
               "bytes_per_drone_step": 304, "hbm_frac": n * 304 / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS,
               "kernel": "k_adaptor_fast (action rows in, observation rows out: one launch per Env.step)"}
        env.close()
        return out
    except Exception as e:          # an extra must not cost the headline
        return {"error": repr(e)[:300]}


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    a = parse(argv)
    if (a.gpus > 1 or (os.environ.get("DSIM_BENCH_FORCE_DIST", "0") != "0" and not a.dry_run)) and "WORLD_SIZE" not in os.environ:
        # not started by a launcher: start the ranks as children — nothing in THIS process has touched the GPU
        sys.exit(launch_ranks(a, argv))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus != world:
        print(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}; the two must agree", file=sys.stderr)
        sys.exit(2)
    # rehearsal knob: DSIM_BENCH_BACKEND=gloo lets several ranks share one GPU (RCCL wants one device per rank)
    backend = os.environ.get("DSIM_BENCH_BACKEND", "nccl")
    import torch
    dist = None
    dist_info = {"world_size": 1, "backend": None}
    if a.dry_run:
        return dry_run(a, rank, world, backend)
    protect_stdout()
    import __graft_entry__ as graft
    if a.lib:
        from dronesim_amd import _native as nat0
        nat0.load(a.lib)            # an A/B build of the same ABI: loaded first, so that everything below binds to it
    if rank == 0:
        graft.build()
    local = local % max(1, torch.cuda.device_count())
    # DSIM_BENCH_FORCE_DIST=1: the process group is made at world size 1 too, so that a one-GPU box runs every collective
    # of the N-rank path once on its real backend (RCCL): barrier, the MAX reduction, the per-rank all-gather
    force_dist = os.environ.get("DSIM_BENCH_FORCE_DIST", "0") != "0" and "RANK" in os.environ
    if world > 1 or force_dist:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
        dist.barrier()
        dist_info = {"world_size": dist.get_world_size(), "backend": dist.get_backend(),
                     "devices_visible": torch.cuda.device_count()}
    if rank != 0:
        graft.build()       # no-op once rank 0 has built
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    torch.cuda.set_device(local)
    from dronesim_amd import _native as nat
    nat.load(a.lib)
    options = {"auto": 0, "on": nat.OPT_STREAM_ON, "off": nat.OPT_STREAM_OFF}[a.stream]
    barrier = (lambda: dist.barrier()) if dist else None
    red_dev = "cuda" if backend == "nccl" else "cpu"

    n_fleet, replicas = {"config2x1024": (4096, 1024), "config2": (4096, 1), "config3": (65536, 1), "config4": (4096, 16),
                         "config5": (65536, 1), "hexa": (4096, 1024), "mixed": (4096, 1024),
                         "mixed_type_major": (4096, 1024), "two_call_loop": (4096, 1024), "dyn": (4096, 1024)}[a.workload]
    tck = a.two_call_kind if a.workload == "two_call_loop" else None
    if tck == "config5":
        n_fleet, replicas = 65536, 1
    if a.replicas > 0:
        replicas = a.replicas
    from dronesim_amd import sharding
    # weak scaling: every rank owns a same-sized contiguous shard of the N-GPU fleet; no data-path collective
    # (config5 only: one halo exchange of positions per step for the neighbour-downwash term)
    fl = Fleet(n_fleet, replicas, local, a.substeps, a.layout, sharding.rank_seed(a.noise_seed, rank),
               waypoints=a.workload == "config3", config5=a.workload == "config5" or tck == "config5",
               dist=(MirrorDist(a.slab_m) if (a.mirror_peer and world == 1) else dist) if a.workload == "config5" else None,
               rank=rank, hexa=a.workload == "hexa" or tck == "hexa",
               mixed=("type_major" if a.workload == "mixed_type_major" else (a.workload == "mixed" or tck == "mixed")), options=options,
               slab_m=a.slab_m, dyn=a.workload == "dyn" or tck == "dyn")
    if a.workload == "two_call_loop":
        fl.make_two_call_loop()

    def repeat_rule(first_wall):
        # the slowest rank's first region decides how many regions every rank runs
        w, _ = sharding.reduce_step_times(dist, red_dev, first_wall, 0.0)
        return max(1, int(np.ceil(MIN_TIMED_S / max(w, 1e-9))))

    wall_local, dev_s_local, regions = fl.timed(a.steps, a.warmup, barrier, min_s=MIN_TIMED_S, repeat_rule=repeat_rule,
                                               settle_s=a.settle_seconds)
    wall, dev_s = sharding.reduce_step_times(dist, red_dev, wall_local, dev_s_local)  # MAX over ranks
    steps_timed = a.steps * regions
    value = sharding.aggregate_throughput([fl.n] * world, steps_timed, wall)
    launch_s = dev_s / steps_timed
    # config 5: half quads (232 B) half hexas (248 B) + 1 B type id + the 12 B downwash force the step kernel reads;
    # its step is a chain of kernels (neighbour query, step + grid binning, WLS fallback), timed as a whole
    bytes_per = {"config5": 253, "hexa": 248, "mixed": 241, "mixed_type_major": 241, "dyn": 256,
                 "two_call_loop": TWO_CALL_BYTES[a.two_call_kind][0]}.get(a.workload, BYTES_PER_DRONE_STEP)
    mixed_k = "k_step_mixed3" if a.layout != "tile64" else "k_step_mixed4"
    if fl.env.order is not None:
        # the interleaved fleet is STORED type-major behind the caller's numbering (fleet.StorageOrder): one single-type
        # launch per type; + 4 B per drone-step for the caller's index that keys the noise stream
        mixed_k = "k_step_runs (type-major storage behind the caller's interleaved order, all runs in one launch)"
    kernel = {"config5": f"k_dw_query_kept (kept lists; k_dw_query_cell one step in {os.environ.get('DSIM_DW_KEEP', '32')}), {mixed_k} "
                         "(+ the lists' refresh / grid binning), k_wls_fallback",
              "hexa": "k_step_hexa (+ k_wls_fallback)",
              "mixed": f"{mixed_k} (+ k_wls_fallback)",
              "mixed_type_major": "k_step_run x2 (+ k_wls_fallback)",
              "dyn": "k_dyn (Physics.DYN: BaseAviary._dynamics + INDI)",
              "two_call_loop": ("k_physics_fast (observation fused) + k_control_fast" if tck == "quad" else
                                "k_dyn (Env.step on Physics.DYN, observation fused) + k_control_fast" if tck == "dyn" else
                                ("k_dw_query_kept / k_dw_query_cell, " if tck == "config5" else "") +
                                "k_physics_runs (observation fused) + k_control_runs (+ k_wls_fallback)")}.get(a.workload, "k_step_fast")
    achieved = fl.n * bytes_per / launch_s / 1e9
    rank_rows = gather_ranks(dist, red_dev, [dev_s_local / steps_timed * 1e6, wall_local / steps_timed * 1e6,
                                             float(torch.cuda.current_device()), float(rank)])
    exchange = exchange_report(fl, dist, red_dev) if (a.workload == "config5" and (world > 1 or a.mirror_peer)) else None

    if rank == 0:
        traffic, traffic_source = None, None
        tp = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tp):
            tj = json.load(open(tp))
            if tj.get("workload") == a.workload and tj.get("layout", "soa") == a.layout:
                traffic = tj.get("hbm_bytes_per_launch")
                traffic_source = (f"{tj.get('source')}: FETCH_SIZE x2 + WRITE_SIZE of this kernel from SEPARATE rocprofv3 --pmc "
                                  "passes of this command (tools/profile_sq.sh), not measured in this run")
        out = {
            "metric": "drone-steps/sec (num_drones x env steps/s)", "value": value, "unit": "drone-steps/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": wall / steps_timed * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "steps_timed": steps_timed, "timed_regions": regions, "settle_seconds": a.settle_seconds,
            "timing_note": f"regions of exactly --steps steps, each bracketed by barrier + device synchronisation, repeated "
                           f"until they cover {MIN_TIMED_S * 1e3:.0f} ms; value and ms_per_step are sums over all of them",
            "config": {"workload": WORKLOAD_TEXT[a.workload],
                       "drones_per_gpu": fl.n, "phys_substeps": a.substeps, "layout": a.layout,
                       "noise_seed": a.noise_seed, "launches_per_step": 1, "parallelism": f"shard{world}",
                       # (fleet.FleetState: a large fleet's state block is allocated with room for one block of targets behind it)
                       "targets_behind_the_state_block": bool(getattr(getattr(fl, "tgt", None), "behind_the_state", False)
                                                              or getattr(getattr(getattr(fl, "ctrl", None), "_targets", None), "behind_the_state", False))},
            "dist": dist_info,
            "ranks": {"launch_us_min": min(r[0] for r in rank_rows), "launch_us_max": max(r[0] for r in rank_rows),
                      "host_us_per_step_min": min(r[1] for r in rank_rows), "host_us_per_step_max": max(r[1] for r in rank_rows),
                      "launch_us_per_rank": [round(r[0], 2) for r in rank_rows], "device_of_rank": [int(r[2]) for r in rank_rows],
                      "seen": sorted(int(r[3]) for r in rank_rows)},
            "self_check": self_check(world, a.gpus, dist_info.get("backend"), backend, [r[2] for r in rank_rows],
                                     [r[0] for r in rank_rows], distinct_devices=backend == "nccl"),
            # drone-steps of the whole run (set-up, warm-up and timed regions) whose collision cylinder reached the ground
            # plane, which the flight kernels of large fleets do not model (DSIM_OPT_PLANE does, for landing-sized fleets):
            # 0 = every step of the workload lies in the domain the flight kernels cover
            "ground_contacts": fl.env.ground_contacts(),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "ratio_to_copy_yardstick": achieved / HBM_COPY_YARDSTICK_GBPS,
                         "copy_yardstick": HBM_COPY_YARDSTICK_GBPS,
                         "copy_yardstick_note": "the float4 device-copy rate of MI355X_MICROARCH.md: a yardstick, not a ceiling (ratios above 1 happen)",
                         "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": kernel, "bytes_per_drone_step": bytes_per,
                         "launch_us": launch_s * 1e6},
        }
        if a.workload == "config5" and world == 1 and not a.mirror_peer:
            try:
                out["roofline_valu"] = valu_roofline(torch, fl, launch_s)
            except Exception as ex:
                out["roofline_valu"] = {"error": repr(ex)[:200]}
        if fl.env.ctx.placement_log:
            # where the fleet-sized arrays beside the state block were put, by measurement (dronesim_amd/placement.py), and
            # what the searches cost (seconds; device memory held at once while walking — transient, straight from the driver)
            out["placement"] = list(fl.env.ctx.placement_log)
            out["placement_cost"] = placement_cost(fl.env.ctx.placement_log)
        if exchange is not None:
            out["exchange"] = exchange
        if force_dist and world == 1:
            out["rccl_selftest"] = rccl_selftest(dist, fl, red_dev)
        if a.mirror_peer:
            out["config"]["synthetic_neighbour"] = ("--mirror-peer: ONE rank whose neighbour is its own reflection across the slab edge "
                                                    "(MirrorDist: a device-side wire); traces the exchange path, not a multi-GPU run")
    else:
        out = {}
    if world > 1 and a.workload == "config2x1024" and not a.no_also:
        with Watchdog(240, rank, out):
            try:
                e5 = config5_all_ranks(a, local, rank, world, dist, red_dev, barrier, options)
            except Exception as e:                # (a rank-local failure; a hang of the others ends in the watchdog)
                e5 = {"error": repr(e)[:300]}
            if rank == 0:
                out["config5_all_ranks"] = e5
                if "overflow_is_zero" in e5:
                    out["self_check"]["checks"]["config5_halo_overflow_is_zero"] = bool(e5["overflow_is_zero"])
                if "error" in e5:
                    out["self_check"]["checks"]["config5_all_ranks_ran"] = False
                out["self_check"]["failed"] = sorted(k for k, v in out["self_check"]["checks"].items() if not v)
                out["self_check"]["ok"] = not out["self_check"]["failed"]
    if rank == 0:
        if world == 1 and not a.no_also:
            # ---- BASELINE.json configs 1-3 at their LITERAL sizes and example settings --------------------------------
            # (fleet, replicas, phys_substeps, waypoint table, Env.steps per launch)
            base = {}
            for name, spec in {
                    "configs[1]_4096_robobee_hover_sub5": (4096, 1, 5, False, 1),
                    "configs[1]_4096_robobee_hover_sub5_32_env_steps_per_launch": (4096, 1, 5, False, 32),
                    "configs[2]_65536_robobee_waypoints_sub2": (65536, 1, 2, True, 1),
                    "configs[2]_65536_robobee_waypoints_sub2_32_env_steps_per_launch": (65536, 1, 2, True, 32),
                    "configs[3]_shard_65536_robobee_hover_sub1": (4096, 16, 1, False, 1)}.items():
                base[name] = measure_variant(torch, local, a.layout, a.noise_seed, a.steps, name, *spec, options=options)
            # configs[1] through the reference's own two surfaces (examples/fly_INDI.py:223-239: obs = env.step(cmd);
            # cmd = ctrl.computeControlFromState(...)): two launches per iteration, paced by the Python around them
            try:
                f1 = Fleet(4096, 1, local, 5, a.layout, a.noise_seed)
                f1.make_two_call_loop()
                k1 = max(200, a.steps)
                w1, d1, r1 = f1.timed(k1, 50, min_s=MIN_TIMED_S)
                base["configs[1]_4096_robobee_hover_sub5_env_step_then_computeControl"] = {
                    "drone_steps_per_s": f1.n * k1 * r1 / w1, "loop_us": w1 / (k1 * r1) * 1e6, "loop_us_device": d1 / (k1 * r1) * 1e6,
                    "drones": f1.n, "phys_substeps": 5, "steps_timed": k1 * r1,
                    "note": "host-paced: the prepared argument blocks of step() / computeControl() are re-launched (round 4: 24 -> 11.5 us)"}
                f1.env.close()
                del f1
            except Exception as e:          # an extra must not cost the headline
                base["configs[1]_4096_robobee_hover_sub5_env_step_then_computeControl"] = {"error": repr(e)[:300]}
            out["baseline_configs"] = base
            also = {}
            # yardsticks of THIS device, from the repo's own probe (tools/membench.hip, a child process): the float4-copy
            # rate, and the floor of the headline kernel's access shape — its 58 streaming dword accesses per lane on the
            # wave-tiled layout with no arithmetic behind them
            also.update(memory_yardstick(fl.n))
            for name, spec in {
                    "config2_single_fleet_4096_sub5_hipgraph_of_32_launches": (4096, 1, 5, False, 32),
                    "config3_65536_waypoints_sub2_hipgraph_of_32_launches": (65536, 1, 2, True, 32),
                    "config2x1024_sub5": (4096, 1024, 5, False, 1),
                    # configs[4] shard at the config's density (128 m slab), and the round-1 definition of the same
                    # line (the 65 536 drones spread over the whole 1024 m box: an eighth of the density)
                    "config5_shard_65536_mixed_downwash": (65536, 1, 1, False, 1),
                    # ... with the neighbour query made from scratch every step (downwash_keep = 0: what every round up to 5 measured)
                    "config5_shard_65536_plain_query_every_step": (65536, 1, 1, False, 1),
                    "config5_lowdensity_r01_definition": (65536, 1, 1, False, 1),
                    # the same shard with a neighbour to exchange positions with — a SYNTHETIC one on this one GPU: the
                    # rank's own reflection across the slab edge (MirrorDist: the wire is two small device ops).  What the
                    # device-paced exchange path costs per step with a wire of no latency; not a multi-GPU number
                    "config5_shard_65536_with_mirrored_neighbour": (65536, 1, 1, False, 1),
                    # DSIM_OPT_CHAINED: the six controller-memory fields that are functions of the stored
                    # rigid state are neither read nor written: 184 B of real traffic per drone-step
                    "config2x1024_chained_184B": (4096, 1024, 1, False, 1),
                    # homogeneous morphing-hexa fleet: 6-DOF INDI + WLS allocation, 248 B/drone-step
                    "hexa_6DOF_4194304_indi6dof_wls": (4096, 1024, 1, False, 1),
                    # ... with the examples' five sub-steps per Env.step (examples/fly_hexa_6DOF.py): the looped k_step_hexa,
                    # bound by vector issue at the sustained clock (round 5: 197-208 us; round 6: wrench-level noise, tables by template)
                    "hexa_6DOF_4194304_sub5": (4096, 1024, 5, False, 1),
                    # Physics.DYN (row D1): BaseAviary._dynamics + INDI in one launch (k_dyn), the headline fleet; 232 B + the
                    # model's own rpy_rates (3 floats in, 3 out) = 256 B per drone-step
                    "physics_dyn_4194304_sub1": (4096, 1024, 1, False, 1),
                    "physics_dyn_4194304_sub5": (4096, 1024, 5, False, 1),
                    # config 5's composition at roofline size, no downwash: 240 B average + 1 B type id.  The caller hands the
                    # fleet over interleaved (even index quad, odd index hexa); the env stores it type-major behind that
                    # numbering (fleet.StorageOrder) — and, for comparison, in the caller's own order (k_step_mixed4)
                    "mixed_quad_hexa_4194304": (4096, 1024, 1, False, 1),
                    "mixed_quad_hexa_4194304_sub5": (4096, 1024, 5, False, 1),
                    "mixed_quad_hexa_4194304_caller_order_storage": (4096, 1024, 1, False, 1),
                    # the same fleet in type-major storage: one single-type launch per type
                    "mixed_quad_hexa_4194304_type_major": (4096, 1024, 1, False, 1)}.items():
                also[name] = measure_variant(torch, local, a.layout, a.noise_seed, a.steps, name, *spec, options=options)
                if name.startswith("config5"):
                    also[name]["note"] = ("a chain of three dependent launches on a 65 536-drone shard (neighbour query, step + grid "
                                          "binning or list refresh, WLS fallback): bound by the vector pipe of the query and by launch "
                                          "latency, not by HBM — its roofline is `roofline` (bound: valu); hbm_frac is kept for "
                                          "comparison with earlier rounds")
                if "mirrored_neighbour" in name:
                    also[name]["note"] = ("the shard above + the halo exchange with a synthetic neighbour (its own reflection): "
                                          "k_halo_pack, a device-side wire, halo binning, one-grid query, step, fallback on one stream")
                if name.startswith("physics_dyn"):
                    also[name]["note"] = ("Physics.DYN: the reference's own explicit model (BaseAviary.py:1767-1828) + INDI, k_dyn; no rotor "
                                          "noise in this model; ang_v reported as R(quat) rpy_rates (dyn_ang_vel='body_rates')")
                if name.startswith("hexa"):
                    also[name]["note"] = ("248 B is the budgeted figure (SURVEY 8d); the 6-DOF law never reads the target "
                                          "acceleration and yaw, measured HBM traffic is 232 B per drone-step")
            # the reference-shaped loop at the same size: obs = env.step(cmd); cmd = ctrl.computeControlFromState(...)
            # (two launches: physics with the observation rows fused, control with the command handed back in place).
            # Measured by a CHILD process that runs `--workload two_call_loop` alone: where the arrays this loop writes lie
            # relative to its state block is worth 10-15 % of it (dronesim_amd/placement.py), and at the end of THIS
            # process — dozens of fleets built and dropped — every fresh allocation is pieced together from fragments of
            # many regions, which a process that runs the loop from its start never sees.
            also["velocity_aviary_env_step_4194304"] = measure_adaptor_env(torch, local, a.layout, a.noise_seed, a.steps, "VelocityAviary")
            also["rpyt_aviary_env_step_4194304"] = measure_adaptor_env(torch, local, a.layout, a.noise_seed, a.steps, "RPYTAviary")
            also["config2x1024_env_step_then_computeControl"] = two_call_child(a)
            # the same loop on the other fleet kinds (examples/fly_hexa_6DOF.py:214-221; BASELINE config 5's composition):
            # Env.step and computeControl on the run kernels, rows / command / errors in the caller's numbering straight from
            # the launches
            also["hexa_env_step_then_computeControl"] = two_call_child(a, "hexa")
            also["mixed_interleaved_env_step_then_computeControl"] = two_call_child(a, "mixed")
            also["config5_shard_two_call"] = two_call_child(a, "config5")
            # Env.step on Physics.DYN (BaseAviary._dynamics) then computeControl: the reference's own explicit model through the loop
            also["physics_dyn_env_step_then_computeControl"] = two_call_child(a, "dyn")
            # placement by trial (dronesim_amd/placement.py; OFF by default since round 5) switched ON for the one loop it
            # moved most, from the same process tree: what the opt-in search is worth on THIS box
            e1 = two_call_child(a, "quad", placement=True)
            also["opt_in_placement_by_trial"] = {"config2x1024_env_step_then_computeControl": {
                k: e1.get(k) for k in ("loop_us_device", "hbm_frac", "drone_steps_per_s", "placement_cost", "error") if k in e1}}
            if "loop_us_device" in also.get("config2x1024_env_step_then_computeControl", {}):
                also["opt_in_placement_by_trial"]["config2x1024_env_step_then_computeControl"]["default_loop_us_device"] = (
                    also["config2x1024_env_step_then_computeControl"]["loop_us_device"])
            if isinstance(also.get("config5_shard_65536_mixed_downwash"), dict) and "loop_us_device" in also["config5_shard_two_call"]:
                also["config5_shard_two_call"]["ratio_to_fused_chain"] = (
                    also["config5_shard_two_call"]["loop_us_device"] / also["config5_shard_65536_mixed_downwash"]["launch_us"])
            out["also"] = also
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.substeps)
        emit(json.dumps(out))
    failed = bool(rank == 0 and world > 1 and not out.get("self_check", {}).get("ok", True))
    if dist:
        dist.barrier()
        dist.destroy_process_group()
    if failed:
        sys.exit(SELF_CHECK_RC)


def dry_run(a, rank, world, backend):
    """The host logic of a run without a device: rendezvous (gloo), barriers, the timing reduction and the JSON line —
    so that the launcher and the N-rank plumbing are testable where there is no GPU.  No kernel runs and no value is
    reported (value = null, "dry_run": true); it is never a benchmark result."""
    dist = None
    dist_info = {"world_size": 1, "backend": None}
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo")
        dist.barrier()
        dist_info = {"world_size": dist.get_world_size(), "backend": dist.get_backend()}
    from dronesim_amd import sharding
    if dist:
        dist.barrier()
    wall = 1e-3 * (rank + 1)                 # synthetic and distinct per rank: the MAX reduction is observable
    wall_max, _ = sharding.reduce_step_times(dist, "cpu", wall, 0.0)
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # (DSIM_DRY_RUN_SHARE_DEVICE=1: every rank reports device 0 — what a mis-launched run would look like — for the test of the self-check)
    dev_id = 0 if os.environ.get("DSIM_DRY_RUN_SHARE_DEVICE", "0") != "0" else local
    rows = gather_ranks(dist, "cpu", [wall * 1e6, float(rank), float(dev_id)])          # the per-rank rows of the real line
    check = self_check(world, a.gpus, dist_info.get("backend"), "gloo", [r[2] for r in rows], [r[0] for r in rows])
    if rank == 0:
        print(json.dumps({"metric": "drone-steps/sec (num_drones x env steps/s)", "value": None, "unit": "drone-steps/s",
                          "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "dry_run": True, "scaling": "weak",
                          "config": {"workload": WORKLOAD_TEXT[a.workload], "parallelism": f"shard{world}"},
                          "dist": dist_info, "wall_max_over_ranks_s": wall_max, "rank0_wall_s": wall,
                          "ranks": {"launch_us_min": min(r[0] for r in rows), "launch_us_max": max(r[0] for r in rows),
                                    "launch_us_per_rank": [r[0] for r in rows], "device_of_rank": [int(r[2]) for r in rows],
                                    "seen": sorted(int(r[1]) for r in rows)},
                          "self_check": check}))
    if dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and world > 1 and not check["ok"]:
        sys.exit(SELF_CHECK_RC)


if __name__ == "__main__":
    main()
