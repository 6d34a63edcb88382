"""bench.py's multi-rank plumbing without a device: `--gpus N` starts N ranks itself (torch.distributed.run as a child
process, before anything touches the GPU), the ranks rendezvous, reduce the timing (MAX over ranks) and rank 0 prints
ONE JSON line with n_gpus = N.  `--dry-run` runs exactly that host logic on gloo and reports no value."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _run(args, env=None, timeout=240):
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None); e.pop("RANK", None); e.pop("LOCAL_RANK", None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                          timeout=timeout, env=e, cwd=ROOT)


def _json_line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_gpus_flag_spawns_that_many_ranks():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run", "--workload", "config4"])
    assert r.returncode == 0, r.stderr[-2000:]
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["dist"] == {"world_size": 2, "backend": "gloo"}
    assert d["dry_run"] is True and d["value"] is None               # never a benchmark number
    assert d["config"]["parallelism"] == "shard2" and "configs[3]" in d["config"]["workload"]
    assert d["rank0_wall_s"] == 1e-3 and d["wall_max_over_ranks_s"] == 2e-3    # rank 1 reports 2 ms: MAX over ranks wins


def test_single_rank_needs_no_launcher():
    r = _run(["--gpus", "1", "--dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 1 and d["dist"]["world_size"] == 1


def test_gpus_must_agree_with_the_launcher():
    r = _run(["--gpus", "2", "--dry-run"], env={"WORLD_SIZE": "3", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and "must agree" in r.stderr


def test_eight_ranks_dry_run_config5():
    """The N = 8 launch of the driver's scaling run, host logic only: eight gloo ranks rendezvous, every rank's row
    reaches rank 0, one line comes out."""
    r = _run(["--gpus", "8", "--steps", "3", "--warmup", "1", "--dry-run", "--workload", "config5"], timeout=400)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 8 and d["dist"]["world_size"] == 8 and "configs[4]" in d["config"]["workload"]
    assert d["ranks"]["seen"] == list(range(8)) and d["self_check"]["ok"] is True
    assert d["ranks"]["launch_us_min"] == 1e3 and d["ranks"]["launch_us_max"] == 8e3 and d["wall_max_over_ranks_s"] == 8e-3


def test_eight_ranks_self_check_passes_and_catches_a_shared_device():
    """The N-rank line checks itself (bench.self_check): world size = --gpus, the backend, one device per rank, every
    rank's row present and timed.  Eight gloo ranks pass; the same run with every rank reporting device 0 — eight processes
    on one GPU, what a mis-launched scaling run would be — prints its line, names the failed check and exits non-zero."""
    import bench
    r = _run(["--gpus", "8", "--steps", "3", "--warmup", "1", "--dry-run"], timeout=400)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _json_line(r.stdout)
    assert d["self_check"]["ok"] is True and d["self_check"]["failed"] == []
    assert d["ranks"]["device_of_rank"] == list(range(8)) and len(d["ranks"]["launch_us_per_rank"]) == 8
    assert set(d["self_check"]["checks"]) >= {"world_size_equals_gpus_flag", "backend_is_gloo", "one_device_per_rank",
                                              "every_rank_reported", "every_rank_timed_something"}
    r = _run(["--gpus", "8", "--steps", "3", "--warmup", "1", "--dry-run"], env={"DSIM_DRY_RUN_SHARE_DEVICE": "1"}, timeout=400)
    d = _json_line(r.stdout)
    assert d["self_check"]["ok"] is False and d["self_check"]["failed"] == ["one_device_per_rank"]
    assert r.returncode != 0
    # the pure function: a wrong backend and a missing rank are named
    c = bench.self_check(8, 8, "gloo", "nccl", list(range(7)), [1.0] * 7)
    assert c["failed"] == ["backend_is_nccl", "every_rank_reported"]
    assert bench.self_check(4, 8, "nccl", "nccl", [0, 1, 2, 3], [1.0] * 4)["failed"] == ["world_size_equals_gpus_flag"]
    assert bench.self_check(1, 1, None, "nccl", [0], [150.0])["ok"]


def test_watchdog_saves_the_headline_when_an_extra_section_hangs():
    """bench.py measures config 5 on all ranks BEHIND the headline of a multi-GPU run; if that collective section hangs,
    rank 0 still prints the line it has — and every rank leaves with a NON-ZERO exit code: a hung collective must not
    read as a successful run."""
    code = ("import sys, time, json; sys.path.insert(0, %r); import bench\n"
            "line = {'metric': 'm', 'value': 1.5}\n"
            "with bench.Watchdog(0.5, 0, line):\n"
            "    time.sleep(30)\n"
            "print('not reached')\n") % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    import bench
    assert bench.WATCHDOG_RC != 0
    assert r.returncode == bench.WATCHDOG_RC and "not reached" not in r.stdout
    d = _json_line(r.stdout)
    assert d["value"] == 1.5 and "abandoned" in d["config5_all_ranks"]["error"]
    code2 = code.replace("bench.Watchdog(0.5, 0, line)", "bench.Watchdog(0.5, 3, line)")
    r = subprocess.run([sys.executable, "-c", code2], capture_output=True, text=True, timeout=60)
    assert r.returncode == bench.WATCHDOG_RC and r.stdout.strip() == ""           # the other ranks print nothing, and fail too
