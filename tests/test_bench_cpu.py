"""bench.py's multi-rank plumbing without a device: `--gpus N` starts N ranks itself (torch.distributed.run as a child
process, before anything touches the GPU), the ranks rendezvous, reduce the timing (MAX over ranks) and rank 0 prints
ONE JSON line with n_gpus = N.  `--dry-run` runs exactly that host logic on gloo and reports no value."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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
