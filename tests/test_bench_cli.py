"""bench.py's launcher contract on the CPU: `--gpus N` without WORLD_SIZE spawns N ranks itself (before any GPU call in the
parent), the ranks rendezvous (gloo in --dry-run), broadcast the intrinsics and rank 0 prints ONE JSON line with n_gpus = N."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(*args):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def test_dry_run_single_process():
    d = run("--dry-run")
    assert d["n_gpus"] == 1 and d["config"]["dry_run"] and d["metric"].startswith("tracker-step frames/sec @1280x720")
    for key in ("value", "unit", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert key in d


def test_gpus_2_spawns_two_ranks():
    d = run("--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1")
    assert d["n_gpus"] == 2 and d["config"]["ranks"] == 2 and d["steps"] == 3 and d["warmup"] == 1


def test_usable_cores_respects_the_container_quota():
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    n = bench.usable_cores()
    assert 1 <= n <= (os.cpu_count() or 1) and n <= len(os.sched_getaffinity(0))
