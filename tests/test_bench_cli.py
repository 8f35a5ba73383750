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
    # the CPU-oracle worker pool of the real run: spawned (fresh interpreters) before the process initialises torch / HIP / RCCL,
    # and it ran one reference-tracker job over the oracle
    assert d["config"]["cpu_pool"]["start_method"] == "spawn" and d["config"]["cpu_pool"]["oracle_steps"] == 2


def test_gpus_2_spawns_two_ranks():
    d = run("--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1")
    assert d["n_gpus"] == 2 and d["config"]["ranks"] == 2 and d["steps"] == 3 and d["warmup"] == 1
    assert d["config"]["cpu_pool"]["oracle_steps"] == 2      # rank 0's pool path works inside a torch.distributed.run rank as well


def test_pmc_csv_parsing_and_profiler_detection(tmp_path, monkeypatch):
    """The PMC probe's plumbing without a GPU: a nested profiler is refused, and a counter CSV of rocprofv3's layout condenses to
    instructions / bytes per point."""
    import bench
    import types
    args = types.SimpleNamespace(width=640, height=480, nfeatures=1000, max_points=None)
    rows = ['"Correlation_Id","Dispatch_Id","Agent_Id","Queue_Id","Process_Id","Thread_Id","Grid_Size","Kernel_Id","Kernel_Name","Workgroup_Size",'
            '"LDS_Block_Size","Scratch_Size","VGPR_Count","Accum_VGPR_Count","SGPR_Count","Counter_Name","Counter_Value","Start_Timestamp","End_Timestamp"']
    for d, (valu, fetch) in enumerate(((2000.0, 4.0), (6000.0, 12.0))):
        for name, v in (("SQ_INSTS_VALU", valu), ("SQ_WAVES", 8.0), ("FETCH_SIZE", fetch), ("WRITE_SIZE", 1.0)):
            rows.append(f'{d},{d},"Agent 2",2,1,1,64,1,"lk_track_kernel(LkArgs)",64,0,0,8,0,8,"{name}",{v},{1000 + d},{2000 + d}')

    def fake_run(cmd, **kw):
        d = cmd[cmd.index("-d") + 1]
        os.makedirs(os.path.join(d, "host"), exist_ok=True)
        open(os.path.join(d, "host", "p_counter_collection.csv"), "w").write("\n".join(rows) + "\n")
        return types.SimpleNamespace(returncode=0, stdout='PMC_PROBE {"points_per_launch": [1, 3]}\n', stderr="")
    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    monkeypatch.setattr(bench.shutil, "which", lambda _: sys.executable)
    out = bench.run_pmc(args)
    assert out["valu_instructions_per_point"] == 2000.0 and out["hbm_read_bytes_per_point"] == 2 * 1024 * 4.0 and out["hbm_write_bytes_per_point"] == 512.0


def test_usable_cores_respects_the_container_quota():
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    n = bench.usable_cores()
    assert 1 <= n <= (os.cpu_count() or 1) and n <= len(os.sched_getaffinity(0))
