#!/usr/bin/env python3
"""bench.py — tracker-step frames/sec of the MI355X-native VO front-end (BASELINE.json metric, config C3).

One "step" = the reference's Tracker::update (src/tracker.cpp:274-333) for every camera stream resident on the GPU, each
stream taking ITS OWN branch on the device (mvo_batch_track, csrc/track.hip): LK + status/err filter -> LOST test ->
solvePnPRansac -> should_add_keyframe -> [findHomography + findFundamentalMat -> has_parallax -> [ORB + knn2/ratio match +
triangulate + landmark hand-over]].  Streams are DISTINCT rendered scenes with true parallax (synth_gpu: own billboard
layout, trajectory and noise per stream), seeded with landmarks from the renderer's depth.  Default workload = config C3:
1280x720, 2000 ORB features; `--width/--height/--nfeatures` select C2 (640x480 / 1000) or C4 (1920x1080 / 4000).
Per GPU: C contexts x B streams (default 4 x 512 up to 1280x720, 4 x 256 above), stepped asynchronously so that the one-wavefront-per-stream RANSAC
chains of one context run beside the wide LK / ORB kernels of another; contexts start 0..10 frames apart so that their
key-frame steps (every 11th frame under the reference's policy) do not coincide.

Phases, all on the same tracker state (rank-0 JSON line):
  value                 reference key-frame policy, frames resident in HBM                       <- the headline
  value_with_ingest     same, frames in pinned host memory and uploaded asynchronously each step (PCIe inside the metric)
  always_on_fps         policy 2: LK + PnP only (no key-frame test)
  keyframe_every_frame_fps   policy 1: key-frame branch on every frame (round 1's headline workload)
  single_stream         ONE camera (1 context x 1 stream, src/mono_vo.cpp:116): ms per step, frames/s
and at N = 1, on rank 0:
  roofline              the LK kernel against the bound that applies to it (integer VALU issue), from THIS run: launch duration
                        by HIP events, points per launch from the step results, VALU instructions per point and HBM bytes
                        (FETCH_SIZE x 2 + WRITE_SIZE) from rocprofv3 PMC passes of a probe child process started by this run
  int_mismatches, rt_max_abs_err    the timed run against the CPU oracle (tests/track_ref.py = the reference's Tracker over oracle/)
  cpu_baseline          the oracle built -O3 -march=native, 1 thread, median of 5 runs x 200 frames of one stream, per stage
  cpu_baseline_all_cores
The line fails (exit 1) if int_mismatches != 0 or rt_max_abs_err > 1e-4.

Multi-GPU: `--gpus N` without WORLD_SIZE spawns N ranks itself (torch.distributed.run, one process per GPU, RCCL);
streams are independent, the only collective is one broadcast of the intrinsics; scaling "weak".  At N > 1
the CPU legs, the PMC probe and the ingest phase are off (the contract asks for them at N = 1 only): a rank renders and
tracks its own streams and nothing else.
"""
import argparse
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

ALGO_BYTES_PER_LK_POINT = 4261  # 4 levels x (24^2 + 22^2) window bytes + 21 B of point I/O
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
N_SIMD = 1024                   # 256 CUs x 4 SIMDs
# ns one wave64 instruction of the LK kernel's integer mix (v_dot2_i32_i16, v_perm_b32, v_alignbyte_b32 at half rate, adds /
# shifts at full rate) occupies a SIMD at full issue: profiles/microbench/valu_mix.hip, wall-clock calibrated (1.49-1.58)
VALU_MIX_NS = 1.55
STAGE_TIMERS = ("frame_fanout", "lk_pyramid", "lk_worklist", "lk_track", "lk_filter", "pnp", "pnp_ransac", "pnp_refine", "ransac_h", "ransac_f",
                "kf_gather", "orb_detect", "orb_select", "orb_blur", "orb_describe", "kf_scatter", "match", "triangulate")
INT_KEYS = ("n_prev", "n_tracked", "pnp_ok", "n_pnp_inliers", "score_h", "score_f", "n_keypoints", "n_matches", "n_triangulated",
            "state", "flags", "tracking_count", "n_tracks")
CPU_STAGES = ("lk_track", "solve_pnp_ransac", "find_homography_ransac", "find_fundamental_ransac", "orb_detect_and_compute",
              "match_knn2_ratio", "triangulate")


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--contexts", type=int, default=4, help="contexts per GPU, stepped asynchronously")
    ap.add_argument("--batch", type=int, default=None, help="independent camera streams per context (default: 512 up to 1280x720, 256 above: "
                    "the frame rings of 4 contexts x 61 frames then take 115 GB / 129 GB of the 288 GB)")
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--nfeatures", type=int, default=2000)
    ap.add_argument("--max-points", type=int, default=None, help="track capacity per stream (default 4096; 8192 above 2000 features)")
    ap.add_argument("--ingest-steps", type=int, default=16, help="steps of the value_with_ingest phase (0 = skip)")
    ap.add_argument("--ingest-ring", type=int, default=3, help="ring entries the ingest phase cycles through (uploads run ring - 1 frames ahead)")
    ap.add_argument("--compute-streams", type=int, default=0, help="HIP streams the contexts' steps are enqueued on (context c uses stream c %% N; 0 = every context its own)")
    ap.add_argument("--depth", type=int, default=1, help="steps kept in flight per context (1 or 2: mvo_batch_track_async pipelines two)")
    ap.add_argument("--extra-steps", type=int, default=5, help="steps of the always-on and key-frame-every-frame phases (0 = skip)")
    ap.add_argument("--single-steps", type=int, default=30, help="steps of the single-stream measurement (0 = skip)")
    ap.add_argument("--cpu-streams", type=int, default=None, help="streams checked against the CPU oracle (default: host cores)")
    ap.add_argument("--cpu-frames", type=int, default=200, help="frames of one stream per CPU-baseline run (BASELINE.md 3)")
    ap.add_argument("--cpu-runs", type=int, default=5, help="runs of the 1-thread CPU baseline (median reported)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 PMC probe (roofline.traffic / instructions stay null)")
    ap.add_argument("--contexts-first", action="store_true", help="diagnostic: create every context before the first upload")
    ap.add_argument("--no-stagger", action="store_true", help="start all contexts on the same frame (key-frame steps coincide)")
    ap.add_argument("--dry-run", action="store_true", help="no GPU work: rendezvous (gloo), the CPU worker pool, and the line skeleton")
    ap.add_argument("--pmc-probe", action="store_true", help=argparse.SUPPRESS)   # child mode: LK launches for the profiler
    ap.add_argument("--dump-stream", type=int, default=None, help="diagnostic: save the frames + depth of this stream of context 0")
    ap.add_argument("--dump-path", default="gpurun_out/stream_dump.npz")
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N fresh ranks (one per GPU) BEFORE this process touches the GPU,
    hand the line through and exit with their status.  Never re-execs a process that has initialised HIP."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def usable_cores():
    """CPUs this process may actually use: the scheduler affinity and the cgroup CPU quota, whichever is smaller (a GPU box
    hands a container 16 of its 256 cores: os.cpu_count() alone overstates what an "all cores" run gets)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


# ---- CPU oracle workers (spawned BEFORE this process touches HIP; they never import torch or the HIP library) ------------
def _oracle_stream(job):
    """Worker: the reference's Tracker over the CPU oracle on one stream.
    -> (per-step result dicts, seconds, steps, per-stage seconds).  `lib`: an alternative build of the oracle (timing)."""
    K, nfeatures, frames, depth0, lib, warm = job
    if lib:
        os.environ["MVO_ORACLE_LIB"] = lib
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from track_ref import TrackRef
    from ros2_mono_vo_amd import synth
    r = TrackRef(K, nfeatures)
    stage = dict.fromkeys(CPU_STAGES, 0.0)

    def timed(name, fn):
        def w(*a, **k):
            t = time.perf_counter()
            try:
                return fn(*a, **k)
            finally:
                stage[name] += time.perf_counter() - t
        return w
    for name in CPU_STAGES:
        setattr(r.backend, name, timed(name, getattr(r.backend, name)))
    r.seed(frames[0], lambda xy: synth.depth_landmarks(K, depth0, xy))
    out = []
    for k in range(1, 1 + warm):
        out.append(r.step(frames[k]))
    for name in CPU_STAGES:
        stage[name] = 0.0
    w0 = time.time()
    t0 = time.perf_counter()
    for k in range(1 + warm, len(frames)):
        out.append(r.step(frames[k]))
    dt = time.perf_counter() - t0
    return out, dt, len(frames) - 1 - warm, stage, w0, time.time()


def build_native_oracle():
    """-O3 -march=native build of the oracle for the timing leg, made on the box that runs it.  -> path or None."""
    try:
        d = tempfile.mkdtemp(prefix="orc_native_")
        out = os.path.join(d, "liborc_native.so")
        srcs = sorted(glob.glob(os.path.join(ROOT, "oracle", "orc_*.cpp")))
        cmd = ["g++", "-O3", "-march=native", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-shared", "-o", out] + srcs
        return subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL), out
    except Exception:
        return None, None


def rodrigues(r):
    r = np.asarray(r, np.float64)
    th = float(np.sqrt(r @ r))
    if th < np.finfo(np.float64).eps:
        return np.eye(3)
    k = r / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.cos(th) * np.eye(3) + (1 - np.cos(th)) * np.outer(k, k) + np.sin(th) * Kx


# ---- rocprofv3 PMC probe -----------------------------------------------------------------------------------------------------
def pmc_probe(args):
    """Child mode (under rocprofv3): one context of 64 of the benchmark's streams, seeded, six tracker steps -> six launches of
    lk_track_kernel whose point counts go to stdout as JSON."""
    import torch
    from ros2_mono_vo_amd import Context, synth, synth_gpu
    W, H, B, NFR = args.width, args.height, 64, 8
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    Kmat = synth.default_K(W, H)
    bank = synth_gpu.SceneBank(dev)
    pitch = (W + 63) // 64 * 64
    frames = torch.zeros((NFR, B, H, pitch), dtype=torch.uint8, device=dev)
    depth0 = np.zeros((B, H, W), np.float32)
    for s in range(B):
        fr, d0 = synth_gpu.render_stream(bank, bank.stream_params(0x5EED0003 + s), Kmat, W, H, NFR)
        frames[:, s, :, :W] = fr
        depth0[s] = d0.cpu().numpy()
    torch.cuda.synchronize()
    pts = []
    with Context(max_width=W, max_height=H, batch=B, nfeatures=args.nfeatures, max_points=args.max_points or (8192 if args.nfeatures > 2000 else 4096),
                 ring_frames=NFR, device=0) as ctx:
        ctx.batch_set_intrinsics(Kmat, np.zeros(5))
        for f in range(NFR):
            ctx.batch_upload_async(f, frames[f].data_ptr(), W, H, pitch, H * pitch)
        ctx.sync()
        ctx.batch_seed(0)
        for s in range(B):
            ctx.batch_set_landmarks(s, synth_gpu.depth_landmarks(Kmat, depth0[s], ctx.batch_get_tracks(s)))
        for k in range(1, NFR - 1):
            pts.append(int(sum(r.n_prev for r in ctx.batch_track(k))))
    print("PMC_PROBE " + json.dumps({"points_per_launch": pts}), flush=True)


def run_pmc(args):
    """Two rocprofv3 PMC passes over the probe (this process has not touched the GPU yet).  -> dict for the roofline block."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {"error": "rocprofv3 not found"}
    out = {"source": "rocprofv3 --kernel-include-regex lk_track_kernel --pmc <counters> -- python3 bench.py --pmc-probe (1 context x 64 of the "
                     "benchmark's streams, six launches), run by this bench.py before its own GPU work",
           "passes": []}
    base = [sys.executable, os.path.abspath(__file__), "--pmc-probe", "--width", str(args.width), "--height", str(args.height),
            "--nfeatures", str(args.nfeatures)] + (["--max-points", str(args.max_points)] if args.max_points else [])
    env = dict(os.environ, TMPDIR="/tmp")
    vals, pts = {}, None
    for counters in (["SQ_INSTS_VALU", "SQ_WAVES", "FETCH_SIZE"], ["WRITE_SIZE"]):
        d = tempfile.mkdtemp(prefix="mvo_pmc_")
        cmd = [exe, "--kernel-include-regex", "lk_track_kernel", "--kernel-trace", "--pmc"] + counters + ["-d", d, "-o", "p", "--output-format", "csv", "--"] + base
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=240)
        except Exception as e:   # noqa: BLE001
            out["passes"].append({"counters": counters, "error": repr(e)})
            continue
        rec = {"counters": counters, "rc": r.returncode, "seconds": round(time.perf_counter() - t0, 1)}
        for line in r.stdout.splitlines():
            if line.startswith("PMC_PROBE "):
                pts = json.loads(line[len("PMC_PROBE "):])["points_per_launch"]
        try:
            import csv
            acc, dur = {}, []
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                seen = set()
                for row in csv.DictReader(open(f)):
                    if not row["Kernel_Name"].startswith("lk_track_kernel"):
                        continue
                    acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
                    if row["Dispatch_Id"] not in seen:
                        seen.add(row["Dispatch_Id"])
                        dur.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-6)
            for c, v in acc.items():
                vals[c] = v
            rec["dispatches"] = len(dur)
            rec["kernel_ms_avg_under_pmc"] = round(sum(dur) / len(dur), 4) if dur else None
        except Exception as e:   # noqa: BLE001
            rec["error"] = repr(e)
        if r.returncode != 0:
            rec["stderr_tail"] = r.stderr[-400:]
        out["passes"].append(rec)
        shutil.rmtree(d, ignore_errors=True)
    if pts and "SQ_INSTS_VALU" in vals and len(vals["SQ_INSTS_VALU"]) == len(pts):
        n = float(sum(pts))
        out["probe_points_per_launch"] = pts
        out["valu_instructions_per_point"] = sum(vals["SQ_INSTS_VALU"]) / n
        if "FETCH_SIZE" in vals:
            out["fetch_kb_per_launch_raw"] = sum(vals["FETCH_SIZE"]) / len(pts)
            # gfx950: FETCH_SIZE counts 128-byte fabric requests at 64 B (MI355X_MICROARCH.md, HBM; profiles/r02_fetch_calibration.json)
            out["hbm_read_bytes_per_point"] = 2.0 * 1024.0 * sum(vals["FETCH_SIZE"]) / n
        if "WRITE_SIZE" in vals and len(vals["WRITE_SIZE"]) == len(pts):
            out["hbm_write_bytes_per_point"] = 1024.0 * sum(vals["WRITE_SIZE"]) / n
    else:
        out["error"] = "probe produced no usable counters"
    return out


def main():
    args = parse_args()
    if args.pmc_probe:
        return pmc_probe(args)
    world = int(os.environ.get("WORLD_SIZE", "0"))
    if world == 0 and args.gpus > 1:
        sys.exit(spawn_ranks(args))
    world = max(world, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    solo = world == 1 and rank == 0          # the N = 1 extras: CPU legs, PMC probe, ingest phase, single stream

    # ---- everything that starts other processes happens BEFORE this process initialises HIP / RCCL ---------------------------
    cpu_on = solo and not args.no_cpu_baseline
    cores = usable_cores()
    pool = native_proc = native_lib = None
    if cpu_on or (args.dry_run and rank == 0):
        import multiprocessing as mp
        pool = mp.get_context("spawn").Pool(max(1, min(cores, 64)))      # fresh interpreters: numpy + the oracle only
        if cpu_on:
            native_proc, native_lib = build_native_oracle()
    pmc = None
    if solo and not args.no_pmc and not args.dry_run:
        if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):
            pmc = {"error": "this process runs under a profiler: the nested PMC probe is skipped"}
        else:
            pmc = run_pmc(args)
        if "valu_instructions_per_point" not in pmc:      # the probe is best effort: fall back to the committed passes, labelled as such
            try:
                old = json.load(open(os.path.join(ROOT, "profiles", "r03_k_lk_pmc.json")))
                pmc.update({k: old[k] for k in ("valu_instructions_per_point", "hbm_read_bytes_per_point", "hbm_write_bytes_per_point") if k in old})
                pmc["fallback"] = "counters from the committed profiles/r03_k_lk_pmc.json (" + old.get("source", "") + "), NOT from this run"
            except Exception:
                pass

    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.dry_run:
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    from ros2_mono_vo_amd import parallel, synth

    if args.batch is None:
        args.batch = 512 if args.width * args.height <= 1280 * 720 else 256
    W, H, B, C, K, Wm = args.width, args.height, args.batch, args.contexts, args.steps, args.warmup
    K2 = args.ingest_steps if solo else 0
    K3 = args.extra_steps
    maxpts = args.max_points or (8192 if args.nfeatures > 2000 else 4096)
    metric = f"tracker-step frames/sec @{W}x{H}, {args.nfeatures} ORB feats"
    base_line = {"metric": metric, "value": None, "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": Wm, "ms_per_step": None,
                 "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8/i32 fixed-point + f32/f64", "data": "synthetic"}
    if args.dry_run:
        Kmat, _ = parallel.broadcast_intrinsics(synth.default_K(W, H) if rank == 0 else np.zeros((3, 3)), np.zeros(5), dist, device="cpu")
        assert abs(Kmat[0, 0] - 0.9 * W) < 1e-9
        dt = parallel.max_over_ranks(0.0, dist, device="cpu")
        if rank == 0:
            # the CPU worker pool the real run uses (spawned above, before any GPU / RCCL initialisation): one tiny oracle job
            fr = synth.gen_stream(320, 240, 0x5EED0009, 3)
            d0 = np.full((240, 320), 10.0, np.float32)
            res = pool.map(_oracle_stream, [(synth.default_K(320, 240), 300, fr, d0, None, 0)])
            pool.close()
            base_line["config"] = {"workload": "dry run (no GPU work)", "dry_run": True, "ranks": world, "dt": dt,
                                   "cpu_pool": {"start_method": "spawn", "workers": max(1, min(cores, 64)), "oracle_steps": res[0][2]}}
            print(json.dumps(base_line), flush=True)
        if dist is not None:
            dist.destroy_process_group()
        return

    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    from ros2_mono_vo_amd import Context, _lib, synth_gpu

    # intrinsics: rank 0 owns them, one RCCL broadcast over xGMI (the path's only collective)
    Kmat, dcoef = parallel.broadcast_intrinsics(synth.default_K(W, H) if rank == 0 else np.zeros((3, 3)), np.zeros(5), dist, device="cuda")

    # ---- streams: rendered on this GPU, distinct per stream (and per rank) ------------------------------------------------
    offs = [0] * C if args.no_stagger else [(11 * c) // C for c in range(C)]           # frames a context starts ahead
    n_main = Wm + K                                                                     # steps every context takes in phase 1
    n_frames = 1 + max(offs) + n_main + K2 + 2 * K3 + 1                                  # + 1: the isolated LK launch at the end
    t_setup = time.perf_counter()
    bank = synth_gpu.SceneBank(dev)
    pitch = (W + 63) // 64 * 64
    cpu_n = 0
    if cpu_on:
        cpu_n = min(B, args.cpu_streams if args.cpu_streams is not None else min(cores, 64))
    seed0 = 0x5EED0003 + 100003 * rank
    # the CPU timing stream: stream 0 of context 0 continued to --cpu-frames frames (+ 20 of warm-up)
    cpu_long = None
    if cpu_on:
        n_long = 1 + 20 + args.cpu_frames
        fr, _ = synth_gpu.render_stream(bank, bank.stream_params(seed0), Kmat, W, H, n_long)
        cpu_long = fr.cpu().numpy()
    single = None
    if solo and args.single_steps:
        fr, d0 = synth_gpu.render_stream(bank, bank.stream_params(seed0), Kmat, W, H, 1 + 2 * args.single_steps)
        single = (fr.cpu().numpy(), d0.cpu().numpy())

    # ---- contexts, one at a time: render its B streams in the ring's layout ([frame][slot][H][pitch]), copy them into the
    # context's device ring, seed with depth landmarks, keep the host copies the later phases need, drop the staging tensor
    # (so the peak is the rings + ONE context's staging copy: 4 x 512 streams x 61 frames is 115 GB of rings)
    ctxs, pins = [], []
    cpu_frames = depth0 = None
    ing0 = [1 + offs[c] + n_main for c in range(C)]                                     # first frame of the ingest phase per context
    shared = [torch.cuda.Stream(device=dev) for _ in range(args.compute_streams)]       # kept alive for the run
    mk = lambda c: Context(max_width=W, max_height=H, batch=B, nfeatures=args.nfeatures, max_points=maxpts, ring_frames=n_frames, device=local_rank,
                           **({"hip_stream": shared[c % len(shared)].cuda_stream} if shared else {}))
    pre = [mk(c) for c in range(C)] if args.contexts_first else None     # diagnostic: creation order must not matter
    for c in range(C):
        frames = torch.zeros((n_frames, B, H, pitch), dtype=torch.uint8, device=dev)
        depth_dev = torch.zeros((B, H, W), dtype=torch.float32, device=dev)              # depth of frame 0: the seed landmarks
        for s in range(B):
            prm = bank.stream_params(seed0 + 1009 * c + s)
            fr, d0 = synth_gpu.render_stream(bank, prm, Kmat, W, H, n_frames)
            frames[:, s, :, :W] = fr
            depth_dev[s] = d0
        torch.cuda.synchronize()
        ctx = pre[c] if pre else mk(c)
        ctx.batch_set_intrinsics(Kmat, dcoef)
        for f in range(n_frames):
            ctx.batch_upload_async(f, frames[f].data_ptr(), W, H, pitch, H * pitch)     # device -> device, one copy per frame
        ctx.sync()
        ctx.batch_seed(0)
        dc = depth_dev.cpu().numpy()
        for s in range(B):
            ctx.batch_set_landmarks(s, synth_gpu.depth_landmarks(Kmat, dc[s], ctx.batch_get_tracks(s)))
        ctxs.append(ctx)
        if c == 0:
            depth0 = dc[:max(cpu_n, 1)].copy()              # host copies only of what the CPU legs need
            cpu_frames = frames[:1 + offs[0] + n_main, :cpu_n, :, :W].permute(1, 0, 2, 3).contiguous().cpu().numpy() if cpu_n else None
            if args.dump_stream is not None and rank == 0:
                np.savez_compressed(args.dump_path, frames=frames[:1 + n_main, args.dump_stream, :, :W].cpu().numpy(),
                                    depth0=depth0[min(args.dump_stream, len(depth0) - 1)], K=Kmat)
        if K2:                                                # frames of the ingest phase go to pinned host memory
            pin = ctx.host_alloc(K2 * B * H * pitch).reshape(K2, B, H, pitch)
            pin[:] = frames[ing0[c]:ing0[c] + K2].cpu().numpy()
            pins.append(pin)
        del frames, depth_dev, dc
    torch.cuda.empty_cache()
    t_setup = time.perf_counter() - t_setup

    RING = max(2, args.ingest_ring)

    def run_phase(n_steps, first_frame, record=None, ingest=False):
        """Every context takes n_steps steps (its frames first_frame[c] ...), kept in flight independently: as soon as a
        context's step is collected its next one is enqueued.  -> elapsed seconds (barrier + device sync on both sides)."""
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        if ingest:
            for j in range(min(RING - 1, n_steps)):
                for c, ctx in enumerate(ctxs):
                    ctx.batch_upload_async(j, pins[c][j].ctypes.data, W, H, pitch, H * pitch)
        done = [0] * C          # steps collected per context
        sent = [0] * C          # steps enqueued per context

        def enqueue(c):
            k = sent[c]
            ctxs[c].batch_track_async(k % RING if ingest else first_frame[c] + k)
            if ingest and k + RING - 1 < n_steps:      # uploads run RING - 1 frames ahead of the step that consumes them
                ctxs[c].batch_upload_async((k + RING - 1) % RING, pins[c][k + RING - 1].ctypes.data, W, H, pitch, H * pitch)
            sent[c] += 1

        for _ in range(max(1, min(2, args.depth))):     # steps in flight per context: the next one is queued behind the running one,
            for c in range(C):                           # so a context's stream never waits for this loop between two steps
                if sent[c] < n_steps:
                    enqueue(c)
        while min(done) < n_steps:
            progressed = False
            for c, ctx in enumerate(ctxs):      # a context whose step has finished is collected and re-armed at once;
                if done[c] < sent[c] and ctx.batch_track_poll():      # a slow step of one context does not hold the others
                    out = ctx.batch_track_wait()
                    if record is not None:
                        record[c].append(out)
                    done[c] += 1
                    if sent[c] < n_steps:
                        enqueue(c)
                    progressed = True
            if not progressed:
                time.sleep(0.0001)
        for ctx in ctxs:
            ctx.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        return parallel.max_over_ranks(time.perf_counter() - t0, dist, device="cuda")

    # ---- phase 0 (untimed): stagger + warm-up.  Context c takes offs[c] extra steps first. -------------------------------
    rec = [[] for _ in range(C)]
    for j in range(max(offs)):
        for c, ctx in enumerate(ctxs):
            if j < offs[c]:
                rec[c].append(ctx.batch_track(1 + j))
    run_phase(Wm, [1 + offs[c] for c in range(C)], rec)
    for ctx in ctxs:
        ctx.profile_reset()
        ctx.profile_enable(True)
    # ---- phase 1 (the metric): reference policy, frames resident ---------------------------------------------------------
    n_before = [len(r) for r in rec]
    dt = run_phase(K, [1 + offs[c] + Wm for c in range(C)], rec)
    for ctx in ctxs:
        ctx.profile_enable(False)
    prof = {}
    for name in STAGE_TIMERS:
        ms = n = 0
        for ctx in ctxs:
            a, b = ctx.profile_read(name)
            ms += a; n += b
        if n:
            prof[name] = {"ms_total": round(ms, 4), "launches": n, "ms_avg": round(ms / n, 5)}
    timed = [r for c in range(C) for step in rec[c][n_before[c]:] for r in step]
    lk_points = sum(r.n_prev for r in timed)
    flags = np.array([r.flags for r in timed])
    states_end = np.array([r.state for c in range(C) for r in rec[c][-1]])
    # ---- phase 2: the same with ingest inside the metric (ring entries 0 .. RING-1 of the resident ring are reused) -------------
    nxt = [ing0[c] + K2 for c in range(C)]
    dt_ing = None
    if K2:
        dt_ing = run_phase(K2, None, None, ingest=True)
        # back to the resident frames: the frame after the last ingested one continues each stream
    # ---- phase 3 / 4: always-on part only, key-frame branch on every frame --------------------------------------------------
    dt_on = dt_kf = None
    if K3:
        for ctx in ctxs:
            ctx.batch_set_policy(2)
        dt_on = run_phase(K3, nxt)
        for ctx in ctxs:
            ctx.batch_set_policy(1)
        dt_kf = run_phase(K3, [n + K3 for n in nxt])
    # ---- LK kernel alone (contexts one at a time): the roofline's launch duration without other contexts beside it ----------
    for ctx in ctxs:
        ctx.batch_set_policy(2)
        ctx.profile_reset()
        ctx.profile_enable(True)
    iso_pts = 0
    for c, ctx in enumerate(ctxs):
        iso_pts += sum(r.n_prev for r in ctx.batch_track(nxt[c] + 2 * K3))
    iso_ms = sum(ctx.profile_read("lk_track")[0] for ctx in ctxs)
    iso_pyr_ms = sum(ctx.profile_read("lk_pyramid")[0] for ctx in ctxs) / C
    for ctx in ctxs:
        ctx.profile_enable(False)
    for ctx in ctxs:
        ctx.close()
    ctxs = []
    torch.cuda.empty_cache()

    # ---- ONE camera: 1 context x 1 stream, synchronous steps (src/mono_vo.cpp:116 is one frame at a time) --------------------------
    single_out = None
    if single is not None:
        fr, d0 = single
        ns = args.single_steps
        with Context(max_width=W, max_height=H, batch=1, nfeatures=args.nfeatures, max_points=maxpts, ring_frames=1 + 2 * ns, device=local_rank) as c1:
            c1.batch_set_intrinsics(Kmat, dcoef)
            for f in range(1 + ns):
                c1.batch_preload_frame(0, f, fr[f])
            c1.batch_seed(0)
            c1.batch_set_landmarks(0, synth_gpu.depth_landmarks(Kmat, d0, c1.batch_get_tracks(0)))
            for k in range(1, 6):                                  # warm-up
                c1.batch_track(k)
            c1.sync()
            t0 = time.perf_counter()
            for k in range(6, 1 + ns):
                c1.batch_track(k)                                 # frame resident in HBM, results in host memory on return
            t_res = (time.perf_counter() - t0) / (ns - 5)
            # the same with the frame coming from host memory inside the step (mvo_batch_preload_frame + mvo_batch_track)
            t0 = time.perf_counter()
            for k in range(1 + ns, 1 + 2 * ns):
                c1.batch_preload_frame(0, k, fr[k])
                c1.batch_track(k)
            t_up = (time.perf_counter() - t0) / ns
            st, _ = c1.batch_get_state()
        single_out = {"contexts": 1, "batch": 1, "ms_per_step": round(t_res * 1e3, 4), "fps": round(1.0 / t_res, 1),
                      "ms_per_step_with_upload": round(t_up * 1e3, 4), "fps_with_upload": round(1.0 / t_up, 1),
                      "steps": ns - 5, "tracking_at_end": bool(st[0] == _lib.TRACK_TRACKING),
                      "note": "synchronous mvo_batch_track per frame; `with_upload` adds the pageable host -> HBM copy of the frame"}

    rcode = 0
    if rank == 0:
        streams = C * B * world
        line = dict(base_line)
        lk = prof.get("lk_track", {"ms_avg": 0.0, "launches": 0})
        pts_per_launch = lk_points / max(lk["launches"], 1)
        algo = ALGO_BYTES_PER_LK_POINT * pts_per_launch
        hbm_achieved = algo / (lk["ms_avg"] * 1e-3) / 1e9 if lk["ms_avg"] else 0.0
        iso_hbm = ALGO_BYTES_PER_LK_POINT * iso_pts / (iso_ms * 1e-3) / 1e9 if iso_ms else 0.0
        ipp = (pmc or {}).get("valu_instructions_per_point")
        rd, wr = (pmc or {}).get("hbm_read_bytes_per_point"), (pmc or {}).get("hbm_write_bytes_per_point")
        traffic = int((rd + (wr or 0.0)) * pts_per_launch) if rd is not None else None
        # VALU-issue roofline: what the launch would take if every SIMD issued the kernel's instructions back to back
        valu_floor_ms = ipp * pts_per_launch * VALU_MIX_NS * 1e-6 / N_SIMD if ipp else None
        valu_frac = valu_floor_ms / lk["ms_avg"] if (ipp and lk["ms_avg"]) else None
        iso_valu_frac = (ipp * iso_pts * VALU_MIX_NS * 1e-6 / N_SIMD) / iso_ms if (ipp and iso_ms) else None
        peak_ginst = N_SIMD / VALU_MIX_NS            # wave-instructions per ns over the chip = G wave-instructions / s
        ach_ginst = ipp * pts_per_launch / (lk["ms_avg"] * 1e6) if (ipp and lk["ms_avg"]) else None
        line.update({
            "value": round(streams * K / dt, 2), "ms_per_step": round(dt / K * 1e3, 4),
            "value_with_ingest": round(streams * K2 / dt_ing, 2) if dt_ing else None,
            "always_on_fps": round(streams * K3 / dt_on, 2) if dt_on else None,
            "keyframe_every_frame_fps": round(streams * K3 / dt_kf, 2) if dt_kf else None,
            "single_stream": single_out,
            "config": {"workload": f"{W}x{H} mono8, {args.nfeatures} ORB; Tracker::update per stream on the device (LK, PnP-RANSAC, key-frame "
                                   f"policy, H/F-RANSAC, ORB + match + triangulate on key-frames); {C} contexts x {B} distinct rendered "
                                   f"true-parallax streams per GPU, frames resident in HBM",
                       "streams_per_gpu": C * B, "contexts_per_gpu": C, "batch_per_context": B, "width": W, "height": H,
                       "nfeatures": args.nfeatures, "max_points": maxpts, "parallelism": f"streams x{world}", "context_start_offsets": offs,
                       "ingest_steps": K2, "ingest_ring": RING, "extra_steps": K3,
                       "mean_tracks_per_frame": round(lk_points / max(len(timed), 1), 1),
                       "keyframe_test_frac": round(float(np.mean((flags & _lib.STEP_KF_CHECKED) != 0)), 4),
                       "keyframe_frac": round(float(np.mean((flags & _lib.STEP_KEYFRAME) != 0)), 4),
                       "pnp_failed_frames": int(np.sum((flags & _lib.STEP_PNP_FAILED) != 0)),
                       "streams_tracking_at_end": int((states_end == _lib.TRACK_TRACKING).sum()), "streams": C * B,
                       "setup_s": round(t_setup, 1)},
            "roofline": {"bound": "valu", "kernel": "lk_track_kernel",
                         "achieved": round(ach_ginst, 2) if ach_ginst else None, "peak": round(peak_ginst, 2), "unit": "G wave-instructions/s",
                         "frac": round(valu_frac, 5) if valu_frac else None,
                         "traffic": traffic,
                         "definition": "integer-VALU issue roofline: achieved = VALU wave-instructions per launch (SQ_INSTS_VALU per point from this run's "
                                       "PMC probe x points per launch of the timed region) / launch duration (HIP events, timed region); peak = 1024 SIMDs / "
                                       f"{VALU_MIX_NS} ns per wave-instruction of this kernel's mix (profiles/microbench/valu_mix.hip)",
                         "valu_instructions_per_point": round(ipp, 1) if ipp else None,
                         "valu_floor_ms_per_launch": round(valu_floor_ms, 4) if valu_floor_ms else None,
                         "points_per_launch": round(pts_per_launch, 1), "avg_launch_ms": lk["ms_avg"], "launches": lk["launches"],
                         "hbm": {"achieved": round(hbm_achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(hbm_achieved / HBM_PEAK_GBS, 5),
                                 "algorithmic_bytes_per_point": ALGO_BYTES_PER_LK_POINT, "algorithmic_bytes_per_launch": int(algo),
                                 "traffic_bytes_per_point": round(rd + (wr or 0.0), 1) if rd is not None else None,
                                 "note": "north_star's stated roofline (HBM) kept as the secondary figure: the kernel is not byte-bound"},
                         "isolated": {"frac": round(iso_valu_frac, 5) if iso_valu_frac else None, "hbm_frac": round(iso_hbm / HBM_PEAK_GBS, 5),
                                      "ms_per_launch": round(iso_ms / C, 4), "points_per_launch": round(iso_pts / C, 1),
                                      "lk_pyramid_ms": round(iso_pyr_ms, 4),
                                      "note": "the same kernel with the GPU to itself (contexts one at a time)"},
                         "pmc": pmc if pmc is not None else {"error": "PMC probe off (--no-pmc or N > 1)"}},
            "stage_ms": prof,
        })
        if cpu_on:
            # the oracle on the same frames: parity of the timed run + the CPU baseline, 1 thread and all cores.  The workers
            # were spawned before this process touched HIP; the jobs run now, after the GPU phases, so neither side disturbs the other
            if native_proc is not None and native_proc.wait() != 0:
                native_lib = None
            jobs = [(Kmat, args.nfeatures, cpu_frames[s], depth0[s], None, 0) for s in range(cpu_n)]
            res = pool.map(_oracle_stream, jobs)
            t_all = max(r[5] for r in res) - min(r[4] for r in res)
            tj = (Kmat, args.nfeatures, cpu_long, depth0[0], native_lib, 20)
            tim = pool.map(_oracle_stream, [tj] * max(1, args.cpu_runs))      # 1 thread each, beside each other (one core per run)
            pool.close()
            allr = [r[0] for r in res]
            # A stream whose ORACLE pose stops being a pose (LM diverged from an ill-conditioned DLT start: |rvec| > pi or |tvec| > 1e3,
            # DESIGN 5) follows garbage from there on, on the CPU and on the device alike but not the same garbage: its later
            # frames are counted, not compared.  Everything before a stream's first such frame must agree.
            mism, rt_err, checked, worst, ill, mism_after, first_bad = 0, 0.0, 0, None, 0, 0, None
            for s in range(cpu_n):
                diverged = False
                for k, e in enumerate(allr[s]):
                    o = rec[0][k][s]
                    checked += 1
                    bad_pose = bool(e["flags"] & _lib.STEP_POSE) and (np.linalg.norm(e["rvec"]) > np.pi or np.linalg.norm(e["tvec"]) > 1e3)
                    if bad_pose:
                        ill += 1
                        diverged = True
                    bad = [key for key in INT_KEYS if int(getattr(o, key)) != int(e[key])]
                    if diverged:
                        mism_after += len(bad)
                        continue
                    mism += len(bad)
                    if bad and first_bad is None:
                        first_bad = {"stream": s, "frame": k + 1, "keys": bad}
                    if e["flags"] & _lib.STEP_POSE:
                        err = max(float(np.abs(rodrigues(o.rvec) - rodrigues(e["rvec"])).max()),
                                  float(np.abs(np.array(o.tvec) - e["tvec"]).max() / max(1.0, float(np.linalg.norm(e["tvec"])))))
                        if err > rt_err:
                            rt_err = err
                            worst = {"stream": s, "frame": k + 1, "err": err, "rvec": list(o.rvec), "tvec": list(o.tvec),
                                     "rvec_cpu": [float(v) for v in e["rvec"]], "tvec_cpu": [float(v) for v in e["tvec"]],
                                     "n_tracked": int(o.n_tracked), "n_pnp_inliers": int(o.n_pnp_inliers)}
            model = "unknown"
            try:
                model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
            except Exception:
                pass
            line["int_mismatches"] = mism
            line["rt_max_abs_err"] = rt_err
            line["rt_worst"] = worst
            line["rt_frames_oracle_pose_diverged"] = ill
            line["int_mismatches_after_oracle_divergence"] = mism_after
            if ill or mism_after:
                line["parity_warning"] = (f"{ill} frame(s) where the ORACLE's own pose is not a pose (|rvec| > pi or |tvec| > 1e3); frames of those streams "
                                          f"after that point are excluded from int_mismatches ({mism_after} differing integers there)")
            line["first_mismatch"] = first_bad
            line["parity_checked"] = f"{cpu_n} streams x {len(allr[0])} frames of context 0 (warm-up + timed steps) vs the CPU oracle: {checked} frame results"
            fps = sorted(r[2] / r[1] for r in tim)
            med = tim[[r[2] / r[1] for r in tim].index(fps[len(fps) // 2])]
            line["cpu_baseline"] = {"value": round(fps[len(fps) // 2], 3), "unit": "frames/s", "cores": 1, "kind": "port",
                                    "runs_fps": [round(v, 3) for v in fps], "frames_per_run": tim[0][2], "warmup_frames": 20,
                                    "build": "g++ -O3 -march=native -ffp-contract=off (built on this box)" if native_lib else "g++ -O2 -ffp-contract=off (native build failed)",
                                    "cpu_model": model,
                                    "stage_ms_per_frame": {k: round(v / med[2] * 1e3, 3) for k, v in med[3].items()},
                                    "end_to_end_ms_per_frame": round(med[1] / med[2] * 1e3, 3),
                                    "sample": f"oracle (CPU restatement of OpenCV-4.6 semantics, NOT OpenCV) behind the reference's Tracker (tests/track_ref.py): "
                                              f"stream 0 of context 0 continued to {tim[0][2]} frames after 20 of warm-up, 1 thread per run, median of "
                                              f"{len(tim)} runs (beside each other, one core each, GPU idle)"}
            nw = min(cores, 64)
            line["cpu_baseline_all_cores"] = {"value": round(sum(r[2] for r in res) / t_all, 3), "unit": "frames/s", "cores": nw,
                                              "host_cores": os.cpu_count(), "cpu_model": model, "kind": "port", "build": "g++ -O2 -ffp-contract=off (the parity build)",
                                              "sample": f"{cpu_n} streams x {res[0][2]} frames (the parity check's jobs) over a pool of {nw} processes = the CPUs this container "
                                                        f"may use (affinity and cgroup quota; the host has {os.cpu_count()}); frames / (last end - first start)"}
            if mism != 0 or rt_err > 1e-4:
                rcode = 1
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()
    sys.exit(rcode)


if __name__ == "__main__":
    main()
