#!/usr/bin/env python3
"""bench.py — tracker-step frames/sec of the MI355X-native VO front-end (BASELINE.json metric, config C3).

One "step" = the reference's Tracker::update (src/tracker.cpp:274-333) for every camera stream resident on the GPU, each
stream taking ITS OWN branch on the device (mvo_batch_track, csrc/track.hip): LK + status/err filter -> LOST test ->
solvePnPRansac -> should_add_keyframe -> [findHomography + findFundamentalMat -> has_parallax -> [ORB + knn2/ratio match +
triangulate + landmark hand-over]].  Streams are DISTINCT rendered 1280x720 scenes with true parallax (synth_gpu: own
billboard layout, trajectory and noise per stream), 2000 ORB features, seeded with landmarks from the renderer's depth.
Per GPU: C contexts x B streams (default 3 x 256), stepped asynchronously so that the one-wavefront-per-stream RANSAC
chains of one context run beside the wide LK / ORB kernels of another; contexts start 0..10 frames apart so that their
key-frame steps (every 11th frame under the reference's policy) do not coincide.

Phases, all on the same tracker state (rank-0 JSON line):
  value                 reference key-frame policy, frames resident in HBM                       <- the headline
  value_with_ingest     same, frames in pinned host memory and uploaded asynchronously each step (PCIe inside the metric)
  always_on_fps         policy 2: LK + PnP only (no key-frame test)
  keyframe_every_frame_fps   policy 1: key-frame branch on every frame (round 1's headline workload)
and, against the CPU oracle on the same frames (tests/track_ref.py = the reference's Tracker over oracle/):
  int_mismatches, rt_max_abs_err (R entries, t relative), cpu_baseline (1 thread) and cpu_baseline_all_cores.
The line fails (exit 1) if int_mismatches != 0 or rt_max_abs_err > 1e-4.

Multi-GPU: `--gpus N` without WORLD_SIZE spawns N ranks itself (torch.distributed.run, one process per GPU, RCCL);
streams are independent, the only collective is one broadcast of the intrinsics; scaling "weak".
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

ALGO_BYTES_PER_LK_POINT = 4261  # 4 levels x (24^2 + 22^2) window bytes + 21 B of point I/O
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
STAGE_TIMERS = ("frame_fanout", "lk_pyramid", "lk_worklist", "lk_track", "lk_filter", "pnp", "pnp_ransac", "pnp_refine", "ransac_h", "ransac_f", "kf_gather", "orb_detect", "orb_select",
                "orb_blur", "orb_describe", "kf_scatter", "match", "triangulate")
INT_KEYS = ("n_prev", "n_tracked", "pnp_ok", "n_pnp_inliers", "score_h", "score_f", "n_keypoints", "n_matches", "n_triangulated",
            "state", "flags", "tracking_count", "n_tracks")


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--contexts", type=int, default=4, help="contexts per GPU, stepped asynchronously")
    ap.add_argument("--batch", type=int, default=256, help="independent camera streams per context")
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--nfeatures", type=int, default=2000)
    ap.add_argument("--ingest-steps", type=int, default=16, help="steps of the value_with_ingest phase (0 = skip)")
    ap.add_argument("--ingest-ring", type=int, default=2, help="ring entries the ingest phase cycles through (uploads run ring - 1 frames ahead)")
    ap.add_argument("--extra-steps", type=int, default=5, help="steps of the always-on and key-frame-every-frame phases (0 = skip)")
    ap.add_argument("--cpu-streams", type=int, default=None, help="streams checked against / timed on the CPU oracle (default: host cores)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--contexts-first", action="store_true", help="diagnostic: create every context before the first upload")
    ap.add_argument("--no-stagger", action="store_true", help="start all contexts on the same frame (key-frame steps coincide)")
    ap.add_argument("--dry-run", action="store_true", help="no GPU work: rendezvous (gloo) and print the line skeleton")
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


def _oracle_stream(job):
    """Worker: the reference's Tracker over the CPU oracle on one stream.  -> (per-step result dicts, seconds, steps)."""
    K, nfeatures, frames, depth0 = job
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from track_ref import TrackRef
    from ros2_mono_vo_amd import synth_gpu
    r = TrackRef(K, nfeatures)
    r.seed(frames[0], lambda xy: synth_gpu.depth_landmarks(K, depth0, xy))
    out = []
    t0 = time.perf_counter()
    for k in range(1, len(frames)):
        out.append(r.step(frames[k]))
    return out, time.perf_counter() - t0, len(frames) - 1


def rodrigues(r):
    r = np.asarray(r, np.float64)
    th = float(np.sqrt(r @ r))
    if th < np.finfo(np.float64).eps:
        return np.eye(3)
    k = r / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.cos(th) * np.eye(3) + (1 - np.cos(th)) * np.outer(k, k) + np.sin(th) * Kx


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "0"))
    if world == 0 and args.gpus > 1:
        sys.exit(spawn_ranks(args))
    world = max(world, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

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

    W, H, B, C, K, Wm = args.width, args.height, args.batch, args.contexts, args.steps, args.warmup
    K2, K3 = args.ingest_steps, args.extra_steps
    metric = f"tracker-step frames/sec @{W}x{H}, {args.nfeatures} ORB feats"
    base_line = {"metric": metric, "value": None, "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": Wm, "ms_per_step": None,
                 "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8/i32 fixed-point + f32/f64", "data": "synthetic"}
    if args.dry_run:
        Kmat, _ = parallel.broadcast_intrinsics(synth.default_K(W, H) if rank == 0 else np.zeros((3, 3)), np.zeros(5), dist, device="cpu")
        assert abs(Kmat[0, 0] - 0.9 * W) < 1e-9
        dt = parallel.max_over_ranks(0.0, dist, device="cpu")
        if rank == 0:
            base_line["config"] = {"workload": "dry run (no GPU work)", "dry_run": True, "ranks": world, "dt": dt}
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
    frames = torch.zeros((C, n_frames, B, H, pitch), dtype=torch.uint8, device=dev)     # ring layout of a context: [frame][slot][H][pitch]
    depth0 = np.zeros((C, B, H, W), np.float32)
    for c in range(C):
        for s in range(B):
            prm = bank.stream_params(0x5EED0003 + 100003 * rank + 1009 * c + s)
            fr, d0 = synth_gpu.render_stream(bank, prm, Kmat, W, H, n_frames)
            frames[c, :, s, :, :W] = fr
            depth0[c, s] = d0.cpu().numpy()
    torch.cuda.synchronize()

    # ---- contexts: frames resident in the device ring, seeded with depth landmarks -------------------------------------
    ctxs = []
    pre = [Context(max_width=W, max_height=H, batch=B, nfeatures=args.nfeatures, max_points=4096, ring_frames=n_frames, device=local_rank)
           for _ in range(C)] if args.contexts_first else None     # diagnostic: creation order must not matter
    for c in range(C):
        ctx = pre[c] if pre else Context(max_width=W, max_height=H, batch=B, nfeatures=args.nfeatures, max_points=4096, ring_frames=n_frames,
                                         device=local_rank)
        ctx.batch_set_intrinsics(Kmat, dcoef)
        for f in range(n_frames):
            ctx.batch_upload_async(f, frames[c, f].data_ptr(), W, H, pitch, H * pitch)     # device -> device, one copy per frame
        ctx.sync()
        ctx.batch_seed(0)
        for s in range(B):
            ctx.batch_set_landmarks(s, synth_gpu.depth_landmarks(Kmat, depth0[c, s], ctx.batch_get_tracks(s)))
        ctxs.append(ctx)
    # frames of the ingest phase go to pinned host memory; CPU-oracle streams are downloaded; then the device copy is dropped
    cpu_n = 0
    if not args.no_cpu_baseline and rank == 0:
        cpu_n = min(B, args.cpu_streams if args.cpu_streams is not None else min(os.cpu_count() or 1, 64))
    cpu_frames = frames[0, :1 + offs[0] + n_main, :cpu_n, :, :W].permute(1, 0, 2, 3).contiguous().cpu().numpy() if cpu_n else None
    if args.dump_stream is not None and rank == 0:
        np.savez_compressed(args.dump_path, frames=frames[0, :1 + n_main, args.dump_stream, :, :W].cpu().numpy(),
                            depth0=depth0[0, args.dump_stream], K=Kmat)
    ing0 = [1 + offs[c] + n_main for c in range(C)]                                     # first frame of the ingest phase per context
    pins = []
    if K2:
        for c in range(C):
            pin = ctxs[c].host_alloc(K2 * B * H * pitch).reshape(K2, B, H, pitch)
            pin[:] = frames[c, ing0[c]:ing0[c] + K2].cpu().numpy()
            pins.append(pin)
    del frames
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

        for c in range(C):
            if n_steps:
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
    # ---- phase 2: the same with ingest inside the metric ---------------------------------------------------------------------
    dt_ing = run_phase(K2, None, None, ingest=True) if K2 else None
    nxt = [ing0[c] + K2 for c in range(C)]
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
    for ctx in ctxs:
        ctx.profile_enable(False)

    if rank == 0:
        streams = C * B * world
        line = dict(base_line)
        lk = prof.get("lk_track", {"ms_avg": 0.0, "launches": 0})
        pts_per_launch = lk_points / max(lk["launches"], 1)
        algo = ALGO_BYTES_PER_LK_POINT * pts_per_launch
        achieved = algo / (lk["ms_avg"] * 1e-3) / 1e9 if lk["ms_avg"] else 0.0
        iso_achieved = ALGO_BYTES_PER_LK_POINT * iso_pts / (iso_ms * 1e-3) / 1e9 if iso_ms else 0.0
        pmc = {}
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r02_lk_pmc.json")))
        except Exception:
            pass
        traffic = int(pmc["hbm_bytes_per_point"] * pts_per_launch) if "hbm_bytes_per_point" in pmc else None
        line.update({
            "value": round(streams * K / dt, 2), "ms_per_step": round(dt / K * 1e3, 4),
            "value_with_ingest": round(streams * K2 / dt_ing, 2) if dt_ing else None,
            "always_on_fps": round(streams * K3 / dt_on, 2) if dt_on else None,
            "keyframe_every_frame_fps": round(streams * K3 / dt_kf, 2) if dt_kf else None,
            "config": {"workload": f"C3: {W}x{H} mono8, {args.nfeatures} ORB; Tracker::update per stream on the device (LK, PnP-RANSAC, key-frame "
                                   f"policy, H/F-RANSAC, ORB + match + triangulate on key-frames); {C} contexts x {B} distinct rendered "
                                   f"true-parallax streams per GPU, frames resident in HBM",
                       "streams_per_gpu": C * B, "contexts_per_gpu": C, "batch_per_context": B, "width": W, "height": H,
                       "nfeatures": args.nfeatures, "parallelism": f"streams x{world}", "context_start_offsets": offs,
                       "ingest_steps": K2, "extra_steps": K3,
                       "mean_tracks_per_frame": round(lk_points / max(len(timed), 1), 1),
                       "keyframe_test_frac": round(float(np.mean((flags & _lib.STEP_KF_CHECKED) != 0)), 4),
                       "keyframe_frac": round(float(np.mean((flags & _lib.STEP_KEYFRAME) != 0)), 4),
                       "streams_tracking_at_end": int((states_end == _lib.TRACK_TRACKING).sum()), "streams": C * B,
                       "setup_s": round(t_setup, 1)},
            "roofline": {"bound": "hbm", "kernel": "lk_track_kernel", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "traffic_source": pmc.get("source", "none: run the rocprofv3 PMC passes of profiles/README"),
                         "algorithmic_bytes_per_launch": int(algo), "points_per_launch": round(pts_per_launch, 1),
                         "avg_launch_ms": lk["ms_avg"], "launches": lk["launches"],
                         "note": "launches of the timed region run beside other contexts' kernels; `isolated` = the same kernel with the GPU to itself",
                         "isolated": {"achieved": round(iso_achieved, 2), "frac": round(iso_achieved / HBM_PEAK_GBS, 5),
                                      "ms_per_launch": round(iso_ms / C, 4), "points_per_launch": round(iso_pts / C, 1)},
                         "valu_instructions_per_point": pmc.get("valu_instructions_per_point"),
                         "valu_issue_frac": pmc.get("valu_issue_frac")},
            "stage_ms": prof,
        })
        rcode = 0
        if cpu_n:
            # the oracle on the same frames: parity of the timed run + the CPU baseline, 1 thread and all cores
            jobs = [(Kmat, args.nfeatures, cpu_frames[s], depth0[0, s]) for s in range(cpu_n)]
            one, t1, n1 = _oracle_stream(jobs[0])
            allr, t_all = [one], None
            cores = usable_cores()
            if cpu_n > 1:
                import multiprocessing as mp
                t0 = time.perf_counter()
                with mp.get_context("fork").Pool(min(cores, cpu_n - 1)) as pool:
                    rest = pool.map(_oracle_stream, jobs[1:])
                t_all = time.perf_counter() - t0
                allr += [r[0] for r in rest]
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
            line["first_mismatch"] = first_bad
            line["parity_checked"] = f"{cpu_n} streams x {len(allr[0])} frames of context 0 (warm-up + timed steps) vs the CPU oracle: {checked} frame results"
            line["cpu_baseline"] = {"value": round(n1 / t1, 3), "unit": "frames/s", "cores": 1, "kind": "port",
                                    "sample": f"oracle (CPU restatement of OpenCV-4.6 semantics, not OpenCV) behind the reference's Tracker "
                                              f"(tests/track_ref.py), stream 0 of context 0, {n1} consecutive frames of the benchmarked run, 1 thread"}
            if t_all:
                nw = min(cores, cpu_n - 1)
                line["cpu_baseline_all_cores"] = {"value": round(sum(r[2] for r in rest) / t_all, 3), "unit": "frames/s", "cores": nw,
                                                  "host_cores": os.cpu_count(), "cpu_model": model, "kind": "port",
                                                  "sample": f"{cpu_n - 1} streams x {n1} frames over a pool of {nw} processes = the CPUs this "
                                                            f"container may use (affinity and cgroup quota; the host has {os.cpu_count()})"}
            if mism != 0 or rt_err > 1e-4:
                rcode = 1
        print(json.dumps(line), flush=True)
    else:
        rcode = 0
    for ctx in ctxs:
        ctx.close()
    if dist is not None:
        dist.destroy_process_group()
    sys.exit(rcode)


if __name__ == "__main__":
    main()
