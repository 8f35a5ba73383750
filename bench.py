#!/usr/bin/env python3
"""bench.py — tracker-step frames/sec of the MI355X-native VO front-end (BASELINE.json metric).

One "step" = the reference's steady-state Tracker::update (src/tracker.cpp:274-333) in its worst case
(key-frame branch every frame) for a batch of B independent 1280x720 mono8 camera streams resident in
HBM:  pyrDown pyramid + LK (2000-feature tracks) + status/err filter + solvePnPRansac +
findHomography/findFundamentalMat RANSAC + ORB(2000) detect/describe + knn2/ratio match + triangulation.
Stages that are not built yet are listed in config["stages_missing"] and make the line a partial one.

Frames are synthetic (ros2_mono_vo_amd.synth) and pre-loaded into the device frame ring
before the timed region.  value = frames processed by all ranks / max-over-ranks wall time.

Multi-GPU (--gpus N under torch.distributed.run): independent streams are sharded across ranks; the only
collective is one RCCL broadcast of the intrinsics {K, d} from rank 0; scaling "weak".
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

ALGO_BYTES_PER_LK_POINT = 4261  # 4 levels x (24^2 + 22^2) window bytes + 21 B of point I/O
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def make_streams(w, h, n_frames, n_distinct, seed0):
    from ros2_mono_vo_amd import synth
    return [synth.gen_stream(w, h, seed0 + 1 + i, n_frames) for i in range(n_distinct)]


def planar_landmarks(K, z=10.0):
    def f(xy):
        zz = np.full(len(xy), z, np.float32)
        return np.stack([(xy[:, 0] - K[0, 2]) / K[0, 0] * zz, (xy[:, 1] - K[1, 2]) / K[1, 1] * zz, zz], 1)
    return f


def cpu_baseline(seed, w, h, K, nfeatures, n_frames):
    """The oracle (CPU restatement, 1 thread) on a bounded sample of the same workload: the first rank's stream 0
    continued for `n_frames` consecutive full steps (tests/pipeline_ref.py: the same stage list and data flow as
    mvo_batch_step).  Test infrastructure used as the baseline only — never on the product path."""
    from pipeline_ref import StreamRef
    from ros2_mono_vo_amd import synth
    fr = synth.gen_stream(w, h, seed, n_frames + 1)
    ref = StreamRef(K, nfeatures)
    ref.seed(fr[0], planar_landmarks(K))
    t0 = time.perf_counter()
    done = 0
    for k in range(1, min(n_frames + 1, len(fr))):
        ref.step(fr[k])
        done += 1
    dt = time.perf_counter() - t0
    return done / dt, done


STAGE_TIMERS = ("frame_fanout", "lk_pyramid", "lk_track", "lk_filter", "orb_detect", "orb_select", "orb_blur", "orb_describe", "match",
                "pnp", "ransac_h", "ransac_f", "triangulate")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=512, help="independent camera streams per GPU (256: 25.3k fps, 512: 26.8k, 1024: 27.5k)")
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--nfeatures", type=int, default=2000)
    ap.add_argument("--distinct", type=int, default=4, help="distinct synthetic streams generated per rank")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=50, help="consecutive oracle steps timed for cpu_baseline (~0.2 s each)")
    ap.add_argument("--per-step", action="store_true", help="diagnostic: add the stage timers of every step (step_stage_ms)")
    ap.add_argument("--stages", type=lambda v: int(v, 0), default=None,
                    help="diagnostic: MVO_STAGE_* mask to run instead of the full step (the line then lists stages_missing)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    import torch
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from ros2_mono_vo_amd import Context, _lib, synth
    from ros2_mono_vo_amd import parallel

    W, H, B, K, Wm = args.width, args.height, args.batch, args.steps, args.warmup
    n_frames = K + Wm + 1
    # intrinsics: rank 0 owns them, one RCCL broadcast over xGMI (the path's only collective)
    Kmat, dcoef = parallel.broadcast_intrinsics(synth.default_K(W, H) if rank == 0 else np.zeros((3, 3)), np.zeros(5), dist,
                                                device="cuda")

    streams = make_streams(W, H, n_frames, min(args.distinct, B), 0x5EED0003 + 64 * rank)
    ctx = Context(max_width=W, max_height=H, batch=B, nfeatures=args.nfeatures, max_points=4096,
                  ring_frames=n_frames, device=local_rank)
    ctx.batch_set_intrinsics(Kmat, dcoef)
    for s in range(B):
        fr = streams[s % len(streams)]
        for f in range(n_frames):
            ctx.batch_preload_frame(s, f, fr[f])
    ctx.sync()
    nk = ctx.batch_seed(0)
    # planar landmarks (Z = 10 m) for the seeded tracks: the similarity-warp stream is a fronto-parallel plane
    lmf = planar_landmarks(Kmat)
    for s in range(B):
        ctx.batch_set_landmarks(s, lmf(ctx.batch_get_tracks(s)))

    stages = _lib.STAGE_ALL if args.stages is None else args.stages
    names = {_lib.STAGE_LK: "lk", _lib.STAGE_PNP: "pnp", _lib.STAGE_HF: "ransac_hf", _lib.STAGE_ORB: "orb",
             _lib.STAGE_MATCH: "match", _lib.STAGE_TRIANG: "triangulate"}
    stages_missing = [n for b, n in names.items() if not stages & b]

    for k in range(Wm):
        ctx.batch_step(1 + k, stages)
    ctx.profile_reset()
    ctx.profile_enable(True)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    lk_points = 0
    last = None
    step_ms, step_stage, prev_cum, results = [], [], {}, []
    for k in range(K):
        ts = time.perf_counter()
        last = ctx.batch_step(1 + Wm + k, stages)   # synchronous: returns when the step's results are on the host
        step_ms.append(round((time.perf_counter() - ts) * 1e3, 3))
        results.append(last)
        if args.per_step:
            cum = {n: ctx.profile_read(n)[0] for n in STAGE_TIMERS}
            step_stage.append({n: round(cum[n] - prev_cum.get(n, 0.0), 3) for n in STAGE_TIMERS})
            prev_cum = cum
    ctx.sync()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    ctx.profile_enable(False)
    lk_points = sum(r.n_prev for res in results for r in res)   # bookkeeping for the roofline, outside the timed region
    dt = parallel.max_over_ranks(dt, dist, device="cuda")

    prof = {}
    for name in STAGE_TIMERS:
        ms, n = ctx.profile_read(name)
        if n:
            prof[name] = {"ms_total": round(ms, 4), "launches": n, "ms_avg": round(ms / n, 5)}

    if rank == 0:
        frames = B * K * world
        value = frames / dt
        lk = prof.get("lk_track", {"ms_avg": 0, "launches": 0})
        algo_bytes = ALGO_BYTES_PER_LK_POINT * (lk_points / max(K, 1))
        achieved = algo_bytes / (lk["ms_avg"] * 1e-3) / 1e9 if lk["ms_avg"] else 0.0
        # HBM traffic of the kernel from the PMC counters (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes;
        # profiles/r01_lk_pmc.json holds the measured bytes per tracked point), scaled to this launch's point count.
        traffic = None
        valu = {}
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_lk_pmc.json")))
            traffic = int((pmc["fetch_bytes_per_point"] + pmc["write_bytes_per_point"]) * (lk_points / max(K, 1)))
            valu = pmc.get("sq_pass", {})
        except Exception:
            pass
        line = {
            "metric": "tracker-step frames/sec @1280x720, 2000 ORB feats",
            "value": round(value, 2), "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": Wm,
            "ms_per_step": round(dt / K * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8/i32 fixed-point + f32/f64", "data": "synthetic",
            "config": {"workload": f"C3: {W}x{H} mono8, {args.nfeatures} ORB + LK + PnP-RANSAC + H/F-RANSAC + match + "
                                   f"triangulate; batch of {B} independent streams per GPU, key-frame branch every frame",
                       "batch_per_gpu": B, "width": W, "height": H, "nfeatures": args.nfeatures,
                       "stages": ["lk_pyramid", "lk_track", "lk_filter", "pnp_ransac+refine", "ransac_h", "ransac_f",
                                  "orb_detect", "orb_describe", "match", "triangulate+landmark_handover"],
                       "stages_missing": stages_missing, "parallelism": f"streams x{world}",
                       "mean_tracks_per_frame": round(lk_points / max(B * K, 1), 1),
                       "mean_keypoints": float(np.mean([r.n_keypoints for r in last])),
                       "mean_matches": float(np.mean([r.n_matches for r in last])),
                       "mean_pnp_inliers": float(np.mean([r.n_pnp_inliers for r in last])),
                       "mean_score_h": float(np.mean([r.score_h for r in last])),
                       "mean_score_f": float(np.mean([r.score_f for r in last])),
                       "mean_triangulated": float(np.mean([r.n_triangulated for r in last]))},
            "roofline": {"bound": "hbm", "kernel": "lk_track_kernel", "achieved": round(achieved, 2),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                         "traffic": traffic,
                         "algorithmic_bytes_per_launch": int(algo_bytes), "avg_launch_ms": lk["ms_avg"],
                         # the kernel is VALU bound, not memory bound (PMC SQ pass, profiles/r01_lk_pmc.json)
                         "valu_instructions_per_point": valu.get("valu_instructions_per_point"),
                         "valu_busy_frac_of_simd_time": valu.get("valu_busy_fraction_of_simd_time_at_2.4GHz")},
            "stage_ms": prof,
            "step_ms": step_ms,
        }
        if step_stage:
            line["step_stage_ms"] = step_stage
        if not args.no_cpu_baseline:
            fps, nfr = cpu_baseline(0x5EED0003 + 1, W, H, Kmat, args.nfeatures, args.cpu_frames)
            line["cpu_baseline"] = {"value": round(fps, 3), "unit": "frames/s", "cores": 1, "kind": "port",
                                    "sample": f"oracle (CPU restatement of OpenCV-4.6 semantics, not OpenCV), stream 0, "
                                              f"{nfr} consecutive full steps (same stage list and data flow), 1 thread"}
        print(json.dumps(line), flush=True)
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
