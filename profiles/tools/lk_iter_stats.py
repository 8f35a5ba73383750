"""Experiment: how much of lk_track_kernel's iteration loop is lost to the four points of a wavefront finishing at different
times.  Needs a variant build: `tools/build_variant.sh iters -DLK_ITER_STATS`, then
`MVO_LIB=build/libmvo_iters.so python3 profiles/tools/lk_iter_stats.py`.  The variant returns, in `err`, own iterations +
100 * loop trips of the point's wavefront + 10000 * search-tile loads (summed over the four levels).  Points go in groups of four in the order given:
ORB order, Morton order (the tracker's work list), and Morton order refined by the own-iteration count of the PREVIOUS frame
pair (what a tracker could carry along)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from ros2_mono_vo_amd import Context, synth, synth_gpu

W, H = 1280, 720
dev = torch.device("cuda", 0)
K = synth.default_K(W, H)
bank = synth_gpu.SceneBank(dev)


def morton(xy, shift=4):
    x = (xy[:, 0].astype(np.int64) >> shift); y = (xy[:, 1].astype(np.int64) >> shift)
    m = np.zeros(len(xy), np.int64)
    for b in range(10):
        m |= ((x >> b) & 1) << (2 * b) | ((y >> b) & 1) << (2 * b + 1)
    return m


def run(ctx, a, b, pts):
    out, st, err = ctx.lk_track(a, b, pts)
    e = err.astype(np.int64)
    own = e % 100; trips = (e // 100) % 100
    run.jloads = getattr(run, "jloads", 0) + int((e // 10000).sum()); run.points = getattr(run, "points", 0) + len(e)
    return out, st, own, trips


res = {}
with Context(max_width=W, max_height=H, max_points=8192) as ctx:
    tot = {k: [0, 0] for k in ("orb", "morton", "morton+prev_iters", "sorted_by_true_iters")}
    for s in range(6):
        fr, _ = synth_gpu.render_stream(bank, bank.stream_params(0x5EED0003 + s), K, W, H, 3)
        fr = fr.cpu().numpy()
        kps, _ = ctx.orb_detect_and_compute(fr[0])
        p0 = np.stack([kps["x"], kps["y"]], 1).astype(np.float32)
        # frame 0 -> 1 in ORB order (gives the previous-frame iteration counts), then frame 1 -> 2 in the orders compared
        p1, st, own01, _ = run(ctx, fr[0], fr[1], p0)
        keep = st > 0
        p1, prev_it = p1[keep], own01[keep]
        orders = {"orb": np.arange(len(p1)), "morton": np.argsort(morton(p1), kind="stable")}
        _, _, own, trips = run(ctx, fr[1], fr[2], p1)
        orders["morton+prev_iters"] = np.lexsort((prev_it, morton(p1, 6)))
        orders["sorted_by_true_iters"] = np.argsort(own, kind="stable")
        for name, o in orders.items():
            _, _, own, trips = run(ctx, fr[1], fr[2], p1[o])
            n4 = len(o) // 4 * 4
            tot[name][0] += int(own[:n4].sum()); tot[name][1] += int(trips[:n4].sum())
    for name, (own, trips) in tot.items():
        res[name] = {"own_iterations": own, "wave_trips_x_rows": trips, "efficiency": round(own / trips, 4)}
res["search_tile_loads_per_point"] = round(run.jloads / run.points, 3)
print(json.dumps(res, indent=1))
