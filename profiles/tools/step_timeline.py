"""Per-context step anatomy from a rocprofv3 kernel trace of bench.py (our kernels only): for every step of every context's stream,
the stream time of each kernel family and the idle gaps between them; plus chip-level coverage (any kernel / LK / wide kernels).
usage: step_timeline.py TRACE.csv [--steps]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"], r["st"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Stream_Id"]
    r["k"] = r["Kernel_Name"].split("(")[0].replace("void ", "")
FAM = {"lk_track_kernel": "lk", "pyr3_kernel": "pyr", "ransac_kernel<PnPModel, 1>": "pnp_ransac", "pnp_refine_kernel<64>": "pnp_refine",
       "ransac_kernel<HModel, 4>": "H", "ransac_kernel<FModel, 1>": "F", "lk_filter_compact_kernel": "filter"}
ORB = ("resize_exact_kernel", "fast_nms_kernel", "blur7_kernel", "orb_", "scan_", "nms_rows", "ic_angle", "brief", "harris", "hamming", "ratio_compact")


def fam(k):
    if k in FAM:
        return FAM[k]
    if k.startswith(ORB) or any(k.startswith(p) for p in ORB):
        return "orb+match"
    return "small"


lk = [r for r in rows if r["k"] == "lk_track_kernel"]
streams = sorted(set(r["st"] for r in lk))
t0 = lk[len(lk) // 3]["s"]
t1 = lk[-1]["e"]
span = (t1 - t0) / 1e6
per_stream = {}
tot = collections.Counter()
nsteps = 0
for st in streams:
    S = sorted((r for r in rows if r["st"] == st), key=lambda r: r["s"])
    idx = [i for i, r in enumerate(S) if r["k"] == "pyr3_kernel" and r["s"] >= t0]
    for a, b in zip(idx[:-1], idx[1:]):
        step = S[a:b]
        acc = collections.Counter()
        prev_e = step[0]["s"]
        for r in step:
            acc[fam(r["k"])] += (r["e"] - r["s"]) / 1e6
            acc["gap"] += max(0, r["s"] - prev_e) / 1e6
            prev_e = max(prev_e, r["e"])
        acc["gap"] += max(0, S[b]["s"] - prev_e) / 1e6
        acc["period"] = (S[b]["s"] - step[0]["s"]) / 1e6
        nsteps += 1
        tot.update(acc)
        if "--steps" in sys.argv:
            print(st, " ".join(f"{k}={v:.2f}" for k, v in sorted(acc.items())))
print(f"window {span:.1f} ms, {nsteps} context-steps over {len(streams)} streams; mean per context-step (ms of stream time):")
for k, v in sorted(tot.items(), key=lambda x: -x[1]):
    print(f"  {k:12s} {v / nsteps:7.3f}")


def union(iv):
    iv = sorted(iv)
    if not iv:
        return 0.0
    total, (cs, ce) = 0, iv[0]
    for s, e in iv[1:]:
        if s > ce:
            total += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    return (total + ce - cs) / 1e6


sel = [r for r in rows if r["s"] >= t0 and r["e"] <= t1]
print(f"chip: any kernel {union([(r['s'], r['e']) for r in sel]) / span:.3f}, LK {union([(r['s'], r['e']) for r in sel if r['k'] == 'lk_track_kernel']) / span:.3f}")
for name in ("ransac_kernel<HModel, 4>", "ransac_kernel<FModel, 1>", "ransac_kernel<PnPModel, 1>", "pnp_refine_kernel<64>", "lk_track_kernel", "pyr3_kernel"):
    d = sorted((r["e"] - r["s"]) / 1e6 for r in sel if r["k"] == name)
    if d:
        print(f"  {name:30s} n {len(d):3d}  min {d[0]:.3f}  median {d[len(d) // 2]:.3f}  p90 {d[int(len(d) * 0.9)]:.3f}  max {d[-1]:.3f}")
