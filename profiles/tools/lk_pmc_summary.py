"""Merge the four PMC passes of tools/profile_round.sh over the one-context LK run into profiles/<TAG>_lk_pmc_condensed.json
and derive the per-point summary profiles/<TAG>_lk_pmc.json (bench.py's labelled fallback when its own PMC probe fails).
usage: lk_pmc_summary.py GPURUN_TAG OUT_TAG   e.g.  lk_pmc_summary.py r03k r03_k"""
import csv, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src, dst = sys.argv[1], sys.argv[2]
G = os.path.join(ROOT, "gpurun_out")
cond = {}
for n in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "SQ_LDS_BANK_CONFLICT"):
    for k, cs in json.load(open(os.path.join(G, f"{src}_pmc_{n}.json"))).items():
        cond.setdefault(k, {}).update(cs)
json.dump(cond, open(os.path.join(ROOT, "profiles", f"{dst}_lk_pmc_condensed.json"), "w"), indent=1)
bench = json.loads([l for l in open(os.path.join(G, f"{src}_1x256_under_rocprof.json")) if l.startswith("{")][0])
stats = {r["Name"].split("(")[0].replace("void ", ""): r for r in csv.DictReader(open(os.path.join(G, f"{src}_kernel_stats_1x256.csv")))}
lk = cond["lk_track_kernel"]
nd = lk["SQ_INSTS_VALU"]["dispatches"]
# the run's launches: 1 warm-up + 4 timed steps + the isolated launch at the end; their point counts differ by < 2 %, so the
# per-point figures use the timed region's mean points per launch x dispatches
pts = bench["roofline"]["points_per_launch"] * nd
s = lambda c: lk[c]["sum"]
ms = float(stats["lk_track_kernel"]["AverageNs"]) * 1e-6
MIX = 1.55
out = {
    "kernel": "lk_track_kernel (four points per wavefront, one per DPP row; level 0 read in place from the frame ring; levels 1..3 in "
              "planes with a reflect-101 border; work list sorted by Morton cell)",
    "source": "tools/profile_round.sh on one MI355X = rocprofv3 --kernel-include-regex 'lk_track_kernel|pyr3_kernel|lk_border_kernel' "
              "--kernel-trace --pmc <set> -- python3 bench.py --contexts 1 --batch 256 --steps 4 --warmup 1 --ingest-steps 0 --extra-steps 0 "
              "--single-steps 0 --no-cpu-baseline --no-pmc (four passes: FETCH_SIZE | WRITE_SIZE | SQ_* | SQ_LDS_*), condensed in "
              f"profiles/{dst}_lk_pmc_condensed.json; kernel time from rocprofv3 --stats of the same command "
              f"(profiles/{dst}_kernel_stats_1x256_lkrun.csv)",
    "points_per_launch": bench["roofline"]["points_per_launch"],
    "kernel_ms_rocprof_stats": ms,
    "valu_instructions_per_point": s("SQ_INSTS_VALU") / pts,
    "salu_instructions_per_point": s("SQ_INSTS_SALU") / pts,
    "lds_instructions_per_point": s("SQ_INSTS_LDS") / pts,
    "hbm_read_bytes_per_point": 2 * 1024 * s("FETCH_SIZE") / pts,
    "hbm_write_bytes_per_point": 1024 * s("WRITE_SIZE") / pts,
    "fetch_size_calibration": "x2 (profiles/r02_fetch_calibration.json: FETCH_SIZE counts 128-byte fabric requests at 64 B on gfx950)",
    "valu_mix_ns_per_wave_instruction_per_simd": MIX,
    "valu_issue_frac_isolated": (s("SQ_INSTS_VALU") / nd) * MIX * 1e-9 / 1024 / (ms * 1e-3),
    "sq_wait_any_over_wave_cycles": s("SQ_WAIT_ANY") / s("SQ_WAVE_CYCLES"),
    "sq_wait_inst_any_over_wave_cycles": s("SQ_WAIT_INST_ANY") / s("SQ_WAVE_CYCLES"),
    "sq_active_inst_valu_over_wave_cycles": s("SQ_ACTIVE_INST_VALU") / s("SQ_WAVE_CYCLES"),
    "lds_bank_conflict_over_lds_active": s("SQ_LDS_BANK_CONFLICT") / s("SQ_LDS_IDX_ACTIVE"),
    "valu_lane_utilisation": s("SQ_THREAD_CYCLES_VALU") / (64 * s("SQ_ACTIVE_INST_VALU")),
}
for k in ("pyr3_kernel", "lk_border_kernel"):
    if k in cond and k in stats:
        c = cond[k]
        n = c["FETCH_SIZE"]["dispatches"]
        t = float(stats[k]["AverageNs"]) * 1e-9
        rd, wr = 2 * 1024 * c["FETCH_SIZE"]["sum"] / n, 1024 * c["WRITE_SIZE"]["sum"] / n
        out[k] = {"launch_us": round(t * 1e6, 1), "hbm_read_MB_per_launch": round(rd / 1e6, 1), "hbm_write_MB_per_launch": round(wr / 1e6, 1),
                  "hbm_TBps": round((rd + wr) / t / 1e12, 2), "valu_instructions_per_launch": c["SQ_INSTS_VALU"]["sum"] / n}
json.dump(out, open(os.path.join(ROOT, "profiles", f"{dst}_lk_pmc.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
