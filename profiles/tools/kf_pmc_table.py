"""Condense the four PMC passes of tools/kf_pmc.sh (key-frame step of one context, policy 1) into one table per kernel:
VALU / SALU / LDS wave-instructions, issue rate against the chip's 1024 SIMDs x 1 instruction per 4 cycles at 2.4 GHz, stall
shares, HBM bytes.  usage: kf_pmc_table.py DIR_WITH_CSVS DURATIONS.txt"""
import collections, csv, glob, json, os, sys

d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(d, "q_*", "**", "*counter_collection.csv"), recursive=True):
    first = "SQ_INSTS_VALU" in f
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if first and r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
out = {}
for k, c in acc.items():
    t_us = sum(dur[k])
    g = lambda n: c.get(n, float("nan"))
    out[k] = {
        "dispatches": len(dur[k]), "total_us_under_pmc": round(t_us, 1),
        "valu_wave_instr": g("SQ_INSTS_VALU"), "salu": g("SQ_INSTS_SALU"), "lds": g("SQ_INSTS_LDS"), "waves": g("SQ_WAVES"),
        "valu_issue_frac_of_chip": round(g("SQ_INSTS_VALU") / (t_us * 1e-6) / (1024 * 0.6e9), 3) if t_us else None,
        "wait_any_frac_of_wave_cycles": round(g("SQ_WAIT_INST_ANY") / max(g("SQ_WAVE_CYCLES"), 1), 3),
        "lds_bank_conflict_cycles": g("SQ_LDS_BANK_CONFLICT"), "lds_active_cycles": g("SQ_ACTIVE_INST_LDS"),
        "hbm_read_MB": round(2 * g("FETCH_SIZE") / 1024, 1), "hbm_write_MB": round(g("WRITE_SIZE") / 1024, 1),
        "hbm_TBps": round((2 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024 / (t_us * 1e-6) / 1e12, 2) if t_us else None,
    }
print(json.dumps(out, indent=1))
