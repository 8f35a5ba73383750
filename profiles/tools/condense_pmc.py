"""Condense a rocprofv3 counter-collection CSV to per-kernel averages (JSON on stdout).  usage: condense_pmc.py FILE [FILE...]"""
import collections, csv, json, sys
out = {}
for path in sys.argv[1:]:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        for c, v in cs.items():
            out.setdefault(k, {})[c] = {"dispatches": len(v), "avg": sum(v) / len(v), "sum": sum(v)}
print(json.dumps(out, indent=1))
