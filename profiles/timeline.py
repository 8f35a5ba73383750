#!/usr/bin/env python3
"""Print the kernel timeline of the last bench step from a rocprofv3 --kernel-trace CSV (diagnostic helper)."""
import csv, sys
tr = list(csv.DictReader(open(sys.argv[1])))
tr.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(tr) if r['Kernel_Name'].startswith('ring_to_level0')]
i0 = idx[-1]
t0 = int(tr[i0]['Start_Timestamp'])
minus = len(sys.argv) > 2
for r in tr[i0:]:
    s = (int(r['Start_Timestamp']) - t0) / 1e3
    e = (int(r['End_Timestamp']) - t0) / 1e3
    if minus and e - s < 20: continue
    print(f"{s:9.1f} {e:9.1f} {e-s:8.1f} q={r['Queue_Id']} {r['Kernel_Name'][:45]}")
