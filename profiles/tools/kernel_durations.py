"""Per-dispatch durations of the kernels matching a regex from a rocprofv3 --kernel-trace CSV.  usage: kernel_durations.py CSV REGEX"""
import csv, re, sys
rx = re.compile(sys.argv[2])
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(sys.argv[1])) if rx.search(r["Kernel_Name"])]
d_sorted = sorted(d)
print(f"{sys.argv[2]}: launches {len(d)} total {sum(d):.2f} ms  mean {sum(d)/max(len(d),1):.3f}  median {d_sorted[len(d)//2]:.3f}  max {d_sorted[-1]:.3f}")
print("  in launch order:", " ".join(f"{x:.2f}" for x in d))
