"""Where a step's time goes BETWEEN kernels: durations and gaps of consecutive dispatches in a rocprofv3 --kernel-trace CSV.
Usage: python tools/trace_gaps.py <dir with *_kernel_trace.csv>   (steady state = the longest run of alternating pyramid / track)"""
import csv, glob, sys, collections
import numpy as np
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("<")[0]) for r in csv.DictReader(open(f))), key=lambda r: r[0])
names = [r[2].replace("void pagk::", "") for r in rows]
gaps = collections.defaultdict(list)
dur = collections.defaultdict(list)
for k in range(1, len(rows)):
    gaps[(names[k - 1], names[k])].append(rows[k][0] - rows[k - 1][1])
    dur[names[k]].append(rows[k][1] - rows[k][0])
print("durations (us): ", {k: (len(v), round(float(np.median(v)) / 1e3, 2)) for k, v in dur.items() if len(v) > 50})
for k, v in sorted(gaps.items(), key=lambda kv: -len(kv[1])):
    if len(v) > 50:
        v = np.array(v) / 1e3
        print(f"gap {k[0]:28s} -> {k[1]:28s}: n {len(v):5d}  median {np.median(v):7.2f} us  p10 {np.percentile(v,10):7.2f}  p90 {np.percentile(v,90):7.2f}")
