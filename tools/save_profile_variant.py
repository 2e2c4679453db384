"""Copy the summaries of gpurun_out/prof_<tag> (tools/profile_variant.sh) into profiles/<tag>/:
kernel_stats.csv (rocprofv3 --kernel-trace --stats) and pmc_summary.json (per-kernel means of every counter pass)."""
import collections, csv, glob, json, os, shutil, sys
tag = sys.argv[1]
src, dst = f"gpurun_out/prof_{tag}", f"profiles/{tag}"
os.makedirs(dst, exist_ok=True)
# gpurun merges a new run's files NEXT TO an older run's: always the newest file of each kind
ks = sorted(glob.glob(f"{src}/trace/*/*_kernel_stats.csv"), key=os.path.getmtime)
if ks:
    shutil.copy(ks[-1], f"{dst}/kernel_stats.csv")
if os.path.exists(f"{src}/trace.log"):
    shutil.copy(f"{src}/trace.log", f"{dst}/run.log")
out = {}
passes = {}
for f in sorted(glob.glob(f"{src}/pmc_*/*/*_counter_collection.csv"), key=os.path.getmtime):
    passes[f.split("/")[-3]] = f   # newest file per pass directory
for f in passes.values():
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        if "pagk" in k:
            for c, v in d.items():
                out.setdefault(k, {})[c] = {"dispatches": len(v), "mean": sum(v) / len(v)}
json.dump(out, open(f"{dst}/pmc_summary.json", "w"), indent=1, sort_keys=True)
for k, d in out.items():
    print(k, {c: round(v["mean"]) for c, v in sorted(d.items())})
