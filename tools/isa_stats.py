"""Static instruction mix of one kernel in a hipcc --save-temps .s file, split at its s_barrier instructions.
   python tools/isa_stats.py <file.s> <kernel-name-substring>"""
import re, sys, collections
path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().splitlines()
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l.split(":")[0] and ":" in l)
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
seg, segs = collections.Counter(), []
def cls(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_") and ("f64" in op): return "valu64"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_"): return "salu"
    return "other"
for l in lines[start + 1:end + 1]:
    m = re.match(r"\s+([a-z_0-9]+)", l)
    if not m: continue
    op = m.group(1)
    seg[cls(op)] += 1
    if op == "s_barrier":
        segs.append(seg); seg = collections.Counter()
segs.append(seg)
tot = collections.Counter()
for k, s in enumerate(segs):
    tot.update(s)
    print(f"segment {k}: " + "  ".join(f"{c}={n}" for c, n in sorted(s.items())))
print("total: " + "  ".join(f"{c}={n}" for c, n in sorted(tot.items())))
