"""Where a kernel's spill stores / reloads sit relative to its barriers, gathers and MFMAs (static, from hipcc -S).
   python tools/isa_spills.py <kernel-name-substring> [extra hipcc flags]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
key = sys.argv[1]
flags = [f for f in g.HIPCC_FLAGS if f not in ("-shared", "-fPIC")]
subprocess.run(["/opt/rocm/bin/hipcc", *flags, *sys.argv[2:], "-S", "--cuda-device-only", "-o", "/tmp/k.s",
                os.path.join(g.CSRC, "pagk_hip.hip")], check=True, capture_output=True)
lines = open("/tmp/k.s").read().splitlines()
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and re.match(r"^_Z\w+:", l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
seg = lines[start:end]
open("/tmp/kb.s", "w").write("\n".join(seg))
pat = re.compile(r"s_barrier|v_mfma_|scratch_store|scratch_load|buffer_load_format|s_setprio|v_fmac_f64_dpp|v_add_f32_dpp|v_sqrt_f64|v_rsq_f64|v_div_fmas_f64")
marks = [(i, l.strip().split()[0]) for i, l in enumerate(seg) if pat.search(l)]
out, prev, cnt, first = [], None, 0, 0
for i, m in marks:
    if m == prev:
        cnt += 1
    else:
        if prev:
            out.append((first, prev, cnt))
        prev, cnt, first = m, 1, i
out.append((first, prev, cnt))
print(len(seg), "lines; full listing in /tmp/kb.s")
for o in out:
    print("%5d  %-28s x%d" % o)
