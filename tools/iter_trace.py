"""Per-level iteration counts of a workload's features, for tools/prio_fluid_model.py: a TEMPORARY copy of oracle/pagk_oracle.c gets one
fprintf behind one_pixel()'s loop (feature, level, iterations), is built into a scratch directory and run once; the repository's oracle is
not touched.  Usage: python tools/iter_trace.py <config> <n> <out.npy>     (e.g. 1 1000 tools/data/iters_cfg1_1000.npy)"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

cfg, n, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
top = tempfile.mkdtemp(prefix="pagk_iter_trace_")
tmp = os.path.join(top, "oracle")          # (the oracle includes ../include/pagk.h)
os.makedirs(tmp)
src = open(os.path.join(ROOT, "oracle", "pagk_oracle.c")).read()
hook = "    if (st->iters) st->iters[i] += iters_done;"
assert src.count(hook) == 1
src = src.replace(hook, hook + '\n    fprintf(stderr, "TR %d %d %d\\n", i, level, iters_done);')
if "#include <stdio.h>" not in src:
    src = "#include <stdio.h>\n" + src
open(os.path.join(tmp, "pagk_oracle.c"), "w").write(src)
for f in ("pagk_oracle.h",):
    open(os.path.join(tmp, f), "w").write(open(os.path.join(ROOT, "oracle", f)).read())
os.makedirs(os.path.join(top, "include"))
open(os.path.join(top, "include", "pagk.h"), "w").write(open(os.path.join(ROOT, "include", "pagk.h")).read())
lib = os.path.join(tmp, "libtrace.so")
subprocess.run(["gcc", "-O2", "-std=c11", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fexcess-precision=standard", "-shared",
                "-o", lib, os.path.join(tmp, "pagk_oracle.c"), "-lm", "-lpthread"], check=True, capture_output=True)
child = f'''
import sys
sys.path.insert(0, {ROOT!r})
from oracle import pagk_oracle as orc
orc.LIB_PATH = {lib!r}
from pixel_aware_gyro_aided_klt_feature_tracker_amd import capi, synth
w = synth.config({cfg}, n={n})
p = capi.make_params(half_patch=w.half_patch, iterations=w.iterations, pyramids=w.pyramids, has_gyro=w.has_gyro, camera=w.camera)
orc.track(p, w.img_ref, w.img_cur, w.pt_ref, w.pt_init, w.affine, w.status_in, nthreads=1)
print(w.pyramids)
'''
r = subprocess.run([sys.executable, "-c", child], capture_output=True, text=True, check=True)
L = int(r.stdout.strip().splitlines()[-1])
it = np.zeros((n, L), dtype=np.uint8)
for line in r.stderr.splitlines():
    if line.startswith("TR "):
        _, i, l, k = line.split()
        it[int(i), int(l)] = int(k)
np.save(out, it)
import shutil
shutil.rmtree(top, ignore_errors=True)
print(f"{out}: {n} features x {L} levels, mean {it.sum(1).mean():.2f} iterations per feature, max {it.sum(1).max()}")
