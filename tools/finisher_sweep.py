"""Kernel time of the four-features-per-wave variant (PAGK_KERNEL=7: its one-level-per-wave form) against (hand-over budget, live finisher workgroups), every setting
measured PAGK_REPS times in alternation (one process each): python tools/finisher_sweep.py [cfg:n ...]
PAGK_SETTINGS="budget:wgs,..."  (budget 0 = no hand-over; wgs 0 = sweep only)."""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
child = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np
from pixel_aware_gyro_aided_klt_feature_tracker_amd import capi, synth
ctx = capi.Context(0)
out = []
for c in sys.argv[1:]:
    cfg, n = (int(v) for v in c.split(":"))
    w = synth.config(cfg, n=n)
    ctx.set_kernel(int(os.environ.get("PAGK_KERNEL", "5")))
    p = capi.make_params(half_patch=10, iterations=30, pyramids=w.pyramids, has_gyro=w.has_gyro, camera=w.camera)
    ts = []
    for _ in range(12):
        ctx.track(p, w.img_ref, w.img_cur, w.pt_ref, w.pt_init, w.affine, w.status_in)
        ts.append(ctx.last_kernel_ms()[0])
    out.append("%%.1f" %% (np.median(ts[3:]) * 1e3))
print(" ".join(out))
''' % ROOT
cases = sys.argv[1:] or ["3:20000", "1:8000", "4:12000", "3:30000"]
settings = os.environ.get("PAGK_SETTINGS", "0:0,16:8,16:16,16:32,20:16,12:16").split(",")
reps = int(os.environ.get("PAGK_REPS", "3"))
res = {s: [] for s in settings}
for rep in range(reps):
    for s in settings:
        b, g = s.split(":")
        env = dict(os.environ, PAGK_QUAD_BUDGET=b, PAGK_FINISHER_WGS=g)
        r = subprocess.run([sys.executable, "-c", child] + cases, capture_output=True, text=True, env=env)
        if r.stdout.strip():
            res[s].append([float(v) for v in r.stdout.strip().splitlines()[-1].split()])
        else:
            print(r.stderr[-300:])
print("budget:wgs   " + "   ".join("%10s" % c for c in cases) + "   (us, median of %d runs)" % reps)
for s in settings:
    if res[s]:
        print("%10s   " % s + "   ".join("%10.1f" % v for v in np.median(np.array(res[s]), axis=0)), flush=True)
