"""Kernel time of the four-features-per-wave variant against the hand-over budget (PAGK_QUAD_BUDGET; 0 = no hand-over):
python tools/budget_sweep.py [cfg:n ...]   (default 3:20000 4:4000x? see CASES).  One child process per budget."""
import os, subprocess, sys
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
    ctx.set_kernel(5)
    p = capi.make_params(half_patch=10, iterations=30, pyramids=3, has_gyro=w.has_gyro, camera=w.camera)
    ts = []
    for _ in range(14):
        ctx.track(p, w.img_ref, w.img_cur, w.pt_ref, w.pt_init, w.affine, w.status_in)
        ts.append(ctx.last_kernel_ms()[0])
    out.append("%%d:%%d %%.1f us" %% (cfg, n, np.median(ts[4:]) * 1e3))
print("   ".join(out))
''' % ROOT
cases = sys.argv[1:] or ["3:20000", "3:8000", "4:12000", "1:8000"]
for budget in os.environ.get("PAGK_BUDGETS", "0,8,12,16,20,24,30").split(","):
    env = dict(os.environ, PAGK_QUAD_BUDGET=budget)
    r = subprocess.run([sys.executable, "-c", child] + cases, capture_output=True, text=True, env=env)
    print("budget %3s  " % budget + (r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:]), flush=True)
