"""k_track_rows against the size of its resident grid (PAGK_ROWS_WAVES; 0 = occupancy x CUs): kernel time per case.
python tools/rows_sweep.py [cfg:n ...]"""
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
    ctx.set_kernel(6)
    p = capi.make_params(half_patch=10, iterations=30, pyramids=3, has_gyro=w.has_gyro, camera=w.camera)
    ts = []
    for _ in range(14):
        ctx.track(p, w.img_ref, w.img_cur, w.pt_ref, w.pt_init, w.affine, w.status_in)
        ts.append(ctx.last_kernel_ms()[0])
    out.append("%%d:%%d %%.1f us" %% (cfg, n, np.median(ts[4:]) * 1e3))
print("   ".join(out))
''' % ROOT
cases = sys.argv[1:] or ["3:20000", "3:8000", "4:12000", "1:8000", "1:4000"]
for waves in os.environ.get("PAGK_WAVES_LIST", "0,3584,3072,2560,2048,1536,1024").split(","):
    env = dict(os.environ, PAGK_ROWS_WAVES=waves)
    r = subprocess.run([sys.executable, "-c", child] + cases, capture_output=True, text=True, env=env)
    print("waves %5s  " % waves + (r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:]), flush=True)
