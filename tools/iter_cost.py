"""Cost of ONE Gauss-Newton iteration of the 4-wave kernel in an UNSTAMPED build: kernel time against the iteration
limit (1..4 iterations per level: no feature converges that early, so every feature runs exactly `iterations` x levels
iterations) -- the slope is three iterations.  python tools/iter_cost.py [libB.so ...]   (PAGK_N=8,250,1000)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
child = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np
from pixel_aware_gyro_aided_klt_feature_tracker_amd import capi, synth
if sys.argv[1] != "-":
    capi.LIB_PATH = sys.argv[1]
ctx = capi.Context(0)
out = []
for n in [int(v) for v in os.environ.get("PAGK_N", "8,250,1000").split(",")]:
    w = synth.config(1, n=n)
    ms, mxs = [], []
    for I in (1, 2, 3, 4):
        p = capi.make_params(half_patch=10, iterations=I, pyramids=3, has_gyro=w.has_gyro, camera=w.camera)
        ts = []
        for _ in range(24):
            o = ctx.track(p, w.img_ref, w.img_cur, w.pt_ref, w.pt_init, w.affine, w.status_in)
            ts.append(ctx.last_kernel_ms()[0])
        ms.append(np.median(ts[6:]) * 1e3)
        mxs.append(int(o["iters"][w.status_in != 0].max()))   # the launch lasts as long as its slowest feature
    slope = (ms[2] - ms[0]) / max(1, mxs[2] - mxs[0])
    out.append("n=%%d: %%s us (max iterations %%s); per iteration %%.3f us = %%.0f cycles @2.4GHz" %% (n, " ".join("%%.1f" %% v for v in ms), mxs, slope, slope * 2400))
print(" | ".join(out))
''' % ROOT
libs = ["-"] + sys.argv[1:]
for rep in range(int(os.environ.get("PAGK_AB_REPS", "2"))):
    for lib in libs:
        r = subprocess.run([sys.executable, "-c", child, lib], capture_output=True, text=True)
        tag = "A (product) " if lib == "-" else "B (%s) " % os.path.basename(lib)
        print(tag + (r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:]), flush=True)
