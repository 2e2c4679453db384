"""Kernel time of chosen variants on chosen launches, one process: python tools/ab_kernels.py 5,6 3:20000 3:40000 ..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pixel_aware_gyro_aided_klt_feature_tracker_amd import capi, synth
kernels = [int(k) for k in sys.argv[1].split(",")]
ctx = capi.Context(0)
for c in sys.argv[2:]:
    cfg, n = (int(v) for v in c.split(":"))
    w = synth.config(cfg, n=n)
    p = capi.make_params(half_patch=10, iterations=30, pyramids=3, has_gyro=w.has_gyro, camera=w.camera)
    line = []
    for k in kernels:
        ctx.set_kernel(k)
        ts = []
        for _ in range(12):
            ctx.track(p, w.img_ref, w.img_cur, w.pt_ref, w.pt_init, w.affine, w.status_in)
            ts.append(ctx.last_kernel_ms()[0])
        t = np.median(ts[3:]) * 1e3
        line.append("kernel %d (variant %d): %8.1f us  %6.2f Mfeat/s" % (k, ctx.last_variant(), t, w.n_active / t))
    print("cfg%d n=%d   " % (cfg, n) + "   ".join(line), flush=True)
