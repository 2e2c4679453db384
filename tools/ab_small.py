"""Kernel time of chosen variants at small launch sizes on configs[1], interleaved: python tools/ab_small.py 0,7 250 500 1000 ..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PAGK_MFMA_MIN"] = os.environ["PAGK_QUAD_MIN"] = os.environ["PAGK_WAVE_MIN"] = "1000000000"
import numpy as np
from pixel_aware_gyro_aided_klt_feature_tracker_amd import capi, synth
kernels = [int(k) for k in sys.argv[1].split(",")]
cfg = int(os.environ.get("PAGK_CFG", "1"))
ctx = capi.Context(0)
for n in (int(v) for v in sys.argv[2:]):
    w = synth.config(cfg, n=n)
    p = capi.make_params(half_patch=10, iterations=30, pyramids=w.pyramids if hasattr(w, "pyramids") else 3, has_gyro=w.has_gyro, camera=w.camera)
    ts = {k: [] for k in kernels}
    for rep in range(24):
        for k in kernels:
            ctx.set_kernel(k)
            ctx.track(p, w.img_ref, w.img_cur, w.pt_ref, w.pt_init, w.affine, w.status_in)
            if rep >= 4:
                ts[k].append(ctx.last_kernel_ms()[0])
    print("cfg%d n=%5d  " % (cfg, n) + "   ".join("kernel %d: %7.1f us (min %7.1f)" % (k, np.median(ts[k]) * 1e3, min(ts[k]) * 1e3) for k in kernels), flush=True)
