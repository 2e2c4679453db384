"""Kernel time vs number of features for the DPP (0) and MFMA (2) variants.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pixel_aware_gyro_aided_klt_feature_tracker_amd import capi, synth
ctx = capi.Context(0)
for cfg, ns in ((1, (250, 500, 1000, 1500, 2000, 3000, 4000, 8000)), (3, (20000,))):
    for n in ns:
        w = synth.config(cfg, n=n)
        p = capi.make_params(half_patch=10, iterations=30, pyramids=3, has_gyro=w.has_gyro, camera=w.camera)
        row = [f"cfg{cfg} n={n:6d} active={w.n_active:6d}"]
        for k in (0, 2, 3):
            ctx.set_kernel(k)
            ts = []
            for _ in range(4):
                ctx.track(p, w.img_ref, w.img_cur, w.pt_ref, w.pt_init, w.affine, w.status_in)
                ts.append(ctx.last_kernel_ms()[0])
            t = min(ts[1:])
            row.append(f"kernel {k}: {t*1e3:8.1f} us = {w.n_active/t/1e3:6.2f} Mfeat/s")
        print("  ".join(row), flush=True)
ctx.close()
