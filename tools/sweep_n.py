"""Kernel time vs number of features for the exact tracking variants (auto-selection thresholds): 4-wave DPP rows,
2-wave MFMA, one wave per feature, four features per wave, four independent rows per wave + queue, one level per wave (7).  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pixel_aware_gyro_aided_klt_feature_tracker_amd import capi, synth
# kernel 0 = "automatic": with the thresholds out of reach it is the 4-wave DPP kernel at every size
os.environ["PAGK_MFMA_MIN"] = os.environ["PAGK_QUAD_MIN"] = os.environ["PAGK_WAVE_MIN"] = os.environ["PAGK_LEVELS_MIN"] = "1000000000"
ctx = capi.Context(0)
for cfg, ns in ((1, (250, 500, 1000, 1500, 2000, 2500, 3000, 4000, 6000, 8000, 12000)), (3, (20000, 30000)), (4, (4000,))):
    for n in ns:
        w = synth.config(cfg, n=n)
        p = capi.make_params(half_patch=10, iterations=30, pyramids=3, has_gyro=w.has_gyro, camera=w.camera)
        row = [f"cfg{cfg} n={n:6d} active={w.n_active:6d}"]
        for k in (k for k in (0, 2, 3, 5, 6, 7) if capi.has_variant(k)):   # (2 and 6 only in a -DPAGK_ALL_VARIANTS build)
            ctx.set_kernel(k)
            ts = []
            for _ in range(4):
                ctx.track(p, w.img_ref, w.img_cur, w.pt_ref, w.pt_init, w.affine, w.status_in)
                ts.append(ctx.last_kernel_ms()[0])
            t = min(ts[1:])
            row.append(f"kernel {k}: {t*1e3:8.1f} us = {w.n_active/t/1e3:6.2f} Mfeat/s")
        print("  ".join(row), flush=True)
ctx.close()
