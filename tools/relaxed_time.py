"""Kernel time of the relaxed-order experiment (kernel 4) beside the exact kernels.  Usage: python tools/relaxed_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pixel_aware_gyro_aided_klt_feature_tracker_amd import capi, synth
from oracle import pagk_oracle as orc
ctx = capi.Context(0)
for cfg, n in ((1, 1000), (1, 4000), (3, 20000)):
    w = synth.config(cfg, n=n)
    p = capi.make_params(half_patch=w.half_patch, iterations=w.iterations, pyramids=w.pyramids, has_gyro=w.has_gyro, camera=w.camera)
    ref = orc.track(p, w.img_ref, w.img_cur, w.pt_ref, w.pt_init, w.affine, w.status_in, nthreads=16)
    line = f"cfg{cfg} n={n}:"
    for kern in (0, 4):
        ctx.set_kernel(kern)
        ts = []
        for _ in range(12):
            out = ctx.track(p, w.img_ref, w.img_cur, w.pt_ref, w.pt_init, w.affine, w.status_in)
            ts.append(ctx.last_kernel_ms()[0])
        t = float(np.median(ts[2:]))
        line += f"  kernel {kern}: {t*1e3:7.1f} us = {w.n_active/t/1e3:6.2f} Mfeat/s"
        if kern == 4:
            m = w.n
            both = (out["status"][:m] == 1) & (ref["status"][:m] == 1)
            d = np.abs(out["pt_un"][:m].astype(np.float64) - ref["pt_un"][:m].astype(np.float64)).max(axis=1)
            line += (f"   [vs oracle: status flips {int((out['status'][:m] != ref['status'][:m]).sum())}, "
                     f"{100*float((d[both] > 1e-3).mean()):.2f}% > 1e-3 px, max {d[both].max():.3g} px, mean iters "
                     f"{out['iters'][:m].mean():.2f} vs {ref['iters'][:m].mean():.2f}]")
    ctx.set_kernel(0)
    print(line, flush=True)
