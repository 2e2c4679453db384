"""Times the literal drop-in call pagk_track() on host buffers (PCIe-inclusive).  Usage: python tools/host_path_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pixel_aware_gyro_aided_klt_feature_tracker_amd import capi, synth
for cfg, n in ((1, 1000), (3, 20000)):
    w = synth.config(cfg, n=n)
    p = capi.make_params(half_patch=w.half_patch, iterations=w.iterations, pyramids=w.pyramids, has_gyro=w.has_gyro, camera=w.camera)
    ctx = capi.Context(0)
    out = capi.alloc_outputs(w.n)
    for _ in range(5):
        ctx.track(p, w.img_ref, w.img_cur, w.pt_ref, w.pt_init, w.affine, w.status_in, out)
    t0 = time.perf_counter()
    K = 100 if n <= 1000 else 30
    for _ in range(K):
        ctx.track(p, w.img_ref, w.img_cur, w.pt_ref, w.pt_init, w.affine, w.status_in, out)
    dt = (time.perf_counter() - t0) / K
    trk, pyr = ctx.last_kernel_ms()
    print(f"cfg{cfg} n={n}: pagk_track {dt*1e6:8.1f} us/call = {w.n_active/dt/1e6:6.2f} Mfeat/s   (tracking kernel {trk*1e3:.1f} us, last pyramid {pyr*1e3:.1f} us)", flush=True)
    ctx.close()
