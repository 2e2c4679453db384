import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pixel_aware_gyro_aided_klt_feature_tracker_amd import capi, synth
lib = sys.argv[1]
if lib != "default":
    capi.LIB_PATH = os.path.abspath(lib)
ctx = capi.Context(0)
ctx.set_kernel(3)
for cfg, n in ((1, 8000), (3, 20000)):
    w = synth.config(cfg, n=n)
    p = capi.make_params(half_patch=10, iterations=30, pyramids=3, has_gyro=w.has_gyro, camera=w.camera)
    ts = []
    for _ in range(4):
        ctx.track(p, w.img_ref, w.img_cur, w.pt_ref, w.pt_init, w.affine, w.status_in)
        ts.append(ctx.last_kernel_ms()[0])
    t = min(ts[1:])
    print(lib, f"cfg{cfg} n={n}: {t*1e3:.1f} us = {w.n_active/t/1e3:.2f} Mfeat/s", flush=True)
