"""Run the tracking kernel a few times on one workload (for rocprofv3 counter collection)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pixel_aware_gyro_aided_klt_feature_tracker_amd import capi, synth
cfg, n, kern = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
w = synth.config(cfg, n=n)
p = capi.make_params(half_patch=10, iterations=30, pyramids=3, has_gyro=w.has_gyro, camera=w.camera)
ctx = capi.Context(0)
ctx.set_kernel(kern)
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
for _ in range(reps):
    out = ctx.track(p, w.img_ref, w.img_cur, w.pt_ref, w.pt_init, w.affine, w.status_in)
print("kernel", kern, "ms", ctx.last_kernel_ms()[0], "iters", out["iters"][:w.n].sum())
ctx.close()
