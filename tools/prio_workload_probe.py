"""The priority threshold on workloads that converge more slowly than BASELINE's: configs[1]'s shape with a gyro whose error is scaled up (the
initial guess further off: more iterations on the coarse levels).  Per gyro-error scale: mean iterations per feature and level, and the
kernel time with the rule off (PAGK_PRIO_K=0), the fixed default 4, other fixed values, and auto (with the K it settled on).  Own process per setting."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
child = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np
from pixel_aware_gyro_aided_klt_feature_tracker_amd import capi, synth
scale = float(sys.argv[1])
w = synth.make_workload("slow", 752, 480, 1000, seed=0x5EED0500, half_patch=10, iterations=30, pyramids=3, camera=synth.EUROC if hasattr(synth, "EUROC") else None,
                        gyro_error=tuple(scale * v for v in (0.004, -0.003, 0.006)))
p = capi.make_params(half_patch=10, iterations=30, pyramids=3, has_gyro=w.has_gyro, camera=w.camera)
ctx = capi.Context(0)
ts = []
for _ in range(20):
    out = ctx.track(p, w.img_ref, w.img_cur, w.pt_ref, w.pt_init, w.affine, w.status_in)
    ts.append(ctx.last_kernel_ms()[0])
it = out["iters"][:w.n]
print("%%.1f us  (mean %%.2f iterations per feature and level, max %%d per feature, tracked %%.0f %%%%, K in force %%d)" %% (
    np.median(ts[6:]) * 1e3, it.mean() / 3.0, it.max(), 100.0 * out["status"][:w.n].mean(), ctx.priority_threshold()))
''' % ROOT
for scale in (1.0, 3.0, 6.0):
    for k in os.environ.get("PAGK_K_LIST", "0,4,6,8,auto").split(","):
        e = dict(os.environ); e["PAGK_PRIO_K"] = k
        r = subprocess.run([sys.executable, "-c", child, str(scale)], capture_output=True, text=True, env=e)
        print(f"gyro error x {scale:3.1f}  PAGK_PRIO_K={k:4s} |", r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:], flush=True)
