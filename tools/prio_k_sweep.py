"""Kernel time of the 4-wave kernels per setting of PAGK_PRIO_K (csrc/pagk_prio.h; 0 = rule off), each setting in its own process,
alternated: configs[1] 250 / 1000 / 2000 features, configs[2] 2000 x 4 levels."""
import os, subprocess, sys
child = r'''
import os, sys
sys.path.insert(0, os.environ["PAGK_ROOT"])
import numpy as np
from pixel_aware_gyro_aided_klt_feature_tracker_amd import capi, synth
ctx = capi.Context(0)
out = []
for cfg, n, L in ((1, 250, 3), (1, 1000, 3), (1, 2000, 3), (2, 2000, 4)):
    w = synth.config(cfg, n=n)
    p = capi.make_params(half_patch=10, iterations=30, pyramids=L, has_gyro=w.has_gyro, camera=w.camera)
    ts = []
    for _ in range(16):
        ctx.track(p, w.img_ref, w.img_cur, w.pt_ref, w.pt_init, w.affine, w.status_in)
        ts.append(ctx.last_kernel_ms()[0])
    out.append("%d: %.1f" % (n, np.median(ts[4:]) * 1e3))
print("   ".join(out))
'''
for rep in range(2):
    for k in os.environ.get("PAGK_K_LIST", "0,3,4,5").split(","):
        e = dict(os.environ); e["PAGK_PRIO_K"] = k; e["PAGK_ROOT"] = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        r = subprocess.run([sys.executable, "-c", child], capture_output=True, text=True, env=e)
        print("PAGK_PRIO_K=" + k, "|", r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:], flush=True)
