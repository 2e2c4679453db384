"""A/B two builds of libpagk_hip.so in one process run each, alternating: python tools/ab_lib.py <libB.so> ...
PAGK_AB_CASES="cfg:n:kernel[:pyramids[:half_patch]],..." chooses the launches (default 1:1000:0,1:4000:0,3:20000:0)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
child = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np
from pixel_aware_gyro_aided_klt_feature_tracker_amd import capi, synth
if sys.argv[1] != "-":
    capi.LIB_PATH = sys.argv[1]
ctx = capi.Context(0)
out = []
cases = [tuple(int(v) for v in c.split(":")) for c in os.environ.get("PAGK_AB_CASES", "1:1000:0,1:4000:0,3:20000:0").split(",")]
for cfg, n, kern, *rest in cases:
    w = synth.config(cfg, n=n)
    ctx.set_kernel(kern)
    p = capi.make_params(half_patch=rest[1] if len(rest) > 1 else 10, iterations=30, pyramids=rest[0] if rest else 3, has_gyro=w.has_gyro, camera=w.camera)
    ts = []
    for _ in range(16):
        ctx.track(p, w.img_ref, w.img_cur, w.pt_ref, w.pt_init, w.affine, w.status_in)
        ts.append(ctx.last_kernel_ms()[0])
    out.append("%%d: %%.1f us" %% (n, np.median(ts[4:]) * 1e3))
print("   ".join(out))
''' % ROOT
libs = ["-"] + sys.argv[1:]
for rep in range(int(os.environ.get("PAGK_AB_REPS", "3"))):
    for lib in libs:
        r = subprocess.run([sys.executable, "-c", child, lib], capture_output=True, text=True)
        print(("A (product) " if lib == "-" else "B (%s) " % os.path.basename(lib)) + r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:], flush=True)
