"""The pipelined 4-wave kernel with a feature's first K iterations summing their cost on the matrix pipe
(TrackArgs::cost_mfma_iters, PAGK_COST_MFMA_ITERS): kernel time against K.  python tools/cost_mfma_sweep.py [lib.so]"""
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
for cfg, n in ((1, 250), (1, 1000), (1, 2000), (2, 2000)):
    w = synth.config(cfg, n=n)
    p = capi.make_params(half_patch=10, iterations=30, pyramids=w.pyramids, has_gyro=w.has_gyro, camera=w.camera)
    ts = []
    for _ in range(20):
        ctx.track(p, w.img_ref, w.img_cur, w.pt_ref, w.pt_init, w.affine, w.status_in)
        ts.append(ctx.last_kernel_ms()[0])
    out.append("cfg%%d/%%d: %%.1f us" %% (cfg, n, np.median(ts[5:]) * 1e3))
print("   ".join(out))
''' % ROOT
lib = sys.argv[1] if len(sys.argv) > 1 else "-"
for rep in range(2):
    for K in (0, 4, 8, 10, 12, 14, 18, 99):
        env = dict(os.environ, PAGK_COST_MFMA_ITERS=str(K))
        r = subprocess.run([sys.executable, "-c", child, lib], capture_output=True, text=True, env=env)
        print("K=%-3d " % K + (r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]), flush=True)
