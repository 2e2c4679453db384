"""The 4-wave kernel between one and three rounds of resident workgroups: automatic selection against k_track_block5 forced
(PAGK_BLOCK5_MIN=1025), configs[1], kernel us per launch size, each setting in its own process, alternated."""
import os, subprocess, sys
child = r'''
import os, sys
sys.path.insert(0, os.environ["PAGK_ROOT"])
import numpy as np
from pixel_aware_gyro_aided_klt_feature_tracker_amd import capi, synth
ctx = capi.Context(0)
out = []
for n in (1100, 1300, 1500, 1800, 2000, 2400, 3000):
    w = synth.config(1, n=n)
    p = capi.make_params(half_patch=10, iterations=30, pyramids=3, has_gyro=w.has_gyro, camera=w.camera)
    ts = []
    for _ in range(16):
        ctx.track(p, w.img_ref, w.img_cur, w.pt_ref, w.pt_init, w.affine, w.status_in)
        ts.append(ctx.last_kernel_ms()[0])
    out.append("%d: %.1f" % (n, np.median(ts[4:]) * 1e3))
print("   ".join(out))
'''
for rep in range(2):
    for name, env in (("default (block5 from 2500)", {}), ("PAGK_BLOCK5_MIN=1025", {"PAGK_BLOCK5_MIN": "1025"})):
        e = dict(os.environ); e.update(env); e['PAGK_ROOT'] = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        r = subprocess.run([sys.executable, "-c", child], capture_output=True, text=True, env=e)
        print(name, "|", r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:], flush=True)
