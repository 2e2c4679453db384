"""Diagnostic for the ordering of the sharded step's gather (runtime._sharded_step): three different frames, each step's gathered result
read while the next step is in flight; says which frame every result belongs to.  PAGK_PROBE_MODE=graph|serial."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pixel_aware_gyro_aided_klt_feature_tracker_amd import capi, synth, distributed, runtime
mode = os.environ.get("PAGK_PROBE_MODE", "graph")
w = synth.config(1, n=600)
p = capi.make_params(half_patch=w.half_patch, iterations=w.iterations, pyramids=w.pyramids, has_gyro=w.has_gyro, camera=w.camera)
g = capi.Multi([0])
frames = [w.img_cur, np.roll(w.img_cur, 1, axis=1).copy(), np.roll(w.img_cur, -1, axis=0).copy(), np.roll(w.img_cur, 2, axis=1).copy()]
plain = runtime.ResidentTracker(p, device=0)
plain.load_pair(w.img_ref, w.img_cur)
plain.set_features(w.pt_ref, w.pt_init, w.affine, w.status_in)
refs = []
for f in frames:
    plain.set_current_image(f)
    res = plain.step(mode="serial")
    plain.synchronize()   # an ungathered result is plain views: read behind the tracker's stream
    refs.append(distributed.to_numpy(res)["pt_un"])
distributed.COMM, distributed.FORCE_COLLECTIVE = g, True
rt = runtime.ResidentTracker(p, device=0)
rt.load_pair(w.img_ref, w.img_cur)
rt.set_features(w.pt_ref, w.pt_init, w.affine, w.status_in)
outs, gots = [], []
for f in frames:
    rt.set_current_image(f)
    outs.append(rt.step(mode=mode))
    if os.environ.get("PAGK_PROBE_SYNC") == "1":
        torch.cuda.synchronize()
    if len(outs) >= 2:
        gots.append(distributed.to_numpy(outs[-2])["pt_un"])
gots.append(distributed.to_numpy(outs[-1])["pt_un"])
rt.synchronize()
def which(a):
    m = [j for j, r in enumerate(refs) if np.array_equal(a, r, equal_nan=True)]
    return m if m else "none"
print("mode", mode, "mode_used", rt.mode_used, ": result k belongs to frame", [which(a) for a in gots])
distributed.COMM, distributed.FORCE_COLLECTIVE = None, False
