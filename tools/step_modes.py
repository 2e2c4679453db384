"""Times step() variants: streams+events, linear graph, fork graph.  Usage: python tools/step_modes.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pixel_aware_gyro_aided_klt_feature_tracker_amd import capi, runtime, synth
w = synth.config(1, n=1000)
p = capi.make_params(half_patch=w.half_patch, iterations=w.iterations, pyramids=w.pyramids, has_gyro=w.has_gyro, camera=w.camera)
for mode in ("streams", "serial", "graph", "fork", "fused", "graph", "fused"):
    rt = runtime.ResidentTracker(p, device=0)
    rt.load_pair(w.img_ref, w.img_cur)
    rt.set_features(w.pt_ref, w.pt_init, w.affine, w.status_in)
    kw = {"mode": mode}
    try:
        for _ in range(20):
            out = rt.step(**kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        K = 300
        for _ in range(K):
            out = rt.step(**kw)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / K
        st = out["status"].cpu().numpy()
        print(f"{mode:8s}: {dt*1e6:7.1f} us/step  {w.n_active/dt/1e6:6.2f} Mfeat/s  tracked {int(st.sum())}", flush=True)
    except Exception as e:
        print(mode, "failed:", repr(e)[:300], flush=True)
    rt.close()
