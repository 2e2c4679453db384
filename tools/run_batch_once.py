"""One process that runs the batched multi-camera launch (pagk_track_device_batch: eight 1280x720 x 4000 streams, one
launch per step) a few times -- the rocprofv3 subject of tools/r04_profiles.sh.  python tools/run_batch_once.py [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pixel_aware_gyro_aided_klt_feature_tracker_amd import capi, runtime, synth
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
w = synth.config(4)
p = capi.make_params(half_patch=w.half_patch, iterations=w.iterations, pyramids=w.pyramids, has_gyro=w.has_gyro, camera=w.camera)
cb = runtime.CameraBatch(p, 8, device=0)
for j in range(8):
    cb.load(j, w.img_ref, w.img_cur, w.pt_ref, w.pt_init, w.affine, w.status_in)
for _ in range(steps):
    cb.step(mode="serial")
cb.synchronize()
print("batch of 8 x", w.n, "features,", steps, "steps, variant", cb.cams[0].ctx.last_variant())
cb.close()
