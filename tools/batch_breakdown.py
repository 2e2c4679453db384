"""Where a batched multi-camera step goes: eight configs[4] streams (pagk_track_device_batch), the batched kernel alone
(HIP events around the launch), the eight pyramids alone, and the whole step directly and as a replayed graph.
python tools/batch_breakdown.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from pixel_aware_gyro_aided_klt_feature_tracker_amd import capi, runtime, synth
w = synth.config(4)
p = capi.make_params(half_patch=w.half_patch, iterations=w.iterations, pyramids=w.pyramids, has_gyro=w.has_gyro, camera=w.camera)
cb = runtime.CameraBatch(p, 8, device=0)
for j in range(8):
    cb.load(j, w.img_ref, w.img_cur, w.pt_ref, w.pt_init, w.affine, w.status_in)


def timed(fn, reps=60, warm=20):
    for _ in range(warm):
        fn()
    cb.stream.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    cb.stream.synchronize()
    return (time.perf_counter() - t) / reps * 1e3


ks = []
for _ in range(30):
    cb.step(mode="serial")
    cb.synchronize()
    ks.append(cb.cams[0].ctx.last_kernel_ms()[0])
print("batched kernel alone (events): %.4f ms" % np.median(ks[10:]))
print("step, direct launches:         %.4f ms" % timed(lambda: cb.step(mode="serial")))
print("step, replayed graph:          %.4f ms" % timed(lambda: cb.step(mode="graph")))


def pyramids():
    with torch.cuda.stream(cb.stream):
        for c in cb.cams:
            c.rebuild_current_pyramid(1)


print("eight pyramids, direct:        %.4f ms" % timed(pyramids))
cb.close()
