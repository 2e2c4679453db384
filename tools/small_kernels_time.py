"""Times the small kernels around the loop (HIP events on the context stream).  Usage: python tools/small_kernels_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from pixel_aware_gyro_aided_klt_feature_tracker_amd import capi, synth
from util import make_geometry_case

ctx = capi.Context(0)
stream = torch.cuda.Stream()
ctx.set_stream(stream.cuda_stream)

def timed(fn, reps=200):
    with torch.cuda.stream(stream):
        for _ in range(10):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            fn()
        e1.record(stream)
    e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

for n in (1000, 20000):
    g = make_geometry_case(7, n, outlier_fraction=0.2)
    with torch.cuda.stream(stream):
        d1, d2 = torch.from_numpy(g["pts1"]).cuda(), torch.from_numpy(g["pts2"]).cuda()
        dH, dF = torch.zeros(n, dtype=torch.uint8, device="cuda"), torch.zeros(n, dtype=torch.uint8, device="cuda")
        ds = torch.zeros(2, device="cuda")
    us = timed(lambda: ctx.geometry_scores_device(g["H21"], g["H12"], g["F21"], n, d1, d2, 1.0, dH, dF, ds))
    print(f"k_geometry_scores n={n}: {us:.1f} us per launch (back to back)")
    w = synth.config(1 if n == 1000 else 3, n=n)
    p = capi.make_params(half_patch=10, iterations=30, pyramids=3, camera=w.camera)
    KRK = np.eye(3, dtype=np.float32); r3 = np.array([0, 0, 1], np.float32)
    with torch.cuda.stream(stream):
        dref = torch.from_numpy(w.pt_ref).cuda()
        dpu, dpd = torch.zeros((n, 2), device="cuda"), torch.zeros((n, 2), device="cuda")
        dst, dA = torch.zeros(n, dtype=torch.uint8, device="cuda"), torch.zeros((n, 4), device="cuda")
    H, W = w.img_ref.shape
    us = timed(lambda: ctx.gyro_predict_device(p, W, H, KRK, r3, n, dref, dpu, dpd, dst, dA))
    print(f"k_gyro_predict    n={n}: {us:.1f} us per launch (back to back)")
for (W, H, L) in ((752, 480, 3), (1920, 1080, 3), (1241, 375, 3)):
    with torch.cuda.stream(stream):
        img = torch.randint(0, 255, (H, W), dtype=torch.uint8, device="cuda")
    us = timed(lambda: ctx.frame_set_device(0, img.data_ptr(), W, H, W, L))
    print(f"pyramid {W}x{H} L={L}: {us:.1f} us per frame (back to back)")
ctx.set_stream(None)
