"""Diagnostic: where the waves of the pipelined 4-wave kernel (pagk_pipe_kernel.h) spend an iteration (separate
-DPAGK_STAMPS build; never quote this build's run time).  Usage: PAGK_N=8 python tools/stamps_pipe.py"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import __graft_entry__ as g
from pixel_aware_gyro_aided_klt_feature_tracker_amd import capi, synth

lib_path = os.environ.get("PAGK_STAMPS_LIB") or os.path.join(ROOT, "tools", "bin", "libpagk_hip_stamps.so")
if not os.path.exists(lib_path):
    subprocess.run(["/opt/rocm/bin/hipcc", *g.HIPCC_FLAGS, "-DPAGK_STAMPS", "-o", lib_path,
                    os.path.join(g.CSRC, "pagk_hip.hip")], check=True)
capi.LIB_PATH = lib_path
for n in [int(v) for v in os.environ.get("PAGK_N", "8,250,1000").split(",")]:
    w = synth.config(1, n=n)
    dbg = torch.zeros(n * 16, dtype=torch.int64, device="cuda")
    os.environ["PAGK_DBG_PTR"] = str(dbg.data_ptr())
    ctx = capi.Context(0)
    p = capi.make_params(half_patch=10, iterations=30, pyramids=3, has_gyro=w.has_gyro, camera=w.camera)
    for _ in range(2):
        dbg.zero_()
        out = ctx.track(p, w.img_ref, w.img_cur, w.pt_ref, w.pt_init, w.affine, w.status_in)
    trk, pyr = ctx.last_kernel_ms()
    d = dbg.cpu().numpy().reshape(n, 16).astype(np.float64)
    it = np.maximum(d[:, 6], 1)
    print(f"== n={n}: kernel {trk*1e3:.1f} us (stamped build), mean iters {it.mean():.2f}, max {it.max():.0f}")
    rows = [(0, "level set-up (per feature)", False), (1, "wave 0: round 0 .. B1", True), (2, "wave 0: B1 .. both chain waves' sums in LDS", True),
            (3, "wave 0: solve .. B2", True), (12, "wave 0: update", True), (4, "wave 1: B1 .. its sums published", True),
            (8, "wave 2: B1 .. batch A published", True), (9, "wave 2: A .. batch C published", True),
            (10, "wave 3: B1 .. batch B published", True), (13, "wave 3: B .. cost published", True)]
    for k, name, per_it in rows:
        v = d[:, k] / (it if per_it else 1)
        print(f"  {name:48s}: {v.mean():8.0f} cycles" + (" / iteration" if per_it else ""))
    print(f"  {'wave 0: whole iteration (sum)':48s}: {((d[:,1]+d[:,2]+d[:,3]+d[:,12])/it).mean():8.0f} cycles / iteration")
    rb, re_ = d[:, 7], d[:, 11]
    t0 = rb.min()
    end = (re_ - t0) / 100.0
    for q in (50, 90, 99, 100):
        print(f"  iters p{q}: {np.percentile(it, q):.0f}   end p{q}: {np.percentile(end, q):.1f} us")
    ctx.close()
    del ctx
