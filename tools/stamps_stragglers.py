"""Diagnostic: the slowest features of a launch of the pipelined 4-wave kernel -- when their workgroup started, when it ended, cycles
per iteration -- and the launch's ramp (when the last workgroup started).  Separate -DPAGK_STAMPS build (inflates every phase; never
quote its run time).  Usage: PAGK_N=1000 [PAGK_STAMPS_FLAGS="-DPAGK_PRIO_MODE=0"] python tools/stamps_stragglers.py"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import __graft_entry__ as g
from pixel_aware_gyro_aided_klt_feature_tracker_amd import capi, synth

flags = os.environ.get("PAGK_STAMPS_FLAGS", "").split()
tag = "".join(c if c.isalnum() else "_" for c in "".join(flags))
lib_path = os.path.join(ROOT, "tools", "bin", f"libpagk_hip_stamps{tag}.so")
if not os.path.exists(lib_path):
    os.makedirs(os.path.dirname(lib_path), exist_ok=True)
    subprocess.run(["/opt/rocm/bin/hipcc", *g.HIPCC_FLAGS, "-DPAGK_STAMPS", *flags, "-o", lib_path,
                    os.path.join(g.CSRC, "pagk_hip.hip")], check=True)
if os.environ.get("PAGK_BUILD_ONLY"):
    sys.exit(0)
capi.LIB_PATH = lib_path
for n in [int(v) for v in os.environ.get("PAGK_N", "1000").split(",")]:
    w = synth.config(1, n=n)
    dbg = torch.zeros(n * 16, dtype=torch.int64, device="cuda")
    os.environ["PAGK_DBG_PTR"] = str(dbg.data_ptr())
    ctx = capi.Context(0)
    p = capi.make_params(half_patch=10, iterations=30, pyramids=3, has_gyro=w.has_gyro, camera=w.camera)
    for _ in range(3):
        dbg.zero_()
        ctx.track(p, w.img_ref, w.img_cur, w.pt_ref, w.pt_init, w.affine, w.status_in)
    trk, pyr = ctx.last_kernel_ms()
    d = dbg.cpu().numpy().reshape(n, 16).astype(np.float64)
    it = np.maximum(d[:, 6], 1)
    rb, re_ = d[:, 7], d[:, 11]
    t0 = rb.min()
    start, end = (rb - t0) / 100.0, (re_ - t0) / 100.0
    print(f"== n={n} flags={flags}: kernel {trk*1e3:.1f} us (stamped build); workgroup starts p50 {np.percentile(start,50):.1f} p99 {np.percentile(start,99):.1f} max {start.max():.1f} us; "
          f"ends p50 {np.percentile(end,50):.1f} p90 {np.percentile(end,90):.1f} p99 {np.percentile(end,99):.1f} max {end.max():.1f} us")
    print("   feature iters start_us end_us cycles/iter(wave0 total) setup_cyc  round0  chains  solve+B2  update  (per iteration)")
    for i in np.argsort(-end)[:12]:
        print(f"   {i:6d} {it[i]:4.0f} {start[i]:7.1f} {end[i]:7.1f} {d[i,5]/it[i]:9.0f} {d[i,0]:9.0f} {d[i,1]/it[i]:7.0f} {d[i,2]/it[i]:7.0f} {d[i,3]/it[i]:7.0f} {d[i,12]/it[i]:7.0f}")
    # cycles per iteration by total iteration count
    for lo, hi in ((1, 9), (10, 12), (13, 17), (18, 40)):
        m = (it >= lo) & (it <= hi)
        if m.any():
            print(f"   features with {lo}-{hi} iterations: {m.sum():4d}, cycles per iteration {np.mean(d[m,5]/it[m]):.0f}, end p50 {np.percentile(end[m],50):.1f} us")
    ctx.close()
    del ctx
