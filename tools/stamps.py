"""Diagnostic: per-phase cycle shares of k_track_block (separate -DPAGK_STAMPS build; never
quote this build's run time -- read the shares).  Usage: python tools/stamps.py"""
import os, sys, subprocess, ctypes as C
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
w = synth.config(1, n=int(os.environ.get("PAGK_N", "1000")))
n = w.n
dbg = torch.zeros(n * 16, dtype=torch.int64, device="cuda")
os.environ["PAGK_DBG_PTR"] = str(dbg.data_ptr())
ctx = capi.Context(0)
p = capi.make_params(half_patch=10, iterations=30, pyramids=3, has_gyro=w.has_gyro, camera=w.camera)
for _ in range(2):
    out = ctx.track(p, w.img_ref, w.img_cur, w.pt_ref, w.pt_init, w.affine, w.status_in)
trk, pyr = ctx.last_kernel_ms()
d = dbg.cpu().numpy().reshape(n, 16).astype(np.float64)
it = d[:, 6]
names = ["level setup", "sampling", "chains", "solve", "(unused)", "total"]
print(f"kernel {trk*1e3:.1f} us (stamped build), mean iters {it.mean():.2f}")
for k in (0, 1, 2, 3, 5):
    val = d[:, k] + (d[:, 8] + d[:, 9] + d[:, 10] + d[:, 12] + d[:, 13] if k == 1 else 0)
    per_it = val / (it if k in (1, 2, 3) else 1)
    print(f"  {names[k]:12s}: mean {val.mean():9.0f} cyc/feature ({100*val.sum()/d[:, 5].sum():5.1f}%)"
          + (f", {per_it.mean():7.0f} cyc/iteration" if k in (1, 2, 3) else ""))
# the sampling phase in pieces (workgroup thread 0's wave): [8] coordinates + gathers issued, [9] gathers returned,
# [10] interpolation + products + LDS stores, [1] the rest = waiting at the barrier for the other waves
for k, name in ((12, "  update/tests"), (13, "  round 0 coords+issue"), (8, "  round 1 coords+issue"), (9, "  gather wait"), (10, "  math + stores"), (1, "  barrier wait")):
    print(f"  {name:14s}: {(d[:, k] / it).mean():7.0f} cyc/iteration")
# schedule of the launch from the 100 MHz wall clock (s_memrealtime): start / end of every workgroup, in us
rb, re_ = d[:, 7], d[:, 11]
t0 = rb.min()
start, end = (rb - t0) / 100.0, (re_ - t0) / 100.0
print(f"  launch span {end.max():.1f} us; workgroups starting after 5 us: {int((start > 5).sum())} (latest start {start.max():.1f} us)")
order = np.argsort(end)
print("  last 5 finishers: " + ", ".join(f"[iters {int(it[k])} start {start[k]:.1f} end {end[k]:.1f}]" for k in order[-5:]))
for q in (50, 90, 99, 100):
    print(f"  iters p{q}: {np.percentile(it, q):.0f}   end p{q}: {np.percentile(end, q):.1f} us")
