"""Diagnostic: per-phase cycle shares of k_track_block (separate -DPAGK_STAMPS build; never
quote this build's run time -- read the shares).  Usage: python tools/stamps.py"""
import os, sys, subprocess, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import __graft_entry__ as g
from pixel_aware_gyro_aided_klt_feature_tracker_amd import capi, synth

lib_path = os.path.join(ROOT, "tools", "bin", "libpagk_hip_stamps.so")
if not os.path.exists(lib_path):
    subprocess.run(["/opt/rocm/bin/hipcc", *g.HIPCC_FLAGS, "-DPAGK_STAMPS", "-o", lib_path,
                    os.path.join(g.CSRC, "pagk_hip.hip")], check=True)
capi.LIB_PATH = lib_path
w = synth.config(1, n=int(os.environ.get("PAGK_N", "1000")))
n = w.n
dbg = torch.zeros(n * 8, dtype=torch.int64, device="cuda")
os.environ["PAGK_DBG_PTR"] = str(dbg.data_ptr())
ctx = capi.Context(0)
p = capi.make_params(half_patch=10, iterations=30, pyramids=3, has_gyro=w.has_gyro, camera=w.camera)
for _ in range(2):
    out = ctx.track(p, w.img_ref, w.img_cur, w.pt_ref, w.pt_init, w.affine, w.status_in)
trk, pyr = ctx.last_kernel_ms()
d = dbg.cpu().numpy().reshape(n, 8).astype(np.float64)
it = d[:, 6]
names = ["level setup", "sampling", "chains", "solve", "(unused)", "total"]
print(f"kernel {trk*1e3:.1f} us (stamped build), mean iters {it.mean():.2f}")
for k in (0, 1, 2, 3, 5):
    per_it = d[:, k] / (it if k in (1, 2, 3) else 1)
    print(f"  {names[k]:12s}: mean {d[:, k].mean():9.0f} cyc/feature ({100*d[:, k].sum()/d[:, 5].sum():5.1f}%)"
          + (f", {per_it.mean():7.0f} cyc/iteration" if k in (1, 2, 3) else ""))
tb = d[:, 7]
print(f"  block start spread: {(tb.max()-tb.min())/100:.1f} us (100 MHz ticks?) raw {tb.max()-tb.min():.0f}; block duration mean {d[:,5].mean():.0f} max {d[:,5].max():.0f} cycles")
# schedule of the launch: when each workgroup started / finished relative to the first start (cycles)
t0 = tb.min()
start, end = tb - t0, tb - t0 + d[:, 5]
late = start > 0.05 * end.max()
print(f"  launch span {end.max():.0f} cycles; workgroups starting later than 5% of the span: {int(late.sum())} "
      f"(their start: mean {start[late].mean() if late.any() else 0:.0f}, max {start.max():.0f})")
order = np.argsort(end)
print("  last 5 finishers: " + ", ".join(f"iters {int(it[k])} start {start[k]:.0f} end {end[k]:.0f}" for k in order[-5:]))
for q in (50, 90, 99, 100):
    print(f"  iters p{q}: {np.percentile(it, q):.0f}   end p{q}: {np.percentile(end, q):.0f}")
