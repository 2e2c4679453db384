"""Diagnostic: per-phase cycle budget of k_track_quad (separate -DPAGK_STAMPS build; shares, not run times).
Usage (GPU box): PAGK_N=<features> PAGK_CFG=<config> python tools/stamps_quad.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from pixel_aware_gyro_aided_klt_feature_tracker_amd import capi, synth
capi.LIB_PATH = os.environ.get("PAGK_STAMPS_LIB") or os.path.join(ROOT, "tools", "bin", "libpagk_hip_stamps.so")
n, cfg = int(os.environ.get("PAGK_N", "20000")), int(os.environ.get("PAGK_CFG", "3"))
w = synth.config(cfg, n=n)
kern = int(os.environ.get("PAGK_KERNEL", "5"))   # 5: k_track_quad, 7: its one-level-per-wave form, 6: k_track_rows (timeline only)
nw = (w.n + 3) // 4 * (3 if kern == 7 else 1)   # wave records; a finisher's per-feature records follow them
dbg = torch.zeros((nw + w.n) * 16, dtype=torch.int64, device="cuda")
os.environ["PAGK_DBG_PTR"] = str(dbg.data_ptr())
ctx = capi.Context(0)
ctx.set_kernel(kern)
p = capi.make_params(half_patch=10, iterations=30, pyramids=3, has_gyro=w.has_gyro, camera=w.camera)
for _ in range(2):
    out = ctx.track(p, w.img_ref, w.img_cur, w.pt_ref, w.pt_init, w.affine, w.status_in)
trk, _ = ctx.last_kernel_ms()
full = dbg.cpu().numpy().reshape(-1, 16).astype(np.float64)
d = full[:nw]
d = d[d[:, 6] > 0]
t0 = d[:, 7].min()
b, e = (d[:, 7] - t0) / 100.0, (d[:, 8] - t0) / 100.0   # us since the first wave started
wi = d[:, 6]
it = out["iters"][:w.n]
print(f"cfg{cfg} n={w.n}: kernel {trk*1e3:.1f} us (stamped build), {len(d)} waves, wave-iterations mean {wi.mean():.1f} "
      f"(features: mean {it[w.status_in > 0].mean():.1f} iterations)")
names = ["level setup", "sampling", "MFMA chain", "cost chain", "solve+update", "total"]
for k in range(6 if kern in (5, 7) else 0):   # k_track_rows records the timeline only
    per = d[:, k] / (wi if k in (1, 2, 3, 4) else 1)
    print(f"  {names[k]:13s}: {100 * d[:, k].sum() / d[:, 5].sum():5.1f} %  mean {d[:, k].mean():10.0f} cycles/wave"
          + (f"  {per.mean():8.0f} cycles per wave-iteration" if k in (1, 2, 3, 4) else ""))
print(f"  timeline (us since the first wave began): last wave begins {b.max():.0f}, waves finished 50 % {np.percentile(e, 50):.0f}  "
      f"90 % {np.percentile(e, 90):.0f}  99 % {np.percentile(e, 99):.0f}  99.9 % {np.percentile(e, 99.9):.0f}  all {e.max():.0f}")
for lo, hi in ((0, 100), (100, 200), (200, 300), (300, 400), (400, 600), (600, 800), (800, 1e9)):
    m = (b >= lo) & (b < hi)
    if m.any():
        print(f"    waves begun in [{lo:.0f}, {hi:.0f}) us: {m.sum():5d}, wave-iterations mean {wi[m].mean():5.1f} max {wi[m].max():3.0f}, "
              f"us per wave-iteration {((e - b)[m] / wi[m]).mean():.2f}, run time mean {(e - b)[m].mean():.0f} max {(e - b)[m].max():.0f} us")
for t in range(0, int(e.max()) + 100, 100):
    print(f"    t = {t:4d} us: {((b <= t) & (e > t)).sum():5d} waves in flight")
r = full[nw:]
r = r[r[:, 7] > 0]
if len(r):
    rb, re_ = (r[:, 7] - t0) / 100.0, (r[:, 11] - t0) / 100.0
    print(f"  k_track_resume: {len(r)} features, begins {rb.min():.0f} us, ends {re_.max():.0f} us; per feature: run time mean "
          f"{(re_ - rb).mean():.0f} max {(re_ - rb).max():.0f} us; iterations at hand-over {os.environ.get('PAGK_QUAD_BUDGET', '20')}, final mean "
          f"{r[:, 6].mean():.1f} max {r[:, 6].max():.0f}")
