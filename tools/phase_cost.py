"""One interval of wave 0's iteration per build (-DPAGK_TIC=a -DPAGK_TOC=b: points 0 iteration top, 1 after B1, 2 chain
done, 3 solve done, 4 after B2, 5 loop bottom), so that the measurement costs the iteration one scalar load instead of
the dozen stamps of the -DPAGK_STAMPS build.  python tools/phase_cost.py  (builds tools/bin/libpagk_pt_<a><b>.so if absent)"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
PAIRS = [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (0, 5)]
NAMES = {(0, 1): "round 0 .. B1", (1, 2): "chain", (2, 3): "solve", (3, 4): "B2", (4, 5): "update", (0, 5): "whole iteration"}
def lib(a, b):
    return os.path.join(ROOT, "tools", "bin", f"libpagk_pt_{a}{b}.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    procs = []
    for a, b in PAIRS:
        if not os.path.exists(lib(a, b)):
            procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc", *g.HIPCC_FLAGS, f"-DPAGK_TIC={a}", f"-DPAGK_TOC={b}", "-o", lib(a, b),
                                           os.path.join(g.CSRC, "pagk_hip.hip")]))
    sys.exit(max([p.wait() for p in procs] + [0]))
child = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch
from pixel_aware_gyro_aided_klt_feature_tracker_amd import capi, synth
capi.LIB_PATH = sys.argv[1]
out = []
for n in [int(v) for v in os.environ.get("PAGK_N", "8,250,1000").split(",")]:
    w = synth.config(1, n=n)
    dbg = torch.zeros(n * 2, dtype=torch.int64, device="cuda")
    os.environ["PAGK_DBG_PTR"] = str(dbg.data_ptr())
    ctx = capi.Context(0)
    p = capi.make_params(half_patch=10, iterations=30, pyramids=3, has_gyro=w.has_gyro, camera=w.camera)
    for _ in range(3):
        ctx.track(p, w.img_ref, w.img_cur, w.pt_ref, w.pt_init, w.affine, w.status_in)
    d = dbg.cpu().numpy().reshape(n, 2).astype(np.float64)
    ok = d[:, 1] > 0
    out.append("n=%%d: %%.0f" %% (n, (d[ok, 0] / d[ok, 1]).mean()))
    ctx.close()
print("  ".join(out))
''' % ROOT
for a, b in PAIRS:
    r = subprocess.run([sys.executable, "-c", child, lib(a, b)], capture_output=True, text=True)
    print(f"{NAMES[(a, b)]:16s} cycles / iteration: " + (r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]), flush=True)
