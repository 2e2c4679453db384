"""How much do the results depend on the guesses the oracle had to make about THIRD-PARTY arithmetic?

The reference calls Eigen (H.llt().solve(b), update.norm()) and OpenCV (cv::resize) -- neither is installed here, so
oracle/pagk_oracle.c restates their arithmetic from the published algorithms (oracle/README.md).  Because the 4x4
system of this algorithm is structurally singular, its results are sensitive to exactly such details.  This tool
re-runs the oracle on BASELINE configs[1] (752x480, 1000 keypoints, h = 10, L = 3, I = 30) with ONE guess at a time
switched to its plausible alternative and reports, against the documented restatement: status flips, features whose
tracked point moves by more than 1e-3 px (the project's coordinate bar), and the largest move.

A maintainer with a real Eigen / OpenCV build can read off which knob to check first.  CPU only (runs anywhere):
    python tools/parity_risk.py [--markdown]
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pagk_oracle as orc  # noqa: E402
from pixel_aware_gyro_aided_klt_feature_tracker_amd import capi, synth  # noqa: E402

ALTERNATIVES = [
    (1, "lower triangular solve, 3-term row: `(c0 + c1) + c2` instead of `c0 + (c1 + c2)`", "Eigen `triangular_solver_unroller` redux shape"),
    (2, "upper triangular solve, 3-term row: `c0 + (c1 + c2)` instead of `(c0 + c1) + c2`", "same, column access"),
    (4, "`update.norm()`: sequential `((x0² + x1²) + x2²) + x3²` instead of the SSE2 packet shape `(x0² + x2²) + (x1² + x3²)`", "Eigen built without vectorisation (`EIGEN_DONT_VECTORIZE`, or a non-SSE target)"),
    (8, "LLT column scaling `A21 *= 1/x` instead of `A21 /= x`", "Eigen <= 3.2 (`llt_inplace::unblocked`)"),
    (32, "4th pivot `A33 - (a0² + (a1² + a2²))` instead of the sequential sum", "a vectorised / tree `squaredNorm` of the 3-element row"),
    (16, "pyramid: exact 2x decimation through the 11-bit fixed-point bilinear kernel instead of the 2x2 box `(a+b+c+d+2)>>2`", "OpenCV versions / builds whose `cv::resize(INTER_LINEAR)` does not switch to the INTER_AREA fast path"),
    (1 | 2 | 4, "all three Eigen association alternatives together", "-"),
]


def run(w, p, flags):
    lib = orc.load()
    lib.pagk_oracle_set_alternatives.restype = None
    lib.pagk_oracle_set_alternatives.argtypes = [C.c_uint32]
    lib.pagk_oracle_set_alternatives(flags)
    try:
        return orc.track(p, w.img_ref, w.img_cur, w.pt_ref, w.pt_init, w.affine, w.status_in, nthreads=8)
    finally:
        lib.pagk_oracle_set_alternatives(0)


def main():
    md = "--markdown" in sys.argv
    orc.build()
    rows = []
    for cfg in (1, 2):
        w = synth.config(cfg)
        p = capi.make_params(half_patch=w.half_patch, iterations=w.iterations, pyramids=w.pyramids, has_gyro=w.has_gyro,
                             camera=w.camera)
        base = run(w, p, 0)
        act = w.status_in > 0
        n_act = int(act.sum())
        for flags, what, when in ALTERNATIVES:
            alt = run(w, p, flags)
            flips = int(np.count_nonzero(alt["status"][:w.n] != base["status"][:w.n]))
            both = act & (alt["status"][:w.n] > 0) & (base["status"][:w.n] > 0)
            d = np.abs(alt["pt_un"][:w.n].astype(np.float64) - base["pt_un"][:w.n].astype(np.float64)).max(axis=1)
            moved = int(np.count_nonzero(d[both] > 1e-3))
            it_changed = int(np.count_nonzero(alt["iters"][:w.n] != base["iters"][:w.n]))
            rows.append((f"configs[{cfg}]", what, when, flips, moved, 100.0 * moved / max(1, int(both.sum())),
                         float(d[both].max()) if both.any() else 0.0, it_changed, n_act))
    if md:
        print("| workload | guess switched to its alternative | when a real build would take the alternative | status flips | points moved > 1e-3 px | largest move (px) | features whose iteration count changes |")
        print("|---|---|---|---|---|---|---|")
        for wl, what, when, flips, moved, pct, dmax, itc, n_act in rows:
            print(f"| {wl} ({n_act} active) | {what} | {when} | {flips} | {moved} ({pct:.1f} %) | {dmax:.3g} | {itc} |")
    else:
        for wl, what, when, flips, moved, pct, dmax, itc, n_act in rows:
            print(f"{wl}: {what}\n    status flips {flips}, moved > 1e-3 px: {moved} of {n_act} ({pct:.1f} %), max move {dmax:.3g} px, "
                  f"iteration count changed for {itc}")


if __name__ == "__main__":
    main()
