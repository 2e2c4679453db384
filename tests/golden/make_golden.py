"""Generates the committed golden vectors under tests/golden/.

What these fixtures pin: the CPU oracle's own outputs (oracle/pagk_oracle.c) on small seeded
cases, one per flag / edge-case combination of the path.  They guard the oracle against
regressions and give the GPU tests fixed expected outputs.  They do NOT pin the oracle to the
reference: the reference cannot be built here and ships no vectors (PARITY UNPINNED, see
oracle/pagk_oracle.h).  Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import pagk_oracle as orc  # noqa: E402
from pixel_aware_gyro_aided_klt_feature_tracker_amd import capi, synth  # noqa: E402

CASES = []


def case(name, w, **flags):
    CASES.append((name, w, flags))


def base(seed, **kw):
    kw.setdefault("half_patch", 5)
    kw.setdefault("iterations", 10)
    kw.setdefault("pyramids", 3)
    return synth.make_workload("g", 160, 120, 64, seed=seed, camera=synth.D435I, omega=(0.2, -0.3, 0.8), **kw)


# flag combinations of GyroAidedTracker::eType (reference src/gyro_aided_tracker.cpp:384-408)
for i, (aff, ill, pen) in enumerate([(a, b, c) for a in (0, 1) for b in (0, 1) for c in (0, 1)]):
    case(f"flags_a{aff}_i{ill}_p{pen}", base(0x601D0000 + i), affine=bool(aff), illumination=bool(ill),
         penalty=bool(pen))
# BASELINE-shaped patch / iteration counts
case("h10_it30_L3", base(0x601D0100, half_patch=10, iterations=30))
case("h10_it30_L4", synth.make_workload("g", 320, 240, 64, seed=0x601D0101, half_patch=10, iterations=30, pyramids=4,
                                        camera=synth.D435I, omega=(0.2, -0.3, 0.8)))
case("h7_L2", base(0x601D0102, half_patch=7, pyramids=2))
case("L1", base(0x601D0103, pyramids=1))
# identity initial guess (the !mbHasGyroPredictInitial branch, :264-270) incl. the penalty NaN case (H7)
case("identity_init", synth.make_workload("g", 160, 120, 64, seed=0x601D0104, half_patch=5, iterations=10,
                                          pyramids=3, motion="translation", has_gyro=False))
case("identity_init_penalty", synth.make_workload("g", 160, 120, 64, seed=0x601D0105, half_patch=5, iterations=10,
                                                  pyramids=3, motion="translation", has_gyro=False), penalty=True)
# features hugging the border (clamped taps, linear-address wrap at the right/bottom edge)
case("edge_features", base(0x601D0106, edge_fraction=1.0))
# PatchMatch::NCC on (bCalculateNCC_), with and without the affine warp (:356-363)
case("ncc_affine", base(0x601D010A), ncc=True)
case("ncc_noaffine_h10", base(0x601D010B, half_patch=10, iterations=30), ncc=True, affine=False)
# some features switched off by the producer (status_in = 0)
w = base(0x601D0107)
w.status_in[::3] = 0
case("status_in_zero", w)
# flat patches: black image region -> H singular -> NaN -> status 0 (:322-326)
w = base(0x601D0108)
w.img_ref[:60, :] = 0
w.img_cur[:60, :] = 0
case("flat_region", w)
# saturated region
w = base(0x601D0109)
w.img_ref[:, :80] = 255
w.img_cur[:, :80] = 255
case("saturated_region", w)


def main():
    for name, w, flags in CASES:
        p = capi.make_params(half_patch=w.half_patch, iterations=w.iterations, pyramids=w.pyramids,
                             has_gyro=w.has_gyro, camera=w.camera,
                             illumination=flags.get("illumination", True), affine=flags.get("affine", True),
                             penalty=flags.get("penalty", False), ncc=flags.get("ncc", False))
        out = orc.track(p, w.img_ref, w.img_cur, w.pt_ref, w.pt_init, w.affine, w.status_in, nthreads=1)
        cam = np.array([w.camera.fx, w.camera.fy, w.camera.cx, w.camera.cy, *w.camera.dist[:4]], np.float64)
        np.savez_compressed(
            os.path.join(HERE, name + ".npz"),
            img_ref=w.img_ref, img_cur=w.img_cur, pt_ref=w.pt_ref, pt_init=w.pt_init, affine=w.affine,
            status_in=w.status_in, camera=cam,
            cfg=np.array([w.half_patch, w.iterations, w.pyramids, int(w.has_gyro), int(flags.get("illumination", True)),
                          int(flags.get("affine", True)), int(flags.get("penalty", False)), int(flags.get("ncc", False))],
                         np.int32),
            **{"out_" + k: v[:w.n] for k, v in out.items()})
        print(f"{name}: n={w.n} ok={int(out['status'][:w.n].sum())} mean iters {out['iters'][:w.n].mean():.2f}")


def main_geometry():
    """Geometry validation scoring (reference src/gyro_aided_tracker.cpp:429-480, 589-768): oracle outputs on
    seeded two-view scenes; the fitted models are numpy stand-ins (tests/util.make_geometry_case)."""
    sys.path.insert(0, os.path.join(HERE, ".."))
    from util import GEOM_DIR, make_geometry_case
    os.makedirs(GEOM_DIR, exist_ok=True)
    cases = {
        "general_300": make_geometry_case(0x6E0A0001, 300),
        "planar_300": make_geometry_case(0x6E0A0002, 300, planar=True),
        "clean_64": make_geometry_case(0x6E0A0003, 64, outlier_fraction=0.0, noise_px=0.1),
        "chunked_4500": make_geometry_case(0x6E0A0004, 4500, outlier_fraction=0.3),
        "few_11": make_geometry_case(0x6E0A0005, 11),   # 9 status-true correspondences: validated (> 8, :445)
        "few_10": make_geometry_case(0x6E0A0006, 10),  # 8 status-true correspondences: left untouched
    }
    for name, g in cases.items():
        inH, sH = orc.check_homography(g["H21"], g["H12"], g["pts1"], g["pts2"], float(g["sigma"]))
        inF, sF = orc.check_fundamental(g["F21"], g["pts1"], g["pts2"], float(g["sigma"]))
        n = g["pts1"].shape[0]
        status = np.ones(n, np.uint8)
        status[::7] = 0
        cnt, st, ts = orc.geometry_validation(g["H21"], g["H12"], g["F21"], g["pts1"], g["pts2"], status, float(g["sigma"]))
        np.savez_compressed(os.path.join(GEOM_DIR, name + ".npz"), **g, status_in=status, out_inl_H=inH, out_inl_F=inF,
                            out_score_H=sH, out_score_F=sF, out_cnt=np.int32(cnt), out_status=st, out_track_score=ts)
        print(f"geometry/{name}: n={n} inliers H {int(inH.sum())} F {int(inF.sum())} scores {sH:.3f} {sF:.3f} "
              f"choose {'H' if orc.geometry_select(sH, sF) else 'F'} validated {cnt}")


def main_neighbors():
    """NCC nearest-neighbour matching (reference src/gyro_aided_tracker.cpp:788-851, 949-1008; free NCC
    src/utils.cpp:110-148): oracle outputs on seeded cases (tests/util.make_neighbor_case)."""
    sys.path.insert(0, os.path.join(HERE, ".."))
    from util import NBR_DIR, make_neighbor_case
    os.makedirs(NBR_DIR, exist_ok=True)
    cases = {
        "h5_ncc": (make_neighbor_case(0x4E420001), dict(use_ncc=True)),
        "h5_distance": (make_neighbor_case(0x4E420002), dict(use_ncc=False)),
        "h10_ncc": (make_neighbor_case(0x4E420003, half_patch=10), dict(use_ncc=True)),
        "h5_noaffine": (make_neighbor_case(0x4E420004), dict(use_ncc=True, affine=False)),
        "h5_pad1": (make_neighbor_case(0x4E420005, pad=1), dict(use_ncc=True)),
        "h5_pad3": (make_neighbor_case(0x4E420006, pad=3), dict(use_ncc=True)),
    }
    for name, (g, opt) in cases.items():
        h = int(g["half_patch"])
        aff = g["affine"] if opt.get("affine", True) else None
        cap = 48
        # level 1, then level 2 for the features level 1 left empty (:912-925)
        r1 = orc.find_near_neighbors(g["img_ref"], g["img_cur"], h, g["keys_ref"], g["pt_predict_un"], g["status"], aff,
                                     g["keys_cur"], g["keys_cur_un"], level=1, use_ncc=opt["use_ncc"], cap=cap)
        r2 = orc.find_near_neighbors(g["img_ref"], g["img_cur"], h, g["keys_ref"], g["pt_predict_un"], g["status"], aff,
                                     g["keys_cur"], g["keys_cur_un"], level=2, use_ncc=opt["use_ncc"], cap=cap,
                                     count=r1["count"])
        assert r1["rc"] == 0 and r2["rc"] == 0
        # level-2 lists of the features level 1 had already filled are not rewritten: merge like mvvNearNeighbors
        keep = r1["count"] > 0
        for k in ("idx", "dist", "ncc"):
            r2[k][keep] = r1[k][keep]
        m1 = orc.match_features(r1["count"], r1["idx"], r1["dist"], r1["ncc"], opt["use_ncc"])
        m2 = orc.match_features(r2["count"], r2["idx"], r2["dist"], r2["ncc"], opt["use_ncc"])
        step = g["img_ref"].strides[0]
        np.savez_compressed(
            os.path.join(NBR_DIR, name + ".npz"),
            **{k: (np.ascontiguousarray(v) if getattr(v, "ndim", 0) > 0 else v) for k, v in g.items()},
            row_step=np.int32(step), use_ncc=np.int32(opt["use_ncc"]), use_affine=np.int32(opt.get("affine", True)),
            cap=np.int32(cap),
            out1_count=r1["count"], out1_idx=r1["idx"], out1_dist=r1["dist"], out1_ncc=r1["ncc"],
            out2_count=r2["count"], out2_idx=r2["idx"], out2_dist=r2["dist"], out2_ncc=r2["ncc"],
            match1_query=m1[0], match1_train=m1[1], match2_query=m2[0], match2_train=m2[1])
        print(f"neighbors/{name}: n={g['keys_ref'].shape[0]} m={g['keys_cur'].shape[0]} lists l1 {int((r1['count']>0).sum())} "
              f"(max {int(r1['count'].max())}) l2 {int((r2['count']>0).sum())} (max {int(r2['count'].max())}) "
              f"matches {len(m1[0])} / {len(m2[0])}")


if __name__ == "__main__":
    if "--geometry-only" not in sys.argv and "--neighbors-only" not in sys.argv:
        main()
    if "--neighbors-only" not in sys.argv:
        main_geometry()
    if "--geometry-only" not in sys.argv:
        main_neighbors()
