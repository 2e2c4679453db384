"""CPU tests of the geometry-validation scoring restatement (oracle/pagk_oracle.c, reference
src/gyro_aided_tracker.cpp:429-480, 589-768): against its golden vectors, against an independent float64
numpy evaluation of the same formulas, and on the edge cases of the bookkeeping."""
import numpy as np
import pytest

from oracle import pagk_oracle as orc
from util import geometry_cases, load_geometry, make_geometry_case


@pytest.mark.parametrize("name", geometry_cases())
def test_oracle_reproduces_geometry_golden(name, built):
    g = load_geometry(name)
    inH, sH = orc.check_homography(g["H21"], g["H12"], g["pts1"], g["pts2"], float(g["sigma"]))
    inF, sF = orc.check_fundamental(g["F21"], g["pts1"], g["pts2"], float(g["sigma"]))
    assert np.array_equal(inH, g["out_inl_H"]) and np.array_equal(inF, g["out_inl_F"])
    assert sH.tobytes() == g["out_score_H"].tobytes() and sF.tobytes() == g["out_score_F"].tobytes()
    cnt, st, ts = orc.geometry_validation(g["H21"], g["H12"], g["F21"], g["pts1"], g["pts2"], g["status_in"],
                                          float(g["sigma"]))
    assert cnt == int(g["out_cnt"]) and np.array_equal(st, g["out_status"])
    assert ts.tobytes() == g["out_track_score"].tobytes()


def _numpy_scores(g):
    """The same formulas in float64 (no float narrowing): chi-squares agree to ~1e-4 relative."""
    p1 = np.c_[g["pts1"].astype(np.float64), np.ones(len(g["pts1"]))]
    p2 = np.c_[g["pts2"].astype(np.float64), np.ones(len(g["pts2"]))]
    q = p1 @ g["H21"].T
    chi2 = ((p2[:, :2] - q[:, :2] / q[:, 2:]) ** 2).sum(1)
    q = p2 @ g["H12"].T
    chi1 = ((p1[:, :2] - q[:, :2] / q[:, 2:]) ** 2).sum(1)
    l2 = p1 @ g["F21"].T
    f2 = (l2 * p2).sum(1) ** 2 / (l2[:, 0] ** 2 + l2[:, 1] ** 2)
    l1 = p2 @ g["F21"]
    f1 = (l1 * p1).sum(1) ** 2 / (l1[:, 0] ** 2 + l1[:, 1] ** 2)
    return chi2, chi1, f2, f1


@pytest.mark.parametrize("seed,planar", [(11, False), (12, True)])
def test_oracle_scoring_against_float64_numpy(seed, planar, built):
    g = make_geometry_case(seed, 500, planar=planar)
    chi2, chi1, f2, f1 = _numpy_scores(g)
    inH, sH = orc.check_homography(g["H21"], g["H12"], g["pts1"], g["pts2"])
    inF, sF = orc.check_fundamental(g["F21"], g["pts1"], g["pts2"])
    # masks: identical wherever the float64 value is not within rounding distance of the threshold
    clear = (np.abs(chi2 - 5.99) > 1e-2) & (np.abs(chi1 - 5.99) > 1e-2)
    assert np.array_equal(inH[clear], ((chi2 <= 5.99) & (chi1 <= 5.99))[clear].astype(np.uint8))
    clear = (np.abs(f2 - 3.84) > 1e-2) & (np.abs(f1 - 3.84) > 1e-2)
    assert np.array_equal(inF[clear], ((f2 <= 3.84) & (f1 <= 3.84))[clear].astype(np.uint8))
    refH = np.where(chi2 <= 5.99, 5.99 - chi2, 0).sum() + np.where(chi1 <= 5.99, 5.99 - chi1, 0).sum()
    refF = np.where(f2 <= 3.84, 5.99 - f2, 0).sum() + np.where(f1 <= 3.84, 5.99 - f1, 0).sum()
    assert abs(float(sH) - refH) <= 2e-3 * max(refH, 1) and abs(float(sF) - refF) <= 2e-3 * max(refF, 1)


def test_exact_correspondences_score_the_maximum(built):
    # points mapped exactly by H (float-representable translation): chi = 0 twice per point
    p1 = np.array([[10, 20], [100, 50], [300, 200], [640, 400]], np.float32)
    H = np.array([[1, 0, 8.0], [0, 1, -4.0], [0, 0, 1]])
    inl, sc = orc.check_homography(H, np.linalg.inv(H), p1, p1 + np.float32([8, -4]))
    assert inl.tolist() == [1, 1, 1, 1]
    acc = np.float32(0)
    for _ in range(8):
        acc = np.float32(acc + np.float32(5.99))
    assert sc == acc
    # epipolar lines of a pure x-translation are the image rows: points moved along x score 5.99 each way
    F = np.array([[0, 0, 0], [0, 0, -1.0], [0, 1.0, 0]])
    inl, sc = orc.check_fundamental(F, p1, p1 + np.float32([13, 0]))
    assert inl.tolist() == [1, 1, 1, 1] and sc == acc
    inl, sc = orc.check_fundamental(F, p1, p1 + np.float32([13, 2]))   # 2 px off the line: 4 > 3.84
    assert inl.tolist() == [0, 0, 0, 0] and sc == 0


def test_degenerate_models_and_nan(built):
    g = make_geometry_case(5, 40)
    # zero fundamental matrix: 0/0 = NaN, `NaN > th` is false -> counted as inlier, score NaN (reference :734-739)
    inl, sc = orc.check_fundamental(np.zeros((3, 3)), g["pts1"], g["pts2"])
    assert inl.all() and np.isnan(sc)
    # a NaN coordinate poisons that point's terms and hence the running score
    p2 = g["pts2"].copy()
    p2[7, 0] = np.nan
    inl, sc = orc.check_homography(g["H21"], g["H12"], g["pts1"], p2)
    assert inl[7] == 1 and np.isnan(sc)
    # sigma = 0: invSigmaSquare = inf; exact matches give 0 * inf = NaN, the rest inf > th
    inl, sc = orc.check_homography(np.eye(3), np.eye(3), g["pts1"], g["pts2"], sigma=0.0)
    assert not inl.any() and sc == 0
    # no correspondences
    inl, sc = orc.check_homography(np.eye(3), np.eye(3), np.zeros((0, 2), np.float32), np.zeros((0, 2), np.float32))
    assert inl.size == 0 and sc == 0


def test_model_choice_threshold(built):
    # RH = sH / (sF + sH) in float, compared with the double 0.45 (:462-465)
    assert orc.geometry_select(45.5, 54.5) and not orc.geometry_select(44.5, 55.5)
    assert not orc.geometry_select(0.0, 0.0)            # 0/0 = NaN -> fundamental
    assert not orc.geometry_select(np.nan, 10.0)
    # float(0.45) = 0.449999988 < 0.45: a ratio that rounds to float 0.45 still selects F
    assert not orc.geometry_select(np.float32(0.45), np.float32(1.0) - np.float32(0.45))


def test_validation_bookkeeping(built):
    g = make_geometry_case(21, 200)
    st = np.ones(200, np.uint8)
    st[50:] = 0
    st[:42] = 0                      # 8 status-true correspondences: nothing happens (:445)
    cnt, out, ts = orc.geometry_validation(g["H21"], g["H12"], g["F21"], g["pts1"], g["pts2"], st)
    assert cnt == 0 and np.array_equal(out, st) and ts == 0
    st[41] = 1                       # 9: validated
    cnt, out, ts = orc.geometry_validation(g["H21"], g["H12"], g["F21"], g["pts1"], g["pts2"], st)
    idx = np.flatnonzero(st)
    inH, sH = orc.check_homography(g["H21"], g["H12"], g["pts1"][idx], g["pts2"][idx])
    inF, sF = orc.check_fundamental(g["F21"], g["pts1"][idx], g["pts2"][idx])
    use = inH if orc.geometry_select(sH, sF) else inF
    exp = st.copy()
    exp[idx[use == 0]] = 0
    assert np.array_equal(out, exp) and cnt == int(use.sum()) and ts == (sH if orc.geometry_select(sH, sF) else sF)
    assert not out[~st.astype(bool)].any()   # status-false entries are never revived
