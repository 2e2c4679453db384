"""GPU parity tests of the geometry-validation scoring kernel (k_geometry_scores, C ABI
pagk_geometry_scores*, reference src/gyro_aided_tracker.cpp:429-480, 589-768) against the CPU oracle and the
committed fixtures.  Bar: inlier masks bit-exact; the float scores are accumulated in the reference's index
order, so they are required to be bit-identical as well."""
import numpy as np
import pytest
import torch

from oracle import pagk_oracle as orc
from pixel_aware_gyro_aided_klt_feature_tracker_amd import capi

from util import geometry_cases, load_geometry, make_geometry_case

pytestmark = pytest.mark.gpu


def same_bits(a, b):
    """Bit-identical floats; any NaN equals any NaN (x86 produces the negative default NaN for 0/0,
    gfx950 the positive one -- sign and payload of a NaN carry no meaning on this path)."""
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    if np.isnan(a).any() or np.isnan(b).any():
        return np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[~np.isnan(a)], b[~np.isnan(b)])
    return a.tobytes() == b.tobytes()


def check_against_oracle(ctx, g, sigma=1.0):
    inH, inF, sH, sF = ctx.geometry_scores(g["H21"], g["H12"], g["F21"], g["pts1"], g["pts2"], sigma)
    rH, rsH = orc.check_homography(g["H21"], g["H12"], g["pts1"], g["pts2"], sigma)
    rF, rsF = orc.check_fundamental(g["F21"], g["pts1"], g["pts2"], sigma)
    assert np.array_equal(inH, rH), "homography inlier mask differs"
    assert np.array_equal(inF, rF), "fundamental inlier mask differs"
    assert same_bits(sH, rsH), f"score_H {sH!r} != {rsH!r}"
    assert same_bits(sF, rsF), f"score_F {sF!r} != {rsF!r}"
    assert capi.geometry_select(sH, sF) == orc.geometry_select(rsH, rsF)
    return inH, inF, sH, sF


@pytest.mark.parametrize("name", geometry_cases())
def test_scores_match_golden_vectors(ctx, name):
    g = load_geometry(name)
    inH, inF, sH, sF = ctx.geometry_scores(g["H21"], g["H12"], g["F21"], g["pts1"], g["pts2"], float(g["sigma"]))
    assert np.array_equal(inH, g["out_inl_H"]) and np.array_equal(inF, g["out_inl_F"])
    assert same_bits(sH, g["out_score_H"]) and same_bits(sF, g["out_score_F"])
    cnt, st, ts = ctx.geometry_validation(g["H21"], g["H12"], g["F21"], g["pts1"], g["pts2"], g["status_in"],
                                          float(g["sigma"]))
    assert cnt == int(g["out_cnt"]) and np.array_equal(st, g["out_status"]) and same_bits(ts, g["out_track_score"])


# sizes around the LDS chunk (2048 correspondences), the 32-term chain blocks and BASELINE's launch sizes
@pytest.mark.parametrize("n", [0, 1, 2, 15, 16, 17, 1000, 2047, 2048, 2049, 4096, 4097, 20000])
def test_scores_match_oracle_at_every_size(ctx, n):
    check_against_oracle(ctx, make_geometry_case(1000 + n, n, outlier_fraction=0.2))


@pytest.mark.parametrize("sigma", [0.5, 1.0, 2.5])
@pytest.mark.parametrize("planar", [False, True])
def test_scores_match_oracle_sigma_and_scene(ctx, sigma, planar):
    check_against_oracle(ctx, make_geometry_case(77, 3000, planar=planar, noise_px=0.8), sigma)


def test_degenerate_inputs_follow_the_reference(ctx):
    g = make_geometry_case(5, 2500)
    z = dict(g, F21=np.zeros((3, 3)))                 # 0/0 -> NaN chi-square -> "inlier", NaN score
    inH, inF, sH, sF = check_against_oracle(ctx, z)
    assert inF.all() and np.isnan(sF)
    p2 = g["pts2"].copy()
    p2[2100] = np.nan                                 # NaN in the second LDS chunk poisons the carried score
    inH, inF, sH, sF = check_against_oracle(ctx, dict(g, pts2=p2))
    assert np.isnan(sH) and np.isnan(sF)
    p2 = g["pts2"].copy()
    p2[5] = np.inf
    check_against_oracle(ctx, dict(g, pts2=p2))
    check_against_oracle(ctx, g, sigma=0.0)           # invSigmaSquare = inf
    sing = dict(g, H21=np.array([[1, 0, 0], [0, 1, 0], [0, 0, 0.0]]), H12=np.full((3, 3), np.inf))
    check_against_oracle(ctx, sing)                   # w = 1/0 for every point


def test_device_resident_entry(ctx):
    g = make_geometry_case(91, 5000, outlier_fraction=0.25)
    n = 5000
    stream = torch.cuda.Stream()   # the inputs are produced and consumed on one stream, like runtime.ResidentTracker
    with torch.cuda.stream(stream):
        d1, d2 = torch.from_numpy(g["pts1"]).cuda(), torch.from_numpy(g["pts2"]).cuda()
        dH, dF = torch.zeros(n, dtype=torch.uint8, device="cuda"), torch.zeros(n, dtype=torch.uint8, device="cuda")
        ds = torch.full((2,), -1.0, dtype=torch.float32, device="cuda")
        ctx.set_stream(stream.cuda_stream)
        try:
            ctx.geometry_scores_device(g["H21"], g["H12"], g["F21"], n, d1, d2, 1.0, dH, dF, ds)
            stream.synchronize()
        finally:
            ctx.set_stream(None)
    rH, rsH = orc.check_homography(g["H21"], g["H12"], g["pts1"], g["pts2"])
    rF, rsF = orc.check_fundamental(g["F21"], g["pts1"], g["pts2"])
    assert np.array_equal(dH.cpu().numpy(), rH) and np.array_equal(dF.cpu().numpy(), rF)
    s = ds.cpu().numpy()
    assert same_bits(s[0], rsH) and same_bits(s[1], rsF)


def test_validation_matches_oracle_and_respects_the_eight_point_rule(ctx):
    g = make_geometry_case(33, 1200, outlier_fraction=0.15)
    rng = np.random.default_rng(3)
    st = (rng.random(1200) < 0.8).astype(np.uint8)
    cnt, out, ts = ctx.geometry_validation(g["H21"], g["H12"], g["F21"], g["pts1"], g["pts2"], st)
    rcnt, rout, rts = orc.geometry_validation(g["H21"], g["H12"], g["F21"], g["pts1"], g["pts2"], st)
    assert cnt == rcnt and np.array_equal(out, rout) and same_bits(ts, rts)
    few = np.zeros(1200, np.uint8)
    few[:8] = 1
    cnt, out, ts = ctx.geometry_validation(g["H21"], g["H12"], g["F21"], g["pts1"], g["pts2"], few)
    assert cnt == 0 and np.array_equal(out, few) and ts == 0


def test_bad_arguments_are_rejected(ctx):
    g = make_geometry_case(1, 16)
    with pytest.raises(ValueError):   # a column-major device array would be mis-paired: refused by the binding
        t = torch.zeros((2, 16), device="cuda").t()
        ctx.geometry_scores_device(g["H21"], g["H12"], g["F21"], 16, t, t, 1.0, None, None, None)
    with pytest.raises(ValueError):
        ctx.geometry_scores(np.eye(2), g["H12"], g["F21"], g["pts1"], g["pts2"])
    lib = ctx.lib
    sc = np.zeros(2, np.float32)
    rc = lib.pagk_geometry_scores(ctx.h, None, g["H12"].ctypes.data, g["F21"].ctypes.data, 16, g["pts1"].ctypes.data,
                                  g["pts2"].ctypes.data, 1.0, None, None, None, None)
    assert rc == capi.PAGK_E_ARG
    rc = lib.pagk_geometry_validation(ctx.h, g["H21"].ctypes.data, g["H12"].ctypes.data, g["F21"].ctypes.data, -1,
                                      None, None, None, 1.0, None)
    assert rc == capi.PAGK_E_ARG
