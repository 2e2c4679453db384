"""NCC nearest-neighbour matching on the device against the CPU oracle: the lists
of FindAndSortNearNeighbor (reference src/gyro_aided_tracker.cpp:788-851) bit for bit -- indices, distances, NCC
scores, order -- and MatchFeatures (:949-1008) on top of them.  Through the C ABI (pagk_find_near_neighbors,
pagk_near_neighbors_device, pagk_ncc_free, pagk_match_features)."""
import numpy as np
import pytest

from util import load_neighbors, make_neighbor_case, neighbor_cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def orc(built):
    from oracle import pagk_oracle
    return pagk_oracle


def same_lists(got, ref, what=""):
    assert np.array_equal(got["count"], ref["count"]), f"{what}: list sizes differ"
    for i, c in enumerate(ref["count"].tolist()):
        for k in ("idx", "dist", "ncc"):
            assert np.array_equal(got[k][i, :c], ref[k][i, :c], equal_nan=True), f"{what}: feature {i} {k}: {got[k][i, :c]} vs {ref[k][i, :c]}"


@pytest.mark.parametrize("name", neighbor_cases())
def test_golden_neighbour_lists_and_matches(ctx, name):
    from pixel_aware_gyro_aided_klt_feature_tracker_amd import capi
    g = load_neighbors(name)
    h, cap, use_ncc = int(g["half_patch"]), int(g["cap"]), bool(g["use_ncc"])
    aff = g["affine"] if int(g["use_affine"]) else None
    r1 = ctx.find_near_neighbors(g["img_ref"], g["img_cur"], h, g["keys_ref"], g["pt_predict_un"], g["status"], aff,
                                 g["keys_cur"], g["keys_cur_un"], level=1, use_ncc=use_ncc, cap=cap)
    assert r1["rc"] == 0
    same_lists(r1, {k: g["out1_" + k] for k in ("count", "idx", "dist", "ncc")}, name + " level 1")
    q, t, _, _ = capi.match_features(r1["count"], r1["idx"], r1["dist"], r1["ncc"], use_ncc)
    assert np.array_equal(q, g["match1_query"]) and np.array_equal(t, g["match1_train"])
    # the wider search only fills the features level 1 left empty (:793, :921-925)
    r2 = ctx.find_near_neighbors(g["img_ref"], g["img_cur"], h, g["keys_ref"], g["pt_predict_un"], g["status"], aff,
                                 g["keys_cur"], g["keys_cur_un"], level=2, use_ncc=use_ncc, cap=cap, count=r1["count"])
    assert r2["rc"] == 0
    keep = r1["count"] > 0
    for k in ("idx", "dist", "ncc"):
        r2[k][keep] = r1[k][keep]
    same_lists(r2, {k: g["out2_" + k] for k in ("count", "idx", "dist", "ncc")}, name + " level 2")
    q, t, _, _ = capi.match_features(r2["count"], r2["idx"], r2["dist"], r2["ncc"], use_ncc)
    assert np.array_equal(q, g["match2_query"]) and np.array_equal(t, g["match2_train"])


@pytest.mark.parametrize("h,n,size,pad", [(10, 300, (752, 480), 0), (7, 200, (640, 480), 0), (5, 500, (640, 480), 2),
                                          (2, 100, (200, 150), 1), (13, 64, (400, 300), 0)])
def test_seeded_cases_match_the_oracle(ctx, orc, h, n, size, pad):
    g = make_neighbor_case(0x4E42A000 + h, n=n, width=size[0], height=size[1], half_patch=h, clutter=3 * n, pad=pad)
    for use_ncc, aff in ((True, g["affine"]), (False, g["affine"]), (True, None)):
        args = (g["img_ref"], g["img_cur"], h, g["keys_ref"], g["pt_predict_un"], g["status"], aff, g["keys_cur"],
                g["keys_cur_un"])
        ref = orc.find_near_neighbors(*args, level=2, use_ncc=use_ncc, cap=96)
        got = ctx.find_near_neighbors(*args, level=2, use_ncc=use_ncc, cap=96)
        assert ref["rc"] == 0 and got["rc"] == 0 and ref["count"].max() > 1
        same_lists(got, ref, f"h={h} use_ncc={use_ncc} affine={aff is not None}")


def test_status_zero_and_prefilled_features_are_left_alone(ctx, orc):
    g = make_neighbor_case(0x4E42A100, n=80)
    st = g["status"].copy()
    st[::4] = 0
    pre = np.zeros(80, np.int32)
    pre[1::4] = 3                      # "neighbours already found at the smaller radius"
    args = (g["img_ref"], g["img_cur"], 5, g["keys_ref"], g["pt_predict_un"], st, g["affine"], g["keys_cur"], g["keys_cur_un"])
    ref = orc.find_near_neighbors(*args, level=1, cap=32, count=pre)
    got = ctx.find_near_neighbors(*args, level=1, cap=32, count=pre)
    assert np.array_equal(got["count"], ref["count"])
    assert np.all(got["count"][::4] == 0) and np.all(got["count"][1::4] == 3)
    assert np.all(got["idx"][::4] == -1) and np.all(got["idx"][1::4] == -1)     # untouched output rows
    live = np.ones(80, bool)
    live[::4] = live[1::4] = False
    same_lists({k: v[live] for k, v in got.items() if k != "rc"}, {k: v[live] for k, v in ref.items() if k != "rc"})


def test_capacity_overflow_returns_the_sizes_needed(ctx, orc):
    from pixel_aware_gyro_aided_klt_feature_tracker_amd import capi
    g = make_neighbor_case(0x4E42A200, n=40, width=160, height=120, half_patch=4, clutter=400)
    args = (g["img_ref"], g["img_cur"], 4, g["keys_ref"], g["pt_predict_un"], g["status"], g["affine"], g["keys_cur"],
            g["keys_cur_un"])
    ref = orc.find_near_neighbors(*args, level=2, cap=512)
    small = ctx.find_near_neighbors(*args, level=2, cap=4)
    assert small["rc"] == capi.PAGK_E_CAPACITY and np.array_equal(small["count"], ref["count"])
    again = ctx.find_near_neighbors(*args, level=2, cap=int(ref["count"].max()))
    assert again["rc"] == 0
    same_lists(again, ref, "retry with the reported capacity")


def test_empty_inputs(ctx):
    g = make_neighbor_case(0x4E42A300, n=16)
    e2 = np.zeros((0, 2), np.float32)
    r = ctx.find_near_neighbors(g["img_ref"], g["img_cur"], 5, g["keys_ref"], g["pt_predict_un"], g["status"], g["affine"],
                                e2, e2, level=1, cap=8)
    assert r["rc"] == 0 and not r["count"].any()          # no current keypoints: every list stays empty
    r = ctx.find_near_neighbors(g["img_ref"], g["img_cur"], 5, e2, e2, np.zeros(0, np.uint8), None, g["keys_cur"],
                                g["keys_cur_un"], level=1, cap=8)
    assert r["rc"] == 0 and r["count"].size == 0


@pytest.mark.parametrize("pad", [0, 1, 3])
def test_free_ncc_of_point_pairs(ctx, orc, pad):
    """pagk_ncc_free = the two-image overload src/utils.cpp:166-200, incl. patches across the right / bottom edge."""
    g = make_neighbor_case(0x4E42A400 + pad, n=120, width=200, height=150, half_patch=6, pad=pad)
    rows, cols = g["img_ref"].shape
    pr = g["keys_ref"].copy()
    pc = g["pt_predict_un"].copy()
    pr[:4] = [(cols - 6, 40), (60, rows - 6), (cols - 6, rows - 6), (2.5, 1.25)]
    pc[:4] = [(cols - 6.5, 41), (61, rows - 5.5), (cols - 5, rows - 5), (0.0, 0.0)]
    for aff in (g["affine"], None):
        got = ctx.ncc_free(g["img_ref"], g["img_cur"], 6, pr, pc, aff)
        want = np.array([orc.ncc_free(g["img_ref"], g["img_cur"], 6, pr[i], pc[i], None if aff is None else aff[i])
                         for i in range(120)], np.float32)
        assert np.array_equal(got, want, equal_nan=True), np.flatnonzero(got != want)


def test_device_entry_point_on_resident_frames(ctx, orc):
    """pagk_near_neighbors_device: frames already in slots (as after tracking), device arrays, asynchronous."""
    import torch
    g = make_neighbor_case(0x4E42A500, n=200, width=640, height=480, half_patch=10, clutter=300)
    n, m, cap = 200, g["keys_cur"].shape[0], 32
    dev = torch.device("cuda", 0)
    ctx.frame_upload(0, np.ascontiguousarray(g["img_ref"]), 3)
    ctx.frame_upload(1, np.ascontiguousarray(g["img_cur"]), 3)
    t = {k: torch.from_numpy(np.ascontiguousarray(g[k])).to(dev) for k in ("keys_ref", "pt_predict_un", "status", "affine",
                                                                            "keys_cur", "keys_cur_un")}
    cnt = torch.zeros(n, dtype=torch.int32, device=dev)
    idx = torch.full((n, cap), -1, dtype=torch.int32, device=dev)
    dist = torch.zeros((n, cap), dtype=torch.float32, device=dev)
    ncc = torch.zeros((n, cap), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    for level in (1, 2):
        ctx.near_neighbors_device(0, 1, 10, n, t["keys_ref"], t["pt_predict_un"], t["status"], t["affine"], m, t["keys_cur"],
                                  t["keys_cur_un"], level, 20.0, True, cap, cnt, idx, dist, ncc)
    ctx.sync()
    got = dict(count=cnt.cpu().numpy(), idx=idx.cpu().numpy(), dist=dist.cpu().numpy(), ncc=ncc.cpu().numpy())
    args = (g["img_ref"], g["img_cur"], 10, g["keys_ref"], g["pt_predict_un"], g["status"], g["affine"], g["keys_cur"],
            g["keys_cur_un"])
    r1 = orc.find_near_neighbors(*args, level=1, cap=cap)
    r2 = orc.find_near_neighbors(*args, level=2, cap=cap, count=r1["count"])
    keep = r1["count"] > 0
    for k in ("idx", "dist", "ncc"):
        r2[k][keep] = r1[k][keep]
    same_lists(got, r2, "device entry point, level 1 then 2")
