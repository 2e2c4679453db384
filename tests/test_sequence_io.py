"""Row f4: the text formats of the reference's demo, parsed by csrc/host/sequence_io.cpp.  The keypoint lists
and corresponds.txt under tests/golden/seq/ are data files taken from the reference's own
Examples/Demo/data/PKUSZ_RealSenseD435i_sequence/SuperPoints.zip (first four frames, first twelve
correspondences); imu.txt / image_file_list.txt are not shipped with the reference (its sequence_1.zip is
listed in .MISSING_LARGE_BLOBS), so those two are written here in the format its readers parse."""
import os

import numpy as np
import pytest

from util import built_variants

from oracle import pagk_oracle as orc
from pixel_aware_gyro_aided_klt_feature_tracker_amd import capi, host_api, synth

SEQ = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "seq")
STAMPS = ["1627889784040685824", "1627889784107402752", "1627889784174119936"]
STAMP4 = "1627889784240837120"   # a fourth list, only used to reach the 2000 keypoints of BASELINE configs[2]


def test_keypoint_lists_of_the_reference(built):
    for s in STAMPS:
        path = os.path.join(SEQ, s + ".txt")
        xy = host_api.load_keypoints(path)
        rows = [ln.split(",") for ln in open(path).read().splitlines()]
        exp = np.array([[float(r[1]), float(r[2])] for r in rows], np.float32)
        assert xy.shape == (500, 2) and np.array_equal(xy, exp)
        assert xy[:, 0].min() >= 0 and xy[:, 0].max() < 640 and xy[:, 1].min() >= 0 and xy[:, 1].max() < 480
    first = host_api.load_keypoints(os.path.join(SEQ, STAMPS[0] + ".txt"))
    assert first[0].tolist() == [455.0, 44.0] and first[499].tolist() == [235.0, 53.0]   # the file's own lines
    with pytest.raises(FileNotFoundError):
        host_api.load_keypoints(os.path.join(SEQ, "missing.txt"))


def test_keypoint_list_tolerates_what_the_reference_tolerates(built, tmp_path):
    p = tmp_path / "k.txt"
    p.write_text("0, 10.5, 20.25\n1,3,4\n\n2, 7\n3, 1e2, -2.5, 9\n")
    xy = host_api.load_keypoints(str(p))
    # atof per comma field; the blank and the two-field line have no point (the reference would index out of range)
    assert xy.tolist() == [[10.5, 20.25], [3.0, 4.0], [100.0, -2.5]]


def test_correspondences_and_time_lookup(built):
    times, names = host_api.load_correspondences(os.path.join(SEQ, "corresponds.txt"))
    assert len(names) == 12 and names[:3] == STAMPS
    assert times[0] == 1627889784.040686 and times[1] == 1627889784.1074028
    for k, s in enumerate(names):   # the two columns are the same instant: seconds vs nanosecond stamp
        assert abs(times[k] - int(s) * 1e-9) < 1e-6
        assert host_api.find_time(times, int(s) * 1e-9) == k          # within 0.1 ms (include/common.h:110)
    assert host_api.find_time(times, times[5] + 0.00009) == 5
    assert host_api.find_time(times, times[5] + 0.02) == -1


def test_image_list_line(built):
    assert host_api.parse_image_line("/cam0/1627889784040685824.png") == 1627889784040685824 * 1e-9
    assert host_api.parse_image_line("1627889784107402752.png") == 1627889784107402752 * 1e-9
    assert host_api.parse_image_line("/cam0/readme.txt") is None
    assert host_api.parse_image_line("/cam0/frame.png") is None


def _write_imu(path, t0_ns, n, rate_hz=200.0):
    rng = np.random.default_rng(4)
    rows = []
    for k in range(n):
        ns = t0_ns + int(round(k * 1e9 / rate_hz))
        v = rng.normal(0, 1, 6)
        rows.append((ns, v))
    with open(path, "w") as f:
        for ns, v in rows:
            f.write(f"{ns} " + " ".join(f"{x:.6f}" for x in v) + "\n")
    return rows


def test_imu_file_and_streaming_window(built, tmp_path):
    t0 = 1627889784000000000
    rows = _write_imu(tmp_path / "imu.txt", t0, 60)
    imu = host_api.load_imu(str(tmp_path / "imu.txt"))
    assert imu.shape == (60, 7)
    for k, (ns, v) in enumerate(rows):
        assert imu[k, 6] == ns * 1e-9
        # members of IMU::Point are float: values narrowed once
        assert np.array_equal(imu[k, :6], np.array([float(f"{x:.6f}") for x in v], np.float64).astype(np.float32))
    # frames every 66.7 ms starting 40 ms in; window of pair (prev, cur) = samples in [first >= prev, < cur)
    ft = np.array([int(s) * 1e-9 for s in STAMPS])
    first, counts = host_api.imu_windows(str(tmp_path / "imu.txt"), ft)
    stamps = imu[:, 6]
    assert counts[0] == 0 and first[0] == -1                       # t_prev == 0: nothing (:208)
    pos = 0
    for k in (1, 2):
        while stamps[pos] < ft[k - 1]:
            pos += 1
        exp_first = pos
        while pos < len(stamps) and stamps[pos] < ft[k]:
            pos += 1
        assert first[k] == exp_first and counts[k] == pos - exp_first and counts[k] in (13, 14)


@pytest.mark.gpu
def test_replay_real_keypoints_through_the_tracker(built):
    # the reference's own SuperPoint lists (integer pixel positions, 640x480) drive the HIP path on a
    # synthetic D435i-like pair; every output bit-identical to the oracle
    w = synth.config(2, n=500, pyramids=3)
    ctx = capi.Context(0)
    try:
        for s in STAMPS:
            kp = host_api.load_keypoints(os.path.join(SEQ, s + ".txt"))
            p = capi.make_params(half_patch=10, iterations=30, pyramids=3, has_gyro=False, camera=synth.D435I)
            st = np.ones(500, np.uint8)
            A = np.tile(np.array([1, 0, 0, 1], np.float32), (500, 1))
            got = ctx.track(p, w.img_ref, w.img_cur, kp, kp, A, st)
            ref = orc.track(p, w.img_ref, w.img_cur, kp, kp, A, st, nthreads=16)
            for k in ("status", "pt_un", "pt_dist", "pix_err", "dist_pred", "iters"):
                assert np.array_equal(got[k][:500], ref[k][:500], equal_nan=True), (s, k)
            assert got["status"][:500].sum() > 300
    finally:
        ctx.close()


@pytest.mark.gpu
def test_baseline_config2_shape_with_superpoint_keypoints(built):
    # BASELINE configs[2]: 640x480, 2000 keypoints "SuperPoint-loaded", 21x21 patch, 4 levels -- the keypoints are
    # the reference's own lists (4 frames x 500), the image pair is the synthetic D435i-like stand-in
    kp = np.concatenate([host_api.load_keypoints(os.path.join(SEQ, s + ".txt")) for s in STAMPS + [STAMP4]])
    assert kp.shape == (2000, 2)
    w = synth.config(2, n=2000)                       # 640x480, L = 4, D435i intrinsics
    p = capi.make_params(half_patch=10, iterations=30, pyramids=4, has_gyro=True, camera=synth.D435I)
    Rp = synth.rodrigues(np.array((0.004, -0.003, 0.006))) @ synth.rodrigues(np.array((0.3, -0.5, 1.0)) * 0.05)
    K32 = synth.D435I.K.astype(np.float32)

    def mul(a, b):
        return (a.astype(np.float64) @ b.astype(np.float64)).astype(np.float32)
    KRK = mul(mul(K32, Rp.astype(np.float32)), np.linalg.inv(K32.astype(np.float64)).astype(np.float32))
    pu, pd, st, A = orc.gyro_predict(p, 640, 480, 10, KRK, Rp.astype(np.float32)[2], kp)
    ctx = capi.Context(0)
    try:
        for kernel in built_variants((0, 2, 3)):
            ctx.set_kernel(kernel)
            got = ctx.track(p, w.img_ref, w.img_cur, kp, pu, A, st)
            ref = orc.track(p, w.img_ref, w.img_cur, kp, pu, A, st, nthreads=16)
            for k in ("status", "pt_un", "pt_dist", "pix_err", "dist_pred", "iters"):
                assert np.array_equal(got[k][:2000], ref[k][:2000], equal_nan=True), (kernel, k)
        assert 1500 < int(st.sum()) <= 2000 and got["status"][:2000].sum() > 1200
    finally:
        ctx.set_kernel(0)
        ctx.close()
