"""ctypes access to the C++ API shell (libpagk_tracker.so: the reference's GyroAidedTracker /
PatchMatch classes over the C ABI).  Used by the tests to drive TrackFeatures() end to end."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import capi

LIB_PATH = os.path.join(capi.PKG_DIR, "libpagk_tracker.so")
_lib = None


def load():
    global _lib
    if _lib is None:
        capi.load()  # libpagk_hip.so first (RTLD_GLOBAL), the shell links against it
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run __graft_entry__.build()")
        lib = C.CDLL(LIB_PATH)
        vp, i = C.c_void_p, C.c_int
        lib.pagk_tracker_track_features.restype = C.c_int
        lib.pagk_tracker_track_features.argtypes = [vp, vp, i, i, C.c_long, i, vp, vp, vp, i, i, i, i, vp, vp, i,
                                                    C.c_double, C.c_double, vp, vp, vp, vp, vp, vp, vp, vp]
        lib.pagk_tracker_geometry_validation.restype = C.c_int
        lib.pagk_tracker_geometry_validation.argtypes = [i, vp, vp, vp, vp, vp, vp, C.POINTER(C.c_float)]
        lib.pagk_seq_load_keypoints.restype = C.c_int
        lib.pagk_seq_load_keypoints.argtypes = [C.c_char_p, vp, i]
        lib.pagk_seq_load_correspondences.restype = C.c_int
        lib.pagk_seq_load_correspondences.argtypes = [C.c_char_p, vp, vp, i, i]
        lib.pagk_seq_find_time.restype = C.c_int
        lib.pagk_seq_find_time.argtypes = [vp, i, C.c_double]
        lib.pagk_seq_parse_image_line.restype = C.c_int
        lib.pagk_seq_parse_image_line.argtypes = [C.c_char_p, C.POINTER(C.c_double)]
        lib.pagk_seq_load_imu.restype = C.c_int
        lib.pagk_seq_load_imu.argtypes = [C.c_char_p, vp, i]
        lib.pagk_seq_imu_windows.restype = C.c_int
        lib.pagk_seq_imu_windows.argtypes = [C.c_char_p, vp, i, vp, vp]
        lib.pagk_tracker_last_error.restype = C.c_char_p
        lib.pagk_tracker_release.restype = None
        _lib = lib
    return _lib


def track_features(img_ref, img_cur, keys_ref, K, dist, *, type=4, half_patch=5, iterations=10, pyramids=3,
                   Rcl=None, imu=None, t_ref=0.0, t_cur=0.0):
    """GyroAidedTracker(ctor #1) -> TrackFeatures() (reference Examples/Demo/RealSenseD435i.cpp:244-251)."""
    lib = load()
    n = int(keys_ref.shape[0])
    nn = max(n, 1)
    K = np.ascontiguousarray(K, np.float32)
    dist = np.ascontiguousarray(dist, np.float32)
    keys_ref = np.ascontiguousarray(keys_ref, np.float32)
    R = None if Rcl is None else np.ascontiguousarray(Rcl, np.float32)
    imu_a = None if imu is None else np.ascontiguousarray(imu, np.float64)
    out = dict(status=np.zeros(nn, np.uint8), pt_predict_un=np.zeros((nn, 2), np.float32),
               pt_predict=np.zeros((nn, 2), np.float32), status_pm=np.zeros(nn, np.uint8),
               pt_pm_un=np.zeros((nn, 2), np.float32), pix_err=np.zeros(nn, np.float64),
               dist_pred=np.zeros(nn, np.float64), affine=np.zeros((nn, 4), np.float32))
    ret = lib.pagk_tracker_track_features(
        img_ref.ctypes.data, img_cur.ctypes.data, img_ref.shape[1], img_ref.shape[0], img_ref.strides[0], n,
        keys_ref.ctypes.data, K.ctypes.data, dist.ctypes.data, type, half_patch, iterations, pyramids,
        None if R is None else R.ctypes.data, None if imu_a is None else imu_a.ctypes.data,
        0 if imu_a is None else int(imu_a.shape[0]), t_ref, t_cur,
        out["status"].ctypes.data, out["pt_predict_un"].ctypes.data, out["pt_predict"].ctypes.data,
        out["status_pm"].ctypes.data, out["pt_pm_un"].ctypes.data, out["pix_err"].ctypes.data,
        out["dist_pred"].ctypes.data, out["affine"].ctypes.data)
    if ret == -100:
        raise RuntimeError("GyroAidedTracker: " + lib.pagk_tracker_last_error().decode())
    return ret, {k: v[:n] for k, v in out.items()}


def geometry_validation(keys_ref_un, pt_predict_un, status, H21, H12, F21):
    """GyroAidedTracker::GeometryValidation() with an installed model fitter (reference
    src/gyro_aided_tracker.cpp:429-480) -> (cnt_inlier, status, track_score)."""
    lib = load()
    k = np.ascontiguousarray(keys_ref_un, np.float32).reshape(-1, 2)
    q = np.ascontiguousarray(pt_predict_un, np.float32).reshape(-1, 2)
    st = np.array(status, np.uint8, copy=True)
    H21, H12, F21 = (np.ascontiguousarray(M, np.float64) for M in (H21, H12, F21))
    ts = C.c_float(0)
    ret = lib.pagk_tracker_geometry_validation(int(st.shape[0]), k.ctypes.data, q.ctypes.data, st.ctypes.data,
                                               H21.ctypes.data, H12.ctypes.data, F21.ctypes.data, C.byref(ts))
    if ret == -100:
        raise RuntimeError("GyroAidedTracker: " + lib.pagk_tracker_last_error().decode())
    return ret, st, np.float32(ts.value)


# ---- sequence formats of the reference's demo (csrc/host/sequence_io.h) ----
def load_keypoints(path: str) -> np.ndarray:
    """SuperPoint keypoint list "idx, x, y" (reference src/frame.cpp:222-240) -> n x 2 float32."""
    lib = load()
    n = lib.pagk_seq_load_keypoints(path.encode(), None, 0)
    if n < 0:
        raise FileNotFoundError(path)
    xy = np.zeros((max(n, 1), 2), np.float32)
    lib.pagk_seq_load_keypoints(path.encode(), xy.ctypes.data, n)
    return xy[:n]


def load_correspondences(path: str):
    """corresponds.txt "<t_seconds>, <stamp>" (reference Examples/Demo/RealSenseD435i.cpp:167-182)."""
    lib = load()
    n = lib.pagk_seq_load_correspondences(path.encode(), None, None, 64, 0)
    if n < 0:
        raise FileNotFoundError(path)
    times = np.zeros(max(n, 1), np.float64)
    names = C.create_string_buffer(64 * max(n, 1))
    lib.pagk_seq_load_correspondences(path.encode(), times.ctypes.data, C.addressof(names), 64, n)
    raw = names.raw
    return times[:n], [raw[64 * k:64 * k + 64].split(b"\0", 1)[0].decode() for k in range(n)]


def find_time(times: np.ndarray, t: float) -> int:
    """findTimeCorrespondenIndex (reference include/common.h:105-114)."""
    times = np.ascontiguousarray(times, np.float64)
    return load().pagk_seq_find_time(times.ctypes.data, int(times.shape[0]), float(t))


def parse_image_line(line: str):
    """One line of image_file_list.txt -> time in seconds, or None (reference RealSenseD435i.cpp:89-94)."""
    t = C.c_double(0)
    return t.value if load().pagk_seq_parse_image_line(line.encode(), C.byref(t)) else None


def load_imu(path: str) -> np.ndarray:
    """imu.txt "<stamp_ns> ax ay az wx wy wz" (reference RealSenseD435i.cpp:102-141) -> n x 7 (ax..wz, t)."""
    lib = load()
    n = lib.pagk_seq_load_imu(path.encode(), None, 0)
    if n < 0:
        raise FileNotFoundError(path)
    out = np.zeros((max(n, 1), 7), np.float64)
    lib.pagk_seq_load_imu(path.encode(), out.ctypes.data, n)
    return out[:n]


def imu_windows(imu_path: str, frame_times):
    """The demo's streaming IMU window per frame pair (reference RealSenseD435i.cpp:196-218) ->
    (first_index, count) per frame."""
    ft = np.ascontiguousarray(frame_times, np.float64)
    first = np.zeros(max(len(ft), 1), np.int32)
    counts = np.zeros(max(len(ft), 1), np.int32)
    n = load().pagk_seq_imu_windows(imu_path.encode(), ft.ctypes.data, len(ft), first.ctypes.data, counts.ctypes.data)
    if n < 0:
        raise FileNotFoundError(imu_path)
    return first[:len(ft)], counts[:len(ft)]
