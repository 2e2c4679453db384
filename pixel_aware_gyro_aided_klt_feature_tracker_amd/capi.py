"""ctypes binding of include/pagk.h (libpagk_hip.so, built by __graft_entry__.build()).

This is the only way Python reaches the product.  There is no CPU fallback: if the
shared library is missing, or no HIP device can be opened, the calls raise.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# PAGK_LIB: an alternative build of the same library (same-session A/B runs of tools/ab_lib.py; never a CPU path)
LIB_PATH = os.environ.get("PAGK_LIB") or os.path.join(PKG_DIR, "libpagk_hip.so")

PAGK_OK = 0
PAGK_E_ARG = -1
PAGK_E_HIP = -2
PAGK_E_NOMEM = -3
PAGK_E_UNSUPPORTED = -4
PAGK_E_NODEVICE = -5
PAGK_E_NCCL = -6
PAGK_E_CAPACITY = -7


class Image(C.Structure):
    _fields_ = [("data", C.c_void_p), ("width", C.c_int32), ("height", C.c_int32), ("step", C.c_int64)]


class Params(C.Structure):
    _fields_ = [
        ("half_patch", C.c_int32), ("iterations", C.c_int32), ("pyramids", C.c_int32),
        ("has_gyro_predict_initial", C.c_uint8), ("inverse", C.c_uint8),
        ("consider_illumination", C.c_uint8), ("consider_affine", C.c_uint8),
        ("regularization_penalty", C.c_uint8), ("calculate_ncc", C.c_uint8),
        ("predict_method", C.c_uint8), ("solver_variant", C.c_uint8),
        ("lambda_", C.c_float), ("alpha", C.c_float), ("max_distance", C.c_int32),
        ("inv_log_max_dist", C.c_float),
        ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
        ("dist_coef", C.c_float * 5), ("n_dist_coef", C.c_int32),
    ]


class Outputs(C.Structure):
    _fields_ = [("pt_un", C.c_void_p), ("pt_dist", C.c_void_p), ("status", C.c_void_p),
                ("pix_err", C.c_void_p), ("dist_pred", C.c_void_p), ("ncc", C.c_void_p),
                ("iters", C.c_void_p)]


def make_params(*, half_patch=5, iterations=10, pyramids=3, has_gyro=True, illumination=True,
                affine=True, penalty=False, ncc=False, inverse=False, camera=None, predict_method=1,
                solver_variant=0) -> Params:
    """pagk_params_default() (reference call site src/gyro_aided_tracker.cpp:276-282)
    with overrides.  `camera` is a synth.Camera or None."""
    p = Params()
    p.half_patch, p.iterations, p.pyramids = half_patch, iterations, pyramids
    p.has_gyro_predict_initial = int(has_gyro)
    p.inverse = int(inverse)
    p.consider_illumination = int(illumination)
    p.consider_affine = int(affine)
    p.regularization_penalty = int(penalty)
    p.calculate_ncc = int(ncc)
    p.predict_method = int(predict_method)   # 1 PIXEL_AWARE_PREDICTION, 2 SINGLE_HOMOGRAPHY (gyro prediction only)
    p.solver_variant = int(solver_variant)   # Eigen association switches (include/pagk.h); 0 = Eigen 3.3 + SSE2
    p.lambda_, p.alpha, p.max_distance = 1.0, 0.5, 25
    p.inv_log_max_dist = 0.0
    if camera is not None:
        p.fx, p.fy, p.cx, p.cy = camera.fx, camera.fy, camera.cx, camera.cy
        for i, v in enumerate(camera.dist[:5]):
            p.dist_coef[i] = v
        p.n_dist_coef = max(4, len(camera.dist))
    else:
        p.fx = p.fy = 1.0
        p.n_dist_coef = 4
    return p


def image_view(a: np.ndarray) -> Image:
    assert a.dtype == np.uint8 and a.ndim == 2 and a.strides[1] == 1
    return Image(a.ctypes.data, a.shape[1], a.shape[0], a.strides[0])


def alloc_outputs(n: int, with_iters: bool = True) -> dict:
    nn = max(n, 1)
    d = dict(pt_un=np.zeros((nn, 2), np.float32), pt_dist=np.zeros((nn, 2), np.float32),
             status=np.zeros(nn, np.uint8), pix_err=np.zeros(nn, np.float64),
             dist_pred=np.zeros(nn, np.float64), ncc=np.zeros(nn, np.float32))
    if with_iters:
        d["iters"] = np.zeros(nn, np.int32)
    return d


def outputs_struct(d: dict) -> Outputs:
    o = Outputs()
    for k in ("pt_un", "pt_dist", "status", "pix_err", "dist_pred", "ncc", "iters"):
        v = d.get(k)
        if v is None:
            setattr(o, k, None)
        elif isinstance(v, np.ndarray):
            setattr(o, k, v.ctypes.data)
        else:  # torch tensor (device pointer)
            setattr(o, k, v.data_ptr())
    return o


def _ptr(a):
    """Raw address of a numpy array or a torch tensor.  The C ABI takes dense row-major arrays: anything
    else (a transposed view, a column-major array) is refused here rather than silently mis-read."""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        if not a.flags.c_contiguous:
            raise ValueError("array must be C-contiguous")
        return a.ctypes.data
    if not a.is_contiguous():
        raise ValueError("tensor must be contiguous")
    return a.data_ptr()


_P = C.POINTER
_lib = None


def declare(lib) -> None:
    """argtypes/restype for every symbol include/pagk.h declares."""
    vp, i32 = C.c_void_p, C.c_int32
    lib.pagk_version.restype = C.c_int
    lib.pagk_version.argtypes = []
    lib.pagk_strerror.restype = C.c_char_p
    lib.pagk_strerror.argtypes = [C.c_int]
    lib.pagk_last_error.restype = C.c_char_p
    lib.pagk_last_error.argtypes = [vp]
    lib.pagk_params_default.restype = None
    lib.pagk_params_default.argtypes = [_P(Params)]
    lib.pagk_inv_log_max_dist.restype = C.c_float
    lib.pagk_inv_log_max_dist.argtypes = [C.c_float, i32]
    lib.pagk_create.restype = C.c_int
    lib.pagk_create.argtypes = [_P(vp), C.c_int]
    lib.pagk_destroy.restype = None
    lib.pagk_destroy.argtypes = [vp]
    lib.pagk_track.restype = C.c_int
    lib.pagk_track.argtypes = [vp, _P(Params), _P(Image), _P(Image), i32, vp, vp, vp, vp, _P(Outputs)]
    lib.pagk_track_pyr.restype = C.c_int
    lib.pagk_track_pyr.argtypes = [vp, _P(Params), i32, _P(Image), _P(Image), i32, vp, vp, vp, vp, _P(Outputs)]
    lib.pagk_frame_upload.restype = C.c_int
    lib.pagk_frame_upload.argtypes = [vp, i32, _P(Image), i32]
    if hasattr(lib, "pagk_frame_upload_pinned"):   # (absent from older builds that tools/ab_lib.py loads for A/B runs)
        lib.pagk_frame_upload_pinned.restype = C.c_int
        lib.pagk_frame_upload_pinned.argtypes = [vp, i32, _P(Image), i32]
    lib.pagk_frame_set_device.restype = C.c_int
    lib.pagk_frame_set_device.argtypes = [vp, i32, vp, i32, i32, C.c_int64, i32]
    lib.pagk_frame_download_level.restype = C.c_int
    lib.pagk_frame_download_level.argtypes = [vp, i32, i32, vp, _P(i32), _P(i32)]
    lib.pagk_track_device.restype = C.c_int
    lib.pagk_track_device.argtypes = [vp, _P(Params), i32, i32, i32, vp, vp, vp, vp, _P(Outputs)]
    if hasattr(lib, "pagk_frame_set_device_batch"):
        lib.pagk_frame_set_device_batch.restype = C.c_int
        lib.pagk_frame_set_device_batch.argtypes = [vp, i32, vp, vp, vp, vp, vp, i32]
    if hasattr(lib, "pagk_track_device_batch"):
        lib.pagk_track_device_batch.restype = C.c_int
        lib.pagk_track_device_batch.argtypes = [vp, i32, _P(Params), vp, vp, vp, vp, vp, vp, vp, vp]
    lib.pagk_track_device_fused.restype = C.c_int
    lib.pagk_track_device_fused.argtypes = [vp, _P(Params), i32, i32, i32, vp, vp, vp, vp, _P(Outputs), i32, vp, i32, i32,
                                            C.c_int64, i32]
    lib.pagk_sync.restype = C.c_int
    lib.pagk_sync.argtypes = [vp]
    lib.pagk_set_stream.restype = C.c_int
    lib.pagk_set_stream.argtypes = [vp, vp]
    lib.pagk_set_kernel.restype = C.c_int
    lib.pagk_set_kernel.argtypes = [vp, i32]
    lib.pagk_last_variant.restype = C.c_int
    lib.pagk_last_variant.argtypes = [vp]
    lib.pagk_set_concurrency.restype = C.c_int
    lib.pagk_set_concurrency.argtypes = [vp, i32]
    lib.pagk_last_handover.restype = C.c_int
    lib.pagk_last_handover.argtypes = [vp]
    lib.pagk_last_kernel_ms.restype = C.c_int
    lib.pagk_last_kernel_ms.argtypes = [vp, _P(C.c_float), _P(C.c_float)]
    lib.pagk_gyro_predict_device.restype = C.c_int
    lib.pagk_gyro_predict_device.argtypes = [vp, _P(Params), i32, i32, vp, vp, i32, vp, vp, vp, vp, vp]
    lib.pagk_gyro_predict_device_rot.restype = C.c_int
    lib.pagk_gyro_predict_device_rot.argtypes = [vp, _P(Params), i32, i32, vp, i32, vp, vp, vp, vp, vp]
    lib.pagk_post_filter.restype = C.c_int
    lib.pagk_post_filter.argtypes = [i32, i32, vp, vp, vp, vp, vp, vp, vp, vp]
    f32 = C.c_float
    for name in ("pagk_graph_begin",):
        getattr(lib, name).restype = C.c_int
        getattr(lib, name).argtypes = [vp]
    lib.pagk_graph_end.restype = C.c_int
    lib.pagk_graph_end.argtypes = [vp, _P(i32)]
    lib.pagk_graph_launch.restype = C.c_int
    lib.pagk_graph_launch.argtypes = [vp, i32]
    lib.pagk_graph_destroy.restype = C.c_int
    lib.pagk_graph_destroy.argtypes = [vp, i32]
    lib.pagk_geometry_scores_device.restype = C.c_int
    lib.pagk_geometry_scores_device.argtypes = [vp, vp, vp, vp, i32, vp, vp, f32, vp, vp, vp]
    lib.pagk_geometry_scores.restype = C.c_int
    lib.pagk_geometry_scores.argtypes = [vp, vp, vp, vp, i32, vp, vp, f32, vp, vp, _P(f32), _P(f32)]
    lib.pagk_geometry_select.restype = C.c_int
    lib.pagk_geometry_select.argtypes = [f32, f32]
    lib.pagk_geometry_validation.restype = C.c_int
    lib.pagk_geometry_validation.argtypes = [vp, vp, vp, vp, i32, vp, vp, vp, f32, _P(f32)]
    lib.pagk_near_neighbors_device.restype = C.c_int
    lib.pagk_near_neighbors_device.argtypes = [vp, i32, i32, i32, i32, vp, vp, vp, vp, i32, vp, vp, i32, f32, i32, i32,
                                               vp, vp, vp, vp]
    lib.pagk_find_near_neighbors.restype = C.c_int
    lib.pagk_find_near_neighbors.argtypes = [vp, _P(Image), _P(Image), i32, i32, vp, vp, vp, vp, i32, vp, vp, i32, f32,
                                             i32, i32, vp, vp, vp, vp]
    lib.pagk_ncc_free.restype = C.c_int
    lib.pagk_ncc_free.argtypes = [vp, _P(Image), _P(Image), i32, i32, vp, vp, vp, vp]
    if hasattr(lib, "pagk_selftest_divide"):   # (absent from older builds that tools/ab_lib.py loads for A/B runs)
        lib.pagk_selftest_divide.restype = C.c_int
        lib.pagk_selftest_divide.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp]
        lib.pagk_selftest_solve.restype = C.c_int
        lib.pagk_selftest_solve.argtypes = [vp, i32, vp, vp, C.c_uint32, vp, vp, vp, vp]
    if hasattr(lib, "pagk_priority_threshold"):
        lib.pagk_priority_threshold.restype = C.c_int
        lib.pagk_priority_threshold.argtypes = [vp]
    if hasattr(lib, "pagk_check_launch"):
        lib.pagk_check_launch.restype = C.c_int
        lib.pagk_check_launch.argtypes = [vp]
    if hasattr(lib, "pagk_selftest_repeat_sum"):
        lib.pagk_selftest_repeat_sum.restype = C.c_int
        lib.pagk_selftest_repeat_sum.argtypes = [vp, i32, vp, i32, vp, vp]
    lib.pagk_match_features.restype = C.c_int
    lib.pagk_match_features.argtypes = [i32, i32, vp, vp, vp, vp, i32, vp, vp, vp, vp]
    # the sharded path
    lib.pagk_multi_create.restype = C.c_int
    lib.pagk_multi_create.argtypes = [_P(vp), _P(i32), i32]
    lib.pagk_multi_unique_id.restype = C.c_int
    lib.pagk_multi_unique_id.argtypes = [vp]
    lib.pagk_multi_create_rank.restype = C.c_int
    lib.pagk_multi_create_rank.argtypes = [_P(vp), vp, i32, i32, i32]
    lib.pagk_multi_destroy.restype = None
    lib.pagk_multi_destroy.argtypes = [vp]
    lib.pagk_multi_world.restype = i32
    lib.pagk_multi_world.argtypes = [vp]
    if hasattr(lib, "pagk_multi_comm_count"):
        lib.pagk_multi_comm_count.restype = i32
        lib.pagk_multi_comm_count.argtypes = [vp]
    lib.pagk_multi_local.restype = i32
    lib.pagk_multi_local.argtypes = [vp]
    lib.pagk_multi_ctx.restype = vp
    lib.pagk_multi_ctx.argtypes = [vp, i32]
    lib.pagk_multi_last_error.restype = C.c_char_p
    lib.pagk_multi_last_error.argtypes = [vp]
    lib.pagk_shard_range.restype = None
    lib.pagk_shard_range.argtypes = [i32, i32, i32, _P(i32), _P(i32)]
    lib.pagk_shard_layout.restype = C.c_size_t
    lib.pagk_shard_layout.argtypes = [i32, _P(C.c_size_t)]
    lib.pagk_multi_allgather.restype = C.c_int
    lib.pagk_multi_allgather.argtypes = [vp, _P(vp), _P(vp), C.c_size_t, _P(vp)]
    lib.pagk_track_sharded.restype = C.c_int
    lib.pagk_track_sharded.argtypes = [vp, _P(Params), _P(Image), _P(Image), i32, vp, vp, vp, vp, _P(Outputs)]


EXPORTED_SYMBOLS = [
    "pagk_version", "pagk_strerror", "pagk_last_error", "pagk_params_default", "pagk_inv_log_max_dist",
    "pagk_create", "pagk_destroy", "pagk_track", "pagk_track_pyr", "pagk_frame_upload", "pagk_frame_upload_pinned",
    "pagk_frame_set_device", "pagk_frame_download_level", "pagk_track_device", "pagk_track_device_fused", "pagk_sync",
    "pagk_set_stream", "pagk_set_kernel", "pagk_last_variant", "pagk_set_concurrency", "pagk_last_handover", "pagk_last_kernel_ms", "pagk_post_filter", "pagk_gyro_predict_device",
    "pagk_gyro_predict_device_rot",
    "pagk_geometry_scores_device", "pagk_geometry_scores", "pagk_geometry_select", "pagk_geometry_validation",
    "pagk_graph_begin", "pagk_graph_end", "pagk_graph_launch", "pagk_graph_destroy",
    "pagk_near_neighbors_device", "pagk_find_near_neighbors", "pagk_ncc_free", "pagk_match_features",
    "pagk_multi_create", "pagk_multi_unique_id", "pagk_multi_create_rank", "pagk_multi_destroy", "pagk_multi_world",
    "pagk_multi_local", "pagk_multi_ctx", "pagk_multi_last_error", "pagk_shard_range", "pagk_shard_layout",
    "pagk_multi_allgather", "pagk_track_sharded", "pagk_selftest_divide", "pagk_selftest_solve",
    "pagk_selftest_repeat_sum", "pagk_check_launch", "pagk_track_device_batch", "pagk_frame_set_device_batch", "pagk_priority_threshold",
    "pagk_multi_comm_count", "pagk_has_variant",
]


def load():
    """Load libpagk_hip.so.  Raises if it has not been built: there is no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). The HIP path is the product; there is no CPU fallback.")
        lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        declare(lib)
        _lib = lib
    return _lib


def has_variant(which: int) -> bool:
    """Can pagk_set_kernel select `which` in the library that is loaded?  (Variants 2 and 6 need -DPAGK_ALL_VARIANTS.)"""
    lib = load()
    if not hasattr(lib, "pagk_has_variant"):
        return True   # (an older build, loaded for an A/B run)
    lib.pagk_has_variant.restype = C.c_int
    lib.pagk_has_variant.argtypes = [C.c_int32]
    return bool(lib.pagk_has_variant(int(which)))


class PagkError(RuntimeError):
    def __init__(self, code: int, where: str, detail: str = ""):
        self.code = code
        msg = load().pagk_strerror(code).decode()
        super().__init__(f"{where}: {msg} ({code}){': ' + detail if detail else ''}")


class Context:
    """pagk_ctx owner.  One per GPU / host thread."""

    def __init__(self, device: int = 0, _borrowed=None):
        self.lib = load()
        self._owned = _borrowed is None
        if _borrowed is not None:          # a member context of a Multi group: the group destroys it
            self.h = C.c_void_p(_borrowed)
            return
        h = C.c_void_p()
        rc = self.lib.pagk_create(C.byref(h), device)
        if rc != PAGK_OK:
            raise PagkError(rc, "pagk_create")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            if self._owned:
                self.lib.pagk_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int, where: str):
        if rc != PAGK_OK:
            raise PagkError(rc, where, self.lib.pagk_last_error(self.h).decode())

    # host-buffer path ------------------------------------------------------------------
    def track(self, params: Params, img_ref, img_cur, pt_ref, pt_init, affine, status_in, out: dict | None = None):
        n = int(pt_ref.shape[0])
        out = out if out is not None else alloc_outputs(n)
        ir, ic = image_view(img_ref), image_view(img_cur)
        o = outputs_struct(out)
        self._check(self.lib.pagk_track(self.h, C.byref(params), C.byref(ir), C.byref(ic), n, _ptr(pt_ref),
                                        _ptr(pt_init), _ptr(affine), _ptr(status_in), C.byref(o)), "pagk_track")
        return out

    def track_pyr(self, params: Params, ref_levels, cur_levels, pt_ref, pt_init, affine, status_in, out=None):
        n = int(pt_ref.shape[0])
        out = out if out is not None else alloc_outputs(n)
        L = len(ref_levels)
        r = (Image * L)(*[image_view(a) for a in ref_levels])
        c = (Image * L)(*[image_view(a) for a in cur_levels])
        o = outputs_struct(out)
        self._check(self.lib.pagk_track_pyr(self.h, C.byref(params), L, r, c, n, _ptr(pt_ref), _ptr(pt_init),
                                            _ptr(affine), _ptr(status_in), C.byref(o)), "pagk_track_pyr")
        return out

    # device-resident path --------------------------------------------------------------
    def frame_upload(self, slot: int, img: np.ndarray, pyramids: int):
        iv = image_view(img)
        self._check(self.lib.pagk_frame_upload(self.h, slot, C.byref(iv), pyramids), "pagk_frame_upload")

    def frame_upload_pinned(self, slot: int, host_ptr: int, width: int, height: int, step: int, pyramids: int):
        """pagk_frame_upload_pinned: asynchronous / capturable upload of a frame that lives in pinned host memory."""
        iv = Image(host_ptr, width, height, step)
        self._check(self.lib.pagk_frame_upload_pinned(self.h, slot, C.byref(iv), pyramids), "pagk_frame_upload_pinned")

    def frame_set_device(self, slot: int, d_ptr: int, width: int, height: int, step: int, pyramids: int):
        self._check(self.lib.pagk_frame_set_device(self.h, slot, d_ptr, width, height, step, pyramids),
                    "pagk_frame_set_device")

    def frame_download_level(self, slot: int, level: int, width: int, height: int) -> np.ndarray:
        w, h = C.c_int32(0), C.c_int32(0)
        buf = np.zeros((height >> level, width >> level), np.uint8)
        self._check(self.lib.pagk_frame_download_level(self.h, slot, level, buf.ctypes.data, C.byref(w), C.byref(h)),
                    "pagk_frame_download_level")
        assert (h.value, w.value) == buf.shape
        return buf

    def track_device(self, params: Params, slot_ref: int, slot_cur: int, n: int, d_pt_ref, d_pt_init, d_affine,
                     d_status, d_out: dict):
        o = outputs_struct(d_out)
        self._check(self.lib.pagk_track_device(self.h, C.byref(params), slot_ref, slot_cur, n, _ptr(d_pt_ref),
                                               _ptr(d_pt_init), _ptr(d_affine), _ptr(d_status), C.byref(o)),
                    "pagk_track_device")

    @staticmethod
    def frame_set_device_batch(ctxs, slots, d_ptrs, widths, heights, steps, pyramids: int):
        """pagk_frame_set_device_batch: the pyramids of k contexts' frames (device images) as ONE launch on ctxs[0]'s
        stream; per frame the same bytes as frame_set_device."""
        k = len(ctxs)
        lib = ctxs[0].lib
        hs = (C.c_void_p * k)(*[c.h for c in ctxs])
        sl = (C.c_int32 * k)(*slots)
        pp = (C.c_void_p * k)(*[int(p) for p in d_ptrs])
        ww, hh = (C.c_int32 * k)(*widths), (C.c_int32 * k)(*heights)
        st = (C.c_int64 * k)(*steps)
        ctxs[0]._check(lib.pagk_frame_set_device_batch(hs, k, sl, pp, ww, hh, st, pyramids), "pagk_frame_set_device_batch")

    @staticmethod
    def track_device_batch(ctxs, params: Params, slots_ref, slots_cur, ns, d_pt_ref, d_pt_init, d_affine, d_status, d_outs):
        """pagk_track_device_batch: k camera streams (contexts `ctxs`, all on one device) as ONE launch, issued on
        ctxs[0]'s stream.  Per stream j: frame slots, feature count and device arrays (torch tensors / addresses) as for
        track_device; d_pt_init / d_affine may be None."""
        k = len(ctxs)
        lib = ctxs[0].lib
        hs = (C.c_void_p * k)(*[c.h for c in ctxs])
        sr, sc, nn = (C.c_int32 * k)(*slots_ref), (C.c_int32 * k)(*slots_cur), (C.c_int32 * k)(*ns)

        def ptrs(lst):
            return None if lst is None else (C.c_void_p * k)(*[_ptr(t) for t in lst])
        outs = (Outputs * k)(*[outputs_struct(o) for o in d_outs])
        rc = lib.pagk_track_device_batch(hs, k, C.byref(params), sr, sc, nn, ptrs(d_pt_ref), ptrs(d_pt_init), ptrs(d_affine),
                                         ptrs(d_status), outs)
        ctxs[0]._check(rc, "pagk_track_device_batch")

    def track_device_fused(self, params: Params, slot_ref: int, slot_cur: int, n: int, d_pt_ref, d_pt_init, d_affine,
                           d_status, d_out: dict, slot_next: int, d_next_ptr: int, width: int, height: int, step: int,
                           pyramids: int):
        """pagk_track_device + the pyramid of another frame into slot_next, one launch when possible."""
        o = outputs_struct(d_out)
        self._check(self.lib.pagk_track_device_fused(self.h, C.byref(params), slot_ref, slot_cur, n, _ptr(d_pt_ref),
                                                     _ptr(d_pt_init), _ptr(d_affine), _ptr(d_status), C.byref(o),
                                                     slot_next, d_next_ptr, width, height, step, pyramids),
                    "pagk_track_device_fused")

    def gyro_predict_device(self, params: Params, width: int, height: int, KRKinv, r3, n: int, d_pt_ref,
                            d_pt_predict_un, d_pt_predict, d_status, d_affine):
        K = np.ascontiguousarray(KRKinv, np.float32)
        r = np.ascontiguousarray(r3, np.float32)
        self._check(self.lib.pagk_gyro_predict_device(self.h, C.byref(params), width, height, K.ctypes.data,
                                                      r.ctypes.data, n, _ptr(d_pt_ref), _ptr(d_pt_predict_un),
                                                      _ptr(d_pt_predict), _ptr(d_status), _ptr(d_affine)),
                    "pagk_gyro_predict_device")

    @staticmethod
    def _mat3(M):
        M = np.ascontiguousarray(M, np.float64)
        if M.size != 9:
            raise ValueError("expected a 3x3 matrix")
        return M

    def geometry_scores_device(self, H21, H12, F21, n: int, d_pts1, d_pts2, sigma: float, d_inl_H, d_inl_F,
                               d_scores):
        """CheckHomography / CheckFundamental scoring loops on device arrays (asynchronous)."""
        H21, H12, F21 = self._mat3(H21), self._mat3(H12), self._mat3(F21)
        self._check(self.lib.pagk_geometry_scores_device(self.h, H21.ctypes.data, H12.ctypes.data, F21.ctypes.data,
                                                         n, _ptr(d_pts1), _ptr(d_pts2), sigma, _ptr(d_inl_H),
                                                         _ptr(d_inl_F), _ptr(d_scores)),
                    "pagk_geometry_scores_device")

    def geometry_scores(self, H21, H12, F21, pts1, pts2, sigma: float = 1.0):
        """Host buffers -> (inliers_H, inliers_F, score_H, score_F); reference
        src/gyro_aided_tracker.cpp:620-676, 704-768."""
        H21, H12, F21 = self._mat3(H21), self._mat3(H12), self._mat3(F21)
        pts1 = np.ascontiguousarray(pts1, np.float32).reshape(-1, 2)
        pts2 = np.ascontiguousarray(pts2, np.float32).reshape(-1, 2)
        n = pts1.shape[0]
        if pts2.shape[0] != n:
            raise ValueError("pts1 / pts2 differ in length")
        inH, inF = np.zeros(max(n, 1), np.uint8), np.zeros(max(n, 1), np.uint8)
        sH, sF = C.c_float(0), C.c_float(0)
        self._check(self.lib.pagk_geometry_scores(self.h, H21.ctypes.data, H12.ctypes.data, F21.ctypes.data, n,
                                                  _ptr(pts1), _ptr(pts2), sigma, _ptr(inH), _ptr(inF),
                                                  C.byref(sH), C.byref(sF)), "pagk_geometry_scores")
        return inH[:n], inF[:n], np.float32(sH.value), np.float32(sF.value)

    def geometry_validation(self, H21, H12, F21, pt_ref_un, pt_predict_un, status, sigma: float = 1.0):
        """GyroAidedTracker::GeometryValidation around externally fitted models
        (src/gyro_aided_tracker.cpp:429-480) -> (cnt_inlier, status, track_score)."""
        H21, H12, F21 = self._mat3(H21), self._mat3(H12), self._mat3(F21)
        p1 = np.ascontiguousarray(pt_ref_un, np.float32).reshape(-1, 2)
        p2 = np.ascontiguousarray(pt_predict_un, np.float32).reshape(-1, 2)
        st = np.array(status, np.uint8, copy=True)
        n = st.shape[0]
        ts = C.c_float(0)
        rc = self.lib.pagk_geometry_validation(self.h, H21.ctypes.data, H12.ctypes.data, F21.ctypes.data, n,
                                               _ptr(p1), _ptr(p2), _ptr(st), sigma, C.byref(ts))
        if rc < 0:
            self._check(rc, "pagk_geometry_validation")
        return rc, st, np.float32(ts.value)

    # NCC nearest-neighbour matching ------------------------------
    def find_near_neighbors(self, img_ref, img_cur, half_patch, keys_ref, pt_predict_un, status, affine, keys_cur,
                            keys_cur_un, level=1, radius_unit=None, use_ncc=True, cap=64, count=None):
        """FindAndSortNearNeighbor (reference src/gyro_aided_tracker.cpp:788-851), host buffers ->
        dict(count, idx, dist, ncc, rc); rc is PAGK_OK or PAGK_E_CAPACITY (count then holds the sizes needed)."""
        n, m = int(keys_ref.shape[0]), int(keys_cur.shape[0])
        ir, ic = image_view(img_ref), image_view(img_cur)
        count = np.zeros(max(n, 1), np.int32) if count is None else np.array(count, np.int32, copy=True)
        idx = np.full((max(n, 1), cap), -1, np.int32)
        dist = np.zeros((max(n, 1), cap), np.float32)
        ncc = np.zeros((max(n, 1), cap), np.float32)
        ru = float(2 * half_patch) if radius_unit is None else float(radius_unit)
        rc = self.lib.pagk_find_near_neighbors(self.h, C.byref(ir), C.byref(ic), half_patch, n, _ptr(keys_ref),
                                               _ptr(pt_predict_un), _ptr(status), _ptr(affine), m, _ptr(keys_cur),
                                               _ptr(keys_cur_un), level, ru, int(use_ncc), cap, _ptr(count), _ptr(idx),
                                               _ptr(dist), _ptr(ncc))
        if rc not in (PAGK_OK, PAGK_E_CAPACITY):
            self._check(rc, "pagk_find_near_neighbors")
        return dict(count=count[:n], idx=idx[:n], dist=dist[:n], ncc=ncc[:n], rc=rc)

    def near_neighbors_device(self, slot_ref, slot_cur, half_patch, n, d_keys_ref, d_pt_predict_un, d_status, d_affine,
                              m, d_keys_cur, d_keys_cur_un, level, radius_unit, use_ncc, cap, d_count, d_idx, d_dist,
                              d_ncc):
        self._check(self.lib.pagk_near_neighbors_device(self.h, slot_ref, slot_cur, half_patch, n, _ptr(d_keys_ref),
                                                        _ptr(d_pt_predict_un), _ptr(d_status), _ptr(d_affine), m,
                                                        _ptr(d_keys_cur), _ptr(d_keys_cur_un), level, float(radius_unit),
                                                        int(use_ncc), cap, _ptr(d_count), _ptr(d_idx), _ptr(d_dist),
                                                        _ptr(d_ncc)), "pagk_near_neighbors_device")

    def ncc_free(self, img_ref, img_cur, half_patch, pt_ref, pt_cur, affine=None) -> np.ndarray:
        """The free NCC of reference src/utils.cpp:166-200 for n point pairs (host buffers)."""
        n = int(pt_ref.shape[0])
        ir, ic = image_view(img_ref), image_view(img_cur)
        out = np.zeros(max(n, 1), np.float32)
        self._check(self.lib.pagk_ncc_free(self.h, C.byref(ir), C.byref(ic), half_patch, n, _ptr(pt_ref), _ptr(pt_cur),
                                           _ptr(affine), _ptr(out)), "pagk_ncc_free")
        return out[:n]

    def selftest_divide(self, num: np.ndarray, den: np.ndarray):
        """(num / den, the same through the prepared-denominator form, sqrt(num), sqrt(num) through the solve's lean form)
        computed on the device."""
        num, den = np.ascontiguousarray(num, np.float64), np.ascontiguousarray(den, np.float64)
        n = int(num.shape[0])
        qp, qq, rt, rl = (np.zeros(max(n, 1), np.float64) for _ in range(4))
        self._check(self.lib.pagk_selftest_divide(self.h, n, _ptr(num), _ptr(den), _ptr(qp), _ptr(qq), _ptr(rt), _ptr(rl)),
                    "pagk_selftest_divide")
        return qp[:n], qq[:n], rt[:n], rl[:n]

    def selftest_repeat_sum(self, c: np.ndarray, count: int):
        """(closed form, loop) of the ordered sum of `count` copies of c*c on the device (pagk_selftest_repeat_sum)."""
        c = np.ascontiguousarray(c, dtype=np.float32)
        n = c.shape[0]
        closed, loop = np.empty(n, np.float64), np.empty(n, np.float64)
        self._check(self.lib.pagk_selftest_repeat_sum(self.h, n, _ptr(c), int(count), _ptr(closed), _ptr(loop)),
                    "pagk_selftest_repeat_sum")
        return closed, loop

    def selftest_solve(self, H: np.ndarray, b: np.ndarray, solver_variant: int = 0):
        """H.llt().solve(b) and the update's norm for n 4x4 systems: (x, norm) of the one-lane form and
        (x, squared norm) of the four-lane form."""
        H, b = np.ascontiguousarray(H, np.float64), np.ascontiguousarray(b, np.float64)
        n = int(H.shape[0])
        xs, xl = np.zeros((max(n, 1), 4)), np.zeros((max(n, 1), 4))
        ns, nl = np.zeros(max(n, 1)), np.zeros(max(n, 1))
        self._check(self.lib.pagk_selftest_solve(self.h, n, _ptr(H), _ptr(b), int(solver_variant), _ptr(xs), _ptr(ns),
                                                 _ptr(xl), _ptr(nl)), "pagk_selftest_solve")
        return xs[:n], ns[:n], xl[:n], nl[:n]

    def gyro_predict_device_rot(self, params: Params, width: int, height: int, d_rot, n: int, d_pt_ref,
                                d_pt_predict_un, d_pt_predict, d_status, d_affine):
        """Prediction with the rotation (KRKinv rows 0-1, r3: 9 floats) in device memory: capturable per frame."""
        self._check(self.lib.pagk_gyro_predict_device_rot(self.h, C.byref(params), width, height, _ptr(d_rot), n,
                                                          _ptr(d_pt_ref), _ptr(d_pt_predict_un), _ptr(d_pt_predict),
                                                          _ptr(d_status), _ptr(d_affine)),
                    "pagk_gyro_predict_device_rot")

    # hipGraph capture of the *_device calls issued on the context stream --------------------
    def graph_begin(self):
        self._check(self.lib.pagk_graph_begin(self.h), "pagk_graph_begin")

    def graph_end(self) -> int:
        gid = C.c_int32(-1)
        self._check(self.lib.pagk_graph_end(self.h, C.byref(gid)), "pagk_graph_end")
        return gid.value

    def graph_launch(self, graph_id: int):
        self._check(self.lib.pagk_graph_launch(self.h, graph_id), "pagk_graph_launch")

    def graph_destroy(self, graph_id: int):
        self._check(self.lib.pagk_graph_destroy(self.h, graph_id), "pagk_graph_destroy")

    def sync(self):
        self._check(self.lib.pagk_sync(self.h), "pagk_sync")

    def check_launch(self):
        """The error state pagk_sync would return, without synchronising: for callers that synchronise the stream
        themselves (a torch stream).  Raises PAGK_E_HIP once for a kernel-7 launch in which a wave gave up its wait."""
        self._check(self.lib.pagk_check_launch(self.h), "pagk_check_launch")

    def set_stream(self, stream_ptr: int | None):
        self._check(self.lib.pagk_set_stream(self.h, stream_ptr), "pagk_set_stream")

    def set_kernel(self, which: int):
        self._check(self.lib.pagk_set_kernel(self.h, which), "pagk_set_kernel")

    VARIANT_NAMES = {0: "4-wave workgroup per feature, DPP row chains", 1: "one thread per feature (cross-check)",
                     2: "2-wave workgroup per feature, f64 MFMA chain", 3: "one wave per feature, f64 MFMA chain",
                     4: "relaxed order (experiment)", 5: "four features per wave, f64 MFMA blocks",
                     6: "four independent rows per wave + work queue",
                     7: "four features per wave, one pyramid level per wave"}

    def priority_threshold(self) -> int:
        """K of the next 4-wave launch's issue priorities (csrc/pagk_prio.h); synchronises."""
        rc = int(self.lib.pagk_priority_threshold(self.h))
        if rc < 0:
            self._check(rc, "pagk_priority_threshold")
        return rc

    def last_variant(self) -> int:
        return int(self.lib.pagk_last_variant(self.h))

    def last_handover(self) -> int:
        """Features the last launch handed from the throughput kernel to the latency kernel (0: hand-over not used)."""
        rc = int(self.lib.pagk_last_handover(self.h))
        if rc < 0:
            self._check(rc, "pagk_last_handover")
        return rc

    def set_concurrency(self, streams: int):
        """`streams` contexts like this one run at the same time on the device: the automatic variant thresholds are
        applied to streams * n (pagk_set_concurrency)."""
        self._check(self.lib.pagk_set_concurrency(self.h, streams), "pagk_set_concurrency")

    def last_kernel_ms(self):
        t, p = C.c_float(0), C.c_float(0)
        self._check(self.lib.pagk_last_kernel_ms(self.h, C.byref(t), C.byref(p)), "pagk_last_kernel_ms")
        return t.value, p.value


def post_filter(half_patch: int, status_pm, pix_err, dist_pred, pt_pm, pt_pm_un):
    """pagk_post_filter: the tracker-side inlier mask (reference src/gyro_aided_tracker.cpp:289-341)."""
    lib = load()
    n = int(status_pm.shape[0])
    status = np.zeros(max(n, 1), np.uint8)
    pp = np.zeros((max(n, 1), 2), np.float32)
    ppu = np.zeros((max(n, 1), 2), np.float32)
    rc = lib.pagk_post_filter(n, half_patch, _ptr(status_pm), _ptr(pix_err), _ptr(dist_pred), _ptr(pt_pm),
                              _ptr(pt_pm_un), _ptr(status), _ptr(pp), _ptr(ppu))
    if rc < 0:
        raise PagkError(rc, "pagk_post_filter")
    return rc, status[:n], pp[:n], ppu[:n]


def geometry_select(score_H: float, score_F: float) -> bool:
    """True = homography chosen (RH > 0.45, reference src/gyro_aided_tracker.cpp:462-470)."""
    return bool(load().pagk_geometry_select(float(score_H), float(score_F)))


def match_features(count, idx, dist, ncc, use_ncc=True):
    """pagk_match_features: GyroAidedTracker::MatchFeatures (reference src/gyro_aided_tracker.cpp:949-1008)
    -> (query, train, dist, ncc) of the accepted matches."""
    n = int(count.shape[0])
    cap = int(idx.shape[1])
    q, t = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.int32)
    d, c = np.zeros(max(n, 1), np.float32), np.zeros(max(n, 1), np.float32)
    count = np.ascontiguousarray(count, np.int32)
    idx, dist, ncc = np.ascontiguousarray(idx), np.ascontiguousarray(dist), np.ascontiguousarray(ncc)
    k = load().pagk_match_features(n, cap, _ptr(count), _ptr(idx), _ptr(dist), _ptr(ncc), int(use_ncc), _ptr(q), _ptr(t),
                                   _ptr(d), _ptr(c))
    if k < 0:
        raise PagkError(k, "pagk_match_features")
    return q[:k], t[:k], d[:k], c[:k]


def shard_range(n: int, rank: int, world: int) -> tuple[int, int]:
    """pagk_shard_range: the contiguous feature block of a rank."""
    lo, hi = C.c_int32(0), C.c_int32(0)
    load().pagk_shard_range(n, rank, world, C.byref(lo), C.byref(hi))
    return lo.value, hi.value


def shard_layout(m: int) -> tuple[list[int], int]:
    """pagk_shard_layout: byte offsets of the seven SoA blocks of a rank's packed slice, and its size."""
    offs = (C.c_size_t * 7)()
    total = load().pagk_shard_layout(m, offs)
    return [int(v) for v in offs], int(total)


class Multi:
    """pagk_multi owner: the group of GPUs one tracking call is sharded over, and its RCCL communicator.
    Multi(devices=[...]) drives all devices from this process; Multi(uid=..., rank=, world=, device=) joins a
    one-process-per-GPU group (uid from Multi.unique_id() on rank 0, handed over by the application)."""

    def __init__(self, devices=None, *, uid: bytes | None = None, rank: int = 0, world: int = 1, device: int = 0):
        self.lib = load()
        self._lent = []   # Contexts handed out by ctx(): invalidated by close()
        h = C.c_void_p()
        if uid is None:
            if devices is None or len(devices) == 0 or any(int(d) < 0 for d in devices):
                raise ValueError("Multi(devices=[...]) needs a non-empty list of device indices (or uid=, rank=, world=)")
            devs = (C.c_int32 * len(devices))(*devices)
            rc = self.lib.pagk_multi_create(C.byref(h), devs, len(devices))
            where = "pagk_multi_create"
        else:
            if len(uid) != 128:
                raise ValueError("the RCCL unique id is 128 bytes")
            buf = (C.c_uint8 * 128).from_buffer_copy(uid)
            rc = self.lib.pagk_multi_create_rank(C.byref(h), buf, rank, world, device)
            where = "pagk_multi_create_rank"
        if rc != PAGK_OK:
            raise PagkError(rc, where)
        self.h = h
        self.world = self.lib.pagk_multi_world(h)
        self.n_local = self.lib.pagk_multi_local(h)
        self.rank = rank if uid is not None else 0

    def comm_count(self) -> int | None:
        """Ranks of the communicator as RCCL reports them (ncclCommCount); None when the library cannot tell."""
        c = int(self.lib.pagk_multi_comm_count(self.h))
        return c if c >= 0 else None

    @staticmethod
    def unique_id() -> bytes:
        buf = (C.c_uint8 * 128)()
        rc = load().pagk_multi_unique_id(buf)
        if rc != PAGK_OK:
            raise PagkError(rc, "pagk_multi_unique_id")
        return bytes(buf)

    def ctx(self, local_index: int = 0) -> Context:
        if not getattr(self, "h", None):
            raise RuntimeError("this Multi has been closed")
        p = self.lib.pagk_multi_ctx(self.h, local_index)
        if not p:
            raise IndexError(local_index)
        c = Context(_borrowed=p)
        self._lent.append(c)
        return c

    def _check(self, rc: int, where: str):
        if rc != PAGK_OK:
            raise PagkError(rc, where, self.lib.pagk_multi_last_error(self.h).decode())

    def allgather(self, d_send, d_recv, nbytes: int, streams=None):
        """One all-gather of `nbytes` bytes per rank; d_send / d_recv: one device buffer per local member."""
        k = self.n_local
        snd = (C.c_void_p * k)(*[_ptr(a) for a in d_send])
        rcv = (C.c_void_p * k)(*[_ptr(a) for a in d_recv])
        st = None if streams is None else (C.c_void_p * k)(*streams)
        self._check(self.lib.pagk_multi_allgather(self.h, snd, rcv, nbytes, st), "pagk_multi_allgather")

    def track_sharded(self, params: Params, img_ref, img_cur, pt_ref, pt_init, affine, status_in, out: dict | None = None):
        n = int(pt_ref.shape[0])
        out = out if out is not None else alloc_outputs(n)
        ir, ic = image_view(img_ref), image_view(img_cur)
        o = outputs_struct(out)
        self._check(self.lib.pagk_track_sharded(self.h, C.byref(params), C.byref(ir), C.byref(ic), n, _ptr(pt_ref),
                                                _ptr(pt_init), _ptr(affine), _ptr(status_in), C.byref(o)),
                    "pagk_track_sharded")
        return out

    def close(self):
        if getattr(self, "h", None):
            for c in getattr(self, "_lent", []):
                c.h = None    # the group owns its member contexts: a borrowed handle must not outlive it
            self._lent = []
            self.lib.pagk_multi_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
