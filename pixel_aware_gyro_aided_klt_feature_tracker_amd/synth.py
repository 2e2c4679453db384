"""Synthetic frame pairs, keypoints and gyro predictions for tests and bench.py.

The reference ships no images (the sequences are missing blobs),
so every workload is generated: an analytic texture sampled under a known warp, which
gives exact sub-pixel ground-truth flow without resampling blur.

Nothing here is on the product path: it only manufactures inputs.
"""
from __future__ import annotations

import dataclasses
import math

import numpy as np

MASK64 = (1 << 64) - 1


class SplitMix64:
    """SplitMix64 (Steele, Lea, Flood 2014); doubles via (x >> 11) * 2^-53."""

    def __init__(self, seed: int):
        self.s = seed & MASK64

    def next_u64(self) -> int:
        self.s = (self.s + 0x9E3779B97F4A7C15) & MASK64
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
        return z ^ (z >> 31)

    def uniform(self, n: int | None = None):
        if n is None:
            return (self.next_u64() >> 11) * (1.0 / (1 << 53))
        return np.array([(self.next_u64() >> 11) * (1.0 / (1 << 53)) for _ in range(n)], dtype=np.float64)


@dataclasses.dataclass
class Camera:
    fx: float
    fy: float
    cx: float
    cy: float
    dist: tuple = (0.0, 0.0, 0.0, 0.0)

    @property
    def K(self) -> np.ndarray:
        return np.array([[self.fx, 0, self.cx], [0, self.fy, self.cy], [0, 0, 1]], dtype=np.float64)


# reference Examples/ROS/ROS_Demo_Feature_Tracking/config/EuRoC.yaml:33-41
EUROC = Camera(458.654, 457.296, 367.215, 248.375, (-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05))
# reference Examples/Demo/RealSenseD435i.yaml:27-35
D435I = Camera(394.5643528049837, 395.2103902700227, 325.2710790421636, 243.20141864231425,
               (-0.0027697209770466296, -0.0007212258451583873, 0.00029903960869777114, 0.0003981049435158156))


def generic_camera(width: int, height: int) -> Camera:
    f = 0.6 * width
    return Camera(f, f, width / 2.0 - 0.5, height / 2.0 - 0.5)


class Texture:
    """T(x,y) = 127.5 + sum_k a_k sin(2 pi (fx_k x + fy_k y) + phi_k), 48 components with
    wavelengths log-uniform in [6, 96] px and a_k = 6 sqrt(lambda_k / 20)."""

    def __init__(self, rng: SplitMix64, n_comp: int = 48):
        lam = 6.0 * (96.0 / 6.0) ** rng.uniform(n_comp)
        ang = 2 * math.pi * rng.uniform(n_comp)
        self.fx = np.cos(ang) / lam
        self.fy = np.sin(ang) / lam
        self.phi = 2 * math.pi * rng.uniform(n_comp)
        self.amp = 6.0 * np.sqrt(lam / 20.0)

    def __call__(self, x: np.ndarray, y: np.ndarray) -> np.ndarray:
        out = np.full(x.shape, 127.5, dtype=np.float64)
        for fx, fy, phi, a in zip(self.fx, self.fy, self.phi, self.amp):
            out += a * np.sin(2 * math.pi * (fx * x + fy * y) + phi)
        return out


def rodrigues(w: np.ndarray) -> np.ndarray:
    th = float(np.linalg.norm(w))
    W = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]], dtype=np.float64)
    if th < 1e-12:
        return np.eye(3) + W
    return np.eye(3) + W * (math.sin(th) / th) + W @ W * ((1 - math.cos(th)) / th ** 2)


def _to_u8(a: np.ndarray) -> np.ndarray:
    return np.clip(np.rint(a), 0, 255).astype(np.uint8)


@dataclasses.dataclass
class Workload:
    """One frame pair plus per-feature inputs, laid out as the C ABI wants them."""

    name: str
    img_ref: np.ndarray      # H x W uint8
    img_cur: np.ndarray      # H x W uint8
    pt_ref: np.ndarray       # n x 2 float32  (mvKeysRefUn)
    pt_init: np.ndarray      # n x 2 float32  (mvPtPredictUn)
    affine: np.ndarray       # n x 4 float32  (mvAffineDeformationMatrix)
    status_in: np.ndarray    # n uint8        (mvStatus after GyroPredictFeatures)
    pt_true: np.ndarray      # n x 2 float64  ground-truth position in img_cur
    camera: Camera
    half_patch: int
    iterations: int
    pyramids: int
    has_gyro: bool
    illumination: bool = True
    affine_on: bool = True
    penalty: bool = False

    @property
    def n(self) -> int:
        return int(self.pt_ref.shape[0])

    @property
    def n_active(self) -> int:
        return int(np.count_nonzero(self.status_in))


def gyro_predict(cam: Camera, H: np.ndarray, r3: np.ndarray, pts: np.ndarray, width: int, height: int,
                 half_patch: int):
    """float32 restatement of the producer of the path's inputs (reference
    src/gyro_aided_tracker.cpp:118-185,194-231): predicted point, border status and the
    2x2 affine from the four predicted patch corners.  Input generator only."""
    f32 = np.float32
    Hf = H.astype(f32)
    r3 = r3.astype(f32)
    fx, fy, cx, cy = f32(cam.fx), f32(cam.fy), f32(cam.cx), f32(cam.cy)
    fxi, fyi = f32(1.0 / float(fx)), f32(1.0 / float(fy))
    k1, k2, p1, p2 = (f32(v) for v in cam.dist[:4])

    def one(px, py):
        xn = (px - cx) * fxi
        yn = (py - cy) * fyi
        lam = (1.0 / (r3[0] * xn + r3[1] * yn + r3[2]).astype(np.float64)).astype(f32)
        ux = (Hf[0, 0] * px + Hf[0, 1] * py + Hf[0, 2]) * lam
        uy = (Hf[1, 0] * px + Hf[1, 1] * py + Hf[1, 2]) * lam
        x = (ux - cx) * fxi
        y = (uy - cy) * fyi
        r2 = x * x + y * y
        r4 = r2 * r2
        xd = x * (1 + k1 * r2 + k2 * r4) + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
        yd = y * (1 + k1 * r2 + k2 * r4) + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
        return ux, uy, fx * xd + cx, fy * yd + cy

    px, py = pts[:, 0].astype(f32), pts[:, 1].astype(f32)
    ux, uy, dx, dy = one(px, py)
    ok = (ux >= 0) & (ux < width) & (uy >= 0) & (uy < height) & (dx >= 0) & (dx < width) & (dy >= 0) & (dy < height)
    h = f32(half_patch)
    corners = [(-h, -h), (h, -h), (-h, h), (h, h)]
    C = np.zeros((pts.shape[0], 2, 4), dtype=np.float64)
    B = np.array([[c[0] for c in corners], [c[1] for c in corners]], dtype=np.float64)
    for j, (ox, oy) in enumerate(corners):
        cux, cuy, _, _ = one(px + ox, py + oy)
        C[:, 0, j] = (cux - ux).astype(np.float64)
        C[:, 1, j] = (cuy - uy).astype(np.float64)
    A = (C @ B.T) @ np.linalg.inv(B @ B.T)
    pt_init = np.stack([ux, uy], axis=1).astype(f32)
    pt_init[~ok] = 0
    return pt_init, A.reshape(-1, 4).astype(f32), ok.astype(np.uint8)


def make_workload(name: str, width: int, height: int, n: int, *, seed: int, half_patch: int = 10,
                  iterations: int = 30, pyramids: int = 3, camera: Camera | None = None,
                  motion: str = "rotation", omega=(0.5, -1.0, 2.0), dt: float = 0.05,
                  gyro_error=(0.004, -0.003, 0.006), translation=(1.37, -0.83), has_gyro: bool = True,
                  gain: float = 1.05, offset: float = 4.0, edge_fraction: float = 0.0,
                  penalty: bool = False) -> Workload:
    """motion='translation': img_cur is img_ref shifted by `translation`, identity init
    (SURVEY.md §8(d) config 1).  motion='rotation': pure-rotation homography K R K^-1 with
    R = exp(omega*dt); the gyro prediction uses R perturbed by `gyro_error` (rad), so the
    initial guess is a couple of pixels off, like a real gyro."""
    rng = SplitMix64(seed)
    cam = camera or generic_camera(width, height)
    tex = Texture(rng)
    yy, xx = np.mgrid[0:height, 0:width].astype(np.float64)
    img_ref = _to_u8(tex(xx, yy))

    if motion == "translation":
        tx, ty = translation
        Hm = np.array([[1, 0, tx], [0, 1, ty], [0, 0, 1]], dtype=np.float64)
        Hpred = np.eye(3)
        r3 = np.array([0.0, 0.0, 1.0])
    else:
        K = cam.K
        Kinv = np.linalg.inv(K)
        R = rodrigues(np.asarray(omega, dtype=np.float64) * dt)
        Hm = K @ R @ Kinv
        Rp = rodrigues(np.asarray(gyro_error, dtype=np.float64)) @ R
        Hpred = K @ Rp @ Kinv
        # pixel-aware form: u2 = (K R K^-1 u1) / (r3 . K^-1 u1)  (reference :216-218)
        r3 = Rp[2, :].copy()
    Hinv = np.linalg.inv(Hm)
    den = Hinv[2, 0] * xx + Hinv[2, 1] * yy + Hinv[2, 2]
    sx = (Hinv[0, 0] * xx + Hinv[0, 1] * yy + Hinv[0, 2]) / den
    sy = (Hinv[1, 0] * xx + Hinv[1, 1] * yy + Hinv[1, 2]) / den
    img_cur = _to_u8(gain * tex(sx, sy) + offset)

    # interior keypoints, margin 2^(L-1) (h+6); optional edge set with margin 2
    m = float((1 << (pyramids - 1)) * (half_patch + 6))
    m = min(m, 0.25 * min(width, height))
    n_edge = int(round(edge_fraction * n))
    u = rng.uniform(2 * n)
    pts = np.empty((n, 2), dtype=np.float64)
    pts[:, 0] = m + u[0::2] * (width - 2 * m)
    pts[:, 1] = m + u[1::2] * (height - 2 * m)
    if n_edge:
        ue = rng.uniform(3 * n_edge)
        side = (ue[0::3] * 4).astype(int)
        t = ue[1::3]
        off = 2.0 + ue[2::3] * (m - 2.0)
        ex = np.where(side == 0, off, np.where(side == 1, width - 1 - off, 2 + t * (width - 4)))
        ey = np.where(side == 2, off, np.where(side == 3, height - 1 - off, 2 + t * (height - 4)))
        pts[:n_edge, 0], pts[:n_edge, 1] = ex, ey
    pt_ref = pts.astype(np.float32)

    pr = pt_ref.astype(np.float64)
    d = Hm[2, 0] * pr[:, 0] + Hm[2, 1] * pr[:, 1] + Hm[2, 2]
    pt_true = np.stack([(Hm[0, 0] * pr[:, 0] + Hm[0, 1] * pr[:, 1] + Hm[0, 2]) / d,
                        (Hm[1, 0] * pr[:, 0] + Hm[1, 1] * pr[:, 1] + Hm[1, 2]) / d], axis=1)

    if has_gyro and motion != "translation":
        # mKRKinv rows are used un-normalised together with lambda (reference :216-218)
        K = cam.K
        KRK = K @ (rodrigues(np.asarray(gyro_error, dtype=np.float64)) @ rodrigues(np.asarray(omega) * dt)) @ np.linalg.inv(K)
        pt_init, A, status = gyro_predict(cam, KRK, r3, pt_ref, width, height, half_patch)
    else:
        # !mbHasGyroPredictInitial branch, reference src/gyro_aided_tracker.cpp:264-270
        pt_init = pt_ref.copy()
        A = np.tile(np.array([1, 0, 0, 1], dtype=np.float32), (n, 1))
        status = np.ones(n, dtype=np.uint8)
    return Workload(name, img_ref, img_cur, pt_ref, np.ascontiguousarray(pt_init), np.ascontiguousarray(A),
                    status, pt_true, cam, half_patch, iterations, pyramids,
                    has_gyro and motion != "translation", penalty=penalty)


# BASELINE.json `configs`, restated as synthetic stand-ins.
def config(idx: int, n: int | None = None, **kw) -> Workload:
    seed = kw.pop("seed", 0x5EED0000 + idx)   # (another seed = another stream of the same shape)
    if idx == 0:   # CPU plumbing: 640x480, 500 kpts, identity init, 21x21, 3 levels
        return make_workload("cfg0_640x480_identity", 640, 480, n or 500, seed=seed, motion="translation",
                             has_gyro=False, **kw)
    if idx == 1:   # EuRoC-like 752x480, ~1000 kpts, gyro-predicted affine init
        return make_workload("cfg1_euroc_752x480", 752, 480, n or 1000, seed=seed, camera=EUROC, **kw)
    if idx == 2:   # D435i-like 640x480, 2000 kpts, 4 levels
        kw.setdefault("pyramids", 4)
        return make_workload("cfg2_d435i_640x480_L4", 640, 480, n or 2000, seed=seed, camera=D435I,
                             omega=(0.3, -0.5, 1.0), **kw)
    if idx == 3:   # 1920x1080 fast rotation, 20000 kpts (sharded over GPUs)
        return make_workload("cfg3_1080p_fastrot", 1920, 1080, n or 20000, seed=seed,
                             omega=(0.8, -1.2, 2.0), **kw)
    if idx == 4:   # one 1280x720 stream x 4000 kpts (one per GPU)
        return make_workload("cfg4_720p_stream", 1280, 720, n or 4000, seed=seed, **kw)
    raise ValueError(idx)
