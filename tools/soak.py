"""One-off soak: the randomized parity sweep of tests/test_parity_gpu.py over many more seeds.
Usage: python tools/soak.py [first_seed] [count]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import pagk_oracle as orc
from pixel_aware_gyro_aided_klt_feature_tracker_amd import capi
import test_parity_gpu as T
from util import assert_parity
first, count = int(sys.argv[1]) if len(sys.argv) > 1 else 1000, int(sys.argv[2]) if len(sys.argv) > 2 else 300
# PAGK_SOAK_KERNELS=0,2,3,5,6 (default); with PAGK_QUAD_BUDGET / PAGK_ROWS_WAVES set, the hand-over and the queue are soaked too
KERNELS = tuple(k for k in (int(v) for v in os.environ.get("PAGK_SOAK_KERNELS", "0,2,3,5,6,7").split(",")) if capi.has_variant(k))
# (2 and 6 only in a -DPAGK_ALL_VARIANTS build: PAGK_LIB=tools/bin/libpagk_hip_all.so, tools/build_all_variants.py)
ctx = capi.Context(0)
bad = 0
feats = 0
for seed in range(first, first + count):
    w, flags = T._random_case(seed)
    if os.environ.get("PAGK_SOAK_CROWDED"):
        # the pipelined 4-wave body (h = 8 / 9 / 10) with every CU holding several workgroups: its LDS flag protocol under
        # contention, all flag combinations (LEAN and generic instantiations), 1-4 levels
        rng = np.random.default_rng(seed ^ 0x5EED)
        L = int(rng.integers(1, 5))
        mult = 1 << (L - 1)
        w = T.synth.make_workload("crowd", mult * int(rng.integers(40, 160)), mult * int(rng.integers(30, 120)),
                                  int(rng.integers(800, 3500)), seed=seed, half_patch=int(rng.integers(8, 11)),
                                  iterations=int(rng.integers(2, 31)), pyramids=L, motion="rotation", has_gyro=True,
                                  omega=tuple(rng.uniform(-1.0, 1.0, 3)), edge_fraction=float(rng.choice([0.0, 0.3])),
                                  gain=float(rng.uniform(0.8, 1.25)), offset=float(rng.uniform(-10, 10)))
    p = capi.make_params(half_patch=w.half_patch, iterations=w.iterations, pyramids=w.pyramids, has_gyro=w.has_gyro,
                         camera=w.camera, **flags)
    ref = orc.track(p, w.img_ref, w.img_cur, w.pt_ref, w.pt_init, w.affine, w.status_in, nthreads=8)
    for kernel in KERNELS:
        ctx.set_kernel(kernel)
        got = ctx.track(p, w.img_ref, w.img_cur, w.pt_ref, w.pt_init, w.affine, w.status_in)
        try:
            assert_parity(got, ref, w.n, exact=True, what=f"seed {seed} kernel {kernel}")
        except AssertionError as e:
            bad += 1
            print("MISMATCH", e, flush=True)
    feats += w.n
    if (seed - first) % 50 == 49:
        print(f"{seed - first + 1} cases, {feats} features, {bad} mismatches", flush=True)
ctx.set_kernel(0)
print(f"soak done: {count} cases x kernels {KERNELS}, {feats} features, {bad} mismatches")
