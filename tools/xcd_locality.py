"""A/B: does the order of the features (which XCD's L2 sees which image region) matter?
Blocks are dealt to the 8 XCDs round-robin (block b -> XCD b % 8).  'banded' order gives XCD k the k-th horizontal
band of the image; 'random' is the generator's order.  Same features, same results, same process.
Usage: python tools/xcd_locality.py   (add rocprofv3 --pmc FETCH_SIZE around it for the traffic)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pixel_aware_gyro_aided_klt_feature_tracker_amd import capi, synth
ctx = capi.Context(0)
for cfg, n in [tuple(int(v) for v in c.split(':')) for c in os.environ.get('PAGK_CASES', '1:1000,1:4000,3:20000').split(',')]:
    w = synth.config(cfg, n=n)
    p = capi.make_params(half_patch=w.half_patch, iterations=w.iterations, pyramids=w.pyramids, has_gyro=w.has_gyro, camera=w.camera)
    order = np.argsort(w.pt_ref[:, 1], kind="stable")            # by y
    bands = np.array_split(order, 8)
    m = min(len(b) for b in bands)
    inter = np.stack([b[:m] for b in bands], axis=1).reshape(-1)  # feature j of band k -> position 8 j + k
    rest = np.concatenate([b[m:] for b in bands])
    banded = np.concatenate([inter, rest]).astype(np.int64)
    rng = np.random.default_rng(0)
    # variant 7 hands QUADS (four consecutive features) to its eight ticket sequences, quad q -> sequence q % 8 -> the XCD
    # that starts with that sequence: band k's features in groups of four at positions 32 j + 4 k + i
    m4 = m // 4 * 4
    bq = np.stack([b[:m4].reshape(-1, 4) for b in bands], axis=1).reshape(-1)
    banded_quads = np.concatenate([bq, np.concatenate([b[m4:] for b in bands])]).astype(np.int64)
    perms = {"generator": np.arange(w.n), "banded": banded, "banded_quads": banded_quads, "sorted_y": order,
             "shuffled": rng.permutation(w.n)}
    line = f"cfg{cfg} n={n}:"
    for rep in range(2):
        for name, perm in perms.items():
            a = [np.ascontiguousarray(x[perm]) for x in (w.pt_ref, w.pt_init, w.affine, w.status_in)]
            ts = []
            for _ in range(14):
                ctx.track(p, w.img_ref, w.img_cur, *a)
                ts.append(ctx.last_kernel_ms()[0])
            if rep == 1:
                line += f"  {name} {np.median(ts[3:])*1e3:7.1f} us"
    print(line, flush=True)
