"""Ad-hoc GPU check: parity of the HIP path against the oracle on the BASELINE-shaped
configs, both kernels, plus kernel timings.  Run on the GPU box via gpurun."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pixel_aware_gyro_aided_klt_feature_tracker_amd import capi, synth
from oracle import pagk_oracle as orc

def compare(tag, got, ref, mask):
    n = mask.shape[0]
    st_bad = int(np.count_nonzero(got["status"][:n] != ref["status"][:n]))
    d = np.abs(got["pt_un"][:n].astype(np.float64) - ref["pt_un"][:n].astype(np.float64)).max(axis=1)
    dm = d[mask > 0]
    it_bad = int(np.count_nonzero(got["iters"][:n] != ref["iters"][:n]))
    pe = np.abs(got["pix_err"][:n] - ref["pix_err"][:n]).max()
    dd = np.abs(got["dist_pred"][:n] - ref["dist_pred"][:n]).max()
    pd = np.abs(got["pt_dist"][:n].astype(np.float64) - ref["pt_dist"][:n]).max()
    print(f"  {tag}: status mismatches {st_bad}, iters mismatches {it_bad}, max|dpt| {dm.max() if dm.size else 0:.3g}, "
          f"n(>1e-3) {int((dm > 1e-3).sum())}, n(!=0) {int((dm != 0).sum())}, pix_err diff {pe:.3g}, dist diff {dd:.3g}, pt_dist diff {pd:.3g}", flush=True)
    return st_bad == 0 and (dm.size == 0 or dm.max() <= 1e-3)

def main():
    ctx = capi.Context(0)
    ok = True
    for idx, n in ((0, 500), (1, 1000), (2, 2000), (3, 4000)):
        w = synth.config(idx, n=n)
        for penalty in (False, True):
            p = capi.make_params(half_patch=w.half_patch, iterations=w.iterations, pyramids=w.pyramids,
                                 has_gyro=w.has_gyro, camera=w.camera, penalty=penalty)
            t = time.time()
            ref = orc.track(p, w.img_ref, w.img_cur, w.pt_ref, w.pt_init, w.affine, w.status_in, nthreads=16)
            tcpu = time.time() - t
            print(f"{w.name} n={w.n} active={w.n_active} penalty={penalty}: oracle {tcpu*1e3:.1f} ms "
                  f"({w.n_active/tcpu:.0f} feat/s), mean iters {ref['iters'][:w.n].mean():.2f}, ok {int(ref['status'].sum())}", flush=True)
            for k in ((0, 1) if idx < 2 else (0,)):
                ctx.set_kernel(k)
                got = ctx.track(p, w.img_ref, w.img_cur, w.pt_ref, w.pt_init, w.affine, w.status_in)
                got = ctx.track(p, w.img_ref, w.img_cur, w.pt_ref, w.pt_init, w.affine, w.status_in)
                trk, pyr = ctx.last_kernel_ms()
                print(f"  kernel {k}: track {trk*1e3:.1f} us, pyramid(last frame) {pyr*1e3:.1f} us -> {w.n_active/trk/1e3:.2f} Mfeat/s", flush=True)
                ok &= compare(f"kernel {k} vs oracle", got, ref, w.status_in)
    # other patch sizes: every (NR, TAIL) instantiation of k_track_block
    for h in (1, 2, 3, 5, 6, 7, 8, 9, 11, 13, 14, 15):
        w = synth.make_workload(f"h{h}", 320, 240, 64, seed=0x5EED0100 + h, half_patch=h, iterations=10, pyramids=3,
                                camera=synth.D435I)
        p = capi.make_params(half_patch=h, iterations=10, pyramids=3, has_gyro=w.has_gyro, camera=w.camera)
        ref = orc.track(p, w.img_ref, w.img_cur, w.pt_ref, w.pt_init, w.affine, w.status_in, nthreads=16)
        ctx.set_kernel(0)
        got = ctx.track(p, w.img_ref, w.img_cur, w.pt_ref, w.pt_init, w.affine, w.status_in)
        trk, _ = ctx.last_kernel_ms()
        print(f"half_patch {h}: track {trk*1e3:.1f} us", flush=True)
        ok &= compare(f"h={h} kernel 0 vs oracle", got, ref, w.status_in)
    ctx.close()
    print("ALL OK" if ok else "MISMATCH")
    return 0 if ok else 1

if __name__ == "__main__":
    sys.exit(main())
