"""Pinning kit, step 1: tests/golden/*.npz -> <outdir>/<case>.in, the raw layout tools/ref_dump/dump_patchmatch.cpp
reads (documented at the top of that file).  Needs numpy only.   python tools/ref_dump/export_cases.py <outdir>"""
import glob
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def export_case(npz_path: str, out_path: str) -> None:
    z = np.load(npz_path, allow_pickle=False)
    cfg = [int(v) for v in z["cfg"]]          # half_patch iterations pyramids has_gyro illumination affine penalty [ncc]
    ncc = cfg[7] if len(cfg) > 7 else 0
    ref, cur = np.ascontiguousarray(z["img_ref"], np.uint8), np.ascontiguousarray(z["img_cur"], np.uint8)
    n = int(z["pt_ref"].shape[0])
    with open(out_path, "wb") as f:
        f.write(b"PAGKIN1\0")
        f.write(struct.pack("<11i", ref.shape[1], ref.shape[0], n, cfg[0], cfg[1], cfg[2], cfg[3], cfg[4], cfg[5], cfg[6], ncc))
        f.write(np.asarray(z["camera"][:8], np.float32).tobytes())
        f.write(ref.tobytes())
        f.write(cur.tobytes())
        for k, dt in (("pt_ref", np.float32), ("pt_init", np.float32), ("affine", np.float32), ("status_in", np.uint8)):
            f.write(np.ascontiguousarray(z[k], dt).tobytes())


def main():
    out = sys.argv[1]
    os.makedirs(out, exist_ok=True)
    for p in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "*.npz"))):
        name = os.path.splitext(os.path.basename(p))[0]
        export_case(p, os.path.join(out, name + ".in"))
        print("wrote", os.path.join(out, name + ".in"))


if __name__ == "__main__":
    main()
