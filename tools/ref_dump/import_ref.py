"""Pinning kit, step 3: <dir>/<case>.ref (written by dump_patchmatch, i.e. by the REAL reference) ->
<golden_ref_dir>/<case>.npz, which tests/test_reference_pin.py compares the oracle and the HIP path with.
python tools/ref_dump/import_ref.py <dir> tests/golden/ref"""
import glob
import os
import struct
import sys

import numpy as np


def read_ref(path: str) -> dict:
    raw = open(path, "rb").read()
    if raw[:8] != b"PAGKREF1":
        raise ValueError(f"{path}: not a PAGKREF1 file")
    n, L = struct.unpack_from("<2i", raw, 8)
    off = [16]

    def take(dtype, count, shape=None):
        a = np.frombuffer(raw, dtype=dtype, count=count, offset=off[0]).copy()
        off[0] += a.nbytes
        return a.reshape(shape) if shape else a
    out = {"pt_un": take(np.float32, 2 * n, (n, 2)), "pt_dist": take(np.float32, 2 * n, (n, 2)), "status": take(np.uint8, n),
           "pix_err": take(np.float64, n), "dist_pred": take(np.float64, n), "ncc": take(np.float32, n)}
    for l in range(1, L):
        w, h = struct.unpack_from("<2i", raw, off[0])
        off[0] += 8
        out[f"ref_level{l}"] = take(np.uint8, w * h, (h, w))
        out[f"cur_level{l}"] = take(np.uint8, w * h, (h, w))
    out["built_with"] = np.frombuffer(raw[off[0]:].strip(), dtype=np.uint8).copy()   # the text tail, as bytes
    return out


def main():
    src, dst = sys.argv[1], sys.argv[2]
    os.makedirs(dst, exist_ok=True)
    for p in sorted(glob.glob(os.path.join(src, "*.ref"))):
        name = os.path.splitext(os.path.basename(p))[0]
        np.savez_compressed(os.path.join(dst, name + ".npz"), **read_ref(p))
        print("imported", name)


if __name__ == "__main__":
    main()
