"""The pinning kit (tools/ref_dump/): oracle and HIP path against outputs of the REAL reference.

Nothing under /root/reference can be built or run in the image this repository is developed in (OpenCV, Eigen, glog
absent), and the reference ships no vectors: parity is unpinned (oracle/README.md).  A maintainer with a working
checkout closes that in three commands (tools/ref_dump/dump_patchmatch.cpp, header): the reference itself runs
PatchMatch::OpticalFlowMultiLevel() (src/patch_match.cpp:79-142) over the committed golden inputs, the results land in
tests/golden/ref/<case>.npz, and the tests below -- skipped until that directory exists -- then
  * name the pagk_params::solver_variant mask under which the CPU restatement reproduces the reference bit for bit
    (the Eigen associations it had to guess, oracle/README.md),
  * check the pyramid levels (the one OpenCV routine on the path) byte for byte,
  * and hold the HIP path, with that mask, to the reference's outputs.
The first test runs always: it drives the kit's file formats end to end with the oracle standing in for the reference,
so that the three commands work the day somebody runs them."""
import glob
import importlib.util
import os
import struct

import numpy as np
import pytest

from oracle import pagk_oracle as orc
from pixel_aware_gyro_aided_klt_feature_tracker_amd import capi

from util import GOLDEN_DIR, golden_cases, load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_DIR = os.path.join(GOLDEN_DIR, "ref")
SOLVER_BITS = (1, 2, 4, 8, 32)
ALL_MASKS = [sum(b for k, b in enumerate(SOLVER_BITS) if m >> k & 1) for m in range(32)]


def _tool(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", "ref_dump", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def ref_cases():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(REF_DIR, "*.npz")))


def _write_ref_like_the_dumper(path, out, levels_ref, levels_cur):
    """The layout dump_patchmatch.cpp writes (its header documents it)."""
    n = out["status"].shape[0]
    with open(path, "wb") as f:
        f.write(b"PAGKREF1")
        f.write(struct.pack("<2i", n, len(levels_ref)))
        for k, dt in (("pt_un", np.float32), ("pt_dist", np.float32), ("status", np.uint8), ("pix_err", np.float64),
                      ("dist_pred", np.float64), ("ncc", np.float32)):
            f.write(np.ascontiguousarray(out[k][:n], dt).tobytes())
        for l in range(1, len(levels_ref)):
            f.write(struct.pack("<2i", levels_ref[l].shape[1], levels_ref[l].shape[0]))
            f.write(levels_ref[l].tobytes())
            f.write(levels_cur[l].tobytes())
        f.write(b"\nEigen 0.0.0; OpenCV none; EIGEN_VECTORIZE off (oracle standing in)\n")


def _pyramid(img, L):
    lv = [np.ascontiguousarray(img)]
    for _ in range(1, L):
        lv.append(orc.pyr_down(lv[-1]))
    return lv


def test_kit_file_formats_round_trip(built, tmp_path):
    export, imp = _tool("export_cases"), _tool("import_ref")
    name = "flags_a1_i1_p0"
    params, inp, exp = load_golden(name)
    fin = str(tmp_path / (name + ".in"))
    export.export_case(os.path.join(GOLDEN_DIR, name + ".npz"), fin)
    raw = open(fin, "rb").read()
    assert raw[:8] == b"PAGKIN1\0"
    W, H, n, h, it, L, gy, il, af, pe, nc = struct.unpack_from("<11i", raw, 8)
    assert (H, W) == inp["img_ref"].shape and n == inp["pt_ref"].shape[0]
    assert (h, it, L) == (params.half_patch, params.iterations, params.pyramids)
    assert (gy, il, af, pe, nc) == (params.has_gyro_predict_initial, params.consider_illumination, params.consider_affine,
                                    params.regularization_penalty, params.calculate_ncc)
    off = 8 + 44 + 32
    assert np.array_equal(np.frombuffer(raw, np.uint8, W * H, off).reshape(H, W), inp["img_ref"])
    off += 2 * W * H
    assert np.array_equal(np.frombuffer(raw, np.float32, 2 * n, off).reshape(n, 2), inp["pt_ref"])
    assert len(raw) == off + n * (8 + 8 + 16 + 1)
    # the oracle plays the reference: what it writes in the dumper's layout comes back through import_ref unchanged
    out = orc.track(params, inp["img_ref"], inp["img_cur"], inp["pt_ref"], inp["pt_init"], inp["affine"], inp["status_in"])
    fref = str(tmp_path / (name + ".ref"))
    _write_ref_like_the_dumper(fref, out, _pyramid(inp["img_ref"], L), _pyramid(inp["img_cur"], L))
    back = imp.read_ref(fref)
    for k in ("pt_un", "pt_dist", "status", "pix_err", "dist_pred", "ncc"):
        assert np.array_equal(back[k], out[k][:n], equal_nan=True), k
    assert back["ref_level1"].shape == (H // 2, W // 2) and "cur_level%d" % (L - 1) in back
    assert b"Eigen" in back["built_with"].tobytes()


def _matching_masks(name):
    """solver_variant masks under which the oracle reproduces the reference's outputs for this case, bit for bit."""
    params, inp, _ = load_golden(name)
    ref = np.load(os.path.join(REF_DIR, name + ".npz"), allow_pickle=False)
    n = ref["status"].shape[0]
    hits = []
    try:
        for mask in ALL_MASKS:
            orc.set_alternatives(mask)
            out = orc.track(params, inp["img_ref"], inp["img_cur"], inp["pt_ref"], inp["pt_init"], inp["affine"], inp["status_in"])
            if all(np.array_equal(out[k][:n], ref[k], equal_nan=True) for k in ("status", "pt_un", "pix_err", "dist_pred")):
                hits.append(mask)
    finally:
        orc.set_alternatives(0)
    return hits, ref, params, inp


@pytest.mark.skipif(not ref_cases(), reason="tests/golden/ref/ is empty: run the pinning kit with the real reference "
                                            "(tools/ref_dump/dump_patchmatch.cpp) to pin the oracle")
@pytest.mark.parametrize("name", ref_cases() or ["-"])
def test_oracle_reproduces_the_reference(built, name):
    hits, ref, params, inp = _matching_masks(name)
    built_with = ref["built_with"].tobytes().decode(errors="replace") if "built_with" in ref.files else "?"
    # pyramid levels: cv::resize as the reference ran it (src/patch_match.cpp:69-70)
    for key, img in (("ref", inp["img_ref"]), ("cur", inp["img_cur"])):
        lv = _pyramid(img, params.pyramids)
        for l in range(1, params.pyramids):
            assert np.array_equal(lv[l], ref[f"{key}_level{l}"]), f"{name}: pyramid level {l} of {key} differs from cv::resize ({built_with})"
    assert hits, (f"{name}: no solver_variant mask makes the CPU restatement reproduce the reference built with {built_with}; "
                  "status / pt_un / pix_err / dist_pred were compared bit for bit")
    print(f"{name}: reference ({built_with}) reproduced with solver_variant in {hits}")


@pytest.mark.gpu
@pytest.mark.skipif(not ref_cases(), reason="tests/golden/ref/ is empty (see test_oracle_reproduces_the_reference)")
@pytest.mark.parametrize("name", ref_cases() or ["-"])
def test_hip_reproduces_the_reference(ctx, name):
    hits, ref, params, inp = _matching_masks(name)
    assert hits, "pin the oracle first"
    n = ref["status"].shape[0]
    params.solver_variant = hits[0]
    got = ctx.track(params, inp["img_ref"], inp["img_cur"], inp["pt_ref"], inp["pt_init"], inp["affine"], inp["status_in"])
    assert np.array_equal(got["status"][:n], ref["status"])                       # masks bit-exact (north_star)
    d = np.abs(got["pt_un"][:n].astype(np.float64) - ref["pt_un"].astype(np.float64))
    assert d.size == 0 or np.nanmax(d) <= 1e-3                                    # coordinates within 1e-3 px
    for k in ("pt_un", "pt_dist", "pix_err", "dist_pred", "ncc"):                  # ... and, by construction, identical
        assert np.array_equal(got[k][:n], ref[k], equal_nan=True), k
