"""libpagk_hip.so WITH variants (b) (2-wave workgroup, f64 MFMA chain: pagk_set_kernel 2) and (e) (four independent rows per
wave + work queue: 6), which the product's build leaves out because nothing selects them and they win at no launch size
 -> tools/bin/libpagk_hip_all.so.  Cross-checks and sweeps load it through PAGK_LIB:
    python tools/build_all_variants.py && PAGK_LIB=tools/bin/libpagk_hip_all.so python -m pytest tests -m gpu -q
(the tests that address variants 2 and 6 skip themselves on the product library)."""
import os
import shutil
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

out = os.path.join(ROOT, "tools", "bin", "libpagk_hip_all.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
scratch = tempfile.mkdtemp(prefix="pagk_build_all_")
try:
    tmp = os.path.join(scratch, "libpagk_hip_all.so")
    g._run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), *g.HIPCC_FLAGS, "-DPAGK_ALL_VARIANTS", "-save-temps=obj", "-o", tmp,
            os.path.join(g.CSRC, "pagk_hip.hip")])
    asm = [f for f in os.listdir(scratch) if f.endswith("gfx950.s")][0]
    res = g._parse_resources_from_asm(open(os.path.join(scratch, asm)).read())
    g.check_resources(res)
    g.check_resources(res, g.RESOURCE_CLAIMS_ALL_VARIANTS)
    shutil.copyfile(tmp, out)
    os.chmod(out, 0o755)
finally:
    shutil.rmtree(scratch, ignore_errors=True)
print("built", out)
