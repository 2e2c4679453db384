"""H.llt().solve(b) / update.norm() (reference src/patch_match.cpp:319,343) on the device against the CPU oracle,
through the C ABI's diagnostics (pagk_selftest_*) and through the tracking kernels, for every
pagk_params::solver_variant bit -- the associations another Eigen version or build would use (oracle/README.md).
Bar: bit-identical."""
import itertools

import numpy as np
import pytest

from oracle import pagk_oracle as orc
from pixel_aware_gyro_aided_klt_feature_tracker_amd import capi, synth

from util import assert_parity, params_for, needs_variant

pytestmark = pytest.mark.gpu

SOLVER_BITS = (1, 2, 4, 8, 32)
MASKS = [0, 1, 2, 4, 8, 32, 1 | 8, 8 | 32, 1 | 2 | 4, 1 | 2 | 4 | 8 | 32]


@pytest.fixture
def alternatives():
    yield orc.set_alternatives
    orc.set_alternatives(0)


def _bits(a):
    return np.ascontiguousarray(a, np.float64).view(np.uint64)


def _operands(rng, n):
    """Numerator / denominator pairs: ordinary magnitudes, the limits of the fast form's range (2^+-400), values
    beyond it, zeros of both signs, denormals, infinities, NaN."""
    mant = 1.0 + rng.random(n)
    sign = np.where(rng.random(n) < 0.5, -1.0, 1.0)
    kind = rng.integers(0, 10, n)
    e = np.where(kind < 6, rng.integers(-40, 40, n),
                 np.where(kind < 8, rng.integers(-420, 421, n), rng.integers(-1074, 1024, n)))
    v = sign * np.ldexp(mant, e)
    special = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 5e-324, -5e-324, 2.2250738585072014e-308,
                        1.7976931348623157e308, np.ldexp(1.0, -400), np.ldexp(1.0, 401), np.ldexp(1.9999, 400),
                        np.ldexp(1.0, -401), 1.0, 3.0])
    pick = rng.random(n) < 0.02
    v[pick] = special[rng.integers(0, len(special), int(pick.sum()))]
    return v


def test_division_by_prepared_denominator(ctx):
    """den_prepare + div_group == the compiler's correctly rounded f64 division, bit for bit, on 2^22 operand pairs
    incl. everything outside the fast form's range; and equal to the host's IEEE division."""
    rng = np.random.default_rng(0xD1F1DE)
    n = 1 << 22
    num, den = _operands(rng, n), _operands(rng, n)
    q_plain, q_prep, root, root_lean = ctx.selftest_divide(np.where(num == 0, num, num), den)
    assert np.array_equal(_bits(q_plain), _bits(q_prep))
    with np.errstate(all="ignore"):
        host = num / den
    nan = np.isnan(host)
    assert np.array_equal(np.isnan(q_prep), nan)
    assert np.array_equal(_bits(q_prep[~nan]), _bits(host[~nan]))
    # the solve's lean square root == the compiler's == the host's, on the same operands (negative -> NaN in all three)
    rn = np.isnan(root)
    assert np.array_equal(np.isnan(root_lean), rn) and np.array_equal(_bits(root_lean[~rn]), _bits(root[~rn]))
    with np.errstate(all="ignore"):
        assert np.array_equal(_bits(root[~rn]), _bits(np.sqrt(num[~rn])))


def test_norm_threshold_is_equivalent_to_the_square_root_test(ctx):
    """sqrt(s) < 1e-2  <=>  s < kNormSqConverged, with the DEVICE's sqrt: checked on the 2^16 doubles either side of
    the threshold and on random squared norms."""
    T = float.fromhex("0x1.a36e2eb1c432cp-14")
    tb = np.array([T]).view(np.uint64)[0]
    near = (tb + np.arange(-(1 << 16), 1 << 16, dtype=np.int64).astype(np.uint64)).view(np.float64)
    rng = np.random.default_rng(7)
    s = np.concatenate([near, rng.random(1 << 16) * 3e-4, np.array([0.0, np.inf, np.nan, 1e-4, T])])
    _, _, root, root_lean = ctx.selftest_divide(s, np.ones_like(s))
    assert np.array_equal(_bits(root[~np.isnan(s)]), _bits(root_lean[~np.isnan(s)]))
    with np.errstate(invalid="ignore"):
        assert np.array_equal(root < 1e-2, s < T)
        assert np.array_equal(_bits(root[~np.isnan(s)]), _bits(np.sqrt(s[~np.isnan(s)])))


def _systems(rng, n):
    """Normal equations like the path's (rank-deficient by construction: J3 = c * J4), plus broken ones."""
    H = np.zeros((n, 4, 4))
    b = np.zeros((n, 4))
    for i in range(n):
        kind = i % 8
        m = 441
        J = rng.normal(0, 8, (m, 4)).astype(np.float32).astype(np.float64)
        c = np.float32(-rng.uniform(20, 220))
        J[:, 2], J[:, 3] = float(c), 1.0
        e = rng.normal(0, 5, m).astype(np.float32).astype(np.float64)
        if kind == 5:
            J[:, 0] = 0.0            # flat in x: first pivot fails
        if kind == 6:
            J[:, :2] *= 1e-3
        Hi = J.T @ J
        bi = -J.T @ e
        if kind == 7:
            Hi = rng.normal(0, 1, (4, 4)) * 10.0 ** rng.integers(-30, 30)   # indefinite / huge / tiny
            bi = rng.normal(0, 1, 4)
        if kind == 4 and i % 16 == 4:
            Hi[rng.integers(0, 4), rng.integers(0, 4)] = np.nan
        H[i], b[i] = Hi, bi
    return H, b


@pytest.mark.parametrize("mask", MASKS)
def test_solve_matches_oracle_for_every_variant(ctx, alternatives, mask):
    rng = np.random.default_rng(1000 + mask)
    H, b = _systems(rng, 4096)
    alternatives(mask)
    ref_x = np.zeros((len(H), 4))
    ref_n = np.zeros(len(H))
    for i in range(len(H)):
        ref_x[i], ref_n[i] = orc.llt_solve4(H[i], b[i])
    alternatives(0)
    xs, ns, xl, nl = ctx.selftest_solve(H.reshape(-1, 16), b, mask)
    for name, got in (("one lane per system", xs), ("four lanes per system", xl)):
        assert np.array_equal(np.isnan(got), np.isnan(ref_x)), name
        ok = ~np.isnan(ref_x)
        assert np.array_equal(_bits(got[ok]), _bits(ref_x[ok])), name
    okn = ~np.isnan(ref_n)
    assert np.array_equal(_bits(ns[okn]), _bits(ref_n[okn]))
    # the four-lane form returns the squared norm: the kernels' convergence test must agree with `norm < 1e-2`
    with np.errstate(invalid="ignore"):
        assert np.array_equal(nl < float.fromhex("0x1.a36e2eb1c432cp-14"), ref_n < 1e-2)


def test_variants_differ_where_the_readme_says_they_do(ctx):
    """The masks are not no-ops: on the path's own systems the reciprocal scaling, the lower solve's association and
    the 4th pivot's association each change some results (oracle/README.md, 'How much each guess matters')."""
    rng = np.random.default_rng(5)
    H, b = _systems(rng, 2048)
    base = ctx.selftest_solve(H.reshape(-1, 16), b, 0)[0]
    for mask in (1, 8, 32):
        alt = ctx.selftest_solve(H.reshape(-1, 16), b, mask)[0]
        assert (_bits(np.nan_to_num(alt)) != _bits(np.nan_to_num(base))).any(), mask


@pytest.mark.parametrize("kernel", [0, 1, pytest.param(2, marks=needs_variant(2)), 3, 5, pytest.param(6, marks=needs_variant(6))])
@pytest.mark.parametrize("mask", [1, 8, 32, 1 | 2 | 4 | 8 | 32])
def test_tracking_kernels_follow_solver_variant(ctx, alternatives, kernel, mask):
    """Every exact tracking variant with pagk_params::solver_variant = mask against the oracle with the same
    alternatives: all outputs bit-identical."""
    w = synth.config(1, n=400)
    p = params_for(w)
    p.solver_variant = mask
    ctx.set_kernel(kernel)
    got = ctx.track(p, w.img_ref, w.img_cur, w.pt_ref, w.pt_init, w.affine, w.status_in)
    ctx.set_kernel(0)
    alternatives(mask)
    ref = orc.track(p, w.img_ref, w.img_cur, w.pt_ref, w.pt_init, w.affine, w.status_in, nthreads=16)
    alternatives(0)
    assert_parity(got, ref, w.n, exact=True, what=f"kernel {kernel} solver_variant {mask}")


def test_solver_variant_changes_tracking_results_and_is_validated(ctx):
    w = synth.config(1, n=400)
    p = params_for(w)
    base = ctx.track(p, w.img_ref, w.img_cur, w.pt_ref, w.pt_init, w.affine, w.status_in)
    p.solver_variant = 8
    alt = ctx.track(p, w.img_ref, w.img_cur, w.pt_ref, w.pt_init, w.affine, w.status_in)
    assert np.array_equal(base["status"], alt["status"])
    assert not np.array_equal(base["pt_un"], alt["pt_un"])
    p.solver_variant = 16   # the oracle's pyramid switch is not a solver bit
    with pytest.raises(capi.PagkError):
        ctx.track(p, w.img_ref, w.img_cur, w.pt_ref, w.pt_init, w.affine, w.status_in)
