"""H(2,2) in closed form (round 4): repeat_sum_f64 of csrc/pagk_device.h against the loop it replaces --
s_k = RN(s_{k-1} + c*c), k = 1..(2h+1)^2, src/patch_match.cpp:296 with J[2] = c (:263).

CPU: a C restatement of the device routine, operation for operation (tests/repeat_sum_check.c), on random floats of the
shapes a sample can take and -- the sums are invariant under scaling c by powers of two -- on a stride of the 2^23
mantissas (tools: the full 2^23 sweep is `repeat_sum_check 0 1`); with the quotient estimate perturbed both ways by far
more than v_rcp_f32's error, which must not matter.  GPU: the device routine itself through pagk_selftest_repeat_sum."""
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("rs") / "repeat_sum_check")
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-o", exe, os.path.join(HERE, "repeat_sum_check.c"), "-lm"], check=True)
    return exe


@pytest.mark.parametrize("scale", ["1.0", "1.00003", "0.99997"])
def test_closed_form_equals_the_loop_on_the_host(checker, scale):
    r = subprocess.run([checker, "300000", "0", scale], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("count", [289, 361, 441])
def test_closed_form_equals_the_loop_on_the_device(ctx, count):
    rng = np.random.default_rng(count)
    n = 1 << 18
    # sample values: bilinear interpolations of bytes (also tiny products of two small weights), exact integers, zero
    c = np.concatenate([
        (rng.random(n // 2) * 255.0).astype(np.float32),
        (rng.random(n // 4) * rng.random(n // 4) * 1e-6).astype(np.float32),
        rng.integers(0, 256, n // 8).astype(np.float32),
        np.ldexp(1.0 + rng.integers(0, 1 << 23, n // 8) / float(1 << 23), rng.integers(-40, 8, n // 8)).astype(np.float32),
    ])
    c[:4] = [0.0, 1.0, 255.0, np.float32(2.0) ** -48]
    c = -c  # de_dg = -I1(pt) (:263); the square does not see the sign
    closed, loop = ctx.selftest_repeat_sum(c, count)
    assert np.array_equal(closed.view(np.uint64), loop.view(np.uint64))
    # ... and the loop is the host's loop
    q = c.astype(np.float64) ** 2
    s = np.zeros_like(q)
    for _ in range(count):
        s = s + q
    assert np.array_equal(s.view(np.uint64), loop.view(np.uint64))
