"""tools/prio_fluid_model.py on the committed per-level iteration counts (tools/data/*.npy, made by tools/iter_trace.py with the oracle): the
model's ranking of the priority rule's thresholds is what DESIGN.md section 4.4 quotes, and the counts are the oracle's."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, ROOT)
import prio_fluid_model as m  # noqa: E402


def test_the_rule_shortens_the_launch_and_k4_is_the_models_choice_at_1000_features():
    it = np.load(os.path.join(ROOT, "tools", "data", "iters_cfg1_1000.npy")).astype(int)
    base = m.simulate(it, m.by_phase_only)
    t = {k: m.simulate(it, m.behind(k)) for k in (3, 4, 5, 6)}
    assert t[4] < base - 8.0                      # ~11 us in the model, ~11 us on the hardware
    assert t[4] <= min(t.values()) + 1e-9 and t[6] > t[5] > t[4] and t[3] > t[4]
    lone = (it.sum(1).max() * m.LONE + it.shape[1] * m.SETUP) / (m.GHZ * 1e3)
    assert lone < t[4] < lone + 10.0              # what no rule can remove: the iterations before the straggler shows


def test_the_committed_counts_are_the_oracles_totals():
    from oracle import pagk_oracle as orc
    from pixel_aware_gyro_aided_klt_feature_tracker_amd import capi, synth
    it = np.load(os.path.join(ROOT, "tools", "data", "iters_cfg1_1000.npy")).astype(int)
    w = synth.config(1, n=1000)
    p = capi.make_params(half_patch=w.half_patch, iterations=w.iterations, pyramids=w.pyramids, has_gyro=w.has_gyro, camera=w.camera)
    ref = orc.track(p, w.img_ref, w.img_cur, w.pt_ref, w.pt_init, w.affine, w.status_in, nthreads=8)
    assert np.array_equal(it.sum(1), np.asarray(ref["iters"][:w.n], dtype=int))
