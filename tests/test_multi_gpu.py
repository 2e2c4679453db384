"""The sharded path behind the C ABI (include/pagk.h: pagk_multi_*, pagk_track_sharded) on the GPU box's one
device: a one-rank RCCL group must reproduce pagk_track bit for bit, the all-gather must go through the library's
own communicator, and the Python runtime must be able to route its gather through it.  (Groups of more than one
GPU cannot be formed on a one-GPU box: RCCL refuses two ranks on one device; the N > 1 partition / packing logic is
covered on CPU by tests/test_distributed_cpu.py.)"""
import numpy as np
import pytest

from pixel_aware_gyro_aided_klt_feature_tracker_amd import capi, synth
from util import assert_parity, params_for

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def group(built):
    g = capi.Multi([0])
    yield g
    g.close()


@pytest.mark.parametrize("n", [0, 1, 5, 1000])
def test_one_rank_group_equals_pagk_track(ctx, group, n):
    w = synth.config(1, n=max(n, 1))
    p = params_for(w)
    sl = slice(0, n)
    args = (w.img_ref, w.img_cur, w.pt_ref[sl].copy(), w.pt_init[sl].copy(), w.affine[sl].copy(), w.status_in[sl].copy())
    ref = ctx.track(p, *args)
    got = group.track_sharded(p, *args)
    assert group.world == 1 and group.n_local == 1
    assert_parity(got, ref, n, exact=True, what=f"pagk_track_sharded, 1 rank, n={n}")


def test_sharded_call_checks_its_arguments(group):
    w = synth.config(1, n=8)
    p = params_for(w)
    with pytest.raises(capi.PagkError) as e:
        group.track_sharded(p, w.img_ref, w.img_cur[:-2], w.pt_ref, w.pt_init, w.affine, w.status_in)
    assert e.value.code == capi.PAGK_E_ARG


def test_allgather_through_the_library_communicator(group):
    import torch
    dev = torch.device("cuda", 0)
    src = torch.arange(4096, dtype=torch.uint8, device=dev) * 3
    dst = torch.zeros(4096, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    group.allgather([src], [dst], 4096)
    group.ctx(0).sync()
    assert torch.equal(src, dst)


def test_runtime_gathers_through_the_c_abi_communicator(built):
    """ResidentTracker with distributed.COMM set: the per-step gather is pagk_multi_allgather (ncclAllGather issued
    by libpagk_hip.so on the tracker's stream), results equal to the ungathered ones."""
    import torch
    from pixel_aware_gyro_aided_klt_feature_tracker_amd import distributed, runtime
    w = synth.config(1, n=600)
    p = params_for(w)
    g = capi.Multi([0])
    # a different frame per step, and every result read WITHOUT a device-wide synchronisation in between: a reader of
    # step()'s Gathered must be ordered after that step's collective (Gathered.done), and the next step's gather must
    # not overwrite a result that is still being read (two buffers + Gathered.consumed)
    frames = [w.img_cur, np.roll(w.img_cur, 1, axis=1).copy(), np.roll(w.img_cur, -1, axis=0).copy()]
    plain = runtime.ResidentTracker(p, device=0)
    plain.load_pair(w.img_ref, w.img_cur)
    plain.set_features(w.pt_ref, w.pt_init, w.affine, w.status_in)
    refs = []
    for f in frames:
        plain.set_current_image(f)
        res = plain.step()
        plain.synchronize()   # (an ungathered result is plain views of the tracker's buffers: read them behind ITS stream --
        refs.append(distributed.to_numpy(res))   # reading first raced with the step whenever the two streams sat on different HW queues)
    assert not np.array_equal(refs[0]["pt_un"], refs[1]["pt_un"])
    distributed.COMM, distributed.FORCE_COLLECTIVE = g, True
    gots = []
    try:
        rt = runtime.ResidentTracker(p, device=0)
        rt.load_pair(w.img_ref, w.img_cur)
        rt.set_features(w.pt_ref, w.pt_init, w.affine, w.status_in)
        outs = []
        for f in frames:
            rt.set_current_image(f)
            out = rt.step()
            assert isinstance(out, distributed.Gathered)
            outs.append(out)
            if len(outs) >= 2:      # read the PREVIOUS step's result while this step's gather is in flight
                gots.append(distributed.to_numpy(outs[-2]))
        gots.append(distributed.to_numpy(outs[-1]))
        rt.synchronize()
        rt.close()
    finally:
        distributed.COMM, distributed.FORCE_COLLECTIVE = None, False
        plain.close()
        g.close()
    for got, ref in zip(gots, refs):
        for k in ("pt_un", "pt_dist", "status", "pix_err", "dist_pred", "ncc", "iters"):
            assert np.array_equal(got[k], ref[k], equal_nan=True), k
