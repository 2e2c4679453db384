"""world_size-2 gloo test of the feature sharding + result all-gather (the N>1 path).
The per-shard computation is stood in by the CPU oracle here (no GPU in this container);
on the GPU box bench.py runs the same module over RCCL with the HIP path."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import pagk_oracle as orc
    from pixel_aware_gyro_aided_klt_feature_tracker_amd import capi, distributed, synth
    w = synth.make_workload("d", 160, 120, n, seed=0x5EEDD157, half_patch=5, iterations=10, pyramids=3)
    p = capi.make_params(half_patch=5, iterations=10, pyramids=3, has_gyro=w.has_gyro, camera=w.camera)
    lo, hi = distributed.shard_range(n, rank, world)
    m = distributed.shard_size(n, world)
    local = distributed.alloc_device_outputs(m, "cpu")
    if hi > lo:
        out = orc.track(p, w.img_ref, w.img_cur, w.pt_ref[lo:hi].copy(), w.pt_init[lo:hi].copy(),
                        w.affine[lo:hi].copy(), w.status_in[lo:hi].copy(), nthreads=1)
        for name, _, _ in distributed.FIELDS:
            local[name][:hi - lo] = torch.from_numpy(out[name][:hi - lo])
    full = distributed.to_numpy(distributed.all_gather_results(local, n))
    if rank == 0:
        ref = orc.track(p, w.img_ref, w.img_cur, w.pt_ref, w.pt_init, w.affine, w.status_in, nthreads=1)
        ok = all(np.array_equal(full[name], ref[name][:n]) for name, _, _ in distributed.FIELDS)
        # the global post-filter (mean pixel error over ALL shards) needs the gathered vectors
        a = capi.post_filter(5, full["status"], full["pix_err"], full["dist_pred"], full["pt_dist"], full["pt_un"])
        b = orc.post_filter(5, ref["status"][:n], ref["pix_err"][:n], ref["dist_pred"][:n], ref["pt_dist"][:n],
                            ref["pt_un"][:n])
        ok = ok and a[0] == b[0] and np.array_equal(a[1], b[1])
        q.put(bool(ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n,world", [(37, 2), (64, 2), (1, 2), (7, 3), (4, 3)])
def test_sharded_gather_equals_single_process(built, n, world):
    # (7, 3): ragged last shard; (4, 3): ceil(4/3) = 2 -> shards of 2, 2 and an EMPTY trailing shard; (1, 2): one
    # rank has nothing to track
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + 17 * n + world) % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(120)
        assert pr.exitcode == 0
    assert q.get(timeout=5) is True


def test_shard_ranges_cover_in_order(built):
    from pixel_aware_gyro_aided_klt_feature_tracker_amd import distributed
    for n in (0, 1, 4, 7, 8, 9, 20000):
        for world in (1, 2, 3, 4, 8):
            spans = [distributed.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert all(hi - lo <= distributed.shard_size(n, world) for lo, hi in spans)


def test_partition_and_slice_layout_behind_the_c_abi(built):
    """pagk_shard_range / pagk_shard_layout (include/pagk.h) against their definitions: contiguous blocks of
    ceil(n / G); seven SoA blocks in SetMatcher order, each padded to 8 bytes."""
    from pixel_aware_gyro_aided_klt_feature_tracker_amd import capi, distributed
    for n in (0, 1, 4, 5, 37, 1000, 20000):
        for world in (1, 2, 3, 8):
            m = -(-n // world)
            for r in range(world):
                lo, hi = capi.shard_range(n, r, world)
                assert (lo, hi) == (min(n, r * m), min(n, r * m + m))
    elem = [8, 8, 1, 8, 8, 4, 4]
    for m in (0, 1, 3, 8, 125, 2500):
        offs, total = capi.shard_layout(m)
        want, t = [], 0
        for e in elem:
            want.append(t)
            t += (e * max(m, 1) + 7) // 8 * 8
        assert offs == want and total == t
        out = distributed.alloc_device_outputs(m, "cpu")     # the Python runtime carves its views from the same layout
        assert out["_buf"].numel() == total
        assert sum(torch.empty(0, dtype=dt).element_size() * k for _, dt, k in distributed.FIELDS) == sum(elem)


def test_group_creation_fails_loudly_without_a_device(built):
    import torch as _t
    if _t.cuda.is_available():
        pytest.skip("needs a machine without a HIP device")
    from pixel_aware_gyro_aided_klt_feature_tracker_amd import capi
    with pytest.raises(capi.PagkError) as e:
        capi.Multi([0])
    assert e.value.code == capi.PAGK_E_NODEVICE


def test_gather_ring_orders_two_alternating_buffers(built):
    """the event ordering of the sharded step's two gather buffers, with stand-ins for the streams and the
    collective (runtime.GatherRing is the code ResidentTracker._sharded_step runs; no GPU needed): a gather into a buffer
    comes after the tracking launch of its step and after the reader of the result that lived in that buffer, the result
    of step k survives step k + 1, and a result whose buffer has been handed to a later gather refuses to unpack."""
    from pixel_aware_gyro_aided_klt_feature_tracker_amd import distributed, runtime

    log = []

    class Stream:
        def wait_event(self, ev):
            log.append(("wait", ev))

    issued = []

    def new_event(stream):
        ev = "done%d" % len(issued)
        log.append(("record", ev))
        return ev

    m = 5
    nbytes = distributed.alloc_device_outputs(m, "cpu")["_buf"].numel()

    def gather(out):
        raw = out if out is not None else torch.zeros(2 * nbytes, dtype=torch.uint8)
        raw.fill_(len(issued) + 1)                         # (the collective writes the buffer)
        log.append(("gather", raw.data_ptr()))
        issued.append(distributed.Gathered(raw, 2, m, 2 * m - 1, nbytes))
        return issued[-1]

    ring, side = runtime.GatherRing(new_event), Stream()
    r0, d0 = ring.issue(side, "tracked0", gather)
    assert log == [("wait", "tracked0"), ("gather", r0.raw.data_ptr()), ("record", "done1")] and r0.done == d0 == "done1"
    r0.consumed = "read0"                                  # (what unpack() records on a device result)
    del log[:]
    r1, d1 = ring.issue(side, "tracked1", gather)
    assert r1.raw.data_ptr() != r0.raw.data_ptr(), "step 1 must not overwrite the result of step 0"
    assert log == [("wait", "tracked1"), ("gather", r1.raw.data_ptr()), ("record", "done2")] and not r0.stale
    assert int(r0.raw[0]) == 1 and set(r0.unpack()) == {name for name, _, _ in distributed.FIELDS}
    del log[:]
    r2, d2 = ring.issue(side, "tracked2", gather)          # back in buffer 0: behind step 0's reader
    assert r2.raw.data_ptr() == r0.raw.data_ptr()
    assert log == [("wait", "tracked2"), ("wait", "read0"), ("gather", r2.raw.data_ptr()), ("record", "done3")]
    assert r0.stale and not r1.stale and not r2.stale
    with pytest.raises(RuntimeError, match="reused"):
        r0.unpack()
    del log[:]
    r3, _ = ring.issue(side, "tracked3", gather)           # buffer 1; step 1's result was never unpacked: nobody to wait for
    assert r3.raw.data_ptr() == r1.raw.data_ptr() and r1.stale
    assert log == [("wait", "tracked3"), ("gather", r3.raw.data_ptr()), ("record", "done4")]
    # a single-rank step returns the local views, not a Gathered: the ring keeps nothing of it
    plain = runtime.GatherRing(new_event)
    res, _ = plain.issue(side, "t", lambda out: {"status": None})  # (world 1: all_gather_results returns the local views)
    assert res == {"status": None} and plain.prev == [None, None] and plain.bufs == [None, None]
