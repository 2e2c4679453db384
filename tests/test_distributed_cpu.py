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


@pytest.mark.parametrize("n", [37, 64, 1])
def test_sharded_gather_equals_single_process(built, n):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + n) % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(120)
        assert pr.exitcode == 0
    assert q.get(timeout=5) is True


def test_shard_ranges_cover_in_order():
    from pixel_aware_gyro_aided_klt_feature_tracker_amd import distributed
    for n in (0, 1, 7, 8, 9, 20000):
        for world in (1, 2, 4, 8):
            spans = [distributed.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert all(hi - lo <= distributed.shard_size(n, world) for lo, hi in spans)
