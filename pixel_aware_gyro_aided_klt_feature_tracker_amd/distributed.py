"""Feature sharding across the GPUs of one node (one process per GPU, torch.distributed).

The path shards by independent units: a feature's Gauss-Newton loop reads only the two
(replicated, read-only) pyramids and its own 33 bytes of input (reference
src/patch_match.cpp:167-367).  Rank r takes the contiguous index block
[r*ceil(n/G), min(n, (r+1)*ceil(n/G))), so index order -- and with it the order of the
tracker's f64 mean-pixel-error sum (src/gyro_aided_tracker.cpp:297-304) -- is the same as
on one GPU.  The only exchange is one all-gather of the packed per-rank result slice
(37 bytes per feature + 4 for the diagnostic iteration count); backend "nccl" is RCCL over
xGMI on ROCm, "gloo" is used by the CPU tests.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

# (name, torch dtype, elements per feature) in SetMatcher order (src/patch_match.cpp:370-388)
FIELDS = (("pt_un", torch.float32, 2), ("pt_dist", torch.float32, 2), ("status", torch.uint8, 1),
          ("pix_err", torch.float64, 1), ("dist_pred", torch.float64, 1), ("ncc", torch.float32, 1),
          ("iters", torch.int32, 1))
FORCE_COLLECTIVE = False  # tests: run the all-gather even with one rank
BYTES_PER_FEATURE = sum(torch.empty(0, dtype=dt).element_size() * k for _, dt, k in FIELDS)  # 41


COMM = None  # a capi.Multi: when set, the all-gather goes through the library's own RCCL communicator (C ABI)


def shard_size(n: int, world: int) -> int:
    return (n + world - 1) // world


def shard_range(n: int, rank: int, world: int) -> tuple[int, int]:
    """The partition is defined once, behind the C ABI (pagk_shard_range)."""
    from . import capi
    return capi.shard_range(n, rank, world)


def alloc_device_outputs(m: int, device) -> dict:
    """Per-rank output arrays for pagk_track_device, carved out of ONE byte buffer so that
    the all-gather ships a single contiguous slice.  SoA blocks, each 8-byte aligned."""
    from . import capi
    m = max(m, 1)
    o, total = capi.shard_layout(m)   # the slice layout is the library's (pagk_shard_layout)
    offs = {name: o[i] for i, (name, _, _) in enumerate(FIELDS)}
    buf = torch.zeros(total, dtype=torch.uint8, device=device)
    out = {"_buf": buf, "_m": m}
    for name, dt, k in FIELDS:
        nbytes = torch.empty(0, dtype=dt).element_size() * k * m
        v = buf[offs[name]:offs[name] + nbytes].view(dt)
        out[name] = v.view(m, k) if k > 1 else v
    return out


def views_of(buf: torch.Tensor, m: int) -> dict:
    """Typed views into one rank's packed slice (inverse of alloc_device_outputs)."""
    out, off = {}, 0
    for name, dt, k in FIELDS:
        nbytes = torch.empty(0, dtype=dt).element_size() * k * m
        v = buf[off:off + nbytes].view(dt)
        out[name] = v.view(m, k) if k > 1 else v
        off += (nbytes + 7) // 8 * 8
    return out


class Gathered:
    """The all-gathered packed slices of every rank, left packed on the device: nothing but the
    collective itself runs per step.  unpack() builds the full-length per-field tensors (feature-index
    order) when the host actually reads them."""

    def __init__(self, raw: torch.Tensor, world: int, m: int, n: int, slice_bytes: int):
        self.raw, self.world, self.m, self.n, self.slice_bytes = raw, world, m, n, slice_bytes
        # set by a caller that issued the collective on another stream than the reader's (runtime._sharded_step):
        # `done` = the collective has written `raw`; `consumed` is recorded by unpack() so that the writer of the NEXT
        # result into the same buffer can wait for this reader
        self.done = None
        self.consumed = None
        self.stale = False   # set when a later step's gather has been issued into `raw` (runtime.GatherRing)

    def unpack(self) -> dict:
        """Full-length per-field tensors on the caller's current stream, ordered after the collective."""
        if self.stale:
            raise RuntimeError("this gathered result's buffer has been reused by a later step; unpack a result before "
                               "the step after next is issued")
        if self.raw.is_cuda:
            cur = torch.cuda.current_stream(self.raw.device)
            if self.done is not None:
                cur.wait_event(self.done)
        parts = [views_of(self.raw[r * self.slice_bytes:(r + 1) * self.slice_bytes], self.m) for r in range(self.world)]
        out = {name: torch.cat([p[name] for p in parts], dim=0)[:self.n] for name, _, _ in FIELDS}
        if self.raw.is_cuda:
            ev = torch.cuda.Event()
            ev.record(cur)
            self.consumed = ev
        return out

    # dict-like access so that callers can treat it as the unpacked result
    def items(self):
        return self.unpack().items()

    def __getitem__(self, key):
        return self.unpack()[key]


def all_gather_results(local: dict, n: int, group=None, out: torch.Tensor | None = None):
    """One all-gather of every rank's packed slice (RCCL `ncclAllGather` of bytes under backend
    "nccl").  Returns the local views when there is a single rank, else a Gathered."""
    world = COMM.world if COMM is not None else (dist.get_world_size(group) if dist.is_initialized() else 1)
    m = local["_m"]
    if world == 1 and not ((dist.is_initialized() or COMM is not None) and FORCE_COLLECTIVE):
        return {name: local[name][:n] for name, _, _ in FIELDS}
    buf = local["_buf"]
    if out is None or out.numel() != world * buf.numel():
        out = torch.empty(world * buf.numel(), dtype=torch.uint8, device=buf.device)
    if COMM is not None:
        # the library's communicator: ncclAllGather issued by libpagk_hip.so on torch's current stream (the
        # tracker's), i.e. ordered after the tracking launch that filled `buf`
        COMM.allgather([buf], [out], buf.numel(), streams=[torch.cuda.current_stream(buf.device).cuda_stream])
    else:
        dist.all_gather_into_tensor(out, buf, group=group)
    return Gathered(out, world, m, n, buf.numel())


def to_numpy(full) -> dict:
    if isinstance(full, Gathered):
        full = full.unpack()
    return {k: (v.cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in full.items()}
