"""Device-resident driver of the hot path for frame streams and benchmarks.

torch is used for what it is good at here -- device buffers, the current HIP stream,
torch.distributed -- and nothing else: every computation goes through the C ABI
(capi.Context -> libpagk_hip.so).  There is no CPU fallback.
"""
from __future__ import annotations

import numpy as np
import torch

from . import capi, distributed


class ResidentTracker:
    """One frame pair resident in HBM; step() = what a tracker does per new frame:
    build the pyramid of the new (current) frame, then run PatchMatch over this rank's
    feature shard.  The reference frame's pyramid is the one the previous step built
    (cur of pair t is ref of pair t+1), so it is not rebuilt."""

    def __init__(self, params: capi.Params, device: int = 0, rank: int = 0, world: int = 1):
        if not torch.cuda.is_available():
            raise RuntimeError("ResidentTracker needs a HIP device (torch.cuda.is_available() is False)")
        self.params = params
        self.rank, self.world = rank, world
        self.dev = torch.device("cuda", device)
        torch.cuda.set_device(self.dev)
        self.ctx = capi.Context(device)
        # run on torch's current stream so that torch.cuda.Event / torch.distributed order with us
        self.main = torch.cuda.current_stream(self.dev)   # the stream current at construction time
        self.ctx.set_stream(self.main.cuda_stream)
        # side stream: the NEXT frame's pyramid is built while the current pair is being tracked (a
        # new frame's pyramid does not depend on any tracking result)
        self.side = torch.cuda.Stream(device=self.dev)
        self.cur_slot = 1           # slots 1 / 2 alternate as "current frame"
        self._pyr_ready = None      # event: pyramid of the frame to track next is built
        self._trk_done = {1: None, 2: None}  # event per slot: last tracking kernel that read it
        self.n = 0

    def close(self):
        self.ctx.close()

    def load_pair(self, img_ref: np.ndarray, img_cur: np.ndarray):
        self.img_ref = torch.from_numpy(np.ascontiguousarray(img_ref)).to(self.dev)
        self.img_cur = torch.from_numpy(np.ascontiguousarray(img_cur)).to(self.dev)
        h, w = img_ref.shape
        self.w, self.h = w, h
        L = self.params.pyramids
        self.ctx.frame_set_device(0, self.img_ref.data_ptr(), w, h, w, L)
        self.ctx.frame_set_device(1, self.img_cur.data_ptr(), w, h, w, L)
        self.ctx.frame_set_device(2, self.img_cur.data_ptr(), w, h, w, L)
        self.cur_slot, self._pyr_ready = 1, None

    def set_features(self, pt_ref, pt_init, affine, status_in):
        """Takes the FULL feature arrays; keeps this rank's contiguous shard on the device."""
        self.n = int(pt_ref.shape[0])
        lo, hi = distributed.shard_range(self.n, self.rank, self.world)
        self.lo, self.hi = lo, hi
        m = distributed.shard_size(self.n, self.world)

        def up(a, width):
            t = torch.zeros((max(m, 1), width) if width > 1 else (max(m, 1),), dtype=torch.from_numpy(a).dtype,
                            device=self.dev)
            if hi > lo:
                t[:hi - lo] = torch.from_numpy(np.ascontiguousarray(a[lo:hi])).to(self.dev)
            return t

        self.d_pt_ref = up(pt_ref, 2)
        self.d_pt_init = up(pt_init, 2)
        self.d_affine = up(affine, 4)
        self.d_status = up(status_in, 1)
        self.out = distributed.alloc_device_outputs(m, self.dev)

    def rebuild_current_pyramid(self, slot: int = 1):
        self.ctx.frame_set_device(slot, self.img_cur.data_ptr(), self.w, self.h, self.w, self.params.pyramids)

    def track_shard(self, slot: int = 1):
        self.ctx.track_device(self.params, 0, slot, self.hi - self.lo, self.d_pt_ref, self.d_pt_init, self.d_affine,
                              self.d_status, self.out)

    def _prefetch_pyramid(self, slot: int):
        """Build the pyramid of the next 'current' frame into `slot` on the side stream."""
        with torch.cuda.stream(self.side):
            if self._trk_done[slot] is not None:
                self.side.wait_event(self._trk_done[slot])   # the kernel that last read this slot
            self.ctx.set_stream(self.side.cuda_stream)
            self.rebuild_current_pyramid(slot)
            ev = torch.cuda.Event()
            ev.record(self.side)
        self.ctx.set_stream(self.main.cuda_stream)
        return ev

    def step(self, gather: bool = True, overlap: bool = True):
        """One pass of the hot path over this rank's shard (+ the result all-gather): pyramid of the
        current frame, PatchMatch.  With overlap (default) the pyramid of step k+1's frame is built on a
        side stream while step k tracks; every step still builds exactly one pyramid and runs one
        tracking launch."""
        if not overlap:
            self.rebuild_current_pyramid(1)
            self.track_shard(1)
        else:
            if self._pyr_ready is None:                       # first step: nothing prefetched yet
                self._pyr_ready = self._prefetch_pyramid(self.cur_slot)
            self.main.wait_event(self._pyr_ready)
            slot = self.cur_slot
            self.track_shard(slot)
            done = torch.cuda.Event()
            done.record(self.main)
            self._trk_done[slot] = done
            self.cur_slot = 3 - slot                          # 1 <-> 2
            self._pyr_ready = self._prefetch_pyramid(self.cur_slot)
        if gather and (self.world > 1 or distributed.FORCE_COLLECTIVE):
            res = distributed.all_gather_results(self.out, self.n, out=getattr(self, "_gather_buf", None))
            if isinstance(res, distributed.Gathered):
                self._gather_buf = res.raw
            return res
        return {name: self.out[name][:self.hi - self.lo] for name, _, _ in distributed.FIELDS}
