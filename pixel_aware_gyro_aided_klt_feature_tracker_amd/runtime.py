"""Device-resident driver of the hot path for frame streams and benchmarks.

torch is used for what it is good at here -- device buffers, the current HIP stream,
torch.distributed -- and nothing else: every computation goes through the C ABI
(capi.Context -> libpagk_hip.so).  There is no CPU fallback.
"""
from __future__ import annotations

import numpy as np
import torch

from . import capi, distributed


def _event_on(stream):
    ev = torch.cuda.Event()
    ev.record(stream)
    return ev


class GatherRing:
    """The two alternating buffers of the sharded step's all-gather and the event ordering around them, apart from the
    streams so that the ordering can be tested without a GPU (tests/test_distributed_cpu.py).  The result of step k stays
    intact while step k+1's gather runs; the gather of step k+2 goes into step k's buffer: it waits for whoever unpacked
    that result (Gathered.consumed), and the old result is marked stale -- unpack() of a stale result raises instead of
    returning a later step's bytes.  `new_event(stream)` returns an event recorded on `stream`."""

    def __init__(self, new_event):
        self.bufs = [None, None]
        self.prev = [None, None]
        self.turn = 0
        self.new_event = new_event

    def issue(self, side, tracked, gather):
        """Order `gather(out_buffer)` on `side` behind `tracked` (the tracking launch that filled the local slice) and
        behind the reader of the buffer it is about to overwrite.  Returns (result, event: the collective is done)."""
        k = self.turn
        self.turn = 1 - k
        side.wait_event(tracked)
        prev = self.prev[k]
        if prev is not None:
            if prev.consumed is not None:
                side.wait_event(prev.consumed)
            prev.stale = True
        res = gather(self.bufs[k])
        done = self.new_event(side)
        if isinstance(res, distributed.Gathered):
            self.bufs[k] = res.raw
            res.done = done              # unpack() / to_numpy() on any stream wait for the collective
            self.prev[k] = res
        return res, done


class ResidentTracker:
    """One frame pair resident in HBM; step() = what a tracker does per new frame:
    build the pyramid of the new (current) frame, then run PatchMatch over this rank's
    feature shard.  The reference frame's pyramid is the one the previous step built
    (cur of pair t is ref of pair t+1), so it is not rebuilt."""

    def __init__(self, params: capi.Params, device: int = 0, rank: int = 0, world: int = 1, concurrency: int = 1):
        if not torch.cuda.is_available():
            raise RuntimeError("ResidentTracker needs a HIP device (torch.cuda.is_available() is False)")
        self.params = params
        self.rank, self.world = rank, world
        self.dev = torch.device("cuda", device)
        torch.cuda.set_device(self.dev)
        self.ctx = capi.Context(device)
        if concurrency > 1:   # this many trackers share the device (one per camera stream): pagk_set_concurrency
            self.ctx.set_concurrency(concurrency)
        # Two explicit streams.  `main` carries the tracking launches (and, because step() makes it torch's
        # current stream, the RCCL all-gather orders after them); `side` builds the NEXT frame's pyramid while
        # the current pair is being tracked (a new frame's pyramid depends on no tracking result).  Never the
        # legacy default stream: its handle is 0, which pagk_set_stream reads as "the context's own stream",
        # and it cannot be captured into a hipGraph.
        self.main = torch.cuda.Stream(device=self.dev)
        self.side = torch.cuda.Stream(device=self.dev)
        self.ctx.set_stream(self.main.cuda_stream)
        self.cur_slot = 1           # slots 1 / 2 alternate as "current frame"
        self._pyr_ready = None      # event: pyramid of the frame to track next is built
        self._trk_done = {1: None, 2: None}  # event per slot: last tracking kernel that read it
        self._graphs = {}           # mode -> graph ids of the captured step ("graph": one; "fork", "fused": two)
        self._graph_failed = False  # capture was refused once: direct launches from then on
        self.mode_used = "serial"   # how the last step's launches were issued
        self._gather_done = None    # event: the previous step's all-gather (side stream) has read self.out
        self._last_gather = None
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
        self._drop_graph()
        self._settle()

    def set_current_image(self, img_cur: np.ndarray):
        """The next frame of the stream: new bytes into the SAME device buffer (captured graphs stay valid), ordered
        on `main` after the launches that still read the old frame."""
        with torch.cuda.stream(self.main):
            self.img_cur.copy_(torch.from_numpy(np.ascontiguousarray(img_cur)).to(self.dev, non_blocking=False))

    def _settle(self):
        """Set-up work ran on torch's current stream and on `main`: let both finish before steps start."""
        torch.cuda.synchronize(self.dev)

    def _drop_graph(self):
        for ids in self._graphs.values():
            for gid in ids:
                self.ctx.graph_destroy(gid)
        self._graphs = {}
        if getattr(self, "_live", None) is not None:
            self.ctx.graph_destroy(self._live[1])
            self._live = None

    @property
    def _graph(self):
        """Graph ids of the default mode (None before its first step)."""
        return self._graphs.get("graph")

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

        same_shape = getattr(self, "d_pt_ref", None) is not None and self.d_pt_ref.shape[0] == max(m, 1) \
            and getattr(self, "_n_alloc", None) == self.n
        if same_shape:
            # a stream of frames with a fixed feature capacity (unused entries carry status_in = 0): new
            # values go into the SAME device arrays, so a captured graph stays valid
            with torch.cuda.stream(self.main):
                for dst, a, width in ((self.d_pt_ref, pt_ref, 2), (self.d_pt_init, pt_init, 2),
                                      (self.d_affine, affine, 4), (self.d_status, status_in, 1)):
                    if hi > lo:
                        dst[:hi - lo].copy_(torch.from_numpy(np.ascontiguousarray(a[lo:hi])).to(self.dev))
            self._settle()
            return
        self.d_pt_ref = up(pt_ref, 2)
        self.d_pt_init = up(pt_init, 2)
        self.d_affine = up(affine, 4)
        self.d_status = up(status_in, 1)
        self.out = distributed.alloc_device_outputs(m, self.dev)
        self._n_alloc = self.n
        self._drop_graph()
        self._settle()

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

    def step(self, gather: bool = True, mode: str = "graph"):
        """One pass of the hot path over this rank's shard (+ the result all-gather): pyramid of the
        current frame, then PatchMatch.  `mode` only chooses how the two launches reach the GPU; every
        mode builds exactly one pyramid and runs one tracking launch per step (measured on MI355X,
        752x480 / 1000 features, tools/step_modes.py, profiles/r01_step_modes.log):
          "graph"   (default) one hipGraphLaunch replaying [pyramid -> PatchMatch], captured on first use
                    (BASELINE configs[4]: "hipGraph-captured iterate"): no inter-launch gap, 121.7 us;
          "serial"  the same two launches issued directly on one stream: 132.9 us (5 us gap per launch);
          "streams" pyramid of step k+1's frame on a side stream while step k tracks, ordered by events:
                    134.6 us -- a cross-stream event wait costs more than the 7.7 us pyramid it hides;
          "fork"    that overlap as two branches of one graph: 143.6 us;
          "fused"   the next frame's pyramid as trailing workgroups of the tracking launch itself
                    (pagk_track_device_fused), one single-node graph per parity: the pyramid costs no launch and no
                    gap.  Like "streams" it needs frame k+1 while pair (k-1, k) is tracked.
        What comes back without a collective is plain VIEWS of the tracker's output buffers, written by launches on `main`:
        read them after synchronize() or on `main` (a Gathered, from the sharded step, orders its reader by itself)."""
        gather = gather and not mode.endswith("-nogather")
        mode = mode.replace("-nogather", "")
        graph = {"graph": True, "fork": "fork", "fused": "fused"}.get(mode, False)
        overlap = mode == "streams"
        if mode not in ("graph", "fork", "fused", "serial", "streams"):
            raise ValueError(mode)
        collective = gather and (self.world > 1 or distributed.FORCE_COLLECTIVE)
        if collective:
            return self._sharded_step(mode)
        with torch.cuda.stream(self.main):
            if graph and not self._graph_failed:
                try:
                    self._graph_step(fork=(graph == "fork"), fused=(graph == "fused"))
                except capi.PagkError as e:
                    # capture refused by the runtime (never seen on MI355X / ROCm 7.2): keep going with the same two
                    # launches issued directly -- still the HIP path -- and say so
                    import sys
                    print(f"pagk: hipGraph capture failed ({e}); step() falls back to direct launches", file=sys.stderr)
                    self._graph_failed = True
                    self._drop_graph()
                    self.ctx.set_stream(self.main.cuda_stream)
                    self.mode_used = "serial"
                    self.rebuild_current_pyramid(1)
                    self.track_shard(1)
                else:
                    self.mode_used = mode
            elif graph:
                self.mode_used = "serial"
                self.rebuild_current_pyramid(1)
                self.track_shard(1)
            elif not overlap:
                self.mode_used = "serial"
                self.rebuild_current_pyramid(1)
                self.track_shard(1)
            else:
                self.mode_used = "streams"
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
        return {name: self.out[name][:self.hi - self.lo] for name, _, _ in distributed.FIELDS}

    def _sharded_step(self, mode: str):
        """A step of the sharded path (world > 1): pyramid, this rank's block of features, then the one exchange of
        the path -- the all-gather of the packed result slices -- on the SIDE stream, so that it runs beside the next
        step's pyramid; the next step's tracking launch waits for it (it rewrites the slice the gather reads, and in
        a tracker its inputs derive from the gathered results, src/gyro_aided_tracker.cpp:289-341).  The tracking
        launch is replayed from a one-node hipGraph when `mode` is a graph mode."""
        with torch.cuda.stream(self.main):
            self.rebuild_current_pyramid(1)                        # needs no result of the previous step
            if self._gather_done is not None:
                self.main.wait_event(self._gather_done)
            if mode in ("graph", "fused", "fork") and not self._graph_failed:
                if "track" not in self._graphs:
                    self.track_shard(1)                            # warm-up
                    self.main.synchronize()
                    try:
                        self.ctx.graph_begin()
                        try:
                            self.track_shard(1)
                        finally:
                            self._graphs["track"] = (self.ctx.graph_end(),)
                    except capi.PagkError:
                        self._graph_failed = True
                        self._graphs.pop("track", None)
                if "track" in self._graphs:
                    self.ctx.graph_launch(self._graphs["track"][0])
                    self.mode_used = "graph"
                else:
                    self.track_shard(1)
                    self.mode_used = "serial"
            else:
                self.track_shard(1)
                self.mode_used = "serial"
            tracked = torch.cuda.Event()
            tracked.record(self.main)
        ring = getattr(self, "_gather_ring", None)
        if ring is None:
            ring = self._gather_ring = GatherRing(_event_on)
        with torch.cuda.stream(self.side):
            res, self._gather_done = ring.issue(self.side, tracked, lambda out: distributed.all_gather_results(self.out, self.n, out=out))
        self._last_gather = res
        return res

    def step_live(self, host_frame: "torch.Tensor"):
        """The step of a live camera loop INCLUDING the frame's way onto the device: `host_frame` is the camera's PINNED
        uint8 buffer (h x w); the graph [host -> device copy of that buffer -> pyramid -> PatchMatch] is captured on
        first use (pagk_frame_upload_pinned is capturable) and replayed with one launch per frame -- the caller writes
        the new frame into the same pinned buffer before each call.  (Examples/Demo/RealSenseD435i.cpp:199-321 is the
        loop this stands for: grab, convert, track.)  Returns the rank's device outputs; nothing is synchronised."""
        key = (host_frame.data_ptr(), tuple(host_frame.shape))
        if getattr(self, "_live", None) is None or self._live[0] != key:
            if getattr(self, "_live", None) is not None:
                self.ctx.graph_destroy(self._live[1])
            if not host_frame.is_pinned():
                raise ValueError("step_live needs the frame in pinned host memory")
            h, w = host_frame.shape
            with torch.cuda.stream(self.main):
                self.ctx.frame_upload_pinned(1, host_frame.data_ptr(), w, h, host_frame.stride(0), self.params.pyramids)
                self.track_shard(1)                      # warm-up: allocations, kernel attributes
                self.main.synchronize()
                self.ctx.graph_begin()
                try:
                    self.ctx.frame_upload_pinned(1, host_frame.data_ptr(), w, h, host_frame.stride(0), self.params.pyramids)
                    self.track_shard(1)
                finally:
                    gid = self.ctx.graph_end()
            self._live = (key, gid, host_frame)          # (keeps the pinned buffer alive as long as the graph)
        self.ctx.graph_launch(self._live[1])
        self.mode_used = "live"
        return self.out

    def finish(self):
        """Order everything issued so far (tracking on `main`, the last gather on `side`) before the caller's
        stream-wide synchronisation."""
        if self._gather_done is not None:
            self.main.wait_event(self._gather_done)

    def gather_report(self, reps: int = 20) -> dict:
        """The all-gather by itself: average duration from events on the stream it runs on."""
        ts = []
        with torch.cuda.stream(self.side):
            for _ in range(reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(self.side)
                distributed.all_gather_results(self.out, self.n, out=(self._gather_ring.bufs[0] if getattr(self, "_gather_ring", None) else None))
                e1.record(self.side)
                e1.synchronize()
                ts.append(e0.elapsed_time(e1))
        return {"gather_ms": float(np.mean(ts[2:])) if len(ts) > 2 else float(np.mean(ts)),
                "bytes_per_rank": int(self.out["_buf"].numel()), "ranks": self.world}

    def track_shard_fused(self, cur: int):
        """PatchMatch(0, cur) and, in the same launch, the pyramid of the next frame into the other slot."""
        self.ctx.track_device_fused(self.params, 0, cur, self.hi - self.lo, self.d_pt_ref, self.d_pt_init, self.d_affine,
                                    self.d_status, self.out, 3 - cur, self.img_cur.data_ptr(), self.w, self.h, self.w,
                                    self.params.pyramids)

    def _graph_step(self, fork: bool, fused: bool = False):
        """Replay (capturing on first use) the step as a hipGraph.  Linear form: [pyramid(slot 1) ->
        PatchMatch(0, 1)].  Fork form: two graphs, one per parity, each [PatchMatch(0, cur) || pyramid(next)]
        -- the side-stream prefetch of step() expressed as two independent branches of one graph.  Fused form:
        two single-node graphs, [PatchMatch(0, cur) + pyramid(next frame -> the other slot) in one launch]."""
        key = "fused" if fused else ("fork" if fork else "graph")
        if key not in self._graphs:
            self.rebuild_current_pyramid(1)      # warm-up: allocations, kernel attributes
            self.rebuild_current_pyramid(2)
            self.track_shard(1)
            self.main.synchronize()
            ids = []
            if fused:
                self.track_shard_fused(1)        # warm-up of the fused kernel
                self.main.synchronize()
                for cur in (1, 2):
                    self.ctx.graph_begin()
                    try:
                        self.track_shard_fused(cur)
                    finally:
                        ids.append(self.ctx.graph_end())
            elif not fork:
                self.ctx.graph_begin()
                try:
                    self.rebuild_current_pyramid(1)
                    self.track_shard(1)
                finally:
                    ids.append(self.ctx.graph_end())
            else:
                for cur in (1, 2):
                    self.ctx.graph_begin()       # capture starts on `main`
                    try:
                        e1 = torch.cuda.Event()
                        e1.record(self.main)
                        self.side.wait_event(e1)                 # fork: `side` joins the capture
                        self.ctx.set_stream(self.side.cuda_stream)
                        self.rebuild_current_pyramid(3 - cur)    # next frame's pyramid
                        self.ctx.set_stream(self.main.cuda_stream)
                        self.track_shard(cur)
                        e2 = torch.cuda.Event()
                        e2.record(self.side)
                        self.main.wait_event(e2)                 # join
                    finally:
                        self.ctx.set_stream(self.main.cuda_stream)
                        ids.append(self.ctx.graph_end())
            self._graphs[key] = tuple(ids)
            self.cur_slot = 1
        ids = self._graphs[key]
        if len(ids) == 1:
            self.ctx.graph_launch(ids[0])
        else:
            # both slots hold a valid pyramid whenever the mode changes: every parity graph leaves them so
            self.ctx.graph_launch(ids[self.cur_slot - 1])
            self.cur_slot = 3 - self.cur_slot

    def synchronize(self):
        """Wait for everything step() has issued (tracking on `main`, prefetch / gather on `side`); then ask the library
        whether one of those launches failed -- these are torch's stream synchronisations, not pagk_sync, so the error
        word of the level-by-level launches (include/pagk.h: pagk_check_launch) would otherwise never be read."""
        self.main.synchronize()
        self.side.synchronize()
        self.ctx.check_launch()


class CameraBatch:
    """k camera streams that share ONE GPU, stepped as one launch: BASELINE configs[4], "batched multi-camera: concurrent
    streams, hipGraph-captured iterate".  The reference builds one PatchMatch per tracker
    (src/gyro_aided_tracker.cpp:276-283); here every stream is a ResidentTracker (its own context: frame slots, feature
    arrays, outputs), all switched to ONE stream, and a step is [the k current frames' pyramids as one launch
    (pagk_frame_set_device_batch), then pagk_track_device_batch] -- replayed as one hipGraph after the first step, or issued directly."""

    def __init__(self, params: capi.Params, k: int, device: int = 0):
        self.params = params
        self.cams = [ResidentTracker(params, device=device) for _ in range(k)]
        self.stream = self.cams[0].main
        for c in self.cams[1:]:
            c.ctx.set_stream(self.stream.cuda_stream)   # one stream for the whole batch: nothing to order across streams
        self._graph = None
        self.mode_used = "serial"

    def close(self):
        self._drop_graph()
        for c in self.cams:
            c.close()

    def _drop_graph(self):
        if self._graph is not None:
            self.cams[0].ctx.graph_destroy(self._graph)
            self._graph = None

    def load(self, j: int, img_ref, img_cur, pt_ref, pt_init, affine, status_in):
        """Stream j's frame pair and features (any image size, any feature count)."""
        c = self.cams[j]
        c.load_pair(img_ref, img_cur)
        c.set_features(pt_ref, pt_init, affine, status_in)
        c.ctx.set_stream(self.stream.cuda_stream)
        self._drop_graph()

    def _issue(self):
        cs = self.cams
        # the k current frames' pyramids as one launch, then the k trackers' PatchMatch as one launch
        capi.Context.frame_set_device_batch([c.ctx for c in cs], [1] * len(cs), [c.img_cur.data_ptr() for c in cs],
                                            [c.w for c in cs], [c.h for c in cs], [c.w for c in cs], self.params.pyramids)
        capi.Context.track_device_batch([c.ctx for c in cs], self.params, [0] * len(cs), [1] * len(cs),
                                        [c.hi - c.lo for c in cs], [c.d_pt_ref for c in cs], [c.d_pt_init for c in cs],
                                        [c.d_affine for c in cs], [c.d_status for c in cs], [c.out for c in cs])

    def step(self, mode: str = "graph"):
        """One frame of every stream; returns the streams' output dicts (device tensors, valid once `stream` has run)."""
        lead = self.cams[0].ctx
        with torch.cuda.stream(self.stream):
            if mode == "graph":
                if self._graph is None:
                    self._issue()                      # warm-up: allocations happen outside the capture
                    self.stream.synchronize()
                    lead.graph_begin()
                    try:
                        self._issue()
                    finally:
                        self._graph = lead.graph_end()
                lead.graph_launch(self._graph)
            else:
                self._issue()
        self.mode_used = mode
        return [c.out for c in self.cams]

    def synchronize(self):
        self.stream.synchronize()
        for c in self.cams:
            c.ctx.check_launch()
