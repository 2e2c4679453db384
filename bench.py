#!/usr/bin/env python3
"""bench.py -- tracked features / second of the pyramidal patch-based KLT refinement
(21x21 patch, 3 levels, <= 30 Gauss-Newton iterations) on N MI355X GPUs of one node.

  python bench.py --gpus N --steps K --warmup W            (N > 1: spawns its N ranks itself)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

A "step" is what a tracker does per new frame: build the pyramid of the current frame
(CreatePyramids) and run PatchMatch over the frame's keypoints, everything already resident in
HBM.  Workload at N=1: BASELINE.json configs[1] (752x480, 1000 keypoints, gyro-predicted affine
init; synthetic stand-in).  For N>1, `--scaling weak` (default) tracks
N x 1000 keypoints on the same pair and `--scaling strong --config 3` the 1080p / 20000-keypoint
pair of configs[3]; either way the keypoints are sharded in contiguous index blocks, one process
per GPU, and every step ends with the RCCL all-gather of the per-rank (pt, status, err) slices.
`--mode replicas --config 4` is BASELINE configs[4] as written: camera stream s lives on GPU s (the whole 1280x720 x
4000-keypoint pair per GPU, the step a replayed hipGraph), no collective on the data path; the line's `value` is the
aggregate over the N streams and `per_gpu_ms_per_step` lists each GPU's own step time.
Rank 0 prints ONE JSON line; `config.mode` says which of the three forms (shard-weak, shard-strong, replicas) ran.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# (before torch or anything else can initialise the HIP runtime: the package sets its process-level runtime defaults on import)
from pixel_aware_gyro_aided_klt_feature_tracker_amd import runtime_env  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
NOMINAL_N = {0: 500, 1: 1000, 2: 2000, 3: 20000, 4: 4000}   # keypoints BASELINE.json quotes per config


def algorithmic_bytes_per_feature(h: int, L: int) -> int:
    """L*[(2h+2)^2 + (2h+4)^2] window bytes + 33 B in + 37 B out."""
    return L * ((2 * h + 2) ** 2 + (2 * h + 4) ** 2) + 70


def host_cores() -> int:
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota (the GPU
    box exposes every logical CPU of the host but gives one GPU's job a 16-core share)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("PAGK_CPU_THREADS", "16"))))


def measured_traffic(workload_name: str, n: int, kernel: str):
    """HBM bytes per launch of `kernel` from the newest committed PMC profile of THIS workload
    (profiles/<tag>/pmc_summary.json next to the bench.json it was collected with; tools/profile.sh,
    separate --pmc passes)."""
    try:
        pdir = os.path.join(ROOT, "profiles")
        for tag in sorted((d for d in os.listdir(pdir) if os.path.isdir(os.path.join(pdir, d))), reverse=True):
            try:
                with open(os.path.join(pdir, tag, "pmc_summary.json")) as f:
                    prof = json.load(f)
                with open(os.path.join(pdir, tag, "bench.json")) as f:
                    ref = json.load(f)
            except Exception:
                continue
            if ref["config"]["workload"].split(":")[0] != workload_name or ref["config"]["features_total"] != n:
                continue
            for name, v in prof.items():
                if f"::{kernel}<" in name and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
                    return (2.0 * v["FETCH_SIZE"]["mean"] + v["WRITE_SIZE"]["mean"]) * 1024.0, tag
    except Exception:
        pass
    return None, None


KERNEL_OF_VARIANT = {0: "k_track_block", 1: "k_track_thread", 2: "k_track_block", 3: "k_track_wave", 4: "k_track_block",
                     5: "k_track_quad", 6: "k_track_rows", 7: "k_track_quad"}


def metric_label(w) -> str:
    """BASELINE.json's metric with the workload's own patch / levels / iteration cap (configs[2] runs 4 levels)."""
    side = 2 * w.half_patch + 1
    return f"tracked features/sec ({side}x{side}, {w.pyramids}-lvl, {w.iterations} iter)"


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as child processes (this parent never
    touches the GPU), one per GPU, rendezvous on 127.0.0.1.  Rank 0's stdout is ours."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


def time_steps(rt, steps: int, warmup: int, mode: str, barrier=None):
    import torch
    for _ in range(warmup):
        out = rt.step(mode=mode)
    torch.cuda.synchronize()
    if barrier:
        barrier()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = rt.step(mode=mode)
    rt.finish()
    torch.cuda.synchronize()
    if barrier:
        barrier()
        torch.cuda.synchronize()
    return time.perf_counter() - t0, out


def kernel_time_ms(rt, reps: int) -> tuple[float, float]:
    """Average duration of the tracking launch and of the pyramid launch: HIP events recorded by the library on
    the stream the kernels run on (pagk_last_kernel_ms); untimed extra launches, outside any graph."""
    import numpy as np
    import torch
    trk, pyr = [], []
    torch.cuda.synchronize()
    with torch.cuda.stream(rt.main):
        for _ in range(reps):
            rt.rebuild_current_pyramid(1)
            rt.track_shard(1)
            a, b = rt.ctx.last_kernel_ms()
            trk.append(a)
            pyr.append(b)
    return float(np.mean(trk)), float(np.mean(pyr))


def config_row(cfg_idx: int, n: int, device: int, steps: int, streams: int = 1) -> dict:
    """One BASELINE config on this GPU: `streams` independent resident trackers of that shape stepped as a replayed
    hipGraph and as direct launches (the faster mode is the row's ms_per_step; both are listed), plus the tracking
    kernel's own time and roofline figures."""
    import numpy as np
    import torch
    from pixel_aware_gyro_aided_klt_feature_tracker_amd import capi, distributed, runtime, synth
    w = synth.config(cfg_idx, n=n)
    p = capi.make_params(half_patch=w.half_patch, iterations=w.iterations, pyramids=w.pyramids, has_gyro=w.has_gyro,
                         camera=w.camera)
    cams = []
    for _ in range(streams):
        rt = runtime.ResidentTracker(p, device=device, concurrency=streams)
        rt.load_pair(w.img_ref, w.img_cur)
        rt.set_features(w.pt_ref, w.pt_init, w.affine, w.status_in)
        cams.append(rt)
    # the step as a replayed hipGraph and as direct launches; the row reports the faster of the two (a long launch
    # gains nothing from the replay, and only a direct launch runs the straggler finisher beside the kernel)
    times = {}
    for mode in ("graph", "serial"):
        # (the same kind of spin-up as the headline's, shorter: the first steps after a pause run slow)
        t_spin = time.perf_counter()
        while time.perf_counter() - t_spin < 0.15:
            for _ in range(3):
                for rt in cams:
                    out = rt.step(mode=mode)
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            for rt in cams:
                out = rt.step(mode=mode)
        torch.cuda.synchronize()
        times[mode] = (time.perf_counter() - t0) / steps
    batch_variant = None
    if streams > 1:
        # BASELINE configs[4] as written -- "batched multi-camera ... hipGraph-captured iterate": the same streams as ONE
        # launch per step (pagk_track_device_batch behind runtime.CameraBatch), replayed as a graph and issued directly
        cb = runtime.CameraBatch(p, streams, device=device)
        for j in range(streams):
            cb.load(j, w.img_ref, w.img_cur, w.pt_ref, w.pt_init, w.affine, w.status_in)
        for mode in ("graph", "serial"):
            t_spin = time.perf_counter()
            while time.perf_counter() - t_spin < 0.15:
                for _ in range(3):
                    bout = cb.step(mode=mode)
                torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                bout = cb.step(mode=mode)
            torch.cuda.synchronize()
            times["batch_" + mode] = (time.perf_counter() - t0) / steps
        cb.synchronize()
        batch_variant = capi.Context.VARIANT_NAMES.get(cb.cams[0].ctx.last_variant(), "?")
        # every stream of the batch against the per-context result of the same workload
        ref_out = distributed.to_numpy(out)
        batch_equal = all(bool(np.array_equal(distributed.to_numpy(o)[key][:w.n], ref_out[key][:w.n]))
                          for o in bout for key in ("status", "pt_un", "pix_err"))
        cb.close()
    step_mode = min(times, key=times.get)
    dt = times[step_mode]
    kms, pms = kernel_time_ms(cams[0], 10)
    res = distributed.to_numpy(out)
    it = res["iters"][:w.n]
    b_alg = algorithmic_bytes_per_feature(w.half_patch, w.pyramids)
    row = {"workload": f"{w.name}: {w.img_ref.shape[1]}x{w.img_ref.shape[0]}, {w.n} keypoints ({w.n_active} active), "
                       f"h={w.half_patch}, L={w.pyramids}, I={w.iterations}" + (f", {streams} concurrent streams" if streams > 1 else ""),
           # ms_per_step = the faster of the two modes (named in step_mode); ms_per_step_graph is the headline's mode
           "ms_per_step": dt * 1e3, "features_per_s": w.n_active * streams / dt, "step_mode": step_mode,
           "ms_per_step_graph": min(times["graph"], times.get("batch_graph", 1e9)) * 1e3,
           "features_per_s_graph": w.n_active * streams / min(times["graph"], times.get("batch_graph", 1e9)),
           "ms_per_step_by_mode": {k: v * 1e3 for k, v in times.items()},
           "mean_iters": float(it[w.status_in > 0].mean()), "max_iters": int(it.max()),
           "variant": batch_variant if step_mode.startswith("batch") else capi.Context.VARIANT_NAMES.get(cams[0].ctx.last_variant(), "?"),
           # features the throughput kernel handed to the latency kernel in the last direct launch (0: rule not applied)
           "handover": int(cams[0].ctx.last_handover()),
           "kernel_ms": kms, "pyramid_ms": pms,
           "roofline_frac": w.n_active * b_alg / (kms * 1e-3) / 1e9 / HBM_PEAK_GBPS}
    if streams > 1:
        row["batch_equals_per_stream_launches"] = batch_equal
        row["variant_per_stream_launches"] = capi.Context.VARIANT_NAMES.get(cams[0].ctx.last_variant(), "?")
    for rt in cams:
        rt.close()
    return row


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--features-per-gpu", type=int, default=None, help="weak scaling: keypoints per GPU (default: the config's own count)")
    ap.add_argument("--config", type=int, default=None, help="synth.config index (default 1 = BASELINE configs[1]; 4 with --mode replicas)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: N x the config's keypoints; strong: the config's keypoints split over N GPUs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the per-config table and the side measurements")
    ap.add_argument("--mode", choices=("shard", "replicas"), default="shard",
                    help="shard: one frame pair, keypoints split over the GPUs + all-gather (--scaling weak|strong); "
                         "replicas: an independent camera stream per GPU, no collective (BASELINE configs[4])")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="wall budget of the CPU baseline sample")
    args = ap.parse_args()
    replicas = args.mode == "replicas"
    if args.config is None:
        args.config = 4 if replicas else 1

    if "RANK" not in os.environ and "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return spawn_ranks(args.gpus)   # before anything touches the GPU in this process

    # stdout carries exactly ONE line, the JSON: whatever libraries print while they initialise (RCCL's version
    # banner, for one) is sent to stderr, and the line is written to the saved descriptor at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        return 2
    if not torch.cuda.is_available():
        print("bench.py: no HIP device visible; the HIP path is the product and has no CPU fallback",
              file=sys.stderr)
        return 3
    torch.cuda.set_device(local_rank)
    force_dist = os.environ.get("PAGK_FORCE_DIST") == "1"  # exercise the collective with 1 rank
    from pixel_aware_gyro_aided_klt_feature_tracker_amd import capi, distributed, runtime, synth

    # The data path's collective belongs to the library (pagk_multi_*: ncclAllGather issued by libpagk_hip.so).
    # torch.distributed only carries the 128-byte RCCL id, the barriers and the max-over-ranks of the timing; with
    # PAGK_GATHER=torch (or if the library's communicator cannot be formed) the gather itself falls back to
    # torch.distributed's RCCL backend -- still RCCL, still on the tracker's stream -- and the JSON line says so.
    gather_via = "none"
    comm = None
    if world > 1 or force_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        # replicas: torch.distributed carries the barriers and the timing reduction only -- no data-path collective
        want_capi = os.environ.get("PAGK_GATHER", "capi") == "capi" and not replicas
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        gather_via = "none (replicas)" if replicas else "torch.distributed (RCCL backend)"
        if want_capi:
            # Every rank must take the same branch (forming a communicator is itself a collective): each rank first
            # checks on its own that the library can reach RCCL at all (an id of its own costs nothing), the ranks
            # agree on that through torch.distributed, and only then is rank 0's id broadcast and the communicator
            # formed; a failure after that point is agreed upon the same way before anyone uses the communicator.
            def all_ok(flag: bool) -> bool:
                t = torch.tensor([1 if flag else 0], dtype=torch.int32, device="cuda")
                dist.all_reduce(t, op=dist.ReduceOp.MIN)
                return bool(t.item())
            why = ""
            try:
                my_uid = capi.Multi.unique_id()
            except Exception as e:   # noqa: BLE001 -- any failure here must not lose the run
                my_uid, why = None, str(e)
            if all_ok(my_uid is not None):
                uid = torch.zeros(128, dtype=torch.uint8, device="cuda")
                if rank == 0:
                    uid.copy_(torch.frombuffer(bytearray(my_uid), dtype=torch.uint8))
                dist.broadcast(uid, src=0)
                try:
                    comm = capi.Multi(uid=bytes(uid.cpu().numpy().tobytes()), rank=rank, world=world, device=local_rank)
                except Exception as e:   # noqa: BLE001
                    comm, why = None, str(e)
                if all_ok(comm is not None):
                    distributed.COMM = comm
                    gather_via = "pagk_multi_allgather (libpagk_hip.so -> ncclAllGather)"
                else:
                    if comm is not None:
                        comm.close()
                    comm = None
            if comm is None:
                print(f"bench.py: library communicator unavailable on some rank ({why or 'another rank'}); gathering "
                      f"through torch.distributed", file=sys.stderr)
    distributed.FORCE_COLLECTIVE = force_dist and not replicas

    # identical inputs on every rank (seeded generator)
    per_gpu = args.features_per_gpu or NOMINAL_N[args.config]
    if replicas:
        # one whole stream per GPU: every rank tracks all `per_gpu` keypoints of ITS frame pair (same shape, a seed
        # of its own) and nothing is exchanged
        n_total = per_gpu
        w = synth.config(args.config, n=n_total, seed=0x5EED0000 + args.config + 0x100 * rank) if rank else \
            synth.config(args.config, n=n_total)
    else:
        n_total = NOMINAL_N[args.config] if args.scaling == "strong" else per_gpu * world
        w = synth.config(args.config, n=n_total)
    p = capi.make_params(half_patch=w.half_patch, iterations=w.iterations, pyramids=w.pyramids,
                         has_gyro=w.has_gyro, camera=w.camera)
    rt = runtime.ResidentTracker(p, device=local_rank, rank=0 if replicas else rank, world=1 if replicas else world)
    rt.load_pair(w.img_ref, w.img_cur)
    rt.set_features(w.pt_ref, w.pt_init, w.affine, w.status_in)
    n_active_total = w.n_active
    n_active_local = int(np.count_nonzero(w.status_in[rt.lo:rt.hi]))
    if replicas and world > 1:
        t = torch.tensor([w.n_active], dtype=torch.int64, device="cuda")
        dist.all_reduce(t)   # (control plane: the aggregate's numerator)
        n_active_total = int(t.item())

    def barrier():
        if world > 1:
            dist.barrier()

    # `value`: the step a live camera loop can use -- "graph": pyramid(current frame) -> PatchMatch replayed as one
    # hipGraph (multi-GPU: the two launches, then the all-gather on a side stream, see runtime.ResidentTracker.step).
    # PAGK_STEP_MODE=fused times the look-ahead form instead (needs frame k+1 while pair (k-1, k) is tracked).
    step_mode = os.environ.get("PAGK_STEP_MODE", "graph")
    # Untimed spin-up (disclosed in the JSON line as `spinup_s`): a fresh process finds the device idle, and the first
    # few milliseconds of work run at the clocks and with the cold queues / caches of an idle device -- 20 steps timed
    # right after 5 warm-up steps read 0.1094 ms per step, the same 20 steps a second later 0.1035
    # (profiles/r03_bench_cold_vs_warm.log).  `value` is the steady-state rate of a running camera loop, so the same
    # step is replayed for a fixed wall time first (synchronised every 20 steps: a deep launch backlog has its own
    # after-effect), THEN the W warm-up steps and the K timed steps follow exactly as the contract says.
    spinup_s = float(os.environ.get("PAGK_BENCH_SPINUP_S", "0.5"))
    # ... and the contract-pure figure beside it: the SAME W warm-up + K timed steps taken first, on the device as this
    # fresh process found it (`ms_per_step_cold` / `value_cold` in the line).  `value` is the
    # steady-state figure and the one compared between rounds; BASELINE.md holds no published number for either.
    elapsed_cold, _ = time_steps(rt, args.steps, args.warmup, step_mode, barrier)
    if world > 1:
        t = torch.tensor([elapsed_cold], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed_cold = float(t.item())
    if spinup_s > 0:
        def burst(k):
            for _ in range(k):
                rt.step(mode=step_mode)
            rt.finish()
            torch.cuda.synchronize()
        t_spin = time.perf_counter()
        burst(20)
        # the number of bursts is agreed between the ranks: a step of the sharded mode contains a collective
        bursts = int(min(1000, max(1, spinup_s / max(time.perf_counter() - t_spin, 1e-6))))
        if world > 1:
            t = torch.tensor([bursts], dtype=torch.int64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            bursts = int(t.item())
        for _ in range(bursts):
            burst(20)
    elapsed, out = time_steps(rt, args.steps, args.warmup, step_mode, barrier)
    per_gpu_ms = [elapsed / args.steps * 1e3]
    if world > 1:
        mine = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_gpu_ms = [float(v.item()) / args.steps * 1e3 for v in every]
        t = mine.clone()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    mode_used = rt.mode_used

    kernel_ms, pyramid_ms = kernel_time_ms(rt, min(50, max(10, args.steps)))
    variant = rt.ctx.last_variant()
    res = distributed.to_numpy(out)  # full length on every rank (gathered when world > 1)

    rt.ctx.check_launch()   # (torch synchronised the streams, not pagk_sync: read the launches' error word)
    extras = {}
    if (world > 1 or force_dist) and not replicas:
        # the collective by itself (events on the stream it runs on), and the same steps without it
        extras["gather"] = rt.gather_report(min(50, max(10, args.steps)))
        extras["gather"]["via"] = gather_via
        t_nog, _ = time_steps(rt, max(10, args.steps // 4), 3, step_mode + "-nogather", barrier)
        extras["gather"]["ms_per_step_without_gather"] = t_nog / max(10, args.steps // 4) * 1e3
        # what RCCL itself says the gather spans (ncclCommCount of the library's communicator), beside torch's world size
        extras["gather"]["rccl_ranks"] = comm.comm_count() if comm is not None else None
        extras["gather"]["torch_world"] = world
        # the gathered result against ONE unsharded launch of the same features on rank 0's own GPU (its own context,
        # host buffers in, results out): the sharded path must reproduce it bit for bit -- the proof, inside the line, that
        # what N GPUs gathered is what one GPU computes
        if rank == 0:
            sctx = capi.Context(local_rank)
            single = sctx.track(p, w.img_ref, w.img_cur, w.pt_ref, w.pt_init, w.affine, w.status_in)
            sctx.close()
            n = w.n
            d = np.abs(res["pt_un"][:n].astype(np.float64) - single["pt_un"][:n].astype(np.float64))
            extras["px_err_vs_single"] = {
                "max": float(d.max()) if d.size else 0.0,
                "status_mismatches": int(np.count_nonzero(res["status"][:n] != single["status"][:n])),
                "pix_err_mismatches": int(np.count_nonzero(res["pix_err"][:n] != single["pix_err"][:n])),
                "compared": int(n),
                "what": f"all-gathered result of {world} rank(s) vs one unsharded pagk_track of the same {n} features on rank 0's GPU"}
    if replicas:
        # every stream's own check (no collective on the data path, so nothing else ties the ranks' results to anything):
        # the first 256 features re-tracked by the reference-shaped one-thread-per-feature kernel (pagk_set_kernel 1, an
        # independent device implementation of the loop nest) on the same GPU; gathered for the line over the control plane
        m = min(256, w.n)
        cctx = capi.Context(local_rank)
        cctx.set_kernel(1)
        chk = cctx.track(p, w.img_ref, w.img_cur, w.pt_ref[:m].copy(), w.pt_init[:m].copy(), w.affine[:m].copy(), w.status_in[:m].copy())
        cctx.close()
        dd = np.abs(res["pt_un"][:m].astype(np.float64) - chk["pt_un"][:m].astype(np.float64))
        mine_chk = torch.tensor([float(dd.max()) if dd.size else 0.0, float(np.count_nonzero(res["status"][:m] != chk["status"][:m])), float(m)],
                                dtype=torch.float64, device="cuda")
        allc = [mine_chk]
        if world > 1:
            allc = [torch.zeros_like(mine_chk) for _ in range(world)]
            dist.all_gather(allc, mine_chk)
        extras["per_stream_check"] = {
            "what": "per GPU: its stream's first features vs the one-thread-per-feature kernel on the same GPU",
            "streams": [{"max_px": float(v[0].item()), "status_mismatches": int(v[1].item()), "compared": int(v[2].item())} for v in allc]}
    if rank == 0 and world == 1 and not args.no_extras:
        # the look-ahead step beside the headline, and how the two launches can reach the GPU
        ksteps = max(20, args.steps // 2)
        modes = {}
        for m in ("graph", "fused", "serial", "streams"):
            t1, _ = time_steps(rt, ksteps, 5, m)
            modes[m] = t1 / ksteps * 1e3
        extras["step_modes_ms"] = modes
        extras["step_modes_note"] = ("graph = value's mode (valid for a live camera); fused = PatchMatch + the NEXT frame's pyramid "
                                     "in one launch (needs one frame of look-ahead)")
        # the live-camera step including the frame's host -> device copy (pinned memory): what `value` leaves out by
        # starting with the frame resident
        pinned = torch.from_numpy(np.ascontiguousarray(w.img_cur)).pin_memory()
        for _ in range(6):
            rt.step_live(pinned)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(ksteps):
            rt.step_live(pinned)
        torch.cuda.synchronize()
        tl = (time.perf_counter() - t1) / ksteps
        extras["live_step"] = {"ms_per_step": tl * 1e3, "value": n_active_total / tl, "unit": "features/s",
                               "what": f"per frame: {w.img_cur.nbytes} B host (pinned) -> device, pyramid, PatchMatch as ONE "
                                       "replayed hipGraph (pagk_frame_upload_pinned is capturable): the step of a live "
                                       "camera loop with the frame's PCIe transfer inside"}
        # (a) the drop-in call itself: pagk_track on HOST buffers -- two frame uploads, pyramids, per-feature
        #     arrays in, results out, synchronous.  PCIe-inclusive; never `value`.
        hctx = capi.Context(local_rank)
        for _ in range(5):
            hctx.track(p, w.img_ref, w.img_cur, w.pt_ref, w.pt_init, w.affine, w.status_in)
        t1 = time.perf_counter()
        reps = 50
        for _ in range(reps):
            hctx.track(p, w.img_ref, w.img_cur, w.pt_ref, w.pt_init, w.affine, w.status_in)
        th = (time.perf_counter() - t1) / reps
        hctx.close()
        extras["host_buffer_path"] = {"value": n_active_total / th, "unit": "features/s", "ms_per_call": th * 1e3,
                                      "what": "pagk_track(): both frames + feature arrays over PCIe, pyramids of both "
                                              "frames, tracking, results back; synchronous"}
        # (b) every BASELINE config at its own size on this one GPU (configs[3] is the 8-GPU config run unsharded,
        #     configs[4] one stream and eight concurrent streams of its shape)
        table = {}
        ksteps = max(10, min(40, args.steps // 5))
        for key, (ci, nn, st) in {"configs[1]": (1, 1000, 1), "configs[2]": (2, 2000, 1), "configs[3] on 1 GPU": (3, 20000, 1),
                                  "configs[4] one stream": (4, 4000, 1), "configs[4] 8 streams on 1 GPU": (4, 4000, 8)}.items():
            table[key] = config_row(ci, nn, local_rank, ksteps, streams=st)
        extras["configs"] = table

    if rank == 0:
        b_alg = algorithmic_bytes_per_feature(w.half_patch, w.pyramids)
        achieved = n_active_local * b_alg / (kernel_ms * 1e-3) / 1e9
        kname = KERNEL_OF_VARIANT.get(variant, "k_track_block")
        traffic, traffic_tag = measured_traffic(w.name, n_total, kname) if world == 1 and not replicas else (None, None)
        P = (2 * w.half_patch + 1) ** 2
        it = res["iters"][:n_total]
        max_it = int(it.max())
        line = {
            "metric": metric_label(w),
            "value": n_active_total * args.steps / elapsed,
            "unit": "features/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak" if replicas else args.scaling, "vs_baseline": None,
            "per_gpu_ms_per_step": per_gpu_ms,
            "ms_per_step_cold": elapsed_cold / args.steps * 1e3, "value_cold": n_active_total * args.steps / elapsed_cold,
            "dtype": "f32 sampling, f64 normal equations", "data": "synthetic", "spinup_s": spinup_s,
            "runtime_env": runtime_env.IN_FORCE,
            "config": {"workload": f"{w.name}: {w.img_ref.shape[1]}x{w.img_ref.shape[0]} pair, "
                                   f"{n_total * world if replicas else n_total} keypoints "
                                   f"({per_gpu if replicas or args.scaling != 'strong' else n_total // world}/GPU), "
                                   f"gyro-predicted affine init, h={w.half_patch}, L={w.pyramids}, I={w.iterations} "
                                   f"(synthetic stand-in for BASELINE configs[{args.config}])",
                       "mode": "replicas" if replicas else f"shard-{args.scaling}",
                       "features_total": n_total * world if replicas else n_total, "features_active": n_active_total,
                       "sharding": (f"none: {world} independent streams, one per GPU, no collective" if replicas else
                                    f"contiguous feature blocks x{world} + all-gather ({gather_via})" if world > 1 else "none"),
                       "step": {"fused": "PatchMatch(all features of the pair) + pyramid(next frame) in ONE launch (trailing "
                                         "workgroups), replayed as a single-node hipGraph; needs one frame of look-ahead",
                                "graph": "pyramid(current frame) -> PatchMatch(all features), replayed as one hipGraph launch; "
                                         "valid for a live camera"
                                }.get(mode_used, f"pyramid + PatchMatch issued as direct launches ({mode_used})")
                               + (" + all-gather of step k on a side stream, awaited before step k+1's tracking"
                                  if world > 1 and not replicas else ""),
                       "step_mode": mode_used},
            # `bound` names the roofline the contract asks to price against; what actually binds the kernel is in
            # `binding_resource` (the fraction of HBM peak is tiny by construction: ~3.2 KB of compulsory bytes per
            # feature against ~10 iterations of 441-step ORDERED f64 accumulation)
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "traffic_source": (f"profiles/{traffic_tag}/pmc_summary.json: (2*FETCH_SIZE + WRITE_SIZE) KiB "
                                            "per launch") if traffic else None,
                         "kernel": kname, "variant": capi.Context.VARIANT_NAMES.get(variant, "?"), "kernel_ms": kernel_ms,
                         "pyramid_ms": pyramid_ms,
                         "algorithmic_bytes_per_feature": b_alg, "features_per_launch": n_active_local,
                         "binding_resource": "latency of the reference-ordered accumulation: the launch lasts as long as its "
                                             "slowest feature (max iterations x [sampling + 441 dependent f64 FMA steps + "
                                             "4x4 LLT solve]); not HBM",
                         "ordered_chain_floor_ms": max_it * P * 5.9 / 2.4e9 * 1e3,
                         "ordered_chain_floor_note": f"{max_it} iterations of the slowest feature x {P} dependent FMA steps x "
                                                     "5.9 cycles (measured DPP chain step) at 2.4 GHz: what the ordered sums "
                                                     "alone cost on its critical path"},
        }
        line["mean_iters_per_feature"] = float(it[w.status_in > 0].mean())
        line["max_iters_per_feature"] = max_it
        line.update(extras)

        if not args.no_cpu_baseline and world == 1:
            from oracle import pagk_oracle as orc   # cpu_baseline leg: the only use of oracle/ here
            threads = host_cores()
            reps, t_cpu = 0, 0.0
            ref = None
            while t_cpu < args.cpu_seconds:
                t1 = time.perf_counter()
                ref = orc.track(p, w.img_ref, w.img_cur, w.pt_ref, w.pt_init, w.affine, w.status_in, nthreads=threads)
                t_cpu += time.perf_counter() - t1
                reps += 1
            n = w.n
            t1 = time.perf_counter()
            sub = min(n, 100)
            orc.track(p, w.img_ref, w.img_cur, w.pt_ref[:sub].copy(), w.pt_init[:sub].copy(), w.affine[:sub].copy(),
                      w.status_in[:sub].copy(), nthreads=1)
            t_one = time.perf_counter() - t1
            cpu_model = ""
            try:
                with open("/proc/cpuinfo") as f:
                    cpu_model = next(l.split(":", 1)[1].strip() for l in f if l.startswith("model name"))
            except Exception:
                pass
            line["cpu_baseline"] = {
                "value": n_active_total * reps / t_cpu, "unit": "features/s", "cores": threads, "kind": "port",
                "sample": f"the same {n} features x {reps} repeats ({t_cpu:.1f} s), oracle/pagk_oracle.c "
                          f"striped over {threads} pthreads like cv::parallel_for_",
                "one_thread_value": float(np.count_nonzero(w.status_in[:sub])) / t_one, "cpu": cpu_model}
            st_bad = int(np.count_nonzero(res["status"][:n] != ref["status"][:n]))
            d = np.abs(res["pt_un"][:n].astype(np.float64) - ref["pt_un"][:n].astype(np.float64))
            live = w.status_in[:n] > 0   # over the features submitted with status_in = 1
            dl = d[live].max(axis=1) if live.any() else np.zeros(1)
            line["px_err_vs_cpu"] = {"max": float(d.max()), "p99": float(np.percentile(dl, 99)), "status_mismatches": st_bad}
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())

    rt.close()
    if comm is not None:
        distributed.COMM = None
        comm.close()
    if world > 1 or force_dist:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
