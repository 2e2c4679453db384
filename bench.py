#!/usr/bin/env python3
"""bench.py -- tracked features / second of the pyramidal patch-based KLT refinement
(21x21 patch, 3 levels, <= 30 Gauss-Newton iterations) on N MI355X GPUs of one node.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

A "step" is what a tracker does per new frame: build the pyramid of the current frame
(CreatePyramids) and run PatchMatch over the frame's keypoints, everything already resident in
HBM.  Workload at N=1: BASELINE.json configs[1] (752x480, 1000 keypoints, gyro-predicted affine
init; synthetic stand-in).  For N>1 the same frame pair carries N x 1000
keypoints, sharded in contiguous index blocks, one process per GPU, and every step ends with the
RCCL all-gather of the per-rank (pt, status, err) slices: weak scaling.  Rank 0 prints ONE JSON
line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


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


def measured_traffic(workload_name: str, n: int, kernel: str = "k_track_block"):
    """HBM bytes per launch of `kernel` from the committed PMC profile of this workload
    (profiles/<tag>/pmc_summary.json; collected by tools/profile.sh in separate --pmc passes)."""
    try:
        pdir = os.path.join(ROOT, "profiles")
        tag = sorted(d for d in os.listdir(pdir) if os.path.isdir(os.path.join(pdir, d)))[-1]
        with open(os.path.join(pdir, tag, "pmc_summary.json")) as f:
            prof = json.load(f)
        with open(os.path.join(pdir, tag, "bench.json")) as f:
            ref = json.load(f)
        if ref["config"]["workload"].split(":")[0] != workload_name or ref["config"]["features_total"] != n:
            return None, None
        k = next(v for name, v in prof.items() if f"::{kernel}<" in name)
        return (2.0 * k["FETCH_SIZE"]["mean"] + k["WRITE_SIZE"]["mean"]) * 1024.0, tag
    except Exception:
        return None, None


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--features-per-gpu", type=int, default=1000)
    ap.add_argument("--config", type=int, default=1, help="synth.config index (1 = BASELINE configs[1])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the host-path and multi-camera side measurements")
    ap.add_argument("--cameras", type=int, default=4, help="independent streams of the multi-camera side measurement")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="wall budget of the CPU baseline sample")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run",
                  file=sys.stderr)
        return 2
    if not torch.cuda.is_available():
        print("bench.py: no HIP device visible; the HIP path is the product and has no CPU fallback",
              file=sys.stderr)
        return 3
    torch.cuda.set_device(local_rank)
    force_dist = os.environ.get("PAGK_FORCE_DIST") == "1" and "RANK" in os.environ  # exercise RCCL with 1 rank
    if world > 1 or force_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))

    from pixel_aware_gyro_aided_klt_feature_tracker_amd import capi, distributed, runtime, synth
    distributed.FORCE_COLLECTIVE = force_dist

    # identical inputs on every rank (seeded generator), n_total = world x features_per_gpu
    n_total = args.features_per_gpu * world
    w = synth.config(args.config, n=n_total)
    p = capi.make_params(half_patch=w.half_patch, iterations=w.iterations, pyramids=w.pyramids,
                         has_gyro=w.has_gyro, camera=w.camera)
    rt = runtime.ResidentTracker(p, device=local_rank, rank=rank, world=world)
    rt.load_pair(w.img_ref, w.img_cur)
    rt.set_features(w.pt_ref, w.pt_init, w.affine, w.status_in)
    n_active_total = w.n_active
    n_active_local = int(np.count_nonzero(w.status_in[rt.lo:rt.hi]))

    # how a step's launches reach the GPU (runtime.ResidentTracker.step): "fused" = PatchMatch of the pair and the
    # pyramid of the following frame in ONE launch, replayed as a single-node hipGraph; "graph" = pyramid then
    # PatchMatch as a two-node graph (valid for a live camera with no frame of look-ahead)
    step_mode = os.environ.get("PAGK_STEP_MODE", "fused")

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        out = rt.step(mode=step_mode)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = rt.step(mode=step_mode)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # dominant kernel (k_track_block): average launch duration, HIP events recorded by the
    # library on the stream the kernel runs on (pagk_last_kernel_ms); untimed extra launches.
    trk, pyr = [], []
    torch.cuda.synchronize()
    fused_kernel = rt.mode_used == "fused"
    with torch.cuda.stream(rt.main):
        for k in range(min(50, max(10, args.steps))):
            if fused_kernel:   # the launch of the timed loop: k_track_block_pyr (tracking + next frame's pyramid)
                rt.track_shard_fused(1 + (k & 1))
            else:
                rt.rebuild_current_pyramid(1)
                rt.track_shard(1)
            a, b = rt.ctx.last_kernel_ms()
            trk.append(a)
            pyr.append(b)
    kernel_ms = float(np.mean(trk))
    pyramid_ms = None if fused_kernel else float(np.mean(pyr))

    res = distributed.to_numpy(out)  # full length on every rank (gathered when world > 1)

    extras = {}
    if rank == 0 and world == 1 and not args.no_extras:
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
        # (b) multi-camera: C independent frame streams in flight on one GPU (BASELINE configs[4] shape).
        #     Steps of DIFFERENT cameras do not depend on each other, so their launches overlap and fill the
        #     tail of each other's slowest features.  Reported beside `value`, not as `value`.
        C = max(1, args.cameras)
        cams = []
        for c in range(C):   # every tracker owns its pair of HIP streams
            r2 = runtime.ResidentTracker(p, device=local_rank)
            r2.load_pair(w.img_ref, w.img_cur)
            r2.set_features(w.pt_ref, w.pt_init, w.affine, w.status_in)
            cams.append(r2)
        def cam_steps(k, **kw):
            for _ in range(k):
                for r2 in cams:
                    r2.step(**kw)
        cam_steps(5, mode=step_mode)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        ksteps = max(20, args.steps // 2)
        cam_steps(ksteps, mode=step_mode)
        torch.cuda.synchronize()
        tc = time.perf_counter() - t1
        extras["multi_camera"] = {"value": n_active_total * C * ksteps / tc, "unit": "features/s", "cameras": C,
                                  "steps_per_camera": ksteps,
                                  "what": f"{C} independent cameras (one hipGraph replay per camera and frame, mode {step_mode}) in "
                                          "flight on one GPU"}
        # (c) how the step's two launches reach the GPU (runtime.ResidentTracker.step modes), one camera:
        #     `value` uses "graph" (one hipGraphLaunch replaying [pyramid -> PatchMatch])
        modes = {}
        for m in ("fused", "graph", "serial", "streams"):
            for _ in range(5):
                cams[0].step(mode=m)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(ksteps):
                cams[0].step(mode=m)
            torch.cuda.synchronize()
            modes[m] = (time.perf_counter() - t1) / ksteps * 1e3
        extras["step_modes_ms"] = modes
        for r2 in cams:
            r2.close()

    if rank == 0:
        b_alg = algorithmic_bytes_per_feature(w.half_patch, w.pyramids)
        achieved = n_active_local * b_alg / (kernel_ms * 1e-3) / 1e9
        traffic, traffic_tag = (measured_traffic(w.name, n_total, "k_track_block_pyr" if fused_kernel else "k_track_block")
                                if world == 1 else (None, None))
        line = {
            "metric": "tracked features/sec (21x21, 3-lvl, 30 iter)",
            "value": n_active_total * args.steps / elapsed,
            "unit": "features/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 sampling, f64 normal equations", "data": "synthetic",
            "config": {"workload": f"{w.name}: {w.img_ref.shape[1]}x{w.img_ref.shape[0]} pair, "
                                   f"{args.features_per_gpu} keypoints/GPU, gyro-predicted affine init, "
                                   f"h={w.half_patch}, L={w.pyramids}, I={w.iterations} "
                                   "(synthetic stand-in for BASELINE configs[1])",
                       "features_total": n_total, "features_active": n_active_total,
                       "sharding": f"contiguous feature blocks x{world} + all-gather" if world > 1 else "none",
                       "step": {"fused": "PatchMatch(all features of the pair) + pyramid(next frame) in ONE launch (trailing "
                                         "workgroups), replayed as a single-node hipGraph",
                                "graph": "pyramid(current frame) -> PatchMatch(all features), replayed as one hipGraph launch"
                                }.get(rt.mode_used, f"pyramid + PatchMatch issued as direct launches ({rt.mode_used})")
                               + (" + all-gather" if world > 1 else ""),
                       "step_mode": rt.mode_used},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "traffic_source": (f"profiles/{traffic_tag}/pmc_summary.json: (2*FETCH_SIZE + WRITE_SIZE) KiB "
                                            "per launch") if traffic else None,
                         "kernel": "k_track_block_pyr" if fused_kernel else "k_track_block", "kernel_ms": kernel_ms,
                         "pyramid_ms": pyramid_ms,
                         "algorithmic_bytes_per_feature": b_alg, "features_per_launch": n_active_local,
                         "note": "compulsory HBM bytes are ~3.2 KB/feature: the kernel is bound by the ordered f64 "
                                 "accumulation chain (dependent-FMA latency), not by HBM; see DESIGN.md"
                                 + ("; the fused launch also builds the next frame's pyramid (6.56 B/pixel, not counted in "
                                    "`achieved`, included in `traffic`)" if fused_kernel else "")},
        }
        # iterations executed (the rate is meaningless without it) and parity, rank 0 shard
        it = res["iters"][:n_total]
        line["mean_iters_per_feature"] = float(it[w.status_in > 0].mean())
        line["max_iters_per_feature"] = int(it.max())
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
            line["px_err_vs_cpu"] = {"max": float(d.max()), "status_mismatches": st_bad}
        print(json.dumps(line), flush=True)

    rt.close()
    if world > 1 or force_dist:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
