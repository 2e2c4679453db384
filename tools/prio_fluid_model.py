"""A fluid model of one CU under the 4-wave kernels' issue-priority rule (csrc/pagk_prio.h), driven by the per-level iteration counts of a
workload (tools/iter_trace.py -> tools/data/*.npy).  A workgroup alone runs one iteration per LONE cycles and uses LONE_SHARE of the CU's VALU
while it does; workgroups share the VALU by strict priority, equals equally; a level set-up costs SETUP lone-equivalent cycles.  Workgroup i
starts on CU i mod 256 (4 slots per CU), a finished one is replaced by the next index.  The model knows nothing of memory, barriers or
the pipeline inside an iteration: it under-states the crowded launch by ~9 us, and it RANKS thresholds as the hardware does
(profiles/r04_ab8_priority_by_remaining_work.log): K = 4 < 3 ~ 5 < 6 at 1000 features.   python tools/prio_fluid_model.py"""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
LONE, SHARE, SETUP, GHZ = 6100.0, 2650.0, 1300.0, 2.4   # measured: lone iteration, crowded iteration / 4, level set-up (cycles)
U = SHARE / LONE

def simulate(it, policy, slots=4, ncu=256, dt=100.0):
    n, L = it.shape
    nxt, live = 0, 0
    cus = [[] for _ in range(ncu)]
    def new(i): return dict(i=i, lv=L - 1, done=0.0, iters=0, cur=0, its=it[i], setup=SETUP)
    for _ in range(slots):
        for c in range(ncu):
            if nxt < n:
                cus[c].append(new(nxt)); nxt += 1; live += 1
    t, end = 0.0, 0.0
    while live:
        for alive in cus:
            if not alive:
                continue
            pr = [policy(s, L) for s in alive]
            cap, rate = 1.0, [0.0] * len(alive)
            for p in sorted(set(pr), reverse=True):
                idx = [k for k in range(len(alive)) if pr[k] == p]
                want = U * len(idx)
                give = U if want <= cap else cap / len(idx)
                for k in idx:
                    rate[k] = give
                cap = max(0.0, cap - want)
            fin = []
            for k, s in enumerate(alive):
                prog = rate[k] / U * dt
                if s["setup"] > 0:
                    s["setup"] -= prog
                    continue
                s["done"] += prog / LONE
                while s["done"] >= 1.0:
                    s["done"] -= 1.0; s["iters"] += 1; s["cur"] += 1
                    if s["cur"] >= s["its"][s["lv"]]:
                        s["cur"] = 0
                        if s["lv"] == 0:
                            fin.append(s); break
                        s["lv"] -= 1; s["setup"] = SETUP
            for s in fin:
                alive.remove(s); live -= 1; end = t + dt
                if nxt < n:
                    alive.append(new(nxt)); nxt += 1; live += 1
        t += dt
    return end / (GHZ * 1e3)

def by_phase_only(s, L): return 0
def behind(K): return lambda s, L: 3 if s["iters"] + 1 > K * (L - s["lv"]) else 0
def by_level(s, L): return s["lv"]

if __name__ == "__main__":
    for name in ("iters_cfg1_1000.npy", "iters_cfg2_2000.npy"):
        it = np.load(os.path.join(HERE, "data", name)).astype(int)
        tot = it.sum(1)
        print(f"{name}: mean {tot.mean():.2f} / max {tot.max()} iterations; the slowest feature alone: {(tot.max() * LONE + it.shape[1] * SETUP) / (GHZ * 1e3):.1f} us")
        print(f"   by phase only: {simulate(it, by_phase_only):6.1f} us")
        for K in (3, 4, 5, 6):
            print(f"   behind, K = {K}: {simulate(it, behind(K)):6.1f} us")
        print(f"   by level     : {simulate(it, by_level):6.1f} us")
