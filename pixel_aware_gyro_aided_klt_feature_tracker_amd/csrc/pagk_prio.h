// pagk_prio.h -- issue priority of a 4-wave workgroup's waves (s_setprio; arithmetic untouched).
//
// A launch of the 4-wave kernels lasts as long as its slowest feature, and while a CU holds four workgroups the
// iteration of every one of them is stretched from 6.1 k to 10.6 k cycles: the feature that
// will need 26 iterations pays for the company of three that need 10.  Which feature that is cannot be known in
// advance, but it shows: all workgroups of a launch start together and iterate at the same pace, so one that has used
// more iterations than K (TrackArgs::prio_k: 4, PAGK_PRIO_K in the environment of pagk_create, 0 = off, auto = the workload's own mean rounded up, below) per pyramid level it has entered is BEHIND its neighbours and has the most work
// left.  Longest-remaining-work-first is the makespan rule: a workgroup that is behind runs every phase at priority 3,
// the others keep the by-phase priorities below it (ordered chains 2, second sampling round 1, cost chain 1, the rest
// 0) and lose only issue slots they had slack for.  K = 4: the mean of the BASELINE workloads is 3.5 iterations per
// level.  Measured in one session (profiles/r04_ab8_priority_by_remaining_work.log): 1000 features 97.9 -> 87.4 us,
// 500: 75.4 -> 72.1, configs[2] 2000 x 4 levels 208 -> 199.5, 250 (nothing to outrank) 69.5 -> 69.9.
// A body states with `constexpr bool kPrioByWork` whether the rule applies to it: not for one-round patches (h <= 7: five
// workgroups of 96 VGPRs per CU, short iterations -- 0..+1.7 % there).
// -DPAGK_PRIO_MODE=0 is the by-phase rule alone (rounds 1-4: chains 3, sampling 2, cost 1).
#pragma once

#ifndef PAGK_PRIO_MODE
#define PAGK_PRIO_MODE 1
#endif

namespace pagk {
// PAGK_PRIO_K=auto (TrackArgs::prio_stats / prio_kbuf non-null): K = the mean number of iterations per feature and level that this
// context's launches have run so far, rounded up -- the threshold sits just above what an ordinary feature needs, whatever the
// imagery (3.5 on the BASELINE workloads: K = 4, the value the sweeps chose).  Nothing of it is on a launch's critical path: a
// workgroup reads K with ONE cached load at its start; a sample of the features adds its counts with fire-and-forget atomics at the
// end, and a few of those that end EARLY (they are not what the launch waits for) re-read the sums and refresh K for the launches to come.
__device__ __forceinline__ void prio_account(const TrackArgs &a, int i, int iters, int levels)
{
    // A SAMPLE of the features keeps the books: a thousand workgroups adding to -- let alone reading at agent scope -- one line
    // queue at the memory side for tens of microseconds (1000 features: 87 -> 154 us when every one did).  Every eighth feature
    // adds its counts, every sixty-fourth that also ends early re-reads the sums and refreshes K.
    if (a.prio_stats && (i & 7) == 0) {
        __hip_atomic_fetch_add(a.prio_stats + 0, (unsigned long long)iters, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(a.prio_stats + 1, (unsigned long long)levels, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((i & 63) == 0 && iters <= 4 * levels) {
            const unsigned long long it = __hip_atomic_load(a.prio_stats + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long lv = __hip_atomic_load(a.prio_stats + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (lv > 0) {
                int k = (int)ceilf((float)it / (float)lv);
                k = k < 3 ? 3 : (k > 12 ? 12 : k);
                __hip_atomic_store(const_cast<int *>(a.prio_kbuf), k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}
}  // namespace pagk
#if PAGK_PRIO_MODE == 0
#define PAGK_PRIO_DECL
#define PAGK_PRIO_TIER
#define PAGK_PRIO_RESET
#define PRIO(n) __builtin_amdgcn_s_setprio(n);
#define PAGK_PRIO_N_CHAIN 3
#define PAGK_PRIO_N_SAMP 2
#define PAGK_PRIO_N_COST 1
#define PAGK_PRIO_N_REST 0
#else
// (one PLAIN load -- the L2 serves a thousand workgroups; an agent-scope load of one line from all of them queues for 60 us -- beside the prologue's other loads -- the feature's points and affine -- and a scalar from then on: a vector register held
// through the body costs the generic instantiations a spill)
#define PAGK_PRIO_DECL   \
    int tier_now = 0;    \
    const int prio_kv = !kPrioByWork ? 0 : (a.prio_kbuf ? __builtin_amdgcn_readfirstlane(*a.prio_kbuf) : a.prio_k);
#define PAGK_STR2(x) #x
#define PAGK_STR(x) PAGK_STR2(x)
// (wave-uniform: iters, level and the kernel argument are the same in every lane; readfirstlane says so to the compiler, and
// the branches are written out -- as C the comparison was carried as a lane mask and every site cost a VALU compare or two)
// The priority is switched once per transition: two scalar instructions on the path of an iteration without one.
#define PAGK_PRIO_TIER                                                                                              \
    {                                                                                                               \
        const int tier = (prio_kv > 0 && __builtin_amdgcn_readfirstlane(iters - prio_kv * (a.n_levels - level)) > 0) ? 3 : 0; \
        asm volatile("s_cmp_eq_u32 %0, %1\n\t"                                                                      \
                     "s_cbranch_scc1 2f\n\t"                                                                        \
                     "s_cmp_eq_u32 %1, 0\n\t"                                                                       \
                     "s_cbranch_scc1 1f\n\t"                                                                        \
                     "s_setprio 3\n\t"                                                                              \
                     "s_branch 2f\n"                                                                                \
                     "1:\n\t"                                                                                       \
                     "s_setprio 0\n"                                                                                \
                     "2:" ::"s"(tier_now), "s"(tier)                                                                \
                     : "scc");                                                                                      \
        tier_now = tier;                                                                                            \
    }
// by-phase priority of a workgroup that is not behind: one guarded s_setprio
#define PRIO(n)                                                                                                     \
    asm volatile("s_cmp_lg_u32 %0, 0\n\t"                                                                           \
                 "s_cbranch_scc1 1f\n\t"                                                                            \
                 "s_setprio " PAGK_STR(n) "\n"                                                                      \
                 "1:" ::"s"(tier_now)                                                                               \
                 : "scc");
// (a finisher runs a body once per feature)
#define PAGK_PRIO_RESET __builtin_amdgcn_s_setprio(0);
#define PAGK_PRIO_N_CHAIN 2
#define PAGK_PRIO_N_SAMP 1
#define PAGK_PRIO_N_COST 1
#define PAGK_PRIO_N_REST 0
#endif
