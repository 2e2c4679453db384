// pagk_pipe_kernel.h -- the 4-wave-per-feature Gauss-Newton loop with its phases OVERLAPPED (round 4).
//
// PatchMatch::OpticalFlowConsideringIlluminationChange_onePixel, src/patch_match.cpp:167-367, for the two-round patches
// (h = 8, 9, 10: 256 < P <= 512).  Same arithmetic, same order, same bits as track_block_body (pagk_kernels.h); what
// changed is WHEN each wave does its share of an iteration.  A launch of this kernel lasts as long as its slowest
// feature, i.e. (iterations of that feature) x (one iteration's critical path), and in track_block_body that path was
// three strictly serial phases: all four waves sample both rounds | barrier | waves 0-2 run the ordered chains (:284-299)
// | barrier | four lanes solve (:319) | barrier.  The chains consume the patch in row-major order at 5.9 cycles per
// pixel, so they can start as soon as pixels 0..255 are in LDS, and nothing but habit made the chain waves sample the
// second round themselves.  Here, per iteration:
//
//   all waves   round 0: pixel p = tid (0..255) -> streams, esq                       __syncthreads (B1)
//   wave 0, 1   eight DPP rows: XX YX YY XE | YE X Y E, pixels 0..255 at once, then -- flags permitting -- 256..P-1
//               (chain_rows_f64_piped: the waits are inside the chain, the ready case costs no bubble)
//               wave 1: acc -> LDS, flag;  then, in a level's first iteration, H22 of the NEXT level (repeat_sum_f64)
//               wave 0: acc -> LDS, waits for wave 1's flag, solves in lanes 0..3, update -> LDS
//   wave 2      samples batch A = [256, 320) and C = [384, P): gathers of both in flight together; flag after each
//   wave 3      samples batch B = [320, 384); flag; then the ordered f32 cost chain (:294) with the same embedded waits.
//               The solve does not read the cost (the penalty's e_pen^2 is added afterwards, :313), so this chain has
//               until the END of the solve
//                                                                                      __syncthreads (B2)
//   all waves   update / exit tests (:322-344)
//
// Two barriers per iteration instead of three, and the critical path is round 0 + chain + solve instead of
// round 0 + round 1 + chain + solve.  H22 = sum of P copies of c^2 (the level's constant, :263) is no chain at all any
// more: repeat_sum_f64 (pagk_device.h) -- wave 1 computes the top level's at level set-up and each further level's in
// the shadow of the previous level's first solve.
//
// Producer -> consumer through LDS: data stores, s_waitcnt lgkmcnt(0), then the flag (a counter for A / B, the
// iteration number for C and for wave 1's accumulators).  The consumer reads the flag BEFORE the data it vouches for;
// LDS executes one wave's instructions in order.  Every wait is on a wave of the same workgroup that needs nothing from
// the waiter (A, B, C depend on B1 only; wave 1's flag on A, B, C), so there is no cycle; B2 closes the iteration and
// separates its readers from the next iteration's writers.
#pragma once

namespace pagk {

// LDS of the pipelined body: the streams / esq / cslot / acc / update area of track_block_body, then
// int flags[4] (A+B counter, C, wave 1's accumulators, pad), double h22[2], double pen (track_block_lds_bytes).

// bounded like the waits inside the chains (kPipeWaitLooks); false: it ran out
__device__ __forceinline__ bool lds_wait_ge(const int *flag, int want)
{
    for (uint32_t look = 0; look < kPipeWaitLooks; look++) {
        if (__builtin_amdgcn_readfirstlane(*reinterpret_cast<const volatile int *>(flag)) >= want) return true;
        __builtin_amdgcn_s_sleep(1);
    }
    return false;
}

// a flag word behind the data it vouches for: every lane stores the same value to the same address (no exec juggling)
__device__ __forceinline__ void lds_store_b32(uint32_t addr, int v)
{
    asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory");
}

template <int H, bool LEAN>
__device__ __forceinline__ void track_pipe_body(const TrackArgs &a, const int i, const SuspState *resume = nullptr)
{
    static_assert(H >= 8 && H <= 10, "two-round patches only");
    constexpr int kBlock = 256;
    constexpr int Wd = 2 * H + 1, P = Wd * Wd, PP = (P + 31) / 32 * 32, PS = PP + 1;
    constexpr int TAIL = P % 32;
    constexpr bool HAS_B = P > 320, HAS_C = P > 384;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;  // (scalar: the roles below are branches, not masks)

    double *stream = reinterpret_cast<double *>(lds_raw);
    float *esq = reinterpret_cast<float *>(stream + (size_t)8 * PS);
    double *cslot = reinterpret_cast<double *>(esq + PP);
    double *acc = cslot + 2;
    double *sh_upd = acc + 16;
    float *sh_cost = reinterpret_cast<float *>(sh_upd + 5);
    int *flags = reinterpret_cast<int *>(sh_cost + 2 + 2);  // (+2: the 8 spare bytes of track_block_lds_bytes)
    double *h22 = reinterpret_cast<double *>(flags + 4);
    double *sh_pen = h22 + 2;

    const float *init = a.has_gyro ? a.pt_init : a.pt_ref;  // :85-89
    float p2x = init[2 * i], p2y = init[2 * i + 1];
    if (!a.status_in[i]) {  // :173 (block-uniform)
        if (tid == 0) write_outputs(a, i, p2x, p2y, 0, 0.0f, 0, 0.0f, 0);
        return;
    }
    float A00 = 1, A01 = 0, A10 = 0, A11 = 1;
    if (a.use_affine) {
        A00 = a.affine[4 * i], A01 = a.affine[4 * i + 1], A10 = a.affine[4 * i + 2], A11 = a.affine[4 * i + 3];
    }
    const float refx = a.pt_ref[2 * i], refy = a.pt_ref[2 * i + 1];
    if (tid < 4) flags[tid] = 0;  // (first use is behind the first iteration's B1)

    // lane -> pixel sets.  set 0: p = tid (every wave);  set 1: wave 2 -> A = 256 + lane, wave 3 -> B = 320 + lane;
    // set 2: wave 2 -> C = 384 + lane.  Lanes past the patch shadow its last pixel and never store.
    int pix[3];
    pix[0] = tid;
    pix[1] = (wave == 3 ? 320 : 256) + lane;
    pix[2] = 384 + lane;
    float px[3], py[3], wx[3], wy[3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        const int p = pix[r] < P ? pix[r] : P - 1;
        const int yy = p / Wd, xx = p - yy * Wd;
        const int x = xx - H, y = yy - H;
        px[r] = (float)x;
        py[r] = (float)y;
        if (a.use_affine) {  // :203-204  A(0,0)*x + A(0,1)*y, int -> float
            wx[r] = A00 * (float)x + A01 * (float)y;
            wy[r] = A10 * (float)x + A11 * (float)y;
        } else {
            wx[r] = (float)x;
            wy[r] = (float)y;
        }
    }
    // extent of the warped patch, for the interior (clamp-free) fast path
    const float fh = (float)H;
    const float ext_x = fabsf(A00) * fh + fabsf(A01) * fh + 2.0f;
    const float ext_y = fabsf(A10) * fh + fabsf(A11) * fh + 2.0f;

    // accumulator rows (waves 0, 1), as in track_block_body: a DPP row broadcasts one stream value per step to its sixteen
    // lanes, the second FMA factor and the accumulator are per lane
    //   row:      0    1    2    3    4    5          6          7
    //   stream:   XX   YX   YY   XE   YE   X          Y          E
    //   entry:    H00  H10  H11  b0   b1   H20 | H30  H21 | H31  b2 | b3        (lane 0 | lane 1 of the row)
    // wave 3's four rows all walk esq (the f32 cost chain; lane 0 is read)
    const int lr = lane & 15;
    const int cid = wave * 4 + (lane >> 4);
    const uint32_t row_addr = wave < 2 ? lds_off(stream + (size_t)cid * PS) + 16u * lr : lds_off(esq) + 8u * lr;
    const uint32_t flag_addr = lds_off(flags);
    // where this lane's sum goes (slots as the solve reads them: H00 H10 H11 b0 b1 H20 H21 H30 H31 b2 b3), or -1:
    // lane 0 of a row -> H00 H10 H11 b0 | b1 H20 H21 b2, lane 1 of rows X Y E -> H30 H31 b3.  Fixed for the kernel, so
    // publishing a chain's result is one masked store.
    const int acc_slot = wave < 2 ? (lr == 0 ? (cid == 7 ? 9 : cid) : ((lr == 1 && cid >= 5) ? (cid == 7 ? 10 : cid + 2) : -1)) : -1;

    int succ = 1, iters = 0, seq = 0;
    [[maybe_unused]] constexpr bool kPrioByWork = true;  // pagk_prio.h
    PAGK_PRIO_DECL
    float lastCost = 0.0f;
#ifdef PAGK_STAMPS
    // diagnostic build only (tools/stamps.py): cycles per phase as wave 0 sees them, summed over iterations.
    // [0] level set-up, [1] round 0 incl. B1, [2] chains (B1 -> accumulators of both chain waves in LDS), [3] solve incl.
    // B2, [12] update, [5] total
    unsigned long long st[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t_begin = __builtin_amdgcn_s_memtime(), t0, t1;
    const unsigned long long rt_begin = __builtin_amdgcn_s_memrealtime();
#define STAMP(k)                                  \
    t1 = __builtin_amdgcn_s_memtime();            \
    st[k] += t1 - t0;                             \
    t0 = t1;
#else
#define STAMP(k)
#endif
#if defined(PAGK_TIC) && defined(PAGK_TOC)
    // diagnostic build (tools/phase_cost.py): ONE interval per build, between two points of wave 0's iteration, so that
    // the measurement perturbs the iteration by one scalar load
    unsigned long long pt_t0 = 0, pt_sum = 0;
#define POINT(k)                                                                   \
    if (wave == 0) {                                                               \
        if ((k) == PAGK_TIC) pt_t0 = __builtin_amdgcn_s_memtime();                 \
        if ((k) == PAGK_TOC) pt_sum += __builtin_amdgcn_s_memtime() - pt_t0;       \
    }
#else
#define POINT(k)
#endif

    const int level_first = resume ? resume->level : a.n_levels - 1;
    for (int level = level_first; level >= 0; level--) {
#ifdef PAGK_STAMPS
        t0 = __builtin_amdgcn_s_memtime();
#endif
        const DevLevel &L1 = a.l1[level];
        const DevLevel L2 = pin_level(a.l2[level]);
        const float ptx = refx * a.scales[level], pty = refy * a.scales[level];  // :177
        float nx, ny;
        if (level == a.n_levels - 1) {  // :180
            nx = p2x * a.scales[level];
            ny = p2y * a.scales[level];
        } else {  // :182
            nx = (float)((double)(p2x * 1.0f) / 0.5);
            ny = (float)((double)(p2y * 1.0f) / 0.5);
        }
        float dx = nx - ptx, dy = ny - pty, dg = 0.0f, db = 0.0f;  // :186-191
        lastCost = 0.0f;                                            // :193
        succ = 1;                                                   // :194
        int iter_first = 0;
        if (resume && level == level_first) {  // pick the feature up where the throughput kernel left it
            dx = resume->dx, dy = resume->dy, dg = resume->dg, db = resume->db;
            lastCost = resume->lastCost;
            iters = resume->iters;
            iter_first = resume->iter;
        }

        // img1 samples are iteration-invariant: once per level (bit-identical to :253, :263)
        const float cneg = -sample<true>(L1, ptx, pty);
        const double cd = (double)cneg;
        float s1[3];
        s1[0] = sample<true>(L1, ptx + px[0], pty + py[0]);
        s1[1] = s1[2] = 0.0f;
        if (wave >= 2) {  // (wave-uniform)
            s1[1] = sample<true>(L1, ptx + px[1], pty + py[1]);
            if (HAS_C && wave == 2) s1[2] = sample<true>(L1, ptx + px[2], pty + py[2]);
        }
        // second FMA factor of this lane's row (:264  J = (Ix, Iy, de_dg, 1))
        double row_s1 = 1.0;  // H00 H10 H11
        switch (cid) {
            case 3: case 4: row_s1 = -1.0; break;                  // b0 b1:  -J * e
            case 5: case 6: row_s1 = lr == 0 ? cd : 1.0; break;    // H20 | H30,  H21 | H31
            case 7: row_s1 = lr == 0 ? -cd : -1.0; break;          // b2 | b3
            default: break;
        }
        // H22 of the first level this body runs; the further levels' were computed one level ahead (below)
        if (level == level_first && wave == 1) {
            const double s = repeat_sum_f64(cd * cd, P);
            if (lane == 0) h22[level & 1] = s;
        }

        STAMP(0)
        for (int iter = iter_first; iter < a.iterations; iter++) {  // :215
            iters++;
            seq++;
            PAGK_PRIO_TIER
            POINT(0)
            // ---- round 0 of the sampling: pixel tid ----------------------------------------------
            const float bx = ptx + dx, by = pty + dy;  // (pt.x + dx), then + wx (:252)
            const float gain = 1.0f + dg;
            // interior test (block-uniform): every tap coordinate of every pixel, +-1 included,
            // lies in [0, cols-1) x [0, rows-1) => clamps are no-ops and can be skipped.
            const bool interior = (bx - ext_x >= 0.0f) & (bx + ext_x < L2.fcols_m1) &
                                  (by - ext_y >= 0.0f) & (by + ext_y < L2.frows_m1);
            // one pixel's products -> LDS (:252-262, :293-296: the exact f64 products the rows accumulate)
            auto emit = [&](const FiveTaps &t, int p, float s1v) {
                const Five s = sample5_finish(t);
                if (p < P) {
                    const float e = s.c + db - gain * s1v;   // :252-253
                    const float Ix = 0.5f * (s.xp - s.xm);   // :259-260
                    const float Iy = 0.5f * (s.yp - s.ym);   // :261-262
                    const double dIx = (double)Ix, dIy = (double)Iy, de = (double)e;
                    stream[0 * PS + p] = dIx * dIx;
                    stream[1 * PS + p] = dIy * dIx;
                    stream[2 * PS + p] = dIy * dIy;
                    stream[3 * PS + p] = dIx * de;
                    stream[4 * PS + p] = dIy * de;
                    stream[5 * PS + p] = dIx;
                    stream[6 * PS + p] = dIy;
                    stream[7 * PS + p] = de;
                    esq[p] = e * e;  // :294
                }
            };
            // The clamp-free and the clamped form are two copies of a sampling step, chosen once (see track_block_body).
            if (interior) {
                const FiveTaps t = sample5_issue<false>(L2, bx + wx[0], by + wy[0]);
                emit(t, pix[0], s1[0]);
            } else {
                const FiveTaps t = sample5_issue<true>(L2, bx + wx[0], by + wy[0]);
                emit(t, pix[0], s1[0]);
            }
            __syncthreads();  // B1: pixels 0..255 are in LDS
            STAMP(1)
            POINT(1)
            if (wave < 2) {
                // ---- ordered accumulation (:284-299), the feature's critical path: wins issue arbitration ----------
                PRIO(PAGK_PRIO_N_CHAIN)
                // (the iteration number is the same in every lane, but the loop's exits depend on values read from LDS, so the
                // compiler may carry it in a vector register -- and an "s" operand of inline asm is taken as is: seen when
                // this body was instantiated inside another kernel, profiles/r04_lead_pyramid_experiment.patch)
                const uint32_t seq_ab = (uint32_t)__builtin_amdgcn_readfirstlane((HAS_B ? 2 : 1) * seq);
                const uint32_t seq_c = (uint32_t)__builtin_amdgcn_readfirstlane(seq);
                const double s = chain_rows_f64_piped<H>(row_addr, row_s1, flag_addr, seq_ab, seq_c);
                if (acc_slot >= 0) acc[acc_slot] = s;
                if (wave == 1) {
                    // the flag right behind the sums: LDS executes one wave's instructions in order, so whoever reads the
                    // flag's new value reads the sums stored before it (no s_waitcnt in between: it would only delay the flag)
                    // (an explicit ds_write: through a volatile pointer the compiler stored with flat_store_dword sc0 sc1 +
                    // s_waitcnt vmcnt(0) -- the flat path into LDS, on the way from wave 1's sums to wave 0's solve)
                    lds_store_b32(flag_addr + 8u, seq);
                    STAMP(4)   // (wave 1) B1 -> its accumulators published
                    PRIO(PAGK_PRIO_N_REST)
                    if (iter == iter_first && level > 0) {
                        // the next level's H22, in the shadow of this solve: c = -I1(pt) there (:263), same expressions
                        // as that level's set-up
                        const float nptx = refx * a.scales[level - 1], npty = refy * a.scales[level - 1];
                        const double ncd = (double)(-sample<true>(a.l1[level - 1], nptx, npty));
                        const double s22 = repeat_sum_f64(ncd * ncd, P);
                        if (lane == 0) h22[(level - 1) & 1] = s22;
                    }
                } else {
                    // the eleven sums, wave 1's among them: the flag is read FIRST and the sums behind it in the same
                    // batch -- one LDS round trip once the flag is up.  Bounded like every wait of this kernel; a wait that
                    // ran out must not pass for a result.
                    typedef double f64x2 __attribute__((ext_vector_type(2)));
                    f64x2 q0, q1, q2, q3, q4, q5;
                    bool acc_there = false;
                    {
                        int f;
                        const uint32_t fad = flag_addr + 8u, aad = lds_off(acc);
                        for (uint32_t look = 0; look < kPipeWaitLooks; look++) {
                            asm volatile("ds_read_b32 %[f], %[fa]\n\t"
                                         "ds_read_b128 %[q0], %[aa]\n\t"
                                         "ds_read_b128 %[q1], %[aa] offset:16\n\t"
                                         "ds_read_b128 %[q2], %[aa] offset:32\n\t"
                                         "ds_read_b128 %[q3], %[aa] offset:48\n\t"
                                         "ds_read_b128 %[q4], %[aa] offset:64\n\t"
                                         "ds_read_b128 %[q5], %[aa] offset:80\n\t"
                                         "s_waitcnt lgkmcnt(0)"
                                         : [f] "=&v"(f), [q0] "=&v"(q0), [q1] "=&v"(q1), [q2] "=&v"(q2), [q3] "=&v"(q3),
                                           [q4] "=&v"(q4), [q5] "=&v"(q5)
                                         : [fa] "v"(fad), [aa] "v"(aad)
                                         : "memory");
                            if (__builtin_amdgcn_readfirstlane(f) >= seq) {
                                acc_there = true;
                                break;
                            }
                            if (look >= 8) __builtin_amdgcn_s_sleep(1);  // (the other chain wave ends within a look or two)
                        }
                    }
                    STAMP(2)
                    POINT(2)
                    // ---- solve (:302-319) --------------------------------------------------------------------------
                    if (tid < 4) {
                        double Hm[4][4], b[4], upd[4];
                        // acc: H00 H10 H11 b0 b1 H20 H21 H30 H31 b2 b3
                        Hm[0][0] = acc_there ? q0.x : __builtin_nan("");
                        Hm[1][0] = q0.y;
                        Hm[1][1] = q1.x;
                        b[0] = q1.y;
                        b[1] = q2.x;
                        Hm[2][0] = q2.y;
                        Hm[2][1] = q3.x;
                        Hm[3][0] = q3.y;
                        Hm[3][1] = q4.x;
                        b[2] = q4.y;
                        b[3] = q5.x;
                        Hm[2][2] = h22[level & 1];
                        Hm[3][2] = (double)P * cd;  // sum of c*1.0: every partial sum k*c is exact
                        Hm[3][3] = (double)P;       // sum of 1.0*1.0
                        double epsq = 0.0;
                        if constexpr (!LEAN) {
                            if (a.penalty) epsq = add_penalty_hb(a, dx, dy, Hm, b);
                        }
                        // four lanes share the divides of each Cholesky column (pagk_device.h); all end with the result
                        const double unorm = llt4_solve_nsq_lanes<true>(Hm, b, tid, upd, LEAN ? 0u : a.solver);  // update.squaredNorm()
                        if (tid == 0) {
                            sh_upd[0] = upd[0];
                            sh_upd[1] = upd[1];
                            sh_upd[2] = upd[2];
                            sh_upd[3] = upd[3];
                            sh_upd[4] = unorm;
                            if constexpr (!LEAN) sh_pen[0] = epsq;
                        }
                    }
                    PRIO(PAGK_PRIO_N_REST)
                    POINT(3)
                }
            } else if (wave == 2) {
                // ---- round 1, batches A and C: both batches' gathers in flight together ----------------------------
                PRIO(PAGK_PRIO_N_SAMP)
                auto round1 = [&](auto clamp_tag) {
                    constexpr bool CL = decltype(clamp_tag)::value;
                    const FiveTaps ta = sample5_issue<CL>(L2, bx + wx[1], by + wy[1]);
                    if constexpr (HAS_C) {
                        const FiveTaps tc = sample5_issue<CL>(L2, bx + wx[2], by + wy[2]);
                        emit(ta, pix[1], s1[1]);
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        if (lane == 0) atomicAdd(flags, 1);
                        STAMP(8)   // (wave 2) B1 -> batch A published
                        emit(tc, pix[2], s1[2]);
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        lds_store_b32(flag_addr + 4u, seq);
                        STAMP(9)   // (wave 2) A published -> C published
                    } else {
                        emit(ta, pix[1], s1[1]);
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        if (lane == 0) atomicAdd(flags, 1);
                    }
                };
                if (interior)
                    round1(std::false_type{});
                else
                    round1(std::true_type{});
                PRIO(PAGK_PRIO_N_REST)
            } else {
                // ---- round 1, batch B; then the ordered f32 cost sum (:294) -----------------------------------------
                if constexpr (HAS_B) {
                    PRIO(PAGK_PRIO_N_SAMP)
                    if (interior) {
                        const FiveTaps tb = sample5_issue<false>(L2, bx + wx[1], by + wy[1]);
                        emit(tb, pix[1], s1[1]);
                    } else {
                        const FiveTaps tb = sample5_issue<true>(L2, bx + wx[1], by + wy[1]);
                        emit(tb, pix[1], s1[1]);
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (lane == 0) atomicAdd(flags, 1);
                    STAMP(10)   // (wave 3) B1 -> batch B published
                }
                PRIO(PAGK_PRIO_N_COST)
                const float c = chain_rows_f32_piped<H>(row_addr, flag_addr, (uint32_t)__builtin_amdgcn_readfirstlane((HAS_B ? 2 : 1) * seq),
                                                        (uint32_t)__builtin_amdgcn_readfirstlane(seq));
                if (lane == 0) sh_cost[0] = c;
                STAMP(13)   // (wave 3) B published -> cost published
                PRIO(PAGK_PRIO_N_REST)
            }
            __syncthreads();  // B2: update, cost (and h22 of the next level) are in LDS; every reader of this iteration's streams is done
            STAMP(3)
            POINT(4)
            // ---- update + termination, identically in every lane (:322-344) -----------------
            const double u0 = sh_upd[0], u1 = sh_upd[1], u2 = sh_upd[2], u3 = sh_upd[3], unorm = sh_upd[4];
            float cost = sh_cost[0];
            if constexpr (!LEAN) {
                if (a.penalty) cost = (float)((double)cost + sh_pen[0]);  // :313
            }
            if (u0 != u0) {  // :322
                succ = 0;
                break;
            }
            if (iter > 0 && cost > lastCost) break;  // :328
            dx = (float)((double)dx + u0);           // :332
            dy = (float)((double)dy + u1);
            if (a.illum) {  // :334-337
                dg = (float)((double)dg + u2);
                db = (float)((double)db + u3);
            }
            lastCost = cost;  // :339
            succ = 1;
            if (unorm < kNormSqConverged) break;  // :343  update.norm() < 1e-2
#ifdef PAGK_STAMPS
            asm volatile("" : "+v"(dx), "+v"(dy), "+v"(dg), "+v"(db));
            STAMP(12)
#endif
#if defined(PAGK_TIC) && defined(PAGK_TOC)
            asm volatile("" : "+v"(dx), "+v"(dy), "+v"(dg), "+v"(db));
            POINT(5)
#endif
        }
        p2x = ptx + dx;  // :348
        p2y = pty + dy;
        // the next level's round 0 overwrites the streams: every lane left the loop after B2
    }
    PAGK_PRIO_RESET
    float ncc = 1.0f;  // :365
    if (a.calc_ncc) {
        // PatchMatch::NCC (:433-469) on the level-0 images at the final point, as in track_block_body: every lane samples
        // its pixels (p = tid, tid + 256), the f32 sums run in the reference's order (x outer, y inner) as DPP row chains.
        float *vref = reinterpret_cast<float *>(stream), *vcur = vref + PP;
        float *tnum = vcur + PP, *td1 = tnum + PP, *td2 = td1 + PP;
        const DevLevel &R0 = a.l1[0], &C0 = a.l2[0];
        constexpr int nfull = P / 32;
        float vr[2], vc[2];
        __syncthreads();  // the last iteration's readers of the streams are done
#pragma unroll
        for (int r = 0; r < 2; r++) {
            int k = tid + kBlock * r;
            k = k < P ? k : P - 1;
            int xi = k / Wd - H, yi = k - (k / Wd) * Wd - H;  // x outer, y inner
            vr[r] = sample<true>(R0, refx + xi, refy + yi);   // :440
            if (a.use_affine) {
                float wxx = A00 * xi + A01 * yi, wyy = A10 * xi + A11 * yi;  // :449-450
                vc[r] = sample<true>(C0, p2x + wxx, p2y + wyy);
            } else {
                vc[r] = sample<true>(C0, p2x + xi, p2y + yi);  // :447
            }
            if (tid + kBlock * r < P) {
                vref[tid + kBlock * r] = vr[r];
                vcur[tid + kBlock * r] = vc[r];
            }
        }
        __syncthreads();
        const int row = lane >> 4;
        float *sh_f = reinterpret_cast<float *>(acc);  // 8 floats of scratch
        if (wave == 0) {  // rows 0/1: mean_ref, mean_cur (:441, :453); rows 2/3 shadow row 0
            float m = chain_rows_f32<TAIL>(lds_off(row == 1 ? vcur : vref) + 8u * lr, 128u, nfull);
            if (lr == 0 && row < 2) sh_f[row] = m;
        }
        __syncthreads();
        const float mean_ref = sh_f[0] / (float)P, mean_cur = sh_f[1] / (float)P;  // :457-458
#pragma unroll
        for (int r = 0; r < 2; r++) {
            int k = tid + kBlock * r;
            if (k < P) {
                float dr = vr[r] - mean_ref, dc = vc[r] - mean_cur;
                tnum[k] = dr * dc;  // :463
                td1[k] = dr * dr;   // :464
                td2[k] = dc * dc;   // :465
            }
        }
        __syncthreads();
        if (wave == 0) {
            const float *src = row == 0 ? tnum : (row == 1 ? td1 : td2);
            float v = chain_rows_f32<TAIL>(lds_off(src) + 8u * lr, 128u, nfull);
            if (lr == 0 && row < 3) sh_f[2 + row] = v;
        }
        __syncthreads();
        // numerator / std::sqrt(d1 * d2 + 1e-10): float product, double sum / sqrt / divide (:468)
        ncc = (float)((double)sh_f[2] / sqrt((double)(sh_f[3] * sh_f[4]) + 1e-10));
    }
    if (tid == 0) write_outputs(a, i, p2x, p2y, succ, lastCost, 1, ncc, iters);
    if (tid == 0 && kPrioByWork && !resume) prio_account(a, i, iters, a.n_levels);
#ifdef PAGK_STAMPS
    if (lane == 0 && a.dbg) {
        // resumed features: after the throughput kernel's per-wave records
        unsigned long long *dbgp = a.dbg + (resume ? (size_t)16 * (a.susp_waves > 0 ? a.susp_waves : (a.n + 3) / 4) : 0);
        if (wave == 0) {
            st[5] = __builtin_amdgcn_s_memtime() - t_begin;
            for (int k = 0; k < 4; k++) dbgp[(size_t)i * 16 + k] = st[k];
            dbgp[(size_t)i * 16 + 5] = st[5];
            dbgp[(size_t)i * 16 + 6] = (unsigned long long)iters;
            dbgp[(size_t)i * 16 + 7] = rt_begin;  // 100 MHz wall clock, common to all XCDs (s_memtime is per XCD)
            dbgp[(size_t)i * 16 + 11] = __builtin_amdgcn_s_memrealtime();
            dbgp[(size_t)i * 16 + 12] = st[12];
        } else if (wave == 1) {
            dbgp[(size_t)i * 16 + 4] = st[4];
        } else if (wave == 2) {
            dbgp[(size_t)i * 16 + 8] = st[8];
            dbgp[(size_t)i * 16 + 9] = st[9];
        } else {
            dbgp[(size_t)i * 16 + 10] = st[10];
            dbgp[(size_t)i * 16 + 13] = st[13];
        }
    }
#endif
#if defined(PAGK_TIC) && defined(PAGK_TOC)
    if (tid == 0 && a.dbg) {
        a.dbg[(size_t)i * 2] = pt_sum;
        a.dbg[(size_t)i * 2 + 1] = (unsigned long long)iters;
    }
#endif
#undef STAMP
#undef POINT
}

}  // namespace pagk
