// pagk_rows_kernel.h -- k_track_rows: k_track_quad's iteration (four features per wavefront: MFMA blocks, DPP cost rows,
// lane = feature solve) with the four ROWS OF A WAVE RUNNING INDEPENDENTLY and a work queue behind them.
//
// k_track_quad steps its four features through the levels in lockstep and the hardware hands a new wave its four
// features only when a whole wave has retired.  Measured on configs[3] (20000 features, tools/stamps_quad.py):
//   * a wave executes 14.4 iterations for features that need 11.6 on average: every level lasts as long as its
//     slowest row, and a row that is done still pays the MFMA chain, the cost chain and the solve of the others;
//   * the launch is 1.4 rounds of resident waves: the second round runs on a machine that is 40 % full, and because
//     a wave that is alone on its SIMD is bound by the latency of its own dependent chains it runs hardly faster
//     there -- 40 % of the launch's duration for 28 % of its work.
// Here a row is a small state machine -- NEED a feature / SETUP a level / ITERate / IDLE -- and each wave-iteration
// advances every iterating row by one Gauss-Newton iteration of ITS level (the per-feature sampling rounds pick up
// the DevLevel of the row's level; MFMA block q, cost row q and the lane = feature solve never cared).  A row that
// finishes a feature writes its outputs and takes the next index from a global counter (TrackArgs::queue), so a
// resident grid (occupancy x CUs waves) stays full until the queue is empty and no row waits for another at a level
// boundary.  The arithmetic of a feature is the same instruction sequence as in k_track_quad in the same order: which
// row runs which feature, and when, cannot change a bit of its result (tests/test_parity_gpu.py).
#pragma once
#include "pagk_quad_kernel.h"

namespace pagk {

template <int NCH>
__global__ void __launch_bounds__(64, 4) k_track_rows(TrackArgs a)
{
    __shared__ __attribute__((aligned(256))) QuadLds S;
    const int lane = threadIdx.x, row = lane >> 4, lr = lane & 15;
    // the kernel is instantiated per patch size (NCH = chunks of 64 pixels: 2 <-> h = 5, 4 <-> h = 7, 7 <-> h = 10): the
    // patch geometry is a compile-time constant (divisions by the patch width, LDS addresses, chunk lengths)
    constexpr int h = NCH == 7 ? 10 : (NCH == 4 ? 7 : 5), Wd = 2 * h + 1, P = Wd * Wd;
    static_assert(NCH == 2 || NCH == 4 || NCH == 7, "instantiated for h = 5, 7, 10");
    const float fh = (float)h;
    const float *init = a.has_gyro ? a.pt_init : a.pt_ref;  // :85-89

    // lane -> patch pixel of chunk c: p = 64 c + lane, row-major (y outer, :233-234)
    auto patch_xy = [&](int c, float &x, float &y) {
        int p = 64 * c + lane;
        p = p < P ? p : P - 1;
        const int yy = p / Wd, xx = p - yy * Wd;
        x = (float)(xx - h);
        y = (float)(yy - h);
    };
    auto first_of = [](unsigned long long m) { return (int)(__builtin_ctzll(m) >> 4); };
    auto without = [](unsigned long long m, int f) { return m & ~(0xffffull << (16 * f)); };

    const QuadOperands ops = quad_operands(S, lane);  // MFMA operand roles of this lane
    const int mk = ops.mk, mq = ops.mq, mi = ops.mi;
    if (lane < 16) S.ones[lane] = 1.0;
    const uint32_t sq_addr = lds_off(&S.sq[quad_sq_row(row)]) + 8u * lr;
    float *ws = a.ws + (size_t)blockIdx.x * (4 * NCH * 64) + lane;  // img1 samples of the rows' current levels

    // ---- the row's state: every lane of a row holds the same values -------------------------------------------
    enum { NEED = 0, SETUP = 1, ITER = 2, IDLE = 3 };
    int state = NEED;
    int fi = 0, level = 0, iter = 0, succ = 1, iters = 0;
    float refx = 0, refy = 0, p2x = 0, p2y = 0, A00 = 1, A01 = 0, A10 = 0, A11 = 1, ext_x = 0, ext_y = 0;
    float ptx = 0, pty = 0, dx = 0, dy = 0, dg = 0, db = 0, lastCost = 0, cneg = 0, fcm1 = 0, frm1 = 0;

#ifdef PAGK_STAMPS
    const unsigned long long qreal0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long wave_iters = 0;
#endif

    for (;;) {
        // ---- NEED: take the next feature (src/patch_match.cpp:166-176) -----------------------------------------
        for (;;) {
            if (__ballot(state == NEED) == 0ull) break;
            // (every index comes from the queue, the first one too: a wave that becomes resident late -- the grid is
            // sized for full occupancy, other work may hold slots -- must not sit on features it was promised)
            int q = 0;
            if (state == NEED && lr == 0) q = atomicAdd(a.queue, 1);
            q = __shfl(q, lane & ~15);
            if (state == NEED) {
                if (q >= a.n) {
                    state = IDLE;  // the queue is empty
                } else {
                    fi = q;
                    p2x = init[2 * fi], p2y = init[2 * fi + 1];
                    if (a.status_in[fi] == 0) {  // :173: nothing to track; the row asks again
                        if (lr == 0) write_outputs(a, fi, p2x, p2y, 0, 0.0f, 0, 0.0f, 0);
                    } else {
                        A00 = 1, A01 = 0, A10 = 0, A11 = 1;
                        if (a.use_affine) {
                            A00 = a.affine[4 * fi], A01 = a.affine[4 * fi + 1], A10 = a.affine[4 * fi + 2];
                            A11 = a.affine[4 * fi + 3];
                        }
                        refx = a.pt_ref[2 * fi], refy = a.pt_ref[2 * fi + 1];
                        ext_x = fabsf(A00) * fh + fabsf(A01) * fh + 2.0f;
                        ext_y = fabsf(A10) * fh + fabsf(A11) * fh + 2.0f;
                        level = a.n_levels - 1;
                        iters = 0;
                        state = SETUP;
                    }
                }
            }
        }

        // ---- SETUP: a row enters a level (:177-194, and the iteration-invariant img1 samples :253, :263) --------
        const unsigned long long setm = __ballot(state == SETUP);
        if (setm) {
            unsigned long long m = setm;
            while (m) {
                const int f = first_of(m);
                m = without(m, f);
                const int src = 16 * f;
                const int lvl = __builtin_amdgcn_readlane(level, src);
                const DevLevel &L1 = a.l1[lvl];
                const float sc = a.scales[lvl];
                const float fptx = rl(refx, src) * sc, fpty = rl(refy, src) * sc;  // :177
                const float cn = -sample<true>(L1, fptx, fpty);
                for (int c = 0; c < NCH; c++) {
                    float x, y;
                    patch_xy(c, x, y);
                    ws[(f * NCH + c) * 64] = sample<true>(L1, fptx + x, fpty + y);
                }
                if (row == f) {
                    ptx = fptx, pty = fpty;
                    cneg = cn;
                    float nx, ny;
                    if (lvl == a.n_levels - 1) {  // :180
                        nx = p2x * sc;
                        ny = p2y * sc;
                    } else {  // :182
                        nx = (float)((double)(p2x * 1.0f) / 0.5);
                        ny = (float)((double)(p2y * 1.0f) / 0.5);
                    }
                    dx = nx - ptx, dy = ny - pty, dg = 0.0f, db = 0.0f;  // :186-191
                    lastCost = 0.0f;                                      // :193
                    succ = 1;                                             // :194
                    iter = 0;
                    fcm1 = a.l2[lvl].fcols_m1, frm1 = a.l2[lvl].frows_m1;
                    state = ITER;
                }
            }
            __syncthreads();  // the previous iteration's readers of cconst are done
            S.cconst[row][lr] = (double)cneg;
            __syncthreads();
        }

        // ---- ITER: one Gauss-Newton iteration of every iterating row, each at its own level (:215-344) ----------
        const bool act = state == ITER;
        const unsigned long long actm = __ballot(act);
        if (actm == 0ull) break;  // every row is IDLE: the queue is empty and this wave's features are written
#ifdef PAGK_STAMPS
        wave_iters++;
#endif
        if (act) iters++;
        const double cd = (double)cneg;
        const float bx = ptx + dx, by = pty + dy;  // (pt.x + dx), then + wx (:252)
        const float gain = 1.0f + dg;
        const bool interior = (bx - ext_x >= 0.0f) && (bx + ext_x < fcm1) && (by - ext_y >= 0.0f) && (by + ext_y < frm1);
        const unsigned long long intm = __ballot(interior);
        double d = 0.0;      // the four 4x4 accumulators (:217-218 H = 0, b = 0)
        float carry = 0.0f;  // cost = 0 (:283)

#pragma nounroll
        for (int c = 0; c < NCH; c++) {
            // ---- sampling: chunk c of every iterating row (see k_track_quad for the shape of the pipelining) -----
            float x, y;
            patch_xy(c, x, y);
            const bool valid = 64 * c + lane < P;
            float s1q[4];
#pragma unroll
            for (int f = 0; f < 4; f++) s1q[f] = ws[(f * NCH + c) * 64];
            auto issue = [&](int f, FiveTaps &tp) {
                const int src = 16 * f;
                const DevLevel &L2 = a.l2[__builtin_amdgcn_readlane(level, src)];  // the row's level
                float wx = x, wy = y;
                if (a.use_affine) {  // :203-204
                    wx = rl(A00, src) * x + rl(A01, src) * y;
                    wy = rl(A10, src) * x + rl(A11, src) * y;
                }
                const float X = rl(bx, src) + wx, Y = rl(by, src) + wy;
                tp = ((intm >> src) & 1ull) ? sample5_issue<false>(L2, X, Y) : sample5_issue<true>(L2, X, Y);
            };
            auto consume = [&](int f, const FiveTaps &tp) {
                const int src = 16 * f;
                const float s1v = f == 0 ? s1q[0] : f == 1 ? s1q[1] : f == 2 ? s1q[2] : s1q[3];
                const Five s = sample5_finish(tp);
                const float e = s.c + rl(db, src) - rl(gain, src) * s1v;  // :252-253
                const float Ix = 0.5f * (s.xp - s.xm);                     // :259-260
                const float Iy = 0.5f * (s.yp - s.ym);                     // :261-262
                S.chunk[0][f][lane] = (double)Ix;
                S.chunk[1][f][lane] = (double)Iy;
                S.chunk[2][f][lane] = -(double)e;
                S.sq[quad_sq_row(f) + 1 + lane] = valid ? e * e : 0.0f;  // :294; past the patch: + 0.0f changes nothing
            };
            int fa = first_of(actm), fb = 0;
            unsigned long long rest = without(actm, fa);
            FiveTaps ta, tb;
            bool last_in_a = true;
            issue(fa, ta);
#pragma nounroll
            while (rest) {
                fb = first_of(rest);
                rest = without(rest, fb);
                issue(fb, tb);
                __builtin_amdgcn_sched_barrier(0);
                consume(fa, ta);
                __builtin_amdgcn_sched_barrier(0);
                if (!rest) {
                    last_in_a = false;
                    break;
                }
                fa = first_of(rest);
                rest = without(rest, fa);
                issue(fa, ta);
                __builtin_amdgcn_sched_barrier(0);
                consume(fb, tb);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (last_in_a) {
                asm volatile("; last taps: set a");
                consume(fa, ta);
            } else {
                asm volatile("; last taps: set b");
                consume(fb, tb);
            }
            if (lr == 0) S.sq[quad_sq_row(row)] = carry;  // running cost = first term of this chunk's chain (0 + s == s)
            __syncthreads();
            carry = quad_chunk_phase(ops, P, c, sq_addr, d);  // H, b and cost of the chunk
            __syncthreads();  // the chunk has been read before the next one is written
        }
        // ---- solve (:302-319): D(q, i, j) sits in lane 16 i + 4 q + j; every lane of row q solves its feature
        quad_acc(S)[mq][mk * 4 + mi] = d;
        __syncthreads();
        double H[4][4], b[4], upd[4];
        quad_read_system(S, row, P, cd, H, b);
        float cost = carry;
        if (a.penalty) add_penalty(a, dx, dy, H, b, cost);
        double unorm = 0.0;
            // (rows without an iterating feature hold stale sums: they sit the solve out, so that only live systems can
            // raise the exception flags that send the wave to the plain-division form)
            if (act) unorm = llt4_solve_nsq(H, b, upd, a.solver);  // update.squaredNorm()
        __syncthreads();  // the accumulators' LDS is the next iteration's first chunk
        // ---- update + termination (:322-344), then the row's next state -------------------------------------------
        if (act) {
            bool cont = true;
            if (upd[0] != upd[0]) {  // :322
                succ = 0;
                cont = false;
            } else if (iter > 0 && cost > lastCost) {  // :328
                cont = false;
            } else {
                dx = (float)((double)dx + upd[0]);  // :332
                dy = (float)((double)dy + upd[1]);
                if (a.illum) {  // :334-337
                    dg = (float)((double)dg + upd[2]);
                    db = (float)((double)db + upd[3]);
                }
                lastCost = cost;  // :339
                succ = 1;
                if (unorm < kNormSqConverged) cont = false;  // :343  update.norm() < 1e-2
            }
            iter++;
            if (!cont || iter >= a.iterations) {  // the level is over (:215, :326, :330, :343)
                p2x = ptx + dx;  // :348
                p2y = pty + dy;
                if (level == 0) {
                    if (lr == 0) write_outputs(a, fi, p2x, p2y, succ, lastCost, 1, 1.0f, iters);  // :365 ncc = 1
                    state = NEED;
                } else {
                    level--;
                    state = SETUP;
                }
            }
        }
    }
#ifdef PAGK_STAMPS
    if (lane == 0 && a.dbg) {
        a.dbg[(size_t)blockIdx.x * 16 + 6] = wave_iters;
        a.dbg[(size_t)blockIdx.x * 16 + 7] = qreal0;
        a.dbg[(size_t)blockIdx.x * 16 + 8] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

}  // namespace pagk
