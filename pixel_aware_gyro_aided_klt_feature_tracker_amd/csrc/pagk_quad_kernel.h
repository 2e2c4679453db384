// pagk_quad_kernel.h -- k_track_quad: the Gauss-Newton loop with FOUR FEATURES PER WAVEFRONT.
//
// The throughput form of the path for launches with (many) more features than the chip has SIMDs.  One wave64
// owns four features, and every part of an iteration that one feature cannot fill a wave with is shared:
//
//   H, b     v_mfma_f64_4x4x4f64 has four independent 4x4 blocks: block q accumulates feature q.  With
//            A = J = (Ix, Iy, c, 1) and B = (Ix, Iy, -e, c) one block yields every entry the solve reads:
//            D[i][0..1] = H[i][0..1], D[i][2] = sum J_i * (-e) = b_i, D[2][3] = sum c*c = H22, D[3][0..1] =
//            H30, H31 (H32 = P*c and H33 = P are exact closed forms).  The instruction is a sequential FMA chain
//            over k = four consecutive patch pixels in ascending order (tools/microbench7.hip), the products are
//            exact, so this is the reference's `H += J*J^T; b += -J*e` (src/patch_match.cpp:293-296) -- one
//            instruction per four pixels for FOUR features, operands read from LDS already widened (the widening
//            is done once per pixel by the sampling lanes, 64 useful lanes per instruction);
//   cost     the ordered f32 sum of e*e (:294) is a 16-lane DPP row chain: the wave's four rows are the four
//            features, one instruction stream -- for a full chunk interleaved into the MFMA chain (four adds
//            between two dependent MFMAs: pagk_chain_asm.h, quad_chunk_full);
//   solve    lane = feature: every lane of row q runs feature q's 4x4 LLT / solve / update (:302-344); four
//            different solves share each f64 divide and sqrt sequence.
//
// Sampling is per feature (64 lanes x one pixel each, a "chunk" of 64 patch pixels at a time, skipped for a
// feature that has left the iteration loop of the level); chunk c of all four features is sampled, then folded
// into the MFMA and cost chains, so LDS holds one chunk: 10000 B per wave, 16 waves per CU.
//
// The four features run the level's iterations in lockstep; a feature that converges early waits for the others
// of its wave at the level boundary (its block / row keeps accumulating values nobody reads).  Results are
// bit-identical to the oracle and to the other variants (tests/test_parity_gpu.py).
#pragma once
#include <cstddef>
#include "pagk_chain_asm.h"
#include "pagk_device.h"

namespace pagk {

constexpr int kLvSeqStride = 1024;  // ints between the counters of two ticket sequences: 4096 bytes apart (128 bytes
                                    // apart they still queue behind each other: 60000 features 1.62 -> 1.55 ms)

__device__ __forceinline__ void write_outputs(const TrackArgs &a, int i, float p2x, float p2y, int succ,
                                              float lastCost, int level0_ran, float ncc, int iters);
struct OutPtrs;
__device__ __forceinline__ void write_outputs_to(const TrackArgs &a, const OutPtrs &o, int i, float p2x, float p2y, int succ,
                                                 float lastCost, int level0_ran, float ncc, int iters);
__device__ __forceinline__ uint32_t lds_off(const void *p);

// 9840 bytes: eight 1280-byte LDS granules, 16 waves per CU (with the accumulators in a member of their own the
// ninth granule cost two resident waves per CU).
// Bank layout (64 banks x 4 B = a 256-B span; a ds_read_b64 is served in two halves of 32 lanes).  Lane 16 mk + 4 mq + mi
// reads, as its MFMA operand, stream mi of feature mq at pixel mk + 4 g: byte 64 mi + 16 mq + 8 mk (mod 256) -- per half
// (mk in {0, 1} or {2, 3}) the streams X / Y / NE occupy [0, 64) [64, 128) [128, 192) resp. [16, 80) [80, 144) [144, 208),
// the per-feature constants c sit at 192 + 16 mq + 8 mk.  Round 4: `ones` (the A operand of entry 3) used to start at
// byte 0 of the span, i.e. on the banks of stream X / feature 0 -- every A read was a 2-way conflict, one LDS cycle in
// three of the chunk phase (SQ_LDS_BANK_CONFLICT 29.8 M of SQ_LDS_IDX_ACTIVE 95.8 M per launch, profiles/r03_f_cfg3).
// It now starts at byte 160: [160, 176) is free in the first half ([128, 192) holds no A operand), [176, 192) in the
// second (tools/microbench16.hip: the A pattern reads conflict-free there, the B pattern always did).
struct QuadLds {
    double chunk[3][4][66];  // X | Y | NE streams of the current chunk: [stream][feature][pixel]; the strides put the
                             // three streams 64 B and the four features 16 B apart modulo the 256-B bank span.
                             // After the last chunk of an iteration its first 64 doubles hold D of the four blocks
                             // (quad_acc) until the solve has read them.
    double cconst[4][34];    // 16 x c per feature (A operand of entry 2, B operand of entry 3), 16 B apart mod 256
    double pad_[20];         // 160 bytes: puts `ones` on banks no A operand of the same half-wave uses
    double ones[16];         // 16 x 1.0 (A operand of entry 3: lane mk reads ones[mk + 4 u] for the four groups of a batch)
    float sq[4 * 129];       // per feature: carry, 64 squares (+ the 32 floats a chain may read past them); rows 129 floats
                             // apart (odd: the two rows of a 32-lane half use the even / the odd banks); last float read: 387 + 95
};
static_assert(offsetof(QuadLds, ones) % 256 == 160 && sizeof(QuadLds) <= 8 * 1280, "QuadLds: bank layout / eight LDS granules");
__host__ __device__ constexpr int quad_sq_row(int r) { return r * 129; }

__device__ __forceinline__ double (*quad_acc(QuadLds &S))[16] { return reinterpret_cast<double (*)[16]>(&S.chunk[0][0][0]); }

// A level description that was loaded from device memory (a batched launch's stream table), moved to scalar registers:
// it is the same for every lane, but a load through a plain pointer comes back in vector registers, and nine of those
// live across the whole level loop are nine spilled elsewhere.
__device__ __forceinline__ DevLevel uniform_level(const DevLevel &v)
{
    DevLevel u;
    const uint64_t q = (uint64_t)(uintptr_t)v.quad;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)q);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(q >> 32));
    u.quad = reinterpret_cast<const uint32_t *>((uintptr_t)(((uint64_t)hi << 32) | lo));
    u.cols = __builtin_amdgcn_readfirstlane(v.cols);
    u.rows = __builtin_amdgcn_readfirstlane(v.rows);
    u.fcols = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v.fcols)));
    u.frows = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v.frows)));
    u.fcols_m1 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v.fcols_m1)));
    u.frows_m1 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v.frows_m1)));
    return u;
}

// Comment lines in the ISA that delimit the fence-free hand-offs; __graft_entry__.build() runs tools/isa_handoff.py over the
// compiler's assembly and refuses a library in which what lies between them has lost its shape (agent scope on every
// store / load, the s_waitcnt vmcnt(0) in front of the publishing atomic and store).  The clobber keeps the compiler from
// moving memory operations across a marker.
#define PAGK_HANDOFF_MARK(text) asm volatile("; pagk-handoff: " text ::: "memory")

__device__ __forceinline__ float rl(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }

// ---- shared by k_track_quad and k_track_rows -----------------------------------------------------------------------
// MFMA operand roles of a lane (layout measured: A(q,i,k) in lane 16k+4q+i, B(q,k,j) in lane 16k+4q+j, D(q,i,j) in lane
// 16i+4q+j): k = pixel within the group of four, q = block = feature / row, i = entry.
//   A = J = (Ix, Iy, c, 1)[i]      B = (Ix, Iy, -e, c)[i]
struct QuadOperands {
    const double *a_src, *b_src;
    int a_step, b_step;  // doubles per four groups: 16 for a lane that walks a stream, 0 for one that re-reads constants
    int mk, mq, mi;
};
__device__ __forceinline__ QuadOperands quad_operands(QuadLds &S, int lane)
{
    QuadOperands o;
    o.mk = lane >> 4, o.mq = (lane >> 2) & 3, o.mi = lane & 3;
    o.a_src = o.mi < 2 ? &S.chunk[o.mi][o.mq][o.mk] : (o.mi == 2 ? &S.cconst[o.mq][o.mk] : &S.ones[o.mk]);
    o.b_src = o.mi < 3 ? &S.chunk[o.mi][o.mq][o.mk] : &S.cconst[o.mq][o.mk];
    o.a_step = o.mi < 2 ? 16 : 0;
    o.b_step = o.mi < 3 ? 16 : 0;
    return o;
}

// H, b and cost of chunk c (after the barrier that publishes the chunk).  A full chunk (64 pixels, 16 MFMA groups): one
// instruction stream in which the four DPP cost adds of a group sit between two dependent MFMAs (pagk_chain_asm.h:
// quad_chunk_full); the patch's last, shorter chunk: the MFMA chain, then the cost chain.  Returns the rows' running cost.
__device__ __forceinline__ float quad_chunk_phase(const QuadOperands &o, int P, int c, uint32_t sq_addr, double &d)
{
    const int left = P - 64 * c;  // valid pixels from this chunk on
    if (left >= 64)
        return quad_chunk_full(d, lds_off(o.a_src), lds_off(o.b_src), 8u * (uint32_t)o.a_step, 8u * (uint32_t)o.b_step, sq_addr);
    const int ng = left >> 2;     // complete groups of four (< 16)
    const double *pa = o.a_src, *pb = o.b_src;
    int m = 0;
    for (; m + 4 <= ng; m += 4) {
        double av[4], bv[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            av[u] = pa[4 * u];
            bv[u] = pb[4 * u];
        }
        pa += o.a_step;
        pb += o.b_step;
#pragma unroll
        for (int u = 0; u < 4; u++) d = __builtin_amdgcn_mfma_f64_4x4x4f64(av[u], bv[u], d, 0, 0, 0);
    }
    for (int u = 0; m < ng; m++, u++) d = __builtin_amdgcn_mfma_f64_4x4x4f64(pa[4 * u], pb[4 * u], d, 0, 0, 0);
    if (left & 3) {
        // last, incomplete group: a pixel past the patch contributes fma(-0.0, 1.0, d) = d exactly
        const int u = ng & 3;
        const bool pad = o.mk >= (left & 3);
        const double av = pa[4 * u], bv = pb[4 * u];
        d = __builtin_amdgcn_mfma_f64_4x4x4f64(pad ? -0.0 : av, pad ? 1.0 : bv, d, 0, 0, 0);
    }
    return chain_rows_f32<1>(sq_addr, 128u, 2);  // cost: ordered f32 sum, row q = feature q
}

// The normal equations of this lane's row from the accumulators (:302-319): D(q, i, j) sits in lane 16 i + 4 q + j; the
// caller has synchronised after quad_acc(S)[mq][mk * 4 + mi] = d.
__device__ __forceinline__ void quad_read_system(QuadLds &S, int row, int P, double cd, double (&H)[4][4], double (&b)[4])
{
    const double *A = quad_acc(S)[row];
    H[0][0] = A[0], H[1][0] = A[4], H[1][1] = A[5];
    H[2][0] = A[8], H[2][1] = A[9], H[2][2] = A[11];
    H[3][0] = A[12], H[3][1] = A[13];
    H[3][2] = (double)P * cd;  // sum of c*1.0: every partial sum k*c is exact
    H[3][3] = (double)P;       // sum of 1.0*1.0
    b[0] = A[2], b[1] = A[6], b[2] = A[10], b[3] = A[14];
}

// LEVELS (variant 7, "four features per wave, one LEVEL per wave"): the launch has n_levels x ceil(n / 4) waves and a
// wave runs ONE pyramid level of four features, a third of the lifetime of a whole-feature wave, so that a launch of one
// to a few rounds of resident waves does not end with the chip half empty behind a few long-lived waves.
//   Work is handed out by ticket counters in the order in which waves START: eight sequences (a.queue, 4096 bytes apart;
// one counter would serialise at ~23 ns per ticket), sequence x owning the quads q = x mod 8; a wave starts with the
// sequence of its XCD and moves on when a sequence is used up.  A sequence lists its work level step by level step,
// coarsest level first.  Ticket j of step 0 IS quad 8 j + x.  Ticket j of step k > 0 is "the j-th quad of this
// sequence to finish step k - 1": a wave that finishes a step appends its quad to the sequence's ready list of the next
// step (a.lv_ready) after storing (p2x, p2y, iteration count, handed-over flag) per feature to a.lv_state, and the
// consumer waits for entry j of that
// list -- not for a particular quad, so it waits only while the sequence really has nothing ready.  Every ticket of
// step k - 1 is lower than every ticket of step k, so the waves that will produce the entries have started, and they
// wait only on still lower steps: the chain ends at step 0, which waits on nothing (no deadlock, whatever order the
// hardware dispatches workgroups in; as many waves as items, so every wave finds one).  The wait is bounded all the same
// (a.lv_polls looks, ~3 us apart): a wave that gives up raises a.lv_error (the host turns it into an error code) and
// the launch drains.  Both sides use agent-scope relaxed atomics ordered by the wave's own instruction order -- state
// stored, stores waited for, entry stored; entry seen, state loaded.  No fences: an acquire per look invalidates the
// CU's vector cache under the waves that are sampling (measured: 4000 features 1.7 ms instead of 0.33), a release per
// item writes back the XCD's L2 (60000 features 2.8 ms instead of 1.5).  Arithmetic per level and feature is
// k_track_quad's, so results are bit-identical.
// The fence-free hand-off below is an argument about gfx942 / gfx950 (in-order return of a wave's loads, what vmcnt
// counts for an sc1 store); it is not the HIP memory model's.  Another target must not build it silently.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx942__) && !defined(__gfx950__)
#error "k_track_quad<.., LEVELS>: the level-to-level hand-off is written for gfx942 / gfx950 (see DESIGN.md section 4.3 (f))"
#endif
// BATCH (with LEVELS; pagk_track_device_batch): ONE launch for several camera streams that share the device -- BASELINE
// configs[4], "batched multi-camera".  The reference builds one PatchMatch per tracker (src/gyro_aided_tracker.cpp:276-283);
// k trackers' calls are k independent feature sets over k image pairs, so their quads share one ticket space and one set
// of ready lists: a quad knows its stream (a.batch[s]: both pyramids, the per-feature arrays), everything else is the
// launch's.  Per stream the arithmetic is the stream's own launch's, hence the same bits.  No hand-over (a batch is a
// launch of several rounds of resident waves, where the hand-over does not pay).
template <int NCH, bool LEAN = false, bool LEVELS = false, bool BATCH = false>   // LEAN: no penalty, solver_variant 0 (see track_block_body)
__global__ void __launch_bounds__(64, 4) k_track_quad(TrackArgs a)
{
    static_assert(!BATCH || LEVELS, "the batched launch is a one-level-per-wave launch");
    __shared__ __attribute__((aligned(256))) QuadLds S;
    const int lane = threadIdx.x, row = lane >> 4, lr = lane & 15;
    int quad = (int)blockIdx.x, lv_step = 0;
    int lv_seq = 0;   // LEVELS: the sequence this wave's item belongs to
    const int lv_cmax = LEVELS ? (((a.n + 3) >> 2) + 7) >> 3 : 0;  // ready-list slots per sequence and step
    if constexpr (LEVELS) {
        const int nq = (a.n + 3) >> 2;
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        int j = -1;
        for (int k = 0; k < 8 && j < 0; k++) {
            const int x = (int)((xcc + (unsigned)a.lv_shift + k) & 7u);
            const int cnt = (nq - x + 7) >> 3;  // quads of sequence x
            int t = 0;
            if (lane == 0) t = atomicAdd(a.queue + kLvSeqStride * x, 1);
            t = __builtin_amdgcn_readfirstlane(t);
            if (t < cnt * a.n_levels) {
                lv_step = t / cnt;
                j = t - lv_step * cnt;
                lv_seq = x;
            }
        }
        if (j < 0) return;  // (cannot happen: the launch has as many waves as there are items)
        if (lv_step == 0) {
            quad = 8 * j + lv_seq;
        } else {  // the j-th quad of the sequence to have finished the level above
            const int *entry = a.lv_ready + ((size_t)(lv_step - 1) * 8 + lv_seq) * lv_cmax + j;
            int polls = 0, e;
            while ((e = ld_agent(entry)) == 0) {
                if (++polls > a.lv_polls) {
                    // (never expected.)  The host reports the launch as failed at its next synchronisation; and because a
                    // caller on its own stream may never pass through one of ours, nothing stale may pass for a result:
                    // the quads this wave -- and, for want of its hand-off, the waves below it -- will not track are
                    // unknown here, so the launch's status array is cleared by the first wave that gives up
                    if (lane == 0) st_agent(a.lv_error, 1);
                    if constexpr (BATCH) {
                        for (int s = 0; s < a.batch_k; s++)
                            for (int k = lane; k < a.batch[s].n; k += 64) a.batch[s].status[k] = 0;
                    } else {
                        for (int k = lane; k < a.n; k += 64) a.status[k] = 0;
                    }
                    return;
                }
                __builtin_amdgcn_s_sleep(127);
            }
            // (what is loaded below from lv_state is indexed by `quad`, i.e. by the VALUE just loaded: the state loads are
            // address-dependent on the entry and a wave's loads return in order -- that dependency is the consumer's
            // ordering; do not index the state by the ticket)
            quad = __builtin_amdgcn_readfirstlane(e) - 1;
#ifdef PAGK_EXPERIMENT_ACQUIRE_ONCE
            // A/B build: ONE agent-scope acquire per wave, behind the look that found the entry -- not one per look
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
        }
    }
    // LEVELS: this wave's quad is through with its level -- the next level's consumers may have it
    auto lv_publish = [&]() {
        PAGK_HANDOFF_MARK("publish begin");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the rows' state has reached the coherence point
        if (lane == 0) {
            const int slot = atomicAdd(a.queue + kLvSeqStride * lv_seq + 64 * (1 + lv_step), 1);
            st_agent(a.lv_ready + ((size_t)lv_step * 8 + lv_seq) * lv_cmax + slot, quad + 1);
        }
        PAGK_HANDOFF_MARK("publish end");
    };
    // hand-over: this wave hands nothing over from here on (the finishers stop waiting once every wave has said so; its
    // list entries were stored with agent-scope atomics behind a fence of their own, this wave's instruction order does
    // the rest: no release here -- it would write back the XCD's L2 once per wave)
    auto wave_ended = [&]() {
        // (LEVELS: only the wave of a quad's LAST level reports -- the waves of its other levels ended before that one
        // started; a third of the atomics on the one counter every wave of the launch shares)
        if (LEVELS && lv_step != a.n_levels - 1) return;
        if (a.iter_budget > 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) atomicAdd(a.susp_count + 2, 1);
        }
    };
    // the kernel is instantiated per patch size (NCH = chunks of 64 pixels: 2 <-> h = 5, 4 <-> h = 7, 7 <-> h = 10): the
    // patch geometry is a compile-time constant (divisions by the patch width, LDS addresses, chunk lengths)
    constexpr int h = NCH == 7 ? 10 : (NCH == 4 ? 7 : 5), Wd = 2 * h + 1, P = Wd * Wd;
    static_assert(NCH == 2 || NCH == 4 || NCH == 7, "instantiated for h = 5, 7, 10");
    // the arrays of this quad's features: the launch's, or -- BATCH -- those of the camera stream the quad belongs to
    // (wave-uniform: scalar loads).  Accessors, not local copies: a local copy of a kernel argument is loaded at once and
    // lives in scalar registers from here on, and the non-batched kernel -- which sits at its register limit -- then
    // spills one register more inside its loop (13 -> 14, +4 % run time; profiles/r04_ab3_batch_refactor_spill.log).
    int quad_local = quad;
    const BatchStream *bs = nullptr;
    if constexpr (BATCH) {
        int s = 0;
        for (int k = 1; k < a.batch_k; k++) s = quad >= a.batch[k].quad_base ? k : s;   // (quad_base ascends)
        bs = a.batch + s;
        quad_local = quad - bs->quad_base;
    }
    auto v_pt_ref = [&]() -> const float * { if constexpr (BATCH) return bs->pt_ref; else return a.pt_ref; };
    auto v_pt_init = [&]() -> const float * { if constexpr (BATCH) return bs->pt_init; else return a.pt_init; };
    auto v_affine = [&]() -> const float * { if constexpr (BATCH) return bs->affine; else return a.affine; };
    auto v_status_in = [&]() -> const uint8_t * { if constexpr (BATCH) return bs->status_in; else return a.status_in; };
    auto v_n = [&]() -> int { if constexpr (BATCH) return bs->n; else return a.n; };
    // `raw` / `fi` index those arrays; BATCH numbers the features through the whole launch as 4 * quad + row (lv_state;
    // rows past a stream's end exist in that numbering and are never read)
    const int raw = 4 * quad_local + row;
    const int fi = raw < v_n() ? raw : v_n() - 1;  // rows past the end shadow the last feature and write nothing
    const bool live = raw < v_n() && v_status_in()[fi] != 0;
    auto emit_outputs = [&](int i, float ox, float oy, int osucc, float ocost, int ran, float oncc, int oiters) {
        if constexpr (BATCH) {
            const OutPtrs o{bs->pt_un, bs->pt_dist, bs->status, bs->pix_err, bs->dist_pred, bs->ncc, bs->iters,
                            bs->pt_init ? bs->pt_init : bs->pt_ref};
            write_outputs_to(a, o, i, ox, oy, osucc, ocost, ran, oncc, oiters);
        } else {
            write_outputs(a, i, ox, oy, osucc, ocost, ran, oncc, oiters);
        }
    };

    const float *init = a.has_gyro ? v_pt_init() : v_pt_ref();  // :85-89
    float p2x = init[2 * fi], p2y = init[2 * fi + 1];
    float A00 = 1, A01 = 0, A10 = 0, A11 = 1;
    if (a.use_affine) {
        A00 = v_affine()[4 * fi], A01 = v_affine()[4 * fi + 1], A10 = v_affine()[4 * fi + 2], A11 = v_affine()[4 * fi + 3];
    }
    const float refx = v_pt_ref()[2 * fi], refy = v_pt_ref()[2 * fi + 1];
    const float fh = (float)h;
    const float ext_x = fabsf(A00) * fh + fabsf(A01) * fh + 2.0f;
    const float ext_y = fabsf(A10) * fh + fabsf(A11) * fh + 2.0f;
    if (__ballot(live) == 0ull) {  // nothing to track in this wave (:173)
        if (LEVELS && lv_step != a.n_levels - 1) {  // (the quad's last item reports)
            lv_publish();
            wave_ended();
            return;
        }
        if (lr == 0 && raw < v_n()) emit_outputs(fi, p2x, p2y, 0, 0.0f, 0, 0.0f, 0);
        wave_ended();
        return;
    }

    // lane -> patch pixel of chunk c: p = 64 c + lane, row-major (y outer, :233-234)
    auto patch_xy = [&](int c, float &x, float &y) {
        int p = 64 * c + lane;
        p = p < P ? p : P - 1;
        const int yy = p / Wd, xx = p - yy * Wd;
        x = (float)(xx - h);
        y = (float)(yy - h);
    };

    const QuadOperands ops = quad_operands(S, lane);  // MFMA operand roles of this lane
    const int mk = ops.mk, mq = ops.mq, mi = ops.mi;
    if (lane < 16) S.ones[lane] = 1.0;
    const uint32_t sq_addr = lds_off(&S.sq[quad_sq_row(row)]) + 8u * lr;

    int succ = 1, iters = 0;
    float lastCost = 0.0f;
    bool susp = false;  // this row's feature was handed to k_track_resume (TrackArgs::iter_budget)
    if constexpr (LEVELS) {
        if (lv_step > 0) {  // take over from the item of the level above
            PAGK_HANDOFF_MARK("take-over begin");
            const int *st = reinterpret_cast<const int *>(a.lv_state + 4 * (size_t)(BATCH ? 4 * quad + row : fi));
            p2x = __int_as_float(ld_agent(st + 0)), p2y = __int_as_float(ld_agent(st + 1));
            iters = ld_agent(st + 2);
            susp = ld_agent(st + 3) != 0;  // handed to the latency kernel on a level above
            PAGK_HANDOFF_MARK("take-over end");
        }
    }
    const int level_first = LEVELS ? a.n_levels - 1 - lv_step : a.n_levels - 1;
    const int level_last = LEVELS ? level_first : 0;
#ifdef PAGK_STAMPS
    // diagnostic build only: cycles per phase of this wave -> a.dbg[16 * wave + k]: [0] level setup, [1] sampling,
    // [2] MFMA chain, [3] cost chain, [4] solve + update, [5] total, [6] wave-iterations
    unsigned long long qst[7] = {0, 0, 0, 0, 0, 0, 0};
    unsigned long long qt0 = __builtin_amdgcn_s_memtime(), qt1;
    const unsigned long long qbegin = qt0;
    const unsigned long long qreal0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz, one clock for the whole device
#define QSTAMP(k)                          \
    qt1 = __builtin_amdgcn_s_memtime();    \
    qst[k] += qt1 - qt0;                   \
    qt0 = qt1;
#else
#define QSTAMP(k)
#endif

    for (int level = level_first; level >= level_last; level--) {
        const DevLevel L1b = BATCH ? uniform_level(bs->l1[level]) : DevLevel();
        const DevLevel &L1 = BATCH ? L1b : a.l1[level];
        const DevLevel L2 = BATCH ? uniform_level(bs->l2[level]) : pin_level(a.l2[level]);
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
        bool act = live && !susp;

        // img1 samples are iteration-invariant (bit-identical to :253, :263): once per level, for all four features
        const float cneg = -sample<true>(L1, ptx, pty);
        const double cd = (double)cneg;
        // ... kept in this wave's slice of the global workspace (4 x NCH x 64 floats, read back coalesced, one load
        // per sampled pixel): 28 registers at h = 10 that would otherwise spill
        float *ws = a.ws + (size_t)blockIdx.x * (4 * NCH * 64) + lane;
        // (a feature's NCH gathers are requested together: one round trip per feature instead of one per chunk)
#pragma nounroll
        for (int f = 0; f < 4; f++) {
            const float fx = rl(ptx, 16 * f), fy = rl(pty, 16 * f);
            OneTap t1[NCH];
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                float x, y;
                patch_xy(c, x, y);
                t1[c] = sample_issue<true>(L1, fx + x, fy + y);
            }
#pragma unroll
            for (int c = 0; c < NCH; c++) ws[(f * NCH + c) * 64] = sample_finish(t1[c]);
        }
        __syncthreads();  // the previous level's readers of cconst are done
        S.cconst[row][lr] = cd;
        __syncthreads();
        QSTAMP(0)

        for (int iter = 0; iter < a.iterations; iter++) {  // :215
            // Continuation: a feature that has used up the launch's iteration budget without finishing leaves this
            // wave here, between two iterations -- everything the loop carries (:186-194, :332-340) goes to
            // susp_state -- and is finished by k_track_resume, the latency kernel.  The slowest features (a handful
            // run 3-5x the mean iteration count) would otherwise keep a whole throughput wave, and the launch, waiting.
            // (susp_lone: only a feature that is the LAST one iterating in its wave leaves -- the rows beside it are
            // waiting for nothing else)
            const bool lone = __popcll(__ballot(act)) == 16;  // one row of sixteen lanes
            if (a.iter_budget > 0 && act && iters >= a.iter_budget && iter > 0 && (!a.susp_lone || lone)) {
                if (lr == 0) {
                    // state first, then the list entry that publishes it (the finisher runs concurrently)
                    int *st = reinterpret_cast<int *>(&a.susp_state[fi]);
                    st_agent(st + 0, level), st_agent(st + 1, iter);
                    st_agent(st + 2, __float_as_int(dx)), st_agent(st + 3, __float_as_int(dy));
                    st_agent(st + 4, __float_as_int(dg)), st_agent(st + 5, __float_as_int(db));
                    st_agent(st + 6, __float_as_int(lastCost)), st_agent(st + 7, iters);
                    __threadfence();
                    const int slot = atomicAdd(a.susp_count, 1);
                    __hip_atomic_store(a.susp_list + slot, fi + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                }
                act = false;
                susp = true;
            }
            const unsigned long long actm = __ballot(act);
            if (actm == 0ull) break;
            if (act) iters++;
            const float bx = ptx + dx, by = pty + dy;  // (pt.x + dx), then + wx (:252)
            const float gain = 1.0f + dg;
            const bool interior = (bx - ext_x >= 0.0f) && (bx + ext_x < L2.fcols_m1) && (by - ext_y >= 0.0f) &&
                                  (by + ext_y < L2.frows_m1);
            const unsigned long long intm = __ballot(interior);
            double d = 0.0;      // the four 4x4 accumulators (:217-218 H = 0, b = 0)
            float carry = 0.0f;  // cost = 0 (:283)

#pragma nounroll
            for (int c = 0; c < NCH; c++) {
                // ---- sampling: chunk c of every active feature --------------------------------------
                float x, y;
                patch_xy(c, x, y);
                const bool valid = 64 * c + lane < P;
                auto issue = [&](int f, FiveTaps &tp, float &s1v) {
                    const int src = 16 * f;
#ifndef PAGK_EXPERIMENT_RECOMPUTE_I1
                    // the feature's img1 sample of this pixel first: loads return in order, so it is there when the taps are
                    s1v = ws[(f * NCH + c) * 64];
#else
                    s1v = sample<true>(L1, rl(ptx, src) + x, rl(pty, src) + y);
#endif
                    float wx = x, wy = y;
                    if (a.use_affine) {  // :203-204
                        wx = rl(A00, src) * x + rl(A01, src) * y;
                        wy = rl(A10, src) * x + rl(A11, src) * y;
                    }
                    const float X = rl(bx, src) + wx, Y = rl(by, src) + wy;
                    tp = ((intm >> src) & 1ull) ? sample5_issue<false>(L2, X, Y) : sample5_issue<true>(L2, X, Y);
                };
                // the active features of the wave, in order (wave-uniform); the gathers of feature k+1 are issued
                // before feature k's are consumed, so that a round's latency hides behind the previous round's math.
                // Two named tap sets alternate and every consume sits on its own control path: a copy "cur = nxt" or a
                // consume shared by the paths with / without a following issue would each make the compiler wait
                // for ALL outstanding loads (s_waitcnt vmcnt(0)) -- the pipelining would exist in the source only.
                auto consume = [&](int f, const FiveTaps &tp, const float s1v) {
                    const int src = 16 * f;
                    const Five s = sample5_finish(tp);
                    const float e = s.c + rl(db, src) - rl(gain, src) * s1v;  // :252-253
                    const float Ix = 0.5f * (s.xp - s.xm);                     // :259-260
                    const float Iy = 0.5f * (s.yp - s.ym);                     // :261-262
                    S.chunk[0][f][lane] = (double)Ix;
                    S.chunk[1][f][lane] = (double)Iy;
                    S.chunk[2][f][lane] = -(double)e;
                    S.sq[quad_sq_row(f) + 1 + lane] = valid ? e * e : 0.0f;  // :294; past the patch: + 0.0f changes nothing
                };
                auto first_of = [](unsigned long long m) { return (int)(__builtin_ctzll(m) >> 4); };
                auto without = [](unsigned long long m, int f) { return m & ~(0xffffull << (16 * f)); };
                int fa = first_of(actm), fb = 0;
                unsigned long long rest = without(actm, fa);
                FiveTaps ta, tb;
                float s1a, s1b;
                bool last_in_a = true;
                issue(fa, ta, s1a);
#pragma nounroll
                while (rest) {  // a next feature exists: its gathers go out before the current taps are consumed
                    fb = first_of(rest);
                    rest = without(rest, fb);
                    issue(fb, tb, s1b);
                    __builtin_amdgcn_sched_barrier(0);
                    consume(fa, ta, s1a);
                    __builtin_amdgcn_sched_barrier(0);
                    if (!rest) {
                        last_in_a = false;
                        break;
                    }
                    fa = first_of(rest);
                    rest = without(rest, fa);
                    issue(fa, ta, s1a);
                    __builtin_amdgcn_sched_barrier(0);
                    consume(fb, tb, s1b);
                    __builtin_amdgcn_sched_barrier(0);
                }
                // the last feature of the chunk: nothing is in flight behind it (kept apart from the consumes in the
                // loop: a shared copy would have to wait as if nothing were in flight there either)
                if (last_in_a) {
                    asm volatile("; last taps: set a");
                    consume(fa, ta, s1a);
                } else {
                    asm volatile("; last taps: set b");
                    consume(fb, tb, s1b);
                }
                if (lr == 0) S.sq[quad_sq_row(row)] = carry;  // running cost = first term of this chunk's chain (0 + s == s)
                __syncthreads();
                QSTAMP(1)
                carry = quad_chunk_phase(ops, P, c, sq_addr, d);  // H, b and cost of the chunk
                __syncthreads();  // the chunk has been read before the next one is written
                QSTAMP(3)
            }
            // ---- solve (:302-319): D(q, i, j) sits in lane 16 i + 4 q + j; every lane of row q solves feature q
            quad_acc(S)[mq][mk * 4 + mi] = d;
            __syncthreads();
            double H[4][4], b[4], upd[4];
            quad_read_system(S, row, P, cd, H, b);
            float cost = carry;
            if constexpr (!LEAN) {
                if (a.penalty) add_penalty(a, dx, dy, H, b, cost);
            }
            double unorm = 0.0;
            // (rows without an iterating feature hold stale sums: they sit the solve out, so that only live systems can
            // raise the exception flags that send the wave to the plain-division form)
            if (act) unorm = llt4_solve_nsq(H, b, upd, LEAN ? 0u : a.solver);  // update.squaredNorm()
            __syncthreads();  // the accumulators' LDS is the next iteration's first chunk
            // ---- update + termination (:322-344), per feature ------------------------------------------
            if (act) {
                if (upd[0] != upd[0]) {  // :322
                    succ = 0;
                    act = false;
                } else if (iter > 0 && cost > lastCost) {  // :328
                    act = false;
                } else {
                    dx = (float)((double)dx + upd[0]);  // :332
                    dy = (float)((double)dy + upd[1]);
                    if (a.illum) {  // :334-337
                        dg = (float)((double)dg + upd[2]);
                        db = (float)((double)db + upd[3]);
                    }
                    lastCost = cost;  // :339
                    succ = 1;
                    if (unorm < kNormSqConverged) act = false;  // :343  update.norm() < 1e-2
                }
            }
#ifdef PAGK_STAMPS
            asm volatile("" : "+v"(dx), "+v"(dy));
            qst[6]++;
            QSTAMP(4)
#endif
        }
        p2x = ptx + dx;  // :348
        p2y = pty + dy;
    }
#ifdef PAGK_STAMPS
    if (lane == 0 && a.dbg) {
        qst[5] = __builtin_amdgcn_s_memtime() - qbegin;
        for (int k = 0; k < 7; k++) a.dbg[(size_t)blockIdx.x * 16 + k] = qst[k];
        a.dbg[(size_t)blockIdx.x * 16 + 7] = qreal0;
        a.dbg[(size_t)blockIdx.x * 16 + 8] = __builtin_amdgcn_s_memrealtime();
    }
#endif
#undef QSTAMP
    if constexpr (LEVELS) {
        if (level_last > 0) {  // hand the quad to the next level: state, then the ready-list entry that publishes it
            PAGK_HANDOFF_MARK("state begin");
            if (lr == 0 && raw < v_n()) {
                int *st = reinterpret_cast<int *>(a.lv_state + 4 * (size_t)(BATCH ? 4 * quad + row : fi));
                st_agent(st + 0, __float_as_int(p2x)), st_agent(st + 1, __float_as_int(p2y)), st_agent(st + 2, iters);
                st_agent(st + 3, susp ? 1 : 0);
            }
            PAGK_HANDOFF_MARK("state end");
            lv_publish();
            wave_ended();
            return;
        }
    }
    if (lr == 0 && raw < v_n() && !susp) {
        if (live)
            emit_outputs(fi, p2x, p2y, succ, lastCost, 1, 1.0f, iters);  // :365 ncc = 1 (calc_ncc runs another variant)
        else
            emit_outputs(fi, init[2 * fi], init[2 * fi + 1], 0, 0.0f, 0, 0.0f, 0);
    }
    wave_ended();
}

}  // namespace pagk
