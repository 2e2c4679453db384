// pagk_device.h -- device-side building blocks of the PatchMatch hot path (gfx950).
//
// Everything here must round exactly like the reference's baseline x86-64 build: one
// IEEE rounding per operation.  The translation unit is compiled with
// -ffp-contract=off (no FMA contraction), f32 denormals preserved (hipcc default on
// gfx9), correctly rounded f32/f64 divide and sqrt (hipcc defaults).
//
// file:line citations are relative to the reference checkout.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pagk {

constexpr int kMaxLevels = 8;

// One pyramid level as the kernels see it: the "quad image".  quad[r*cols + c] packs the
// four bilinear taps of pixel (r, c) exactly as PatchMatch::GetPixelValue addresses them
// (src/patch_match.cpp:399-403): byte0 = data[off], byte1 = data[off+1],
// byte2 = data[off+step], byte3 = data[off+step+1], off = r*step + c, linear addressing
// (so column cols-1 pairs with the next row's first byte when the image is continuous),
// bytes past the buffer = 0.  One aligned dword load per bilinear sample.
struct DevLevel {
    const uint32_t *quad;
    int cols, rows;
    float fcols, frows;     // (float)cols, (float)rows        -- the `x >= img.cols` compare
    float fcols_m1, frows_m1; // (float)(cols-1), (float)(rows-1) -- the clamp target
};

// Device-scope loads / stores that bypass the per-CU vector cache: the hand-over list is written by one kernel and read
// by another one running at the same time.
__device__ __forceinline__ int ld_agent(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// A feature handed over mid-flight by a throughput kernel to the latency kernel (k_track_resume): everything the
// Gauss-Newton loop carries from one iteration to the next (src/patch_match.cpp:186-194, :332-340).  The level's
// constants (pt, the img1 samples) are recomputed by the resuming kernel: they depend on the inputs only.
struct SuspState {
    int level;     // pyramid level being iterated
    int iter;      // index of the next iteration at that level (>= 1: at least one iteration ran there)
    float dx, dy, dg, db, lastCost;
    int iters;     // iterations executed so far, all levels (diagnostic output)
};

// One camera stream of a batched launch (pagk_track_device_batch: k_track_quad<.., LEVELS, BATCH>): what differs from
// stream to stream -- both pyramids, the per-feature arrays and their count.  Everything else (patch size, iteration
// limit, flags, camera model) is the launch's.  Quads are numbered through the batch: stream s owns the quads
// [quad_base, quad_base + ceil(n / 4)).
struct BatchStream {
    DevLevel l1[kMaxLevels], l2[kMaxLevels];
    const float *pt_ref, *pt_init, *affine;
    const uint8_t *status_in;
    float *pt_un, *pt_dist;
    uint8_t *status;
    double *pix_err, *dist_pred;
    float *ncc;
    int *iters;
    int n;
    int quad_base;
};

struct TrackArgs {
    DevLevel l1[kMaxLevels], l2[kMaxLevels];
    float scales[kMaxLevels];
    int n_levels;
    int n;
    const float *pt_ref;
    const float *pt_init;
    const float *affine;
    const uint8_t *status_in;
    float *pt_un;
    float *pt_dist;
    uint8_t *status;
    double *pix_err;
    double *dist_pred;
    float *ncc;
    int *iters;
    unsigned long long *dbg;  // diagnostic builds only (PAGK_STAMPS)
    float *ws;                // k_track_quad: per-wave scratch for the iteration-invariant img1 samples
    // continuation (large launches): a feature that has executed `iter_budget` iterations without finishing is
    // suspended -- its state goes to susp_state[feature], its index is appended to susp_list -- and finished by
    // k_track_resume.  iter_budget = 0: never suspend.
    int iter_budget;
    int *susp_count;   // [0] entries published, [1] tickets taken by the finisher, [2] throughput waves that have ended
    int *susp_list;    // entry k: 0 = not (yet) published, feature + 1 = waiting, -(feature + 1) = finished
    SuspState *susp_state;
    int susp_lone;     // suspend a feature only when no other row of its wave is iterating
    int susp_waves;    // waves of the throughput launch (what susp_count[2] reaches)
    int susp_polls;    // how often a finisher workgroup looks for its entry before it gives up (bounded: never a hang)
    // k_track_rows: the next feature index to hand out (zeroed before every launch); k_track_quad<.., LEVELS>: the ticket
    int *queue;
    // k_track_quad<.., LEVELS>: queue = eight sequences' counters, 1024 ints apart ([0] tickets, [64 (1 + k)] quads that
    // have finished level step k: a cache line each); lv_ready[(k * 8 + sequence) * ceil(quads / 8) + slot] = quad + 1, in the order in which the
    // sequence's quads finished step k (all zeroed before every launch); lv_state[4 feature] = (p2x, p2y, iterations so
    // far, -) handed from one level's wave to the next; lv_error: a wait ran out
    int *lv_ready;
    float *lv_state;
    int *lv_error;
    int lv_shift;      // test knob: a wave starts with the sequence of XCD (its own + lv_shift) mod 8 -- every hand-off then crosses XCDs
    int lv_polls;      // looks (~3 us apart) a wave takes at its ready-list entry before it gives up (bounded: never a hang)
    const BatchStream *batch;  // batched launch: batch_k streams (device memory); n is then 4 x the batch's quads
    int batch_k;
    int half, iterations;
    int prio_k;  // pagk_prio.h: a 4-wave workgroup past prio_k iterations per level entered is behind (0: never)
    // PAGK_PRIO_K=auto: the threshold follows the workload.  prio_stats[0] / [1] = iterations / feature-levels this context's 4-wave
    // launches have run (device memory, cumulative, seeded with the BASELINE mean), prio_kbuf = ceil of their ratio in 3..12, refreshed by the
    // workgroups that finish early and read -- one load, consumed a level set-up later -- by every workgroup of the NEXT launches
    unsigned long long *prio_stats;
    const int *prio_kbuf;
    int has_gyro, illum, use_affine, penalty, calc_ncc;
    uint32_t solver;        // pagk_params::solver_variant (SV_* bits)
    float lam_invlog;       // mLambda * mInvLogMaxDist            (f32 product, :305)
    float lam_invlog_alpha; // mLambda * mInvLogMaxDist * mAlpha   (f32 product, :307)
    float alpha;
    double win_size_inv;    // mWinSizeInv (:57)
    int distort_on;         // mDistCoef(0) != 0 (:410)
    float fx, fy, cx, cy, fx_inv, fy_inv, k1, k2, p1, p2, k3;
};

// A level's description copied out of the kernel-argument segment.  Pinning the fields in scalar registers with
// `asm volatile("" : "+s"(x))` (so that the compiler cannot re-fetch them with s_load inside the Gauss-Newton loop)
// was measured and dropped: the extra live SGPRs spill into VGPR lanes and the launch got 2-3 % slower (126.0 vs
// 123.0 us at 1000 features, 312 vs 304 us at 4000; profiles/r02_ab_runs.md).  The re-fetches are issued early and
// their latency is covered by the coordinate arithmetic.
__device__ __forceinline__ DevLevel pin_level(const DevLevel &src) { return src; }

// ---- bilinear sampler ---------------------------------------------------------------------------
// One coordinate of PatchMatch::GetPixelValue (src/patch_match.cpp:394-401): clamp, integer
// part, fraction and its complement.
struct Coord {
    int i;      // int(x)
    float f;    // xx = x - floor(x)
    float omf;  // 1.0f - xx
};

template <bool CLAMP>
__device__ __forceinline__ Coord prep_coord(float x, float fmax, float fmax_m1)
{
    if (CLAMP) {
        // `if (x < 0) x = 0;` -- fmaxf also maps NaN to 0 (defined behaviour of this
        // implementation; the reference's int(NaN) is undefined).  -0.0 vs +0.0 is
        // immaterial: both give i = 0, xx = 0.
        x = fmaxf(x, 0.0f);
        x = (x >= fmax) ? fmax_m1 : x;  // `if (x >= img.cols) x = img.cols - 1;`
    }
    Coord c;
    c.i = (int)x;  // truncation == floor for x >= 0
    // xx = x - floor(x) (:400).  For x >= 0 (guaranteed: the clamp, or the interior test) the subtraction
    // is exact and v_fract_f32 returns exactly that value; its only deviation from x - floor(x) is a
    // clamp below 1.0 for tiny negative x, which cannot occur here.  One instruction instead of two.
    c.f = __builtin_amdgcn_fractf(x);
    c.omf = 1.0f - c.f;
    return c;
}

// b*(a*d0 + xx*d1) + yy*(a*d2 + xx*d3), src/patch_match.cpp:402-403, this association.
__device__ __forceinline__ float bilerp(uint32_t q, const Coord &cx, const Coord &cy)
{
    float d0 = (float)(q & 0xffu);
    float d1 = (float)((q >> 8) & 0xffu);
    float d2 = (float)((q >> 16) & 0xffu);
    float d3 = (float)(q >> 24);
    float top = cx.omf * d0 + cx.f * d1;
    float bot = cx.omf * d2 + cx.f * d3;
    return cy.omf * top + cy.f * bot;
}

// ---- the sampler's two hardware-specific forms -------------------------------------------------------------------
// Default build: tap loads through a typed buffer resource (the texture addresser converts the four bytes to floats)
// and the interpolation in packed f32 (the quad's two rows side by side in v_pk_mul_f32 / v_pk_add_f32).  Both are
// bit-identical to the classic form (byte unpacking with v_cvt_f32_ubyte, scalar f32 arithmetic), which
// -DPAGK_CLASSIC_SAMPLER builds for A/B runs.  Measured on MI355X (same-session A/B, profiles/r02_ab_runs.md): 53
// instead of 135 VALU instructions per patch pixel; -1.6 % launch time at 1000 features (the launch is latency-bound
// there), -5 % at 20000 (four features per wave), -7.5 % at 8000.
#if !defined(PAGK_CLASSIC_SAMPLER)
#ifndef PAGK_PK_BILERP
#define PAGK_PK_BILERP
#endif
#ifndef PAGK_TYPED_TAPS
#define PAGK_TYPED_TAPS
#endif
#endif
typedef float pagk_f32x4 __attribute__((ext_vector_type(4)));
typedef float pagk_f32x2 __attribute__((ext_vector_type(2)));
typedef int pagk_i32x4 __attribute__((ext_vector_type(4)));

#ifdef PAGK_TYPED_TAPS
// The four taps of a quad arrive as four floats: the texture addresser converts them.  A quad dword is read
// through a buffer resource whose element format is 8_8_8_8 / USCALED (unsigned byte -> float, exact), so
// `buffer_load_format_xyzw` returns (float)byte0..3 and the 4 x v_cvt_f32_ubyte per sample disappear from the
// VALU stream; `idxen` with a 4-byte stride takes the element index as is (no 64-bit address arithmetic).  The
// destination swizzle delivers (d0, d2 | d1, d3).  clang has no builtin for the format loads; the LLVM intrinsic is
// reached through its asm label.  Bit-identical to the byte unpacking on MI355X (the parity suite runs through it).
// Price: 16 bytes per lane come back from the texture unit instead of 4, and 4 VGPRs per tap in flight.
__device__ pagk_f32x4 pagk_buffer_load_format_xyzw(pagk_i32x4 rsrc, int vindex, int voffset, int soffset, int aux)
    __asm("llvm.amdgcn.struct.buffer.load.format.v4f32");

struct TapSrc {
    pagk_i32x4 rs;
};
__device__ __forceinline__ TapSrc tap_src(const DevLevel &L)
{
    const uint64_t base = (uint64_t)(uintptr_t)L.quad;
    TapSrc t;
    t.rs.x = (int)(uint32_t)base;
    t.rs.y = (int)((uint32_t)(base >> 32) & 0xffffu) | (4 << 16);  // stride 4 bytes
    t.rs.z = L.cols * L.rows;                                       // records
    // dst_sel x,y,z,w = byte0 (R), byte2 (B), byte1 (G), byte3 (A); num_format USCALED (2); data_format 8_8_8_8 (10)
    t.rs.w = 4 | (6 << 3) | (5 << 6) | (7 << 9) | (2 << 12) | (10 << 15);
    return t;
}
typedef pagk_f32x4 Taps;   // (d0, d2, d1, d3) as floats
__device__ __forceinline__ Taps load_taps(const TapSrc &t, int idx) { return pagk_buffer_load_format_xyzw(t.rs, idx, 0, 0, 0); }
__device__ __forceinline__ pagk_f32x4 taps_f32(Taps q) { return q; }
#else
struct TapSrc {
    const uint32_t *quad;
};
__device__ __forceinline__ TapSrc tap_src(const DevLevel &L) { return TapSrc{L.quad}; }
typedef uint32_t Taps;     // the packed quad dword; converted when consumed
// unsigned 32-bit element offsets: SGPR base + VGPR offset addressing, no 64-bit pointer math
__device__ __forceinline__ Taps load_taps(const TapSrc &t, int idx) { return t.quad[(uint32_t)idx]; }
__device__ __forceinline__ pagk_f32x4 taps_f32(Taps q)
{
    pagk_f32x4 d;   // (d0, d2, d1, d3): v_cvt_f32_ubyte0/2/1/3 straight into the two register pairs
    d.x = (float)(q & 0xffu);
    d.y = (float)((q >> 16) & 0xffu);
    d.z = (float)((q >> 8) & 0xffu);
    d.w = (float)(q >> 24);
    return d;
}
#endif

// b*(a*d0 + xx*d1) + yy*(a*d2 + xx*d3) (:402-403) on the pairs (d0, d2), (d1, d3): the same seven roundings in
// the same association, the two rows of the quad side by side in v_pk_mul_f32 / v_pk_add_f32 (IEEE per component,
// no contraction).  cx = (a, xx) = (1 - xx, xx) of the x coordinate, cy = (b, yy) of the y coordinate.
__device__ __forceinline__ float bilerp_pk(pagk_f32x4 d, pagk_f32x2 cx, pagk_f32x2 cy)
{
    const pagk_f32x2 rows = d.xy * cx.xx + d.zw * cx.yy;  // (a*d0 + xx*d1, a*d2 + xx*d3)
    const pagk_f32x2 t = rows * cy;                       // (b*top, yy*bot)
    return t.x + t.y;
}

struct CoordPk {
    int i;            // int(x)
    pagk_f32x2 w;     // (1 - xx, xx)
};

template <bool CLAMP>
__device__ __forceinline__ CoordPk prep_coord_pk(float x, float fmax, float fmax_m1)
{
    if (CLAMP) {
        x = fmaxf(x, 0.0f);
        x = (x >= fmax) ? fmax_m1 : x;
    }
    CoordPk c;
    c.i = (int)x;
    const float f = __builtin_amdgcn_fractf(x);
    c.w.y = f;
    c.w.x = 1.0f - f;
    return c;
}

template <bool CLAMP>
__device__ __forceinline__ float sample(const DevLevel &L, float x, float y)
{
#ifdef PAGK_PK_BILERP
    const CoordPk cx = prep_coord_pk<CLAMP>(x, L.fcols, L.fcols_m1);
    const CoordPk cy = prep_coord_pk<CLAMP>(y, L.frows, L.frows_m1);
    return bilerp_pk(taps_f32(load_taps(tap_src(L), __mul24(cy.i, L.cols) + cx.i)), cx.w, cy.w);
#else
    Coord cx = prep_coord<CLAMP>(x, L.fcols, L.fcols_m1);
    Coord cy = prep_coord<CLAMP>(y, L.frows, L.frows_m1);
    return bilerp(L.quad[(uint32_t)(__mul24(cy.i, L.cols) + cx.i)], cx, cy);
#endif
}

// One sample in two halves (tap load requested / interpolated), so that a batch of independent samples has all its
// gathers in flight at once.  Same arithmetic as sample<CLAMP>().
struct OneTap {
    Taps q;
    float fx, fy;  // the fractions xx, yy; (1 - xx) is formed at use: the same single rounding
};
template <bool CLAMP>
__device__ __forceinline__ OneTap sample_issue(const DevLevel &L, float x, float y)
{
#ifdef PAGK_PK_BILERP
    const CoordPk cx = prep_coord_pk<CLAMP>(x, L.fcols, L.fcols_m1);
    const CoordPk cy = prep_coord_pk<CLAMP>(y, L.frows, L.frows_m1);
    OneTap t;
    t.q = load_taps(tap_src(L), __mul24(cy.i, L.cols) + cx.i);
    t.fx = cx.w.y, t.fy = cy.w.y;
    return t;
#else
    const Coord cx = prep_coord<CLAMP>(x, L.fcols, L.fcols_m1);
    const Coord cy = prep_coord<CLAMP>(y, L.frows, L.frows_m1);
    OneTap t;
    t.q = L.quad[(uint32_t)(__mul24(cy.i, L.cols) + cx.i)];
    t.fx = cx.f, t.fy = cy.f;
    return t;
#endif
}
__device__ __forceinline__ float sample_finish(const OneTap &t)
{
#ifdef PAGK_PK_BILERP
    pagk_f32x2 wx, wy;
    wx.x = 1.0f - t.fx, wx.y = t.fx;
    wy.x = 1.0f - t.fy, wy.y = t.fy;
    return bilerp_pk(taps_f32(t.q), wx, wy);
#else
    const Coord cx{0, t.fx, 1.0f - t.fx}, cy{0, t.fy, 1.0f - t.fy};
    return bilerp(t.q, cx, cy);
#endif
}

// The five img2 samples one pixel of the GN loop needs (src/patch_match.cpp:252,259-262):
// centre, x+1, x-1, y+1, y-1.  Each coordinate is prepared once (X+-1 share Y and vice versa).
struct Five {
    float c, xp, xm, yp, ym;
};

// The same in two halves, so that a kernel can put other work (or a phase stamp) between the gathers and
// their first use: sample5_issue computes the six coordinates and issues the five tap loads, sample5_finish
// interpolates.
#ifdef PAGK_PK_BILERP
// five tap loads in flight and the six fractions; the (1 - xx, xx) pairs are formed at use
struct FiveTaps {
    Taps q0, q1, q2, q3, q4;
    float fx, fxp, fxm, fy, fyp, fym;
};
template <bool CLAMP>
__device__ __forceinline__ FiveTaps sample5_issue(const DevLevel &L, float X, float Y)
{
    const CoordPk cx = prep_coord_pk<CLAMP>(X, L.fcols, L.fcols_m1);
    const CoordPk cxp = prep_coord_pk<CLAMP>(X + 1.0f, L.fcols, L.fcols_m1);
    const CoordPk cxm = prep_coord_pk<CLAMP>(X - 1.0f, L.fcols, L.fcols_m1);
    const CoordPk cy = prep_coord_pk<CLAMP>(Y, L.frows, L.frows_m1);
    const CoordPk cyp = prep_coord_pk<CLAMP>(Y + 1.0f, L.frows, L.frows_m1);
    const CoordPk cym = prep_coord_pk<CLAMP>(Y - 1.0f, L.frows, L.frows_m1);
    // row offsets: both factors are < 2^24 (checked at upload), full-rate 24-bit multiply-add
    const int rc = __mul24(cy.i, L.cols), rp = __mul24(cyp.i, L.cols), rm = __mul24(cym.i, L.cols);
    const TapSrc ts = tap_src(L);
    FiveTaps t;
    t.q0 = load_taps(ts, rc + cx.i);
    t.q1 = load_taps(ts, rc + cxp.i);
    t.q2 = load_taps(ts, rc + cxm.i);
    t.q3 = load_taps(ts, rp + cx.i);
    t.q4 = load_taps(ts, rm + cx.i);
    t.fx = cx.w.y, t.fxp = cxp.w.y, t.fxm = cxm.w.y, t.fy = cy.w.y, t.fyp = cyp.w.y, t.fym = cym.w.y;
    return t;
}
__device__ __forceinline__ pagk_f32x2 weights_pk(float f)
{
    pagk_f32x2 w;
    w.x = 1.0f - f;
    w.y = f;
    return w;
}
__device__ __forceinline__ Five sample5_finish(const FiveTaps &t)
{
    const pagk_f32x2 cx = weights_pk(t.fx), cxp = weights_pk(t.fxp), cxm = weights_pk(t.fxm);
    const pagk_f32x2 cy = weights_pk(t.fy), cyp = weights_pk(t.fyp), cym = weights_pk(t.fym);
    Five r;
    r.c = bilerp_pk(taps_f32(t.q0), cx, cy);
    r.xp = bilerp_pk(taps_f32(t.q1), cxp, cy);
    r.xm = bilerp_pk(taps_f32(t.q2), cxm, cy);
    r.yp = bilerp_pk(taps_f32(t.q3), cx, cyp);
    r.ym = bilerp_pk(taps_f32(t.q4), cx, cym);
    return r;
}
#else
// five packed quads and the six fractions: 11 registers between issue and use (1 - xx is recomputed at use: the
// same single rounding)
struct FiveTaps {
    uint32_t q0, q1, q2, q3, q4;
    float fx, fxp, fxm, fy, fyp, fym;
};
template <bool CLAMP>
__device__ __forceinline__ FiveTaps sample5_issue(const DevLevel &L, float X, float Y)
{
    const Coord cx = prep_coord<CLAMP>(X, L.fcols, L.fcols_m1);
    const Coord cxp = prep_coord<CLAMP>(X + 1.0f, L.fcols, L.fcols_m1);
    const Coord cxm = prep_coord<CLAMP>(X - 1.0f, L.fcols, L.fcols_m1);
    const Coord cy = prep_coord<CLAMP>(Y, L.frows, L.frows_m1);
    const Coord cyp = prep_coord<CLAMP>(Y + 1.0f, L.frows, L.frows_m1);
    const Coord cym = prep_coord<CLAMP>(Y - 1.0f, L.frows, L.frows_m1);
    const int rc = __mul24(cy.i, L.cols), rp = __mul24(cyp.i, L.cols), rm = __mul24(cym.i, L.cols);
    const uint32_t *q = L.quad;
    FiveTaps t;
    // unsigned 32-bit element offsets: SGPR base + VGPR offset addressing, no 64-bit pointer math
    t.q0 = q[(uint32_t)(rc + cx.i)];
    t.q1 = q[(uint32_t)(rc + cxp.i)];
    t.q2 = q[(uint32_t)(rc + cxm.i)];
    t.q3 = q[(uint32_t)(rp + cx.i)];
    t.q4 = q[(uint32_t)(rm + cx.i)];
    t.fx = cx.f, t.fxp = cxp.f, t.fxm = cxm.f, t.fy = cy.f, t.fyp = cyp.f, t.fym = cym.f;
    return t;
}
__device__ __forceinline__ Five sample5_finish(const FiveTaps &t)
{
    const Coord cx{0, t.fx, 1.0f - t.fx}, cxp{0, t.fxp, 1.0f - t.fxp}, cxm{0, t.fxm, 1.0f - t.fxm};
    const Coord cy{0, t.fy, 1.0f - t.fy}, cyp{0, t.fyp, 1.0f - t.fyp}, cym{0, t.fym, 1.0f - t.fym};
    Five r;
    r.c = bilerp(t.q0, cx, cy);
    r.xp = bilerp(t.q1, cxp, cy);
    r.xm = bilerp(t.q2, cxm, cy);
    r.yp = bilerp(t.q3, cx, cyp);
    r.ym = bilerp(t.q4, cx, cym);
    return r;
}
#endif

template <bool CLAMP>
__device__ __forceinline__ Five sample5(const DevLevel &L, float X, float Y)
{
    return sample5_finish(sample5_issue<CLAMP>(L, X, Y));
}

// ---- software log (src/patch_match.cpp:305 calls std::log(double)) ------------------------------
// glibc's log is neither available on the device nor bit-reproducible across hosts; this
// is a fixed sequence of IEEE double operations (k*ln2 + log1p(f), 7-term series in
// s = f/(2+f)), < 1 ulp.  The oracle carries the same sequence.
__device__ __forceinline__ double soft_log(double x)
{
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                 Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
                 Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                 Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    if (!(x < __builtin_inf())) return x;
    if (x <= 0.0) return x == 0.0 ? -__builtin_inf() : __builtin_nan("");
    uint64_t u = (uint64_t)__double_as_longlong(x);
    int k = (int)(u >> 52) - 1023;
    u = (u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
    double m = __longlong_as_double((long long)u);
    if (m > 1.4142135623730951) {
        m = m * 0.5;
        k += 1;
    }
    double f = m - 1.0;
    double s = f / (2.0 + f);
    double z = s * s;
    double w = z * z;
    double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    double R = t2 + t1;
    double hfsq = 0.5 * f * f;
    double dk = (double)k;
    return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
}

// ---- H.llt().solve(b), update.norm()  (src/patch_match.cpp:319,343) -----------------------------
// Operation order of Eigen 3.3's fixed-size 4x4 path (unblocked LLT that stops at a non-positive pivot and leaves the
// rest of the matrix untouched, fully unrolled triangular solves, SSE2-shaped squaredNorm).  That order is restated
// from memory of a third-party library (oracle/README.md); the places where another Eigen version or build would
// associate differently are switchable at run time -- pagk_params::solver_variant, the same bits as the oracle's
// pagk_oracle_set_alternatives -- so that a host can make this library match ITS Eigen:
enum : uint32_t {
    SV_LOWER_SEQ = 1,   // lower solve, row 3: (c0 + c1) + c2 instead of c0 + (c1 + c2)
    SV_UPPER_TREE = 2,  // upper solve, row 0: c0 + (c1 + c2) instead of (c0 + c1) + c2
    SV_NORM_SEQ = 4,    // squaredNorm: ((x0^2 + x1^2) + x2^2) + x3^2 instead of the SSE2 packet shape
    SV_LLT_RECIP = 8,   // Eigen <= 3.2: A21 *= 1/x instead of A21 /= x
    SV_PIVOT_TREE = 32, // 4th pivot: A33 - (a0^2 + (a1^2 + a2^2)) instead of the sequential sum
};

// update.norm() < 1e-2 (:343) without the square root: sqrt is monotonic and correctly rounded, so
// sqrt(s) < 0.01  <=>  s < T  with T the smallest double whose square root rounds to >= 0.01 (0x3f1a36e2eb1c432c;
// tests/test_capi_cpu.py re-derives it with the host's correctly rounded sqrt, tests/test_parity_gpu.py checks the
// device's sqrt on both sides of it).  NaN compares false in both forms.
constexpr double kNormSqConverged = 0x1.a36e2eb1c432cp-14;

// Division by a denominator that is used more than once.  hipcc expands an f64 `n / d` into v_div_scale x2, v_rcp_f64,
// two Newton steps on the reciprocal, q0 = n * r, the residual fma(-d, q0, n), v_div_fmas and v_div_fixup: correctly
// rounded.  For operands well inside the normal range the scale / fmas / fixup instructions are identities, so the
// same instruction sequence can keep the refined reciprocal r(d) and spend three dependent FMAs per further numerator
// -- bit for bit the compiler's quotient (tests/test_solver_gpu.py::test_division_by_prepared_denominator, 2^22
// operand pairs incl. the range limits).  The solve runs in that form WITHOUT a branch per division: every division
// records whether its operands were in range, and a solve in which one was not (zero, denormal, huge, inf, NaN: a
// degenerate system) is repeated with plain divisions.
struct Den {
    double d, r;
};
// Exponent range of every operand of a solve's divisions, as ONE integer accumulator beside the dependent FMA chain
// (a lone wave pays ~6 cycles per instruction of any kind, so the check is counted in instructions): per operand
// t = (high word without sign) - (biased exponent 1023 - 400 in place), unsigned -- an exponent below the range wraps
// to a huge t, one above it is large -- folded with v_max3_u32.  in_range(): every operand finite and normal with
// 2^-400 <= |v| < 2^401.
struct OperandRange {
    static constexpr uint32_t kLow = (1023u - 400u) << 20, kSpan = (801u << 20) - 1u;
    uint32_t t = 0u;
    __device__ __forceinline__ void add(double v)
    {
        const uint32_t u = ((uint32_t)__double2hiint(v) & 0x7fffffffu) - kLow;
        t = u > t ? u : t;
    }
    // a denominator: a negative one is out of range by definition (pivots and sums of squares are not negative)
    __device__ __forceinline__ void add_positive(double v)
    {
        const uint32_t u = (uint32_t)__double2hiint(v) - kLow;
        t = u > t ? u : t;
    }
    // a numerator that may be exactly +0.0 (see div_by): counted as in range
    __device__ __forceinline__ void add_or_zero(double v)
    {
        uint32_t u = ((uint32_t)__double2hiint(v) & 0x7fffffffu) - kLow;
        u = ((uint32_t)__double2hiint(v) | (uint32_t)__double2loint(v)) == 0u ? 0u : u;
        t = u > t ? u : t;
    }
    __device__ __forceinline__ bool in_range() const { return t <= kSpan; }
};
template <bool FAST>
__device__ __forceinline__ Den den_prepare(double d, OperandRange &rg)
{
    Den D;
    D.d = d;
    D.r = 0.0;
    if constexpr (FAST) {
        rg.add_positive(d);
        double r = __builtin_amdgcn_rcp(d);
        double e = __builtin_fma(-d, r, 1.0);
        r = __builtin_fma(r, e, r);
        e = __builtin_fma(-d, r, 1.0);
        D.r = __builtin_fma(r, e, r);
    }
    return D;
}
// n / D.d; FAST: through the prepared reciprocal, `rg` collects the operands' exponents.  ZERO_OK: a numerator that is
// exactly +0.0 counts as in range -- the three FMAs then return +-0 with the quotient's sign (-0.0 would not:
// fma(+0, r, -0) = +0).  Used where a zero is an ordinary outcome: H is structurally singular, so the last row's
// right-hand side b3 - sum is a difference of nearly equal numbers and cancels to +0 in a fair share of solves.
template <bool FAST, bool ZERO_OK = false>
__device__ __forceinline__ double div_by(double n, const Den &D, OperandRange &rg)
{
    if constexpr (FAST) {
        if constexpr (ZERO_OK)
            rg.add_or_zero(n);
        else
            rg.add(n);
        const double q = n * D.r;
        return __builtin_fma(__builtin_fma(-D.d, q, n), D.r, q);
    } else {
        return n / D.d;
    }
}
// pivot -> diagonal entry: sqrt(x) for an accepted pivot, the untouched H(k,k) otherwise.  hipcc expands an f64 sqrt
// into a range test + v_ldexp (scaling for x < 2^-767), v_rsq_f64, two coupled Newton steps on (g, h) ~ (sqrt x,
// 1 / (2 sqrt x)), two residual corrections, the scaling back and a v_cmp_class fix-up for 0 / inf: 22 instructions, of
// which the scalings and the fix-up are identities for a pivot inside the range the divisions require anyway.  FAST: the
// ten instructions that remain, in the compiler's order, hence its bits (tests/test_solver_gpu.py); the pivot joins the
// operand range (a rejected pivot does not: its root is never formed, and H is singular -- the 4th pivot fails in most solves).
template <bool FAST>
__device__ __forceinline__ double pivot_root(bool ok, double x, double hkk, OperandRange &rg)
{
    if constexpr (FAST) {
        const double xs = ok ? x : 1.0;
        rg.add(xs);
        const double y = __builtin_amdgcn_rsq(xs);
        double g = xs * y, h = y * 0.5;
        const double r = __builtin_fma(-h, g, 0.5);
        g = __builtin_fma(g, r, g);
        h = __builtin_fma(h, r, h);
        double d = __builtin_fma(-g, g, xs);
        g = __builtin_fma(d, h, g);
        d = __builtin_fma(-g, g, xs);
        g = __builtin_fma(d, h, g);
        return ok ? g : hkk;
    } else {
        return ok ? sqrt(x) : hkk;
    }
}
// The first three pivots in the FAST form: taken as accepted.  They fail for flat or saturated patches only (the 4th
// fails routinely: H is singular), so instead of carrying `ok ? ... : ...` selects through every column the pivot joins
// the operand range as a POSITIVE value -- zero, negative or NaN sends the solve to the plain form, which implements
// the reference's early return.
template <bool FAST>
__device__ __forceinline__ double pivot_root_early(bool ok, double x, double hkk, OperandRange &rg)
{
    if constexpr (FAST) {
        rg.add_positive(x);
        const double y = __builtin_amdgcn_rsq(x);
        double g = x * y, h = y * 0.5;
        const double r = __builtin_fma(-h, g, 0.5);
        g = __builtin_fma(g, r, g);
        h = __builtin_fma(h, r, h);
        double d = __builtin_fma(-g, g, x);
        g = __builtin_fma(d, h, g);
        d = __builtin_fma(-g, g, x);
        return __builtin_fma(d, h, g);
    } else {
        return ok ? sqrt(x) : hkk;
    }
}
// the lean square root by itself, with the plain one where the operand is out of its range (diagnostics)
__device__ __forceinline__ double sqrt_one(double x)
{
    OperandRange rg;
    const double g = pivot_root<true>(true, x, x, rg);
    return (rg.in_range() && x > 0.0) ? g : sqrt(x);
}
// the checked single division (diagnostics: pagk_selftest_divide)
__device__ __forceinline__ double div_one(double n, double d)
{
    OperandRange rg;
    const double q = div_by<true>(n, den_prepare<true>(d, rg), rg);
    return rg.in_range() ? q : n / d;
}

// The back substitution L^T x = y and the squared norm, shared by the two forms below.  D0..D3 hold the diagonal.
template <bool FAST>
__device__ __forceinline__ double llt4_upper_nsq(double r0, double r1, double r2, double r3, double L10, double L20,
                                                 double L21, double L30, double L31, double L32, const Den &D0,
                                                 const Den &D1, const Den &D2, const Den &D3, uint32_t sv,
                                                 double (&x)[4], OperandRange &rg)
{
    r3 = div_by<FAST, true>(r3, D3, rg);
    r2 -= L32 * r3;
    r2 = div_by<FAST>(r2, D2, rg);
    r1 -= L21 * r2 + L31 * r3;
    r1 = div_by<FAST>(r1, D1, rg);
    const double c0 = L10 * r1, c1 = L20 * r2, c2 = L30 * r3;
    r0 -= (sv & SV_UPPER_TREE) ? c0 + (c1 + c2) : (c0 + c1) + c2;
    r0 = div_by<FAST>(r0, D0, rg);
    x[0] = r0;
    x[1] = r1;
    x[2] = r2;
    x[3] = r3;
    const double s0 = r0 * r0, s1 = r1 * r1, s2 = r2 * r2, s3 = r3 * r3;
    return (sv & SV_NORM_SEQ) ? ((s0 + s1) + s2) + s3 : (s0 + s2) + (s1 + s3);
}

// One lane solves one system.  M: lower triangle is read.  Returns update.squaredNorm() (compare with
// kNormSqConverged).  `ok_k` = "column k was factorised": the reference returns at the first pivot with x <= 0 (a NaN
// pivot does NOT stop it: `x <= 0` is false), leaving that column and everything right of it untouched; values
// computed speculatively past a failed pivot are dropped by the selects.
template <bool FAST>
__device__ __forceinline__ double llt4_solve_nsq_form(const double (&M)[4][4], const double (&b)[4], double (&x)[4],
                                                      uint32_t sv, OperandRange &rg)
{
    const double H00 = M[0][0], H10 = M[1][0], H11 = M[1][1], H20 = M[2][0], H21 = M[2][1], H22 = M[2][2];
    const double H30 = M[3][0], H31 = M[3][1], H32 = M[3][2], H33 = M[3][3];
    const bool recip = (sv & SV_LLT_RECIP) != 0;
    // a column's scaling: A21 /= x (Eigen 3.3) or A21 *= 1/x (<= 3.2)
    auto scale = [&](double n, const Den &D, double rx) { return recip ? n * rx : div_by<FAST>(n, D, rg); };
    // column 0
    const bool ok0 = FAST || (!(H00 <= 0.0));
    const double d0 = pivot_root_early<FAST>(ok0, H00, H00, rg);
    const Den D0 = den_prepare<FAST>(d0, rg);
    double rx = recip ? div_by<FAST>(1.0, D0, rg) : 0.0;
    const double q10 = scale(H10, D0, rx), q20 = scale(H20, D0, rx), q30 = scale(H30, D0, rx);
    const double r0 = div_by<FAST>(b[0], D0, rg);
    const double L10 = ok0 ? q10 : H10, L20 = ok0 ? q20 : H20, L30 = ok0 ? q30 : H30;
    // column 1
    const double x1 = H11 - L10 * L10;
    const bool ok1 = FAST || (ok0 && !(x1 <= 0.0));
    const double d1 = pivot_root_early<FAST>(ok1, x1, H11, rg);
    const Den D1 = den_prepare<FAST>(d1, rg);
    rx = recip ? div_by<FAST>(1.0, D1, rg) : 0.0;
    const double q21 = scale(H21 - L20 * L10, D1, rx), q31 = scale(H31 - L30 * L10, D1, rx);
    const double r1 = div_by<FAST>(b[1] - L10 * r0, D1, rg);
    const double L21 = ok1 ? q21 : H21, L31 = ok1 ? q31 : H31;
    // column 2
    double s = L20 * L20;
    s += L21 * L21;
    const double x2 = H22 - s;
    const bool ok2 = FAST || (ok1 && !(x2 <= 0.0));
    const double d2 = pivot_root_early<FAST>(ok2, x2, H22, rg);
    const Den D2 = den_prepare<FAST>(d2, rg);
    s = L30 * L20;
    s += L31 * L21;
    rx = recip ? div_by<FAST>(1.0, D2, rg) : 0.0;
    const double q32 = scale(H32 - s, D2, rx);
    const double r2 = div_by<FAST>(b[2] - (L20 * r0 + L21 * r1), D2, rg);
    const double L32 = ok2 ? q32 : H32;
    // column 3
    const double a0 = L30 * L30, a1 = L31 * L31, a2 = L32 * L32;
    const double x3 = H33 - ((sv & SV_PIVOT_TREE) ? a0 + (a1 + a2) : (a0 + a1) + a2);
    const bool ok3 = ok2 && !(x3 <= 0.0);
    const double d3 = pivot_root<FAST>(ok3, x3, H33, rg);
    const Den D3 = den_prepare<FAST>(d3, rg);
    // L y = b, last row (the rows above are interleaved with their columns): c0 + (c1 + c2)
    const double c0 = L30 * r0, c1 = L31 * r1, c2 = L32 * r2;
    const double r3 = div_by<FAST, true>(b[3] - ((sv & SV_LOWER_SEQ) ? (c0 + c1) + c2 : c0 + (c1 + c2)), D3, rg);
    return llt4_upper_nsq<FAST>(r0, r1, r2, r3, L10, L20, L21, L30, L31, L32, D0, D1, D2, D3, sv, x, rg);
}

__device__ __forceinline__ double llt4_solve_nsq(const double (&M)[4][4], const double (&b)[4], double (&x)[4],
                                                 uint32_t sv)
{
    OperandRange rg;
    double nsq = llt4_solve_nsq_form<true>(M, b, x, sv, rg);
    if (!rg.in_range()) nsq = llt4_solve_nsq_form<false>(M, b, x, sv, rg);  // a degenerate system: plain divisions
    return nsq;
}

// The same with the square root taken (k_track_thread, the reference-shaped cross-check, tests `norm < 1e-2` as the
// reference writes it: an on-device check of the threshold form the other kernels use) and plain divisions only.
__device__ __forceinline__ double llt4_solve_norm(const double (&M)[4][4], const double (&b)[4], double (&x)[4],
                                                  uint32_t sv)
{
    OperandRange rg;
    return sqrt(llt4_solve_nsq_form<false>(M, b, x, sv, rg));
}

// The same solve spread over four lanes (l = 0..3 of one wave, all holding the same H and b): per
// Cholesky column the divides of the rows below the pivot and the forward-substitution divide of that
// column are ONE divide executed by four lanes instead of up to four sequences in one lane;
// results travel by DPP row broadcasts.  Operation for operation the arithmetic of llt4_solve_nsq_form above.
__device__ __forceinline__ double lane_bcast(double v, int src)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, src);
    hi = __builtin_amdgcn_readlane(hi, src);
    return __hiloint2double(hi, lo);
}
// The same for a source lane K < 16 of the caller's own 16-lane row: ONE v_mov_b64_dpp row_newbcast:K instead of two
// v_readlane and two v_mov back from the scalar registers (round 4: the solve's nine broadcasts cost it 36 of its ~240
// instructions, and a lone wave pays ~7 cycles per instruction of any kind).  The four solving lanes are lanes 0..3 of
// their wave in every kernel, i.e. of row 0.
template <int K>
__device__ __forceinline__ double row_bcast(double v)
{
    return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + K, 0xf, 0xf, false);  // row_newbcast:K
}

#ifdef PAGK_COUNT_REDO
static __device__ uint32_t g_redo_lo, g_redo_hi;   // (diagnostic build: racy by design, last writer wins)
#endif
// ROWB: the lane-to-lane results travel by row_bcast (one DPP instruction, the value stays in a vector register) instead
// of lane_bcast (two v_readlane, the value in scalar registers): fewer instructions, ~20 more live VGPRs -- for kernels
// that have them (the pipelined 4-wave kernel; the others sit at their register limit and would spill).
template <int K, bool ROWB>
__device__ __forceinline__ double solve_bcast(double v)
{
    if constexpr (ROWB)
        return row_bcast<K>(v);
    else
        return lane_bcast(v, K);
}
template <bool FAST, bool ROWB = false>
__device__ __forceinline__ double llt4_solve_nsq_lanes_form(const double (&M)[4][4], const double (&b)[4], int l,
                                                            double (&x)[4], uint32_t sv, OperandRange &rg)
{
    const double H00 = M[0][0], H10 = M[1][0], H11 = M[1][1], H20 = M[2][0], H21 = M[2][1], H22 = M[2][2];
    const double H30 = M[3][0], H31 = M[3][1], H32 = M[3][2], H33 = M[3][3];
    const bool recip = (sv & SV_LLT_RECIP) != 0;
    // a column's scaling: n / d (Eigen 3.3) or n * (1 / d) (<= 3.2); `fwd`: this lane's quotient is the forward
    // substitution's rhs[k] /= L(k,k), a true division in every version
    auto column = [&](double n, const Den &D, bool fwd) {
        const double q = div_by<FAST>(n, D, rg);
        if (!recip) return q;
        const double rx = div_by<FAST>(1.0, D, rg);
        return fwd ? q : n * rx;
    };
    // column 0: lane 0 -> r0 = b0 / d0, lane i -> L(i,0) = H(i,0) / d0
    const bool ok0 = FAST || (!(H00 <= 0.0));
    const double d0 = pivot_root_early<FAST>(ok0, H00, H00, rg);
    const Den D0 = den_prepare<FAST>(d0, rg);
    const double n0 = l == 0 ? b[0] : (l == 1 ? H10 : (l == 2 ? H20 : H30));
    const double q0 = column(n0, D0, l == 0);
    const double own0 = (l == 0 || ok0) ? q0 : n0;
    double r0 = solve_bcast<0, ROWB>(own0);
    const double L10 = solve_bcast<1, ROWB>(own0), L20 = solve_bcast<2, ROWB>(own0), L30 = solve_bcast<3, ROWB>(own0);
    // column 1: lane 1 -> r1, lanes 2, 3 -> L(i,1)
    const double x1 = H11 - L10 * L10;
    const bool ok1 = FAST || (ok0 && !(x1 <= 0.0));
    const double d1 = pivot_root_early<FAST>(ok1, x1, H11, rg);
    const Den D1 = den_prepare<FAST>(d1, rg);
    const double h1 = l == 2 ? H21 : H31;
    const double n1 = l == 1 ? b[1] - L10 * r0 : h1 - (l == 2 ? L20 : L30) * L10;
    const double q1 = column(n1, D1, l == 1);
    const double own1 = (l == 1 || ok1) ? q1 : h1;
    double r1 = solve_bcast<1, ROWB>(own1);
    const double L21 = solve_bcast<2, ROWB>(own1), L31 = solve_bcast<3, ROWB>(own1);
    // column 2: lane 2 -> r2, lane 3 -> L(3,2)
    double s = L20 * L20;
    s += L21 * L21;
    const double x2 = H22 - s;
    const bool ok2 = FAST || (ok1 && !(x2 <= 0.0));
    const double d2 = pivot_root_early<FAST>(ok2, x2, H22, rg);
    const Den D2 = den_prepare<FAST>(d2, rg);
    s = L30 * L20;
    s += L31 * L21;
    const double n2 = l == 2 ? b[2] - (L20 * r0 + L21 * r1) : H32 - s;
    const double q2 = column(n2, D2, l == 2);
    const double own2 = (l == 2 || ok2) ? q2 : H32;
    double r2 = solve_bcast<2, ROWB>(own2);
    const double L32 = solve_bcast<3, ROWB>(own2);
    // column 3
    const double a0 = L30 * L30, a1 = L31 * L31, a2 = L32 * L32;
    const double x3 = H33 - ((sv & SV_PIVOT_TREE) ? a0 + (a1 + a2) : (a0 + a1) + a2);
    const bool ok3 = ok2 && !(x3 <= 0.0);
    const double d3 = pivot_root<FAST>(ok3, x3, H33, rg);
    const Den D3 = den_prepare<FAST>(d3, rg);
    const double c0 = L30 * r0, c1 = L31 * r1, c2 = L32 * r2;
    const double r3 = div_by<FAST, true>(b[3] - ((sv & SV_LOWER_SEQ) ? (c0 + c1) + c2 : c0 + (c1 + c2)), D3, rg);
    // L^T x = y  (sequential by nature; every lane computes it)
    return llt4_upper_nsq<FAST>(r0, r1, r2, r3, L10, L20, L21, L30, L31, L32, D0, D1, D2, D3, sv, x, rg);
}

// (the calling lanes -- four, or a whole wave whose lanes 0..3 matter -- take the decision together: the form
// broadcasts between them)
template <bool ROWB = false>
__device__ __forceinline__ double llt4_solve_nsq_lanes(const double (&M)[4][4], const double (&b)[4], int l,
                                                       double (&x)[4], uint32_t sv)
{
    OperandRange rg;
    double nsq = llt4_solve_nsq_lanes_form<true, ROWB>(M, b, l, x, sv, rg);
#ifndef PAGK_EXPERIMENT_NO_REDO
    if (__builtin_amdgcn_ballot_w64(!rg.in_range() && l < 4) != 0) nsq = llt4_solve_nsq_lanes_form<false, ROWB>(M, b, l, x, sv, rg);
#endif
#ifdef PAGK_COUNT_REDO
    g_redo_lo = rg.t;
#endif
    return nsq;
}

// H22 = the ordered sum of P copies of q = c * c (src/patch_match.cpp:296 with J[2] = de_dg = c constant over the patch,
// :263): s_0 = 0, s_k = RN(s_{k-1} + q).  The value depends on the level only, and it needs no 441-step chain: q is the
// exact product of two floats (48 significant bits), so
//   - the first 32 partial sums are exact (k q < 2^6 q needs at most 53 bits): s_32 = 32 q;
//   - while s stays inside one binade [2^F, 2^(F+1)) every step adds the same increment I_F = q rounded to that binade's
//     spacing u -- s is a multiple of u, so RN(s + q) = s + RN_u(q); a tie (q mod u == u/2) goes to the even multiple,
//     which after ONE step inside the binade is again "s + the same I_F" (s / u is even from then on) -- and
//     I_F = (q + 1.5 * 2^F) - 1.5 * 2^F is that rounding done by the adder itself;
//   - so per binade: n steps at once as fma(n, I_F, s) (exact: the result is a multiple of u below 2^(F+1)), with n
//     chosen to stop at least one increment short of the binade's top, then four plain steps that carry s across the
//     top and once more inside the next binade (the parity step).  32 q lies in binade E + 5, P q < 2^(E+10): five
//     binades, and the step count is made to come out at exactly P by capping n.
// Bit-identical to the loop for every float c (tests/test_repeat_sum.py: all 2^23 mantissas x P = 289 / 361 / 441 on the
// host, with the quotient estimate perturbed both ways; pagk_selftest_repeat_sum on the device).  ~100 instructions
// instead of a 441-step dependent chain.  P in (57, 480].
__device__ __forceinline__ double repeat_sum_f64(double q, int P)
{
    if (!(q > 0.0)) return (double)P * q;  // c == 0: every partial sum is 0
    double x = 32.0 * q;
    int k = 32;
    double top = __hiloint2double((__double2hiint(q) & 0x7ff00000) + (6 << 20), 0);  // 2^(E+6)
    double M = 0.75 * top;                                                             // 1.5 * 2^(E+5)
#pragma unroll
    for (int b = 0; b < 5; b++) {
        const double I = (q + M) - M;
        const float est = (float)(top - x) * __builtin_amdgcn_rcpf((float)I);
        int n = (int)est - 1;
        const int lim = (P - k) - 4 * (5 - b);
        n = n > lim ? lim : n;
        n = n < 0 ? 0 : n;
        x = __builtin_fma((double)n, I, x);
        k += n + 4;
        x = x + q;
        x = x + q;
        x = x + q;
        x = x + q;
        M *= 2.0;
        top *= 2.0;
    }
    return x;
}

// Gyro regularisation penalty, src/patch_match.cpp:302-314.  Adds to H (lower triangle
// only: LLT reads nothing else), b and cost.
// add_penalty_hb: the H and b terms; returns e_pen * e_pen, which the caller adds to the cost (:313) -- the pipelined
// 4-wave kernel solves before its cost chain has finished.
__device__ __forceinline__ double add_penalty_hb(const TrackArgs &a, float dx, float dy, double (&H)[4][4],
                                                 double (&b)[4])
{
    double d = (double)sqrtf(dx * dx + dy * dy);                                       // :304
    double e_pen = (double)a.lam_invlog * soft_log((double)a.alpha * d + 1);           // :305
    double jx = (double)a.lam_invlog_alpha / ((double)a.alpha * d + 1) * ((double)dx / d); // :307
    double jy = (double)a.lam_invlog_alpha / ((double)a.alpha * d + 1) * ((double)dy / d); // :308
    H[0][0] += jx * jx;                                                                // :311
    H[1][0] += jy * jx;
    H[1][1] += jy * jy;
    // rows/cols 2,3 of JPenalty are 0: H += 0*x adds +0.0 and changes nothing unless jx/jy
    // are NaN/inf (d == 0 gives 0/0): then 0*NaN = NaN lands in those entries too.
    H[2][0] += 0.0 * jx;
    H[2][1] += 0.0 * jy;
    H[2][2] += 0.0 * 0.0;
    H[3][0] += 0.0 * jx;
    H[3][1] += 0.0 * jy;
    H[3][2] += 0.0 * 0.0;
    H[3][3] += 0.0 * 0.0;
    b[0] += jx * e_pen;                                                                // :312 (PLUS)
    b[1] += jy * e_pen;
    b[2] += 0.0 * e_pen;
    b[3] += 0.0 * e_pen;
    return e_pen * e_pen;
}
__device__ __forceinline__ void add_penalty(const TrackArgs &a, float dx, float dy, double (&H)[4][4],
                                            double (&b)[4], float &cost)
{
    const double epsq = add_penalty_hb(a, dx, dy, H, b);
    cost = (float)((double)cost + epsq);                                               // :313
}

// DistortVecPoints, src/utils.cpp:49-76, one point.
__device__ __forceinline__ void distort_point(const TrackArgs &a, float ux, float uy, float &ox, float &oy)
{
    if (!a.distort_on) {  // src/patch_match.cpp:410-411
        ox = ux;
        oy = uy;
        return;
    }
    float x = (ux - a.cx) * a.fx_inv;
    float y = (uy - a.cy) * a.fy_inv;
    float r2 = x * x + y * y;
    float r4 = r2 * r2;
    float r6 = r4 * r2;
    float xd = x * (1 + a.k1 * r2 + a.k2 * r4 + a.k3 * r6) + 2 * a.p1 * x * y + a.p2 * (r2 + 2 * x * x);
    float yd = y * (1 + a.k1 * r2 + a.k2 * r4 + a.k3 * r6) + a.p1 * (r2 + 2 * y * y) + 2 * a.p2 * x * y;
    ox = a.fx * xd + a.cx;
    oy = a.fy * yd + a.cy;
}

}  // namespace pagk
