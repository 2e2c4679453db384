// pagk_wave_kernel.h -- k_track_wave: the Gauss-Newton loop with ONE WAVEFRONT PER FEATURE.
//
// The throughput-shaped form of the path, for launches with (many) more features than the chip holds
// workgroups.  In k_track_block most of a workgroup's waves wait at barriers while one of them walks the
// ordered chain or solves; those waiting waves still occupy wave slots, so the CU runs at ~30 % issue
// utilisation.  Here a feature is a single wave: nothing ever waits for another wave, 16 features are
// resident per CU (4 waves/SIMD at <= 128 VGPRs, 9 KB of LDS each) and the SIMD interleaves them.
//
// Per iteration, all in the one wave:
//   1. sampling   lane l owns patch pixels l, l+64, ... (row-major = the reference's order); per pixel the
//                 five img2 samples, e, Ix, Iy in f32 (:252-262); streams X = Ix, Y = Iy, NE = -e go to LDS
//                 as f32 (they ARE f32 values; widening happens at the matrix pipe's doorstep);
//   2. H, b       a chain of v_mfma_f64_4x4x4f64, four pixels per instruction: block 0 accumulates J J^T,
//                 block 1 accumulates J * (-e) = -J e (:293-296).  The instruction is a sequential FMA chain
//                 over k in ascending order (tools/microbench7.hip: 128000/128000 outputs bit-identical), the
//                 products are exact, so this is the reference's summation;
//   3. cost       sum of e*e in f32, in order: DPP row chain over the squares of the NE stream (:294);
//   4. solve + update in lane 0 / every lane, as in k_track_block.
// Bit-identical to the oracle and to the other variants (tests/test_parity_gpu.py).
#pragma once
#include <type_traits>
#include "pagk_chain_asm.h"
#include "pagk_device.h"

#ifndef PAGK_WAVE_OCC
#define PAGK_WAVE_OCC 4  // waves per SIMD the register allocation targets
#endif

namespace pagk {

__device__ __forceinline__ void write_outputs(const TrackArgs &a, int i, float p2x, float p2y, int succ,
                                              float lastCost, int level0_ran, float ncc, int iters);
__device__ __forceinline__ uint32_t lds_off(const void *p);
__host__ __device__ inline int track_block_pp(int half);

// LDS: max(3 f32 streams of PP + 8 plus 16 constants, 5 f32 arrays of PP for the NCC epilogue),
// then double acc[24], upd[5], float cost[2].
__host__ __device__ inline size_t track_wave_lds_bytes(int half)
{
    size_t PP = (size_t)track_block_pp(half);
    size_t region = 3 * (PP + 8) + 16;
    if (5 * PP > region) region = 5 * PP;
    return region * 4 + 8 + 24 * 8 + 5 * 8 + 2 * 4 + 8;
}

template <int NR, int TAIL, bool LEAN = false>   // LEAN: no penalty, solver_variant 0 (see track_block_body)
__global__ void __launch_bounds__(64, PAGK_WAVE_OCC) k_track_wave(TrackArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int i = blockIdx.x;
    const int lane = threadIdx.x;
    const int h = a.half, Wd = 2 * h + 1, P = Wd * Wd, PP = track_block_pp(h), PS = PP + 8;
    const int nfull = P / 32;

    float *fs = reinterpret_cast<float *>(lds_raw);  // X | Y | NE, stride PS
    float *cst = fs + 3 * PS;                        // 4 x c, 4 x 1.0f, 4 x 0.0f
    size_t region = 3 * (size_t)PS + 16;
    if ((size_t)5 * PP > region) region = (size_t)5 * PP;
    double *acc = reinterpret_cast<double *>(lds_raw + ((region * 4 + 7) / 8) * 8);
    double *sh_upd = acc + 24;
    float *sh_cost = reinterpret_cast<float *>(sh_upd + 5);

    const float *init = a.has_gyro ? a.pt_init : a.pt_ref;  // :85-89
    float p2x = init[2 * i], p2y = init[2 * i + 1];
    if (!a.status_in[i]) {  // :173
        if (lane == 0) write_outputs(a, i, p2x, p2y, 0, 0.0f, 0, 0.0f, 0);
        return;
    }
    float A00 = 1, A01 = 0, A10 = 0, A11 = 1;
    if (a.use_affine) {
        A00 = a.affine[4 * i], A01 = a.affine[4 * i + 1], A10 = a.affine[4 * i + 2], A11 = a.affine[4 * i + 3];
    }
    const float refx = a.pt_ref[2 * i], refy = a.pt_ref[2 * i + 1];

    // lane -> pixels p = lane + 64 r, row-major (y outer, :233-234); lanes past the patch shadow the last pixel
    float wx[NR], wy[NR];
    int pxy[NR];  // x and y packed (two int16)
#pragma unroll
    for (int r = 0; r < NR; r++) {
        int p = lane + 64 * r;
        p = p < P ? p : P - 1;
        int yy = p / Wd, xx = p - yy * Wd;
        int x = xx - h, y = yy - h;
        pxy[r] = (x & 0xffff) | (y << 16);
        if (a.use_affine) {  // :203-204
            wx[r] = A00 * (float)x + A01 * (float)y;
            wy[r] = A10 * (float)x + A11 * (float)y;
        } else {
            wx[r] = (float)x;
            wy[r] = (float)y;
        }
    }
    const float fh = (float)h;
    const float ext_x = fabsf(A00) * fh + fabsf(A01) * fh + 2.0f;
    const float ext_y = fabsf(A10) * fh + fabsf(A11) * fh + 2.0f;

    // MFMA operand roles (layout measured: A(q,i,k) lane 16k+4q+i, B(q,k,j) lane 16k+4q+j, D(q,i,j) lane
    // 16i+4q+j).  q = 0: A = B = J[i];  q = 1: A = J[i], B = (i == 0 ? -e : 0);  q >= 2: zeros.
    const int mk = lane >> 4, mq = (lane >> 2) & 3, mi = lane & 3;
    const float *j_src = mi < 2 ? fs + mi * PS + mk : cst + (mi == 2 ? 0 : 4) + mk;
    const int j_step = mi < 2 ? 4 : 0;
    const float *a_src = mq < 2 ? j_src : cst + 8 + mk;
    const int a_step = mq < 2 ? j_step : 0;
    const float *b_src = mq == 0 ? j_src : ((mq == 1 && mi == 0) ? fs + 2 * PS + mk : cst + 8 + mk);
    const int b_step = mq == 0 ? j_step : ((mq == 1 && mi == 0) ? 4 : 0);
    const uint32_t cost_addr = lds_off(fs + 2 * PS) + 8u * (lane & 15);
    if (lane < 4) {
        cst[4 + lane] = 1.0f;
        cst[8 + lane] = 0.0f;
    }

    int succ = 1, iters = 0;
    float lastCost = 0.0f;

    for (int level = a.n_levels - 1; level >= 0; level--) {
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
        lastCost = 0.0f;
        succ = 1;

        const float cneg = -sample<true>(L1, ptx, pty);  // :263
        float s1[NR];
#pragma unroll
        for (int r = 0; r < NR; r++) {
            const float x = (float)(short)(pxy[r] & 0xffff), y = (float)(pxy[r] >> 16);
            s1[r] = sample<true>(L1, ptx + x, pty + y);  // :253, iteration-invariant
        }
        __syncthreads();  // the previous level's readers of cst are done
        if (lane < 4) cst[lane] = cneg;

        for (int iter = 0; iter < a.iterations; iter++) {  // :215
            iters++;
            // ---- 1. sampling ------------------------------------------------------------------
            const float bx = ptx + dx, by = pty + dy;
            const float gain = 1.0f + dg;
            const bool interior = (bx - ext_x >= 0.0f) && (bx + ext_x < L2.fcols_m1) &&
                                  (by - ext_y >= 0.0f) && (by + ext_y < L2.frows_m1);
            // two rounds' gathers in flight at a time; the clamp-free / clamped choice is made once for the phase, not
            // per round (see k_track_block: chosen per round, the compiler serialises the rounds)
            auto sampling = [&](auto clamp_tag) {
                constexpr bool CLAMP = decltype(clamp_tag)::value;
#pragma unroll
                for (int r0 = 0; r0 < NR; r0 += 2) {
                    FiveTaps taps[2];
#pragma unroll
                    for (int u = 0; u < 2; u++) {
                        const int r = r0 + u < NR ? r0 + u : NR - 1;
                        float X = bx + wx[r], Y = by + wy[r];
                        taps[u] = sample5_issue<CLAMP>(L2, X, Y);
                    }
#pragma unroll
                    for (int u = 0; u < 2; u++) {
                        const int r = r0 + u;
                        if (r < NR) {
                            const Five s = sample5_finish(taps[u]);
                            const int p = lane + 64 * r;
                            if (p < P) {
                                float e = s.c + db - gain * s1[r];  // :252-253
                                float Ix = 0.5f * (s.xp - s.xm);    // :259-260
                                float Iy = 0.5f * (s.yp - s.ym);    // :261-262
                                fs[0 * PS + p] = Ix;
                                fs[1 * PS + p] = Iy;
                                fs[2 * PS + p] = -e;
                            }
                        }
                    }
                }
            };
            if (interior)
                sampling(std::false_type{});
            else
                sampling(std::true_type{});
            __syncthreads();
            __builtin_amdgcn_s_setprio(3);  // chain + solve first (see k_track_block); -2 % at 20000 features
            // ---- 2. H and b: ordered MFMA chain -------------------------------------------------
            {
                constexpr int kU = 4;
                double d = 0.0;
                const int M = (P + 3) >> 2, Mfull = P >> 2;  // PS >= 4 * (M + kU): prefetch stays inside the streams
                float an[kU], bn[kU];
                const float *pa = a_src, *pb = b_src;
#pragma unroll
                for (int u = 0; u < kU; u++) {
                    an[u] = *pa;
                    bn[u] = *pb;
                    pa += a_step;
                    pb += b_step;
                }
                int m = 0;
                for (; m + kU <= Mfull; m += kU) {
                    double a0[kU], b0[kU];
#pragma unroll
                    for (int u = 0; u < kU; u++) {
                        a0[u] = (double)an[u];
                        b0[u] = (double)bn[u];
                    }
#pragma unroll
                    for (int u = 0; u < kU; u++) {
                        an[u] = *pa;
                        bn[u] = *pb;
                        pa += a_step;
                        pb += b_step;
                    }
#pragma unroll
                    for (int u = 0; u < kU; u++) d = __builtin_amdgcn_mfma_f64_4x4x4f64(a0[u], b0[u], d, 0, 0, 0);
                }
                const int rem = Mfull - m;
#pragma unroll
                for (int u = 0; u < kU; u++)
                    if (u < rem) d = __builtin_amdgcn_mfma_f64_4x4x4f64((double)an[u], (double)bn[u], d, 0, 0, 0);
                if (M > Mfull) {
                    // pixels past the patch contribute fma(-0.0, 1.0, d) = d exactly
                    const bool pad = 4 * Mfull + mk >= P;
                    const double av = (double)a_src[Mfull * a_step], bv = (double)b_src[Mfull * b_step];
                    d = __builtin_amdgcn_mfma_f64_4x4x4f64(pad ? -0.0 : av, pad ? 1.0 : bv, d, 0, 0, 0);
                }
                const int di = lane >> 4, dq = (lane >> 2) & 3, dj = lane & 3;
                if (dq == 0) acc[di * 4 + dj] = d;         // H
                if (dq == 1 && dj == 0) acc[16 + di] = d;  // b
            }
            // ---- 3. cost (:283, :294): ordered f32 sum of the squares of the -e stream ------------
            {
                float c = chain_rows_sq_f32<TAIL>(cost_addr, 128u, nfull);
                if (lane == 0) sh_cost[0] = c;
            }
            __syncthreads();
            // ---- 4. solve (:302-319) ------------------------------------------------------------
            if (lane < 4) {
                double H[4][4], b[4], upd[4];
                for (int r = 0; r < 4; r++)
                    for (int c = 0; c <= r; c++) H[r][c] = acc[r * 4 + c];
                for (int r = 0; r < 4; r++) b[r] = acc[16 + r];
                float cost = sh_cost[0];
                if constexpr (!LEAN) {
                    if (a.penalty) add_penalty(a, dx, dy, H, b, cost);
                }
                // four lanes share the divides of each Cholesky column (pagk_device.h); all end with the result
                double unorm = llt4_solve_nsq_lanes(H, b, lane, upd, LEAN ? 0u : a.solver);  // update.squaredNorm()
                if (lane == 0) {
                    sh_upd[0] = upd[0];
                    sh_upd[1] = upd[1];
                    sh_upd[2] = upd[2];
                    sh_upd[3] = upd[3];
                    sh_upd[4] = unorm;
                    sh_cost[1] = cost;
                }
            }
            __syncthreads();
            __builtin_amdgcn_s_setprio(0);
            // ---- 5. update + termination (:322-344) -----------------------------------------------
            const double u0 = sh_upd[0], u1 = sh_upd[1], u2 = sh_upd[2], u3 = sh_upd[3], unorm = sh_upd[4];
            const float cost = sh_cost[1];
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
            lastCost = cost;
            succ = 1;
            if (unorm < kNormSqConverged) break;  // :343  update.norm() < 1e-2
        }
        p2x = ptx + dx;  // :348
        p2y = pty + dy;
    }

    float ncc = 1.0f;  // :365
    if (a.calc_ncc) {
        // PatchMatch::NCC (:433-469), as in k_track_block: x outer / y inner order, ordered f32 sums
        float *vref = fs, *vcur = vref + PP, *tnum = vcur + PP, *td1 = tnum + PP, *td2 = td1 + PP;
        const DevLevel &R0 = a.l1[0], &C0 = a.l2[0];
        float vr[NR], vc[NR];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < NR; r++) {
            int k = lane + 64 * r;
            k = k < P ? k : P - 1;
            int xi = k / Wd - h, yi = k - (k / Wd) * Wd - h;
            vr[r] = sample<true>(R0, refx + xi, refy + yi);
            if (a.use_affine) {
                float wxx = A00 * xi + A01 * yi, wyy = A10 * xi + A11 * yi;
                vc[r] = sample<true>(C0, p2x + wxx, p2y + wyy);
            } else {
                vc[r] = sample<true>(C0, p2x + xi, p2y + yi);
            }
            if (lane + 64 * r < P) {
                vref[lane + 64 * r] = vr[r];
                vcur[lane + 64 * r] = vc[r];
            }
        }
        __syncthreads();
        const int row = lane >> 4, lr = lane & 15;
        float *sh_f = reinterpret_cast<float *>(acc);
        {
            float m = chain_rows_f32<TAIL>(lds_off(row == 1 ? vcur : vref) + 8u * lr, 128u, nfull);
            if (lr == 0 && row < 2) sh_f[row] = m;
        }
        __syncthreads();
        const float mean_ref = sh_f[0] / (float)P, mean_cur = sh_f[1] / (float)P;
#pragma unroll
        for (int r = 0; r < NR; r++) {
            int k = lane + 64 * r;
            if (k < P) {
                float dr = vr[r] - mean_ref, dc = vc[r] - mean_cur;
                tnum[k] = dr * dc;
                td1[k] = dr * dr;
                td2[k] = dc * dc;
            }
        }
        __syncthreads();
        {
            const float *src = row == 0 ? tnum : (row == 1 ? td1 : td2);
            float v = chain_rows_f32<TAIL>(lds_off(src) + 8u * lr, 128u, nfull);
            if (lr == 0 && row < 3) sh_f[2 + row] = v;
        }
        __syncthreads();
        ncc = (float)((double)sh_f[2] / sqrt((double)(sh_f[3] * sh_f[4]) + 1e-10));
    }
    if (lane == 0) write_outputs(a, i, p2x, p2y, succ, lastCost, 1, ncc, iters);
}

}  // namespace pagk
