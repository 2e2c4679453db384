// pagk_score_kernel.h -- the per-correspondence scoring loops of GyroAidedTracker::GeometryValidation
// (reference src/gyro_aided_tracker.cpp:589-768): CheckHomography (symmetric transfer error, chi-square
// 5.99) and CheckFundamental (point-to-epipolar-line distance, 3.84 / 5.99).  The RANSAC model fit that
// precedes them is third-party (cv::findHomography / cv::findFundamentalMat) and stays with the caller:
// the fitted matrices come in as arguments.
//
// One launch scores both models, block 0 the homography and block 1 the fundamental matrix (the
// reference runs the two loops on two std::threads, :455-460).  The per-point arithmetic is
// data-parallel; the score is a float accumulated in index order (`score += th - chiSquare`), so it is
// summed by the same ordered DPP row chain as the tracker's cost: per point two terms in the reference's
// order, an outlier's skipped addition written as +0.0f (x + 0 == x for the non-negative or NaN running
// score), the running score carried into the next chunk as the chunk's first term (0 + s == s).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "pagk_chain_asm.h"

namespace pagk {

struct ScoreArgs {
    double H21[9], H12[9], F21[9];  // row-major 3x3
    const float *pts1, *pts2;       // n x 2
    int n;
    float sigma;
    uint8_t *inl_H, *inl_F;  // n each
    float *scores;           // [0] = score_H, [1] = score_F
};

constexpr int kScoreChunk = 2048;  // correspondences per LDS pass (2 terms each + the carry = 128 * 32 + 1)

// CheckHomography, one correspondence (:629-676).  h double, points float: products and sums in
// double, one narrowing per `const float` initialiser.
__device__ __forceinline__ int score_h_point(const double *h, const double *hi, float u1, float v1, float u2,
                                             float v2, float invSigmaSquare, float &t2, float &t1)
{
    const float th = 5.99;
    int bIn = 1;
    const float w1in2inv = (float)(1.0 / (h[6] * (double)u1 + h[7] * (double)v1 + h[8]));  // :641
    const float u1in2 = (float)((h[0] * (double)u1 + h[1] * (double)v1 + h[2]) * (double)w1in2inv);
    const float v1in2 = (float)((h[3] * (double)u1 + h[4] * (double)v1 + h[5]) * (double)w1in2inv);
    const float squareDist2 = (u2 - u1in2) * (u2 - u1in2) + (v2 - v1in2) * (v2 - v1in2);
    const float chiSquare2 = squareDist2 * invSigmaSquare;
    if (chiSquare2 > th) {  // :648 (a NaN takes the else branch, as in the reference)
        bIn = 0;
        t2 = 0.0f;
    } else {
        t2 = th - chiSquare2;
    }
    const float w2in1inv = (float)(1.0 / (hi[6] * (double)u2 + hi[7] * (double)v2 + hi[8]));  // :657
    const float u2in1 = (float)((hi[0] * (double)u2 + hi[1] * (double)v2 + hi[2]) * (double)w2in1inv);
    const float v2in1 = (float)((hi[3] * (double)u2 + hi[4] * (double)v2 + hi[5]) * (double)w2in1inv);
    const float squareDist1 = (u1 - u2in1) * (u1 - u2in1) + (v1 - v2in1) * (v1 - v2in1);
    const float chiSquare1 = squareDist1 * invSigmaSquare;
    if (chiSquare1 > th) {  // :664
        bIn = 0;
        t1 = 0.0f;
    } else {
        t1 = th - chiSquare1;
    }
    return bIn;
}

// CheckFundamental, one correspondence (:713-768).
__device__ __forceinline__ int score_f_point(const double *f, float u1, float v1, float u2, float v2,
                                             float invSigmaSquare, float &t2, float &t1)
{
    const float th = 3.84, thScore = 5.99;
    int bIn = 1;
    const float a2 = (float)(f[0] * (double)u1 + f[1] * (double)v1 + f[2]);  // :725-727
    const float b2 = (float)(f[3] * (double)u1 + f[4] * (double)v1 + f[5]);
    const float c2 = (float)(f[6] * (double)u1 + f[7] * (double)v1 + f[8]);
    const float num2 = a2 * u2 + b2 * v2 + c2;  // :730
    const float squareDist2 = num2 * num2 / (a2 * a2 + b2 * b2);
    const float chiSquare2 = squareDist2 * invSigmaSquare;
    if (chiSquare2 > th) {  // :734
        bIn = 0;
        t2 = 0.0f;
    } else {
        t2 = thScore - chiSquare2;
    }
    const float a1 = (float)((double)u2 * f[0] + (double)v2 * f[3] + f[6]);  // :743-745
    const float b1 = (float)((double)u2 * f[1] + (double)v2 * f[4] + f[7]);
    const float c1 = (float)((double)u2 * f[2] + (double)v2 * f[5] + f[8]);
    const float num1 = a1 * u1 + b1 * v1 + c1;
    const float squareDist1 = num1 * num1 / (a1 * a1 + b1 * b1);
    const float chiSquare1 = squareDist1 * invSigmaSquare;
    if (chiSquare1 > th) {  // :752
        bIn = 0;
        t1 = 0.0f;
    } else {
        t1 = thScore - chiSquare1;
    }
    return bIn;
}

__global__ void __launch_bounds__(256) k_geometry_scores(ScoreArgs a)
{
    // [0] carry, [1 .. 2*chunk] terms, + one block the chain may read past the end
    __shared__ __attribute__((aligned(16))) float terms[1 + 2 * kScoreChunk + 32];
    const int model = blockIdx.x;  // 0: homography, 1: fundamental
    const int tid = threadIdx.x;
    const float invSigmaSquare = (float)(1.0 / (double)(a.sigma * a.sigma));  // :625 / :709
    uint8_t *inl = model == 0 ? a.inl_H : a.inl_F;
    float carry = 0.0f;  // :623 / :706
    for (int base = 0; base < a.n; base += kScoreChunk) {
        const int cnt = a.n - base < kScoreChunk ? a.n - base : kScoreChunk;
        const int nfull = (2 * cnt + 31) / 32;  // chain length = 32 * nfull + 1
        if (tid == 0) terms[0] = carry;
        for (int k = tid; k < 16 * nfull; k += 256) {
            float t2 = 0.0f, t1 = 0.0f;  // padding past cnt: +0.0f
            if (k < cnt) {
                const int i = base + k;
                const float u1 = a.pts1[2 * i], v1 = a.pts1[2 * i + 1], u2 = a.pts2[2 * i], v2 = a.pts2[2 * i + 1];
                const int bIn = model == 0 ? score_h_point(a.H21, a.H12, u1, v1, u2, v2, invSigmaSquare, t2, t1)
                                           : score_f_point(a.F21, u1, v1, u2, v2, invSigmaSquare, t2, t1);
                inl[i] = (uint8_t)bIn;
            }
            terms[1 + 2 * k] = t2;
            terms[2 + 2 * k] = t1;
        }
        __syncthreads();
        if (tid < 64) {  // wave 0, all lanes; its four DPP rows fold the same array
            const uint32_t addr = (uint32_t)(uintptr_t)terms + 8u * (uint32_t)(tid & 15);
            carry = chain_rows_f32<1>(addr, 128u, nfull);
        }
        __syncthreads();  // the chain has read the chunk before the next one is written
    }
    if (tid == 0) a.scores[model] = carry;
}

}  // namespace pagk
