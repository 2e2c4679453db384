// pagk_selftest_kernel.h -- the arithmetic building blocks of the solve, callable on arbitrary operands
// (pagk_selftest_divide / pagk_selftest_solve in include/pagk.h).  Diagnostics: the tracking path never calls these;
// they exist so that a host can check, on ITS device and with ITS solver_variant, the two claims the solve rests on:
//   * a division by a prepared denominator (pagk_device.h: den_prepare / div_group) returns the bits of the plain
//     correctly rounded f64 division for every operand pair, and the square-root-free convergence test is
//     equivalent to `norm < 1e-2`;
//   * H.llt().solve(b) and update.norm() (src/patch_match.cpp:319,343) -- one lane per system and four lanes per
//     system -- return the bits of the CPU restatement for any matrix, including failed pivots, zeros, NaN and
//     out-of-range magnitudes.
#pragma once
#include "pagk_device.h"

namespace pagk {

// H22 of the pipelined 4-wave kernel: the ordered sum of P copies of c * c in closed form (repeat_sum_f64) beside the
// P-step loop it replaces (src/patch_match.cpp:296 with J[2] = c).
__global__ void __launch_bounds__(256) k_selftest_repeat_sum(int n, const float *c, int P, double *closed, double *loop)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double cd = (double)c[i];
    closed[i] = repeat_sum_f64(cd * cd, P);
    double s = 0.0;
    for (int k = 0; k < P; k++) s = __builtin_fma(cd, cd, s);
    loop[i] = s;
}

__global__ void __launch_bounds__(256) k_selftest_divide(int n, const double *num, const double *den, double *q_plain,
                                                         double *q_prepared, double *root, double *root_lean)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double a = num[i], d = den[i];
    q_plain[i] = a / d;
    q_prepared[i] = div_one(a, d);
    root[i] = sqrt(a);
    root_lean[i] = sqrt_one(a);
}

// block = 64 threads = 16 systems: thread t solves system blockIdx * 16 + t / 4 with the four-lane form (lane t % 4);
// threads 0..15 also solve systems blockIdx * 16 + t with the one-lane form.
__global__ void __launch_bounds__(64) k_selftest_solve(int n, const double *H, const double *b, uint32_t sv,
                                                       double *x_serial, double *norm_serial, double *x_lanes,
                                                       double *nsq_lanes)
{
    const int t = threadIdx.x;
    auto load = [&](int sys, double (&M)[4][4], double (&rhs)[4]) {
        sys = sys < n ? sys : n - 1;
        for (int r = 0; r < 4; r++) {
            for (int c = 0; c < 4; c++) M[r][c] = H[(size_t)sys * 16 + r * 4 + c];
            rhs[r] = b[(size_t)sys * 4 + r];
        }
    };
    {
        // the four-lane form broadcasts with v_readlane from FIXED lanes, so the systems of a wave take turns
        for (int g = 0; g < 16; g++) {
            const int sys = blockIdx.x * 16 + g;
            double M[4][4], rhs[4], x[4];
            load(sys, M, rhs);
            const double nsq = llt4_solve_nsq_lanes(M, rhs, t & 3, x, sv);  // every lane holds the same system
            if (t == 0 && sys < n) {
                for (int k = 0; k < 4; k++) x_lanes[(size_t)sys * 4 + k] = x[k];
                nsq_lanes[sys] = nsq;
            }
        }
    }
    if (t < 16) {
        const int sys = blockIdx.x * 16 + t;
        double M[4][4], rhs[4], x[4];
        load(sys, M, rhs);
        const double nrm = llt4_solve_norm(M, rhs, x, sv);
        if (sys < n) {
            for (int k = 0; k < 4; k++) x_serial[(size_t)sys * 4 + k] = x[k];
            norm_serial[sys] = nrm;
        }
    }
}

}  // namespace pagk
