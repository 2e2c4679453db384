// microbench15.hip -- issue cost of packed f32 arithmetic on gfx950: are v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32
// full-rate (one issue slot for two results per lane) as the 157 TFLOP/s "vector FP32" figure needs, or does a packed
// instruction cost two slots?  Independent chains (8 accumulators per lane), 1 / 2 / 4 waves per SIMD.
// hipcc --offload-arch=gfx950 -O2 -o tools/bin/microbench15 tools/microbench15.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ void __launch_bounds__(256) k_rate(float *out, unsigned long long *cyc, int n)
{
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    f32x2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    const float m = 1.0000001f;
    const f32x2 pm = {1.0000001f, 0.9999999f};
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int k = 0; k < n; k++) {
        if (KIND == 0) {  // 8 independent v_mul_f32
            asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                         "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
        } else if (KIND == 1) {  // 8 independent v_pk_mul_f32
            asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                         "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pm));
        } else if (KIND == 2) {  // 8 independent v_pk_add_f32
            asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                         "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pm));
        } else if (KIND == 3) {  // 8 independent v_pk_fma_f32
            asm volatile("v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n"
                         "v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pm));
        } else if (KIND == 4) {  // 8 independent v_fma_f32
            asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                         "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
        } else if (KIND == 5) {  // 8 independent v_fma_f64
            double *d = nullptr; (void)d;
            asm volatile("v_cvt_i32_f32 %0, %0\n v_cvt_i32_f32 %1, %1\n v_cvt_i32_f32 %2, %2\n v_cvt_i32_f32 %3, %3\n"
                         "v_fract_f32 %4, %4\n v_fract_f32 %5, %5\n v_fract_f32 %6, %6\n v_fract_f32 %7, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else {  // 8 v_readlane_b32 into SGPRs
            int s0, s1, s2, s3, s4, s5, s6, s7;
            asm volatile("v_readlane_b32 %0, %8, 3\n v_readlane_b32 %1, %9, 5\n v_readlane_b32 %2, %8, 7\n v_readlane_b32 %3, %9, 9\n"
                         "v_readlane_b32 %4, %8, 11\n v_readlane_b32 %5, %9, 13\n v_readlane_b32 %6, %8, 15\n v_readlane_b32 %7, %9, 17"
                         : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3), "=s"(s4), "=s"(s5), "=s"(s6), "=s"(s7) : "v"(a0), "v"(a1));
            asm volatile("" :: "s"(s0), "s"(s1), "s"(s2), "s"(s3), "s"(s4), "s"(s5), "s"(s6), "s"(s7));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main()
{
    float *out; unsigned long long *cyc;
    CHK(hipMalloc(&out, 4 * 256 * 4096)); CHK(hipMalloc(&cyc, 8 * 4096));
    const char *names[7] = {"v_mul_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_pk_fma_f32", "v_fma_f32", "v_cvt_i32_f32 / v_fract_f32", "v_readlane_b32"};
    const int n = 4096;
    // one workgroup of 256 threads = one wave per SIMD of a CU; k workgroups per CU need k x 256 CUs' worth of blocks:
    // launch blocks = 256 CUs x k and read the slowest block
    for (int kind = 0; kind < 7; kind++)
        for (int per_cu = 1; per_cu <= 8; per_cu *= 2) {
            const int blocks = 256 * per_cu;
            for (int rep = 0; rep < 2; rep++) {
                switch (kind) {
                    case 0: hipLaunchKernelGGL(k_rate<0>, dim3(blocks), dim3(256), 0, 0, out, cyc, n); break;
                    case 1: hipLaunchKernelGGL(k_rate<1>, dim3(blocks), dim3(256), 0, 0, out, cyc, n); break;
                    case 2: hipLaunchKernelGGL(k_rate<2>, dim3(blocks), dim3(256), 0, 0, out, cyc, n); break;
                    case 3: hipLaunchKernelGGL(k_rate<3>, dim3(blocks), dim3(256), 0, 0, out, cyc, n); break;
                    case 4: hipLaunchKernelGGL(k_rate<4>, dim3(blocks), dim3(256), 0, 0, out, cyc, n); break;
                    case 5: hipLaunchKernelGGL(k_rate<5>, dim3(blocks), dim3(256), 0, 0, out, cyc, n); break;
                    default: hipLaunchKernelGGL(k_rate<6>, dim3(blocks), dim3(256), 0, 0, out, cyc, n); break;
                }
                CHK(hipDeviceSynchronize());
            }
            static unsigned long long h[4096];  // (256 CUs x 8 blocks = 2048)
            CHK(hipMemcpy(h, cyc, 8 * blocks, hipMemcpyDeviceToHost));
            unsigned long long mx = 0, sum = 0;
            for (int b = 0; b < blocks; b++) { mx = h[b] > mx ? h[b] : mx; sum += h[b]; }
            // cycles per instruction per wave, and per instruction per SIMD (= the former / waves per SIMD)
            printf("%-30s %d wave(s) per SIMD: %.2f cycles per instruction per wave (mean; max %.2f) = %.2f per SIMD\n", names[kind], per_cu,
                   (double)sum / blocks / n / 8, (double)mx / n / 8, (double)sum / blocks / n / 8 / per_cu);
        }
    return 0;
}
