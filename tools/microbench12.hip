// microbench12 -- issue cost (independent stream) and dependent latency of the VALU instructions the sampler is made
// of, for ONE wave alone on its SIMD (the straggler regime of the tracking kernels) and for two waves sharing a SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/microbench12 tools/microbench12.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

// independent: 8 different destination registers round-robin; dependent: one register chain
#define KERNEL(name, indep8, dep1)                                                                              \
    __global__ void __launch_bounds__(64) name(unsigned long long *out, float seed)                             \
    {                                                                                                           \
        float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, \
              a7 = seed + 7, b = seed * 0.5f + 3.25f;                                                           \
        double d0 = seed, d1 = seed + 1.5;                                                                      \
        int i0 = (int)seed + threadIdx.x;                                                                       \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                   \
        for (int k = 0; k < 16; k++) { asm volatile(REP16(indep8) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(d0), "+v"(d1), "+v"(i0) : "v"(b)); } \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                   \
        for (int k = 0; k < 16; k++) { asm volatile(REP64(dep1) REP64(dep1) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(d0), "+v"(d1), "+v"(i0) : "v"(b)); } \
        unsigned long long t2 = __builtin_amdgcn_s_memtime();                                                   \
        if (threadIdx.x == 0) {                                                                                 \
            out[2 * blockIdx.x] = t1 - t0;                                                                      \
            out[2 * blockIdx.x + 1] = t2 - t1;                                                                  \
        }                                                                                                       \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)d0 + (float)d1 + (float)i0 == 1.2345f) out[0] = 0;  \
    }

KERNEL(k_add, "v_add_f32 %0, %0, %11\n v_add_f32 %1, %1, %11\n v_add_f32 %2, %2, %11\n v_add_f32 %3, %3, %11\n v_add_f32 %4, %4, %11\n v_add_f32 %5, %5, %11\n v_add_f32 %6, %6, %11\n v_add_f32 %7, %7, %11\n",
       "v_add_f32 %0, %0, %11\n")
KERNEL(k_mul, "v_mul_f32 %0, %0, %11\n v_mul_f32 %1, %1, %11\n v_mul_f32 %2, %2, %11\n v_mul_f32 %3, %3, %11\n v_mul_f32 %4, %4, %11\n v_mul_f32 %5, %5, %11\n v_mul_f32 %6, %6, %11\n v_mul_f32 %7, %7, %11\n",
       "v_mul_f32 %0, %0, %11\n")
KERNEL(k_fract, "v_fract_f32 %0, %0\n v_fract_f32 %1, %1\n v_fract_f32 %2, %2\n v_fract_f32 %3, %3\n v_fract_f32 %4, %4\n v_fract_f32 %5, %5\n v_fract_f32 %6, %6\n v_fract_f32 %7, %7\n",
       "v_fract_f32 %0, %0\n")
KERNEL(k_cvti, "v_cvt_i32_f32 %0, %0\n v_cvt_i32_f32 %1, %1\n v_cvt_i32_f32 %2, %2\n v_cvt_i32_f32 %3, %3\n v_cvt_i32_f32 %4, %4\n v_cvt_i32_f32 %5, %5\n v_cvt_i32_f32 %6, %6\n v_cvt_i32_f32 %7, %7\n",
       "v_cvt_i32_f32 %0, %0\n")
KERNEL(k_ubyte, "v_cvt_f32_ubyte0 %0, %0\n v_cvt_f32_ubyte1 %1, %1\n v_cvt_f32_ubyte2 %2, %2\n v_cvt_f32_ubyte3 %3, %3\n v_cvt_f32_ubyte0 %4, %4\n v_cvt_f32_ubyte1 %5, %5\n v_cvt_f32_ubyte2 %6, %6\n v_cvt_f32_ubyte3 %7, %7\n",
       "v_cvt_f32_ubyte0 %0, %0\n")
KERNEL(k_mad24, "v_mad_i32_i24 %0, %0, %11, %0\n v_mad_i32_i24 %1, %1, %11, %1\n v_mad_i32_i24 %2, %2, %11, %2\n v_mad_i32_i24 %3, %3, %11, %3\n v_mad_i32_i24 %4, %4, %11, %4\n v_mad_i32_i24 %5, %5, %11, %5\n v_mad_i32_i24 %6, %6, %11, %6\n v_mad_i32_i24 %7, %7, %11, %7\n",
       "v_mad_i32_i24 %0, %0, %11, %0\n")
KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %11, vcc\n v_cndmask_b32 %1, %1, %11, vcc\n v_cndmask_b32 %2, %2, %11, vcc\n v_cndmask_b32 %3, %3, %11, vcc\n v_cndmask_b32 %4, %4, %11, vcc\n v_cndmask_b32 %5, %5, %11, vcc\n v_cndmask_b32 %6, %6, %11, vcc\n v_cndmask_b32 %7, %7, %11, vcc\n",
       "v_cndmask_b32 %0, %0, %11, vcc\n")
KERNEL(k_mulf64, "v_mul_f64 %8, %8, %9\n v_mul_f64 %8, %8, %9\n v_mul_f64 %8, %8, %9\n v_mul_f64 %8, %8, %9\n v_mul_f64 %8, %8, %9\n v_mul_f64 %8, %8, %9\n v_mul_f64 %8, %8, %9\n v_mul_f64 %8, %8, %9\n",
       "v_mul_f64 %8, %8, %9\n")
KERNEL(k_cvtf64, "v_cvt_f64_f32 %8, %0\n v_cvt_f64_f32 %9, %1\n v_cvt_f64_f32 %8, %2\n v_cvt_f64_f32 %9, %3\n v_cvt_f64_f32 %8, %4\n v_cvt_f64_f32 %9, %5\n v_cvt_f64_f32 %8, %6\n v_cvt_f64_f32 %9, %7\n",
       "v_cvt_f64_f32 %8, %0\n v_cvt_f32_f64 %0, %8\n")

// packed f32: operands must be 64-bit register pairs -> use the two doubles' registers as float pairs
KERNEL(k_pkmul, "v_pk_mul_f32 %8, %8, %9\n v_pk_mul_f32 %9, %9, %8\n v_pk_mul_f32 %8, %8, %9\n v_pk_mul_f32 %9, %9, %8\n v_pk_mul_f32 %8, %8, %9\n v_pk_mul_f32 %9, %9, %8\n v_pk_mul_f32 %8, %8, %9\n v_pk_mul_f32 %9, %9, %8\n",
       "v_pk_mul_f32 %8, %8, %9\n")
KERNEL(k_pkadd, "v_pk_add_f32 %8, %8, %9\n v_pk_add_f32 %9, %9, %8\n v_pk_add_f32 %8, %8, %9\n v_pk_add_f32 %9, %9, %8\n v_pk_add_f32 %8, %8, %9\n v_pk_add_f32 %9, %9, %8\n v_pk_add_f32 %8, %8, %9\n v_pk_add_f32 %9, %9, %8\n",
       "v_pk_add_f32 %8, %8, %9\n")

template <typename K>
void run(const char *name, K kern, unsigned long long *d)
{
    for (int wpc : {1, 2}) {  // blocks of one wave; 256 blocks -> one per CU ... x8 -> two per SIMD
        const int blocks = wpc == 1 ? 256 : 2048;
        for (int r = 0; r < 2; r++) {
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, d, 1.0f);
            (void)hipDeviceSynchronize();
        }
        std::vector<unsigned long long> h(2 * blocks);
        (void)hipMemcpy(h.data(), d, 2 * blocks * 8, hipMemcpyDeviceToHost);
        double a = 0, b = 0;
        for (int i = 0; i < blocks; i++) a += h[2 * i], b += h[2 * i + 1];
        printf("%-22s %s: independent %5.2f cycles/instr   dependent %5.2f cycles/instr\n", name,
               wpc == 1 ? "1 wave / SIMD " : "2 waves / SIMD", a / blocks / (16.0 * 16 * 8), b / blocks / (16.0 * 128));
    }
}

int main()
{
    unsigned long long *d;
    (void)hipMalloc(&d, 2 * 2048 * 8);
    run("v_add_f32", k_add, d);
    run("v_mul_f32", k_mul, d);
    run("v_fract_f32", k_fract, d);
    run("v_cvt_i32_f32", k_cvti, d);
    run("v_cvt_f32_ubyteN", k_ubyte, d);
    run("v_mad_i32_i24", k_mad24, d);
    run("v_cndmask_b32", k_cndmask, d);
    run("v_pk_mul_f32", k_pkmul, d);
    run("v_pk_add_f32", k_pkadd, d);
    run("v_mul_f64", k_mulf64, d);
    run("v_cvt_f64_f32 (+back)", k_cvtf64, d);
    return 0;
}
