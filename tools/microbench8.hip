// microbench8.hip -- issue cost of v_mul_f32 vs v_pk_mul_f32 (and add) for one wave / two waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2v __attribute__((ext_vector_type(2)));
__device__ inline unsigned long long now() { return __builtin_amdgcn_s_memtime(); }

#define REP8(x) x x x x x x x x
__global__ void k_scalar(float *out, unsigned long long *cyc, float m, int n)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    unsigned long long t0 = now();
    for (int k = 0; k < n; k++) {
        asm volatile(REP8("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                          "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n")
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
    }
    unsigned long long t1 = now();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void k_packed(float *out, unsigned long long *cyc, float m, int n)
{
    float2v a0 = {(float)threadIdx.x, 1}, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float2v mm = {m, m};
    unsigned long long t0 = now();
    for (int k = 0; k < n; k++) {
        asm volatile(REP8("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                          "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n")
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(mm));
    }
    unsigned long long t1 = now();
    float2v s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main()
{
    float *d_out; unsigned long long *d_cyc, c;
    hipMalloc(&d_out, 1 << 20); hipMalloc(&d_cyc, 1 << 12);
    auto rd = [&]() { hipDeviceSynchronize(); hipMemcpy(&c, d_cyc, 8, hipMemcpyDeviceToHost); return (double)c; };
    const int n = 200; const double instrs = n * 64.0;
    for (int threads : {64, 128, 256, 512}) {
        hipLaunchKernelGGL(k_scalar, dim3(1), dim3(threads), 0, 0, d_out, d_cyc, 1.0000001f, n);
        double s = rd() / instrs;
        hipLaunchKernelGGL(k_packed, dim3(1), dim3(threads), 0, 0, d_out, d_cyc, 1.0000001f, n);
        double p = rd() / instrs;
        printf("%d threads in one workgroup (%d wave(s) per SIMD): v_mul_f32 %.2f cyc/instr, v_pk_mul_f32 %.2f cyc/instr\n", threads, (threads + 255) / 256, s, p);
    }
    return 0;
}
