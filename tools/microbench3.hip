// microbench3.hip -- inline-asm ordered chain: 8-step groups, ping-pong, counted lgkmcnt
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__device__ inline unsigned long long now() { return __builtin_amdgcn_s_memtime(); }

// n16 = number of 16-step blocks; src must have 8 readable doubles of slack past the end
__device__ __forceinline__ double chain_asm_f64(const double *src, int n16)
{
    uint32_t addr = (uint32_t)(uintptr_t)src;  // LDS byte address (low 32 bits of the flat LDS pointer)
    double s = 0.0;
    double a0, a1, a2, a3, a4, a5, a6, a7, b0, b1, b2, b3, b4, b5, b6, b7;
    asm volatile(
        "ds_read_b64 %[a0], %[ad] offset:0\n\t"
        "ds_read_b64 %[a1], %[ad] offset:8\n\t"
        "ds_read_b64 %[a2], %[ad] offset:16\n\t"
        "ds_read_b64 %[a3], %[ad] offset:24\n\t"
        "ds_read_b64 %[a4], %[ad] offset:32\n\t"
        "ds_read_b64 %[a5], %[ad] offset:40\n\t"
        "ds_read_b64 %[a6], %[ad] offset:48\n\t"
        "ds_read_b64 %[a7], %[ad] offset:56\n\t"
        "1:\n\t"
        "ds_read_b64 %[b0], %[ad] offset:64\n\t"
        "ds_read_b64 %[b1], %[ad] offset:72\n\t"
        "ds_read_b64 %[b2], %[ad] offset:80\n\t"
        "ds_read_b64 %[b3], %[ad] offset:88\n\t"
        "ds_read_b64 %[b4], %[ad] offset:96\n\t"
        "ds_read_b64 %[b5], %[ad] offset:104\n\t"
        "ds_read_b64 %[b6], %[ad] offset:112\n\t"
        "ds_read_b64 %[b7], %[ad] offset:120\n\t"
        "s_waitcnt lgkmcnt(8)\n\t"
        "v_add_f64 %[s], %[s], %[a0]\n\t"
        "v_add_f64 %[s], %[s], %[a1]\n\t"
        "v_add_f64 %[s], %[s], %[a2]\n\t"
        "v_add_f64 %[s], %[s], %[a3]\n\t"
        "v_add_f64 %[s], %[s], %[a4]\n\t"
        "v_add_f64 %[s], %[s], %[a5]\n\t"
        "v_add_f64 %[s], %[s], %[a6]\n\t"
        "v_add_f64 %[s], %[s], %[a7]\n\t"
        "v_add_u32 %[ad], 0x80, %[ad]\n\t"
        "ds_read_b64 %[a0], %[ad] offset:0\n\t"
        "ds_read_b64 %[a1], %[ad] offset:8\n\t"
        "ds_read_b64 %[a2], %[ad] offset:16\n\t"
        "ds_read_b64 %[a3], %[ad] offset:24\n\t"
        "ds_read_b64 %[a4], %[ad] offset:32\n\t"
        "ds_read_b64 %[a5], %[ad] offset:40\n\t"
        "ds_read_b64 %[a6], %[ad] offset:48\n\t"
        "ds_read_b64 %[a7], %[ad] offset:56\n\t"
        "s_waitcnt lgkmcnt(8)\n\t"
        "v_add_f64 %[s], %[s], %[b0]\n\t"
        "v_add_f64 %[s], %[s], %[b1]\n\t"
        "v_add_f64 %[s], %[s], %[b2]\n\t"
        "v_add_f64 %[s], %[s], %[b3]\n\t"
        "v_add_f64 %[s], %[s], %[b4]\n\t"
        "v_add_f64 %[s], %[s], %[b5]\n\t"
        "v_add_f64 %[s], %[s], %[b6]\n\t"
        "v_add_f64 %[s], %[s], %[b7]\n\t"
        "s_sub_u32 %[n], %[n], 1\n\t"
        "s_cmp_lg_u32 %[n], 0\n\t"
        "s_cbranch_scc1 1b\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        : [s] "+v"(s), [ad] "+v"(addr), [n] "+s"(n16), [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2),
          [a3] "=&v"(a3), [a4] "=&v"(a4), [a5] "=&v"(a5), [a6] "=&v"(a6), [a7] "=&v"(a7), [b0] "=&v"(b0),
          [b1] "=&v"(b1), [b2] "=&v"(b2), [b3] "=&v"(b3), [b4] "=&v"(b4), [b5] "=&v"(b5), [b6] "=&v"(b6),
          [b7] "=&v"(b7)
        :
        : "memory", "scc");
    return s;
}

__global__ void k_chain(double *out, unsigned long long *cyc, int PP, int nl)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    for (int k = threadIdx.x; k < 11 * PP + 64; k += blockDim.x) lds[k] = 1.0 + 1e-9 * k;
    __syncthreads();
    unsigned long long t0 = now();
    if (threadIdx.x < nl) {
        double s = chain_asm_f64(lds + threadIdx.x * PP, PP / 16);
        out[threadIdx.x] = s;
    }
    unsigned long long t1 = now();
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_ref(double *out, int PP)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    for (int k = threadIdx.x; k < 11 * PP + 64; k += blockDim.x) lds[k] = 1.0 + 1e-9 * k;
    __syncthreads();
    if (threadIdx.x < 11) {
        double s = 0.0;
        for (int k = 0; k < PP; k++) s += lds[threadIdx.x * PP + k];
        out[16 + threadIdx.x] = s;
    }
}
int main()
{
    double *d_out; unsigned long long *d_cyc, c;
    CHK(hipMalloc(&d_out, 1 << 16)); CHK(hipMalloc(&d_cyc, 64));
    auto rd = [&]() { hipDeviceSynchronize(); hipMemcpy(&c, d_cyc, 8, hipMemcpyDeviceToHost); return (double)c; };
    const int PP = 448;
    size_t lds = (11 * PP + 64) * 8;
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(k_chain, dim3(1), dim3(256), lds, 0, d_out, d_cyc, PP, 11);
        printf("asm f64 chain, 11 lanes: %.2f cyc/step (%.0f)\n", rd() / PP, (double)c);
        hipLaunchKernelGGL(k_chain, dim3(1), dim3(256), lds, 0, d_out, d_cyc, PP, 1);
        printf("asm f64 chain, 1 lane: %.2f cyc/step (%.0f)\n", rd() / PP, (double)c);
    }
    hipLaunchKernelGGL(k_chain, dim3(1), dim3(256), lds, 0, d_out, d_cyc, PP, 11);
    hipLaunchKernelGGL(k_ref, dim3(1), dim3(256), lds, 0, d_out, PP);
    double h[32];
    hipDeviceSynchronize();
    hipMemcpy(h, d_out, sizeof h, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int k = 0; k < 11; k++) bad += h[k] != h[16 + k];
    printf("asm vs plain loop: %d mismatches (%.17g %.17g)\n", bad, h[3], h[19]);
    return 0;
}
