
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__device__ inline unsigned long long now() { return __builtin_amdgcn_s_memtime(); }
__device__ __forceinline__ double chain_b64(const double *src, int n16)
{
    uint32_t addr = (uint32_t)(uintptr_t)src; double s = 0.0;
    double a0,a1,a2,a3,a4,a5,a6,a7,b0,b1,b2,b3,b4,b5,b6,b7;
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
        "s_waitcnt lgkmcnt(7)\n\t"
        "v_add_f64 %[s], %[s], %[a0]\n\t"
        "ds_read_b64 %[b0], %[ad] offset:64\n\t"
        "s_waitcnt lgkmcnt(7)\n\t"
        "v_add_f64 %[s], %[s], %[a1]\n\t"
        "ds_read_b64 %[b1], %[ad] offset:72\n\t"
        "s_waitcnt lgkmcnt(7)\n\t"
        "v_add_f64 %[s], %[s], %[a2]\n\t"
        "ds_read_b64 %[b2], %[ad] offset:80\n\t"
        "s_waitcnt lgkmcnt(7)\n\t"
        "v_add_f64 %[s], %[s], %[a3]\n\t"
        "ds_read_b64 %[b3], %[ad] offset:88\n\t"
        "s_waitcnt lgkmcnt(7)\n\t"
        "v_add_f64 %[s], %[s], %[a4]\n\t"
        "ds_read_b64 %[b4], %[ad] offset:96\n\t"
        "s_waitcnt lgkmcnt(7)\n\t"
        "v_add_f64 %[s], %[s], %[a5]\n\t"
        "ds_read_b64 %[b5], %[ad] offset:104\n\t"
        "s_waitcnt lgkmcnt(7)\n\t"
        "v_add_f64 %[s], %[s], %[a6]\n\t"
        "ds_read_b64 %[b6], %[ad] offset:112\n\t"
        "s_waitcnt lgkmcnt(7)\n\t"
        "v_add_f64 %[s], %[s], %[a7]\n\t"
        "ds_read_b64 %[b7], %[ad] offset:120\n\t"
        "v_add_u32 %[ad], 0x80, %[ad]\n\t"
        "s_waitcnt lgkmcnt(7)\n\t"
        "v_add_f64 %[s], %[s], %[b0]\n\t"
        "ds_read_b64 %[a0], %[ad] offset:0\n\t"
        "s_waitcnt lgkmcnt(7)\n\t"
        "v_add_f64 %[s], %[s], %[b1]\n\t"
        "ds_read_b64 %[a1], %[ad] offset:8\n\t"
        "s_waitcnt lgkmcnt(7)\n\t"
        "v_add_f64 %[s], %[s], %[b2]\n\t"
        "ds_read_b64 %[a2], %[ad] offset:16\n\t"
        "s_waitcnt lgkmcnt(7)\n\t"
        "v_add_f64 %[s], %[s], %[b3]\n\t"
        "ds_read_b64 %[a3], %[ad] offset:24\n\t"
        "s_waitcnt lgkmcnt(7)\n\t"
        "v_add_f64 %[s], %[s], %[b4]\n\t"
        "ds_read_b64 %[a4], %[ad] offset:32\n\t"
        "s_waitcnt lgkmcnt(7)\n\t"
        "v_add_f64 %[s], %[s], %[b5]\n\t"
        "ds_read_b64 %[a5], %[ad] offset:40\n\t"
        "s_waitcnt lgkmcnt(7)\n\t"
        "v_add_f64 %[s], %[s], %[b6]\n\t"
        "ds_read_b64 %[a6], %[ad] offset:48\n\t"
        "s_waitcnt lgkmcnt(7)\n\t"
        "v_add_f64 %[s], %[s], %[b7]\n\t"
        "ds_read_b64 %[a7], %[ad] offset:56\n\t"
        "s_sub_u32 %[n], %[n], 1\n\t"
        "s_cmp_lg_u32 %[n], 0\n\t"
        "s_cbranch_scc1 1b\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        : [s] "+v"(s), [ad] "+v"(addr), [n] "+s"(n16), [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3), [a4] "=&v"(a4), [a5] "=&v"(a5), [a6] "=&v"(a6), [a7] "=&v"(a7), [b0] "=&v"(b0), [b1] "=&v"(b1), [b2] "=&v"(b2), [b3] "=&v"(b3), [b4] "=&v"(b4), [b5] "=&v"(b5), [b6] "=&v"(b6), [b7] "=&v"(b7) : : "memory", "scc");
    return s;
}
__device__ __forceinline__ double chain_b128(const double *src, int n16)
{
    uint32_t addr = (uint32_t)(uintptr_t)src; double s = 0.0;
    asm volatile(
        "ds_read_b128 v[96:99], %[ad] offset:0\n\t"
        "ds_read_b128 v[100:103], %[ad] offset:16\n\t"
        "ds_read_b128 v[104:107], %[ad] offset:32\n\t"
        "ds_read_b128 v[108:111], %[ad] offset:48\n\t"
        "1:\n\t"
        "s_waitcnt lgkmcnt(3)\n\t"
        "v_add_f64 %[s], %[s], v[96:97]\n\t"
        "ds_read_b128 v[112:115], %[ad] offset:64\n\t"
        "v_add_f64 %[s], %[s], v[98:99]\n\t"
        "s_waitcnt lgkmcnt(3)\n\t"
        "v_add_f64 %[s], %[s], v[100:101]\n\t"
        "ds_read_b128 v[116:119], %[ad] offset:80\n\t"
        "v_add_f64 %[s], %[s], v[102:103]\n\t"
        "s_waitcnt lgkmcnt(3)\n\t"
        "v_add_f64 %[s], %[s], v[104:105]\n\t"
        "ds_read_b128 v[120:123], %[ad] offset:96\n\t"
        "v_add_f64 %[s], %[s], v[106:107]\n\t"
        "s_waitcnt lgkmcnt(3)\n\t"
        "v_add_f64 %[s], %[s], v[108:109]\n\t"
        "ds_read_b128 v[124:127], %[ad] offset:112\n\t"
        "v_add_f64 %[s], %[s], v[110:111]\n\t"
        "v_add_u32 %[ad], 0x80, %[ad]\n\t"
        "s_waitcnt lgkmcnt(3)\n\t"
        "v_add_f64 %[s], %[s], v[112:113]\n\t"
        "ds_read_b128 v[96:99], %[ad] offset:0\n\t"
        "v_add_f64 %[s], %[s], v[114:115]\n\t"
        "s_waitcnt lgkmcnt(3)\n\t"
        "v_add_f64 %[s], %[s], v[116:117]\n\t"
        "ds_read_b128 v[100:103], %[ad] offset:16\n\t"
        "v_add_f64 %[s], %[s], v[118:119]\n\t"
        "s_waitcnt lgkmcnt(3)\n\t"
        "v_add_f64 %[s], %[s], v[120:121]\n\t"
        "ds_read_b128 v[104:107], %[ad] offset:32\n\t"
        "v_add_f64 %[s], %[s], v[122:123]\n\t"
        "s_waitcnt lgkmcnt(3)\n\t"
        "v_add_f64 %[s], %[s], v[124:125]\n\t"
        "ds_read_b128 v[108:111], %[ad] offset:48\n\t"
        "v_add_f64 %[s], %[s], v[126:127]\n\t"
        "s_sub_u32 %[n], %[n], 1\n\t"
        "s_cmp_lg_u32 %[n], 0\n\t"
        "s_cbranch_scc1 1b\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        : [s] "+v"(s), [ad] "+v"(addr), [n] "+s"(n16) : : "memory", "scc", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127");
    return s;
}
template <int V>
__global__ void k_chain(double *out, unsigned long long *cyc, int PP, int nl)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    for (int k = threadIdx.x; k < 11 * PP + 64; k += blockDim.x) lds[k] = 1.0 + 1e-9 * k;
    __syncthreads();
    unsigned long long t0 = now();
    if (threadIdx.x < nl) {
        double s = V == 0 ? chain_b64(lds + threadIdx.x * PP, PP / 16) : chain_b128(lds + threadIdx.x * PP, PP / 16);
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
    if (threadIdx.x < 11) { double s = 0.0; for (int k = 0; k < PP; k++) s += lds[threadIdx.x * PP + k]; out[16 + threadIdx.x] = s; }
}
int main()
{
    double *d_out; unsigned long long *d_cyc, c;
    CHK(hipMalloc(&d_out, 1 << 16)); CHK(hipMalloc(&d_cyc, 64));
    auto rd = [&]() { hipDeviceSynchronize(); hipMemcpy(&c, d_cyc, 8, hipMemcpyDeviceToHost); return (double)c; };
    const int PP = 448; size_t lds = (11 * PP + 64) * 8; double h[32];
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(k_chain<0>, dim3(1), dim3(256), lds, 0, d_out, d_cyc, PP, 11); printf("interleaved b64, 11 lanes: %.2f cyc/step\n", rd() / PP);
        hipLaunchKernelGGL(k_chain<1>, dim3(1), dim3(256), lds, 0, d_out, d_cyc, PP, 11); printf("interleaved b128, 11 lanes: %.2f cyc/step\n", rd() / PP);
    }
    for (int v = 0; v < 2; v++) {
        if (v == 0) hipLaunchKernelGGL(k_chain<0>, dim3(1), dim3(256), lds, 0, d_out, d_cyc, PP, 11);
        else hipLaunchKernelGGL(k_chain<1>, dim3(1), dim3(256), lds, 0, d_out, d_cyc, PP, 11);
        hipLaunchKernelGGL(k_ref, dim3(1), dim3(256), lds, 0, d_out, PP);
        hipDeviceSynchronize(); hipMemcpy(h, d_out, sizeof h, hipMemcpyDeviceToHost);
        int bad = 0; for (int k = 0; k < 11; k++) bad += h[k] != h[16 + k];
        printf("variant %d vs plain loop: %d mismatches\n", v, bad);
    }
    return 0;
}
