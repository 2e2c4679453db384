// microbench.hip -- latencies that bound the ordered-accumulation chain on gfx950.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/microbench.hip -o /tmp/microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ inline unsigned long long now() { return __builtin_amdgcn_s_memtime(); }

// N dependent v_add_f64 in one lane
__global__ void k_dep_add_f64(double *out, unsigned long long *cyc, double inc, int n)
{
    double s = out[0];
    unsigned long long t0 = now();
#pragma unroll 16
    for (int k = 0; k < n; k++) s += inc;
    unsigned long long t1 = now();
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void k_dep_add_f32(float *out, unsigned long long *cyc, float inc, int n)
{
    float s = out[0];
    unsigned long long t0 = now();
#pragma unroll 16
    for (int k = 0; k < n; k++) s += inc;
    unsigned long long t1 = now();
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// dependent chain of f64 divides / sqrt
__global__ void k_dep_div_f64(double *out, unsigned long long *cyc, double d, int n)
{
    double s = out[0];
    unsigned long long t0 = now();
    for (int k = 0; k < n; k++) s = s / d;
    unsigned long long t1 = now();
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void k_dep_sqrt_f64(double *out, unsigned long long *cyc, double d, int n)
{
    double s = out[0];
    unsigned long long t0 = now();
    for (int k = 0; k < n; k++) s = sqrt(s + d);
    unsigned long long t1 = now();
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// LDS chain: lane c walks its own array of doubles, depth-D register prefetch
template <int D>
__global__ void k_lds_chain(double *out, unsigned long long *cyc, int n)
{
    extern __shared__ double lds[];
    for (int k = threadIdx.x; k < 11 * n; k += blockDim.x) lds[k] = 1.0 + 1e-9 * k;
    __syncthreads();
    unsigned long long t0 = now();
    if (threadIdx.x < 11) {
        const double *src = lds + threadIdx.x * n;
        double s = 0.0;
        double buf[D];
#pragma unroll
        for (int j = 0; j < D; j++) buf[j] = src[j];
        int k = 0;
        for (; k + 2 * D <= n; k += D) {
            double nxt[D];
#pragma unroll
            for (int j = 0; j < D; j++) nxt[j] = src[k + D + j];
#pragma unroll
            for (int j = 0; j < D; j++) s += buf[j];
#pragma unroll
            for (int j = 0; j < D; j++) buf[j] = nxt[j];
        }
#pragma unroll
        for (int j = 0; j < D; j++) s += buf[j];
        k += D;
        for (; k < n; k++) s += src[k];
        out[threadIdx.x] = s;
    }
    unsigned long long t1 = now();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// naive (compiler-scheduled) version, as in the first kernel
__global__ void k_lds_chain_naive(double *out, unsigned long long *cyc, int n)
{
    extern __shared__ double lds[];
    for (int k = threadIdx.x; k < 11 * n; k += blockDim.x) lds[k] = 1.0 + 1e-9 * k;
    __syncthreads();
    unsigned long long t0 = now();
    if (threadIdx.x < 11) {
        const double *src = lds + threadIdx.x * n;
        double s = 0.0;
#pragma unroll 8
        for (int k = 0; k < n; k++) s += src[k];
        out[threadIdx.x] = s;
    }
    unsigned long long t1 = now();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// occupancy probe: blocks of 256 threads with a given dynamic LDS size spin ~fixed work
__global__ void __launch_bounds__(256) k_occ(double *out, int n)
{
    extern __shared__ double lds[];
    double s = threadIdx.x;
    for (int k = 0; k < n; k++) s = s * 1.0000001 + 1e-9;
    if (s == 12345.0) lds[threadIdx.x] = s;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main()
{
    double *d_out;
    float *f_out;
    unsigned long long *d_cyc;
    CHK(hipMalloc(&d_out, 1 << 24));
    CHK(hipMalloc(&f_out, 1 << 16));
    CHK(hipMalloc(&d_cyc, 1 << 16));
    CHK(hipMemset(d_out, 0, 1 << 24));
    unsigned long long c;
    const int N = 4096;
    auto rd = [&]() { hipDeviceSynchronize(); hipMemcpy(&c, d_cyc, 8, hipMemcpyDeviceToHost); return (double)c; };
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k_dep_add_f64, dim3(1), dim3(64), 0, 0, d_out, d_cyc, 1e-3, N);
        printf("dep v_add_f64 (1 wave): %.2f cyc/op\n", rd() / N);
        hipLaunchKernelGGL(k_dep_add_f64, dim3(1), dim3(256), 0, 0, d_out, d_cyc, 1e-3, N);
        printf("dep v_add_f64 (4 waves/CU): %.2f cyc/op\n", rd() / N);
        hipLaunchKernelGGL(k_dep_add_f32, dim3(1), dim3(64), 0, 0, f_out, d_cyc, 1e-3f, N);
        printf("dep v_add_f32 (1 wave): %.2f cyc/op\n", rd() / N);
        hipLaunchKernelGGL(k_dep_div_f64, dim3(1), dim3(64), 0, 0, d_out, d_cyc, 1.0000001, 512);
        printf("dep f64 div: %.1f cyc/op\n", rd() / 512);
        hipLaunchKernelGGL(k_dep_sqrt_f64, dim3(1), dim3(64), 0, 0, d_out, d_cyc, 1.5, 512);
        printf("dep f64 sqrt(+add): %.1f cyc/op\n", rd() / 512);
        int n = 441;
        size_t lds = 11 * n * 8;
        hipLaunchKernelGGL(k_lds_chain_naive, dim3(1), dim3(256), lds, 0, d_out, d_cyc, n);
        printf("lds chain naive unroll8: %.2f cyc/step (%.0f total)\n", rd() / n, (double)c);
        hipLaunchKernelGGL(k_lds_chain<4>, dim3(1), dim3(256), lds, 0, d_out, d_cyc, n);
        printf("lds chain prefetch D=4: %.2f cyc/step\n", rd() / n);
        hipLaunchKernelGGL(k_lds_chain<8>, dim3(1), dim3(256), lds, 0, d_out, d_cyc, n);
        printf("lds chain prefetch D=8: %.2f cyc/step\n", rd() / n);
        hipLaunchKernelGGL(k_lds_chain<16>, dim3(1), dim3(256), lds, 0, d_out, d_cyc, n);
        printf("lds chain prefetch D=16: %.2f cyc/step\n", rd() / n);
    }
    // occupancy step: time vs number of blocks for LDS = 40.6 KB and 20 KB
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (size_t lds : {(size_t)40668, (size_t)32768, (size_t)20480, (size_t)1024}) {
        hipFuncSetAttribute((const void *)k_occ, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        printf("occupancy probe, %zu B LDS per 256-thread block:", lds);
        for (int blocks : {256, 512, 768, 1024, 1280, 1536, 2048, 4096}) {
            hipLaunchKernelGGL(k_occ, dim3(blocks), dim3(256), lds, 0, d_out, 20000);
            hipDeviceSynchronize();
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k_occ, dim3(blocks), dim3(256), lds, 0, d_out, 20000);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            printf(" %d:%.0fus", blocks, ms * 1e3);
        }
        printf("\n");
    }
    // launch overhead: back-to-back empty-ish kernels
    hipLaunchKernelGGL(k_occ, dim3(1), dim3(256), 1024, 0, d_out, 1);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int k = 0; k < 100; k++) hipLaunchKernelGGL(k_occ, dim3(64), dim3(256), 1024, 0, d_out, 1);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("100 back-to-back tiny launches: %.1f us each\n", ms * 10);
    return 0;
}
