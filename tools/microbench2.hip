// microbench2.hip -- chain-loop forms: goal ~6.5 cyc/step (dependent v_add_f64 latency)
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__device__ inline unsigned long long now() { return __builtin_amdgcn_s_memtime(); }

// ping-pong, G double2 loads (2G steps) per group
template <int G>
__device__ __forceinline__ double chain_pp(const double *src, int ngroups2)
{
    const double2 *s2 = reinterpret_cast<const double2 *>(src);
    double2 a[G], b[G];
#pragma unroll
    for (int j = 0; j < G; j++) a[j] = s2[j];
    double s = 0.0;
    for (int g = 0; g < ngroups2; g++) {
#pragma unroll
        for (int j = 0; j < G; j++) b[j] = s2[G + j];
#pragma unroll
        for (int j = 0; j < G; j++) { s += a[j].x; s += a[j].y; }
        s2 += 2 * G;
#pragma unroll
        for (int j = 0; j < G; j++) a[j] = s2[j];
#pragma unroll
        for (int j = 0; j < G; j++) { s += b[j].x; s += b[j].y; }
    }
    return s;
}
template <int G>
__device__ __forceinline__ float chain_pp_f32(const float *src, int ngroups2)
{
    const float4 *s4 = reinterpret_cast<const float4 *>(src);
    float4 a[G], b[G];
#pragma unroll
    for (int j = 0; j < G; j++) a[j] = s4[j];
    float s = 0.0f;
    for (int g = 0; g < ngroups2; g++) {
#pragma unroll
        for (int j = 0; j < G; j++) b[j] = s4[G + j];
#pragma unroll
        for (int j = 0; j < G; j++) { s += a[j].x; s += a[j].y; s += a[j].z; s += a[j].w; }
        s4 += 2 * G;
#pragma unroll
        for (int j = 0; j < G; j++) a[j] = s4[j];
#pragma unroll
        for (int j = 0; j < G; j++) { s += b[j].x; s += b[j].y; s += b[j].z; s += b[j].w; }
    }
    return s;
}

template <int G>
__global__ void k_chain(double *out, unsigned long long *cyc, int PP)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    for (int k = threadIdx.x; k < 11 * PP + 64; k += blockDim.x) lds[k] = 1.0 + 1e-9 * k;
    __syncthreads();
    unsigned long long t0 = now();
    if (threadIdx.x < 11) {
        double s = chain_pp<G>(lds + threadIdx.x * PP, PP / (4 * G));
        out[threadIdx.x] = s;
    }
    unsigned long long t1 = now();
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int G>
__global__ void k_chain32(float *out, unsigned long long *cyc, int PP)
{
    extern __shared__ __attribute__((aligned(16))) float ldsf[];
    for (int k = threadIdx.x; k < PP + 64; k += blockDim.x) ldsf[k] = 1.0f + 1e-6f * k;
    __syncthreads();
    unsigned long long t0 = now();
    if (threadIdx.x < 1) {
        float s = chain_pp_f32<G>(ldsf, PP / (8 * G));
        out[threadIdx.x] = s;
    }
    unsigned long long t1 = now();
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main()
{
    double *d_out; float *f_out; unsigned long long *d_cyc, c;
    CHK(hipMalloc(&d_out, 1 << 16)); CHK(hipMalloc(&f_out, 1 << 16)); CHK(hipMalloc(&d_cyc, 64));
    auto rd = [&]() { hipDeviceSynchronize(); hipMemcpy(&c, d_cyc, 8, hipMemcpyDeviceToHost); return (double)c; };
    const int PP = 448;
    size_t lds = (11 * PP + 64) * 8;
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k_chain<2>, dim3(1), dim3(256), lds, 0, d_out, d_cyc, PP); printf("f64 pp G=2 (4 steps/grp): %.2f cyc/step (%.0f)\n", rd() / PP, (double)c);
        hipLaunchKernelGGL(k_chain<4>, dim3(1), dim3(256), lds, 0, d_out, d_cyc, PP); printf("f64 pp G=4 (8 steps/grp): %.2f cyc/step (%.0f)\n", rd() / PP, (double)c);
        hipLaunchKernelGGL(k_chain<8>, dim3(1), dim3(256), lds, 0, d_out, d_cyc, PP); printf("f64 pp G=8 (16 steps/grp): %.2f cyc/step (%.0f)\n", rd() / PP, (double)c);
        hipLaunchKernelGGL(k_chain32<2>, dim3(1), dim3(256), lds, 0, f_out, d_cyc, PP); printf("f32 pp G=2 (8 steps/grp): %.2f cyc/step (%.0f)\n", rd() / PP, (double)c);
        hipLaunchKernelGGL(k_chain32<4>, dim3(1), dim3(256), lds, 0, f_out, d_cyc, PP); printf("f32 pp G=4 (16 steps/grp): %.2f cyc/step (%.0f)\n", rd() / PP, (double)c);
    }
    return 0;
}
