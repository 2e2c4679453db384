// microbench9.hip -- f64 division with a shared, pre-refined reciprocal.
//
// hipcc expands `n / d` (f64) to v_div_scale x2, v_rcp_f64, four FMAs refining the reciprocal, a multiply,
// a residual FMA, v_div_fmas and v_div_fixup: ten dependent operations, ~100 cycles.  When neither operand
// needs scaling (both well inside the exponent range) the scale steps are identities, v_div_fmas is a plain
// FMA and v_div_fixup passes its input through, so
//     r = refine(rcp(d));  q = n * r;  e = fma(-d, q, n);  result = fma(e, r, q)
// is instruction for instruction the same arithmetic -- and the reciprocal part depends on d only, so one
// Cholesky column's quotients can share it.  This program
//   (1) checks bit-identity of that shortcut against the compiler's division on random operands whose
//       exponents lie in [-340, 340] (the range the kernel accepts before falling back to `/`), with
//       adversarial mantissas mixed in;
//   (2) reports where identity is lost outside that range (information only);
//   (3) times dependent chains of both forms on one lane of one wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>

__device__ inline unsigned long long now() { return __builtin_amdgcn_s_memtime(); }

struct Recip {
    double d, r;
};
__device__ __forceinline__ Recip make_recip(double d)
{
    double r0 = __builtin_amdgcn_rcp(d);
    double e0 = __builtin_fma(-d, r0, 1.0);
    double r1 = __builtin_fma(r0, e0, r0);
    double e1 = __builtin_fma(-d, r1, 1.0);
    double r2 = __builtin_fma(r1, e1, r1);
    return Recip{d, r2};
}
__device__ __forceinline__ double fast_div(double n, const Recip &R)
{
    double q = n * R.r;
    double e = __builtin_fma(-R.d, q, n);
    return __builtin_fma(e, R.r, q);
}

__device__ inline uint64_t mix(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// a double with a random sign, exponent uniform in [-emax, emax], and a mantissa that is random, all ones,
// all zeros, or has a short run of random low/high bits
__device__ inline double make_operand(uint64_t s, int emax)
{
    uint64_t a = mix(s), b = mix(a);
    int e = (int)(a % (uint64_t)(2 * emax + 1)) - emax;
    uint64_t m = b & 0x000fffffffffffffull;
    switch ((a >> 40) & 7) {
        case 0: m = 0; break;
        case 1: m = 0x000fffffffffffffull; break;
        case 2: m &= 0xffull; break;
        case 3: m &= 0x000ff00000000000ull; break;
        case 4: m |= 0x000ffffffff00000ull; break;
        default: break;
    }
    uint64_t u = ((a >> 63) << 63) | ((uint64_t)(e + 1023) << 52) | m;
    double v;
    memcpy(&v, &u, 8);
    return v;
}

__global__ void k_verify(unsigned long long *bad, unsigned long long *first_bad, uint64_t seed, int emax, int per_thread)
{
    uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long nb = 0;
    for (int k = 0; k < per_thread; k++) {
        uint64_t s = seed + (g * per_thread + k) * 2;
        double n = make_operand(s, emax), d = make_operand(s + 1, emax);
        if ((k & 7) == 7) n = d * (double)(1 + (k & 0xff));  // exact quotients
        double ref = n / d;
        double got = fast_div(n, make_recip(d));
        if (__double_as_longlong(ref) != __double_as_longlong(got)) {
            if (nb == 0 && atomicAdd(bad + 1, 1ull) == 0) {
                first_bad[0] = (unsigned long long)__double_as_longlong(n);
                first_bad[1] = (unsigned long long)__double_as_longlong(d);
                first_bad[2] = (unsigned long long)__double_as_longlong(ref);
                first_bad[3] = (unsigned long long)__double_as_longlong(got);
            }
            nb++;
        }
    }
    if (nb) atomicAdd(bad, nb);
}

// dependent chains on lane 0 (other lanes run the same code)
__global__ void k_chain_div(double *out, unsigned long long *cyc, double x, double d, int n)
{
    unsigned long long t0 = now();
    for (int k = 0; k < n; k++) {
#pragma unroll
        for (int u = 0; u < 8; u++) x = x / d;
        x = x * 1e8;
    }
    unsigned long long t1 = now();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_chain_fast(double *out, unsigned long long *cyc, double x, double d, int n)
{
    unsigned long long t0 = now();
    Recip R = make_recip(d);
    for (int k = 0; k < n; k++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            x = fast_div(x, R);
            asm volatile("" : "+v"(x));
        }
        x = x * 1e8;
    }
    unsigned long long t1 = now();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
// reciprocal on the chain as well: d changes every step (x = d / x form: each quotient is the next denominator)
__global__ void k_chain_recip(double *out, unsigned long long *cyc, double x, double d, int n)
{
    unsigned long long t0 = now();
    for (int k = 0; k < n; k++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            x = fast_div(d, make_recip(x));
            asm volatile("" : "+v"(x));
        }
    }
    unsigned long long t1 = now();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_chain_sqrt(double *out, unsigned long long *cyc, double x, int n)
{
    unsigned long long t0 = now();
    for (int k = 0; k < n; k++) {
#pragma unroll
        for (int u = 0; u < 8; u++) x = sqrt(x) + 3.0;
    }
    unsigned long long t1 = now();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main()
{
    unsigned long long *d_bad, *d_first, *d_cyc, h[4], nb[2], c;
    double *d_out;
    hipMalloc(&d_bad, 16);
    hipMalloc(&d_first, 32);
    hipMalloc(&d_cyc, 8);
    hipMalloc(&d_out, 8 * 64);
    const int blocks = 4096, threads = 256, per = 128;
    for (int emax : {20, 100, 340, 380, 500, 700, 1000}) {
        hipMemset(d_bad, 0, 16);
        hipLaunchKernelGGL(k_verify, dim3(blocks), dim3(threads), 0, 0, d_bad, d_first, 0x1234567ull + emax, emax, per);
        hipDeviceSynchronize();
        hipMemcpy(nb, d_bad, 16, hipMemcpyDeviceToHost);
        printf("exponents in [-%d, %d]: %llu of %llu quotients differ from n / d", emax, emax, nb[0],
               (unsigned long long)blocks * threads * per);
        if (nb[0]) {
            hipMemcpy(h, d_first, 32, hipMemcpyDeviceToHost);
            printf("   e.g. n=%016llx d=%016llx ref=%016llx got=%016llx", h[0], h[1], h[2], h[3]);
        }
        printf("\n");
    }
    auto rd = [&]() { hipDeviceSynchronize(); hipMemcpy(&c, d_cyc, 8, hipMemcpyDeviceToHost); return (double)c; };
    const int n = 200;
    hipLaunchKernelGGL(k_chain_div, dim3(1), dim3(64), 0, 0, d_out, d_cyc, 3.0, 7.0, n);
    printf("dependent x = x / d            : %.1f cycles per divide\n", rd() / (n * 8.0));
    hipLaunchKernelGGL(k_chain_fast, dim3(1), dim3(64), 0, 0, d_out, d_cyc, 3.0, 7.0, n);
    printf("dependent x = fast_div(x, R)   : %.1f cycles per divide\n", rd() / (n * 8.0));
    hipLaunchKernelGGL(k_chain_recip, dim3(1), dim3(64), 0, 0, d_out, d_cyc, 3.0, 7.0, n);
    printf("dependent x = d / x via recip  : %.1f cycles per divide\n", rd() / (n * 8.0));
    hipLaunchKernelGGL(k_chain_sqrt, dim3(1), dim3(64), 0, 0, d_out, d_cyc, 3.0, n);
    printf("dependent x = sqrt(x) + 3      : %.1f cycles per step\n", rd() / (n * 8.0));
    return 0;
}
