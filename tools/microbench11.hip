// microbench11 -- how long a wave waits for a BATCH of patch-shaped gathers, by batch size and load width.
// (microbench10 gave the throughput; this is the latency side: a lone tracking workgroup issues 10 dword gathers per
// lane-pair of pixels and then needs all of them.)  One wave per CU and four waves per CU (one per SIMD).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/microbench11 tools/microbench11.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int W = 752, H = 480, ITERS = 200;
struct u3 { unsigned x, y, z; };

template <int BATCH, int WIDTH>
__global__ void __launch_bounds__(64) k(const unsigned *img, unsigned long long *cycles, float *sink, int n_elems)
{
    const int lane = threadIdx.x, wave = blockIdx.x;
    unsigned s = wave * 2654435761u + 12345u;
    const int x0 = 30 + (s >> 8) % (W - 80), y0 = 30 + (s >> 20) % (H - 80);
    const int px = lane % 21, py = lane / 21;
    float acc = 0.f;
    unsigned long long total = 0;
    for (int it = 0; it < ITERS; it++) {
        const int sh = it & 3;
        unsigned v[BATCH];
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int u = 0; u < BATCH; u++) {
            const int dx = (u % 5 == 1) - (u % 5 == 2), dy = (u % 5 == 3) - (u % 5 == 4) + (u / 5) * 3;
            const size_t idx = (size_t)((y0 + py + dy + sh) * W + x0 + px + dx + sh) * WIDTH;
            if (WIDTH == 1) v[u] = img[idx];
            else if (WIDTH == 3) { u3 q = *reinterpret_cast<const u3 *>(img + idx); v[u] = q.x ^ q.y ^ q.z; }
            else { uint4 q = *reinterpret_cast<const uint4 *>(img + idx); v[u] = q.x ^ q.y ^ q.z ^ q.w; }
        }
#pragma unroll
        for (int u = 0; u < BATCH; u++) acc += __uint_as_float(v[u]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        total += __builtin_amdgcn_s_memtime() - t0;
        asm volatile("" : "+v"(acc));
    }
    if (lane == 0) cycles[wave] = total;
    if (acc == 12345.678f) sink[0] = acc;
}

template <int BATCH, int WIDTH>
void run(const unsigned *img, unsigned long long *d_cyc, float *sink, int n)
{
    for (int wpc : {1, 4}) {
        const int waves = 256 * wpc;
        for (int rep = 0; rep < 3; rep++) {
            hipLaunchKernelGGL((k<BATCH, WIDTH>), dim3(waves), dim3(64), 0, 0, img, d_cyc, sink, n);
            hipDeviceSynchronize();
        }
        std::vector<unsigned long long> c(waves);
        hipMemcpy(c.data(), d_cyc, waves * 8, hipMemcpyDeviceToHost);
        double sum = 0;
        for (auto v : c) sum += (double)v;
        printf("batch of %2d gathers, %2d B per lane, %d wave(s)/CU: %7.1f cycles per batch (%6.1f per gather)\n", BATCH, 4 * WIDTH, wpc,
               sum / waves / ITERS, sum / waves / ITERS / BATCH);
    }
}

int main()
{
    const int n = W * (H + 64);
    unsigned *img;
    unsigned long long *d_cyc;
    float *sink;
    hipMalloc(&img, (size_t)n * 16);
    hipMalloc(&d_cyc, 1024 * 8);
    hipMalloc(&sink, 4);
    hipMemset(img, 1, (size_t)n * 16);
    run<1, 1>(img, d_cyc, sink, n);
    run<2, 1>(img, d_cyc, sink, n);
    run<5, 1>(img, d_cyc, sink, n);
    run<6, 1>(img, d_cyc, sink, n);
    run<10, 1>(img, d_cyc, sink, n);
    run<1, 3>(img, d_cyc, sink, n);
    run<2, 3>(img, d_cyc, sink, n);
    run<1, 4>(img, d_cyc, sink, n);
    run<2, 4>(img, d_cyc, sink, n);
    return 0;
}
