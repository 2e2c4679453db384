// microbench10 -- what a vector-memory GATHER costs on MI355X, by address pattern and width.
// The sampling phase of the tracking kernels issues 5 dword gathers per patch pixel; this measures the
// cycles one wave-instruction occupies (throughput, many in flight) for: coalesced dwords, the patch-shaped
// gather of the kernels (64 lanes = ~3 rows of 21 neighbouring 4-byte quads), the same as one 16-byte load per
// lane, typed buffer loads, and fully random dwords -- at 1, 4 and 16 waves per CU.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/microbench10 tools/microbench10.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ f32x4 buffer_load_format_xyzw(i32x4 rsrc, int vindex, int voffset, int soffset, int aux)
    __asm("llvm.amdgcn.struct.buffer.load.format.v4f32");

constexpr int W = 752, H = 480, ITERS = 256, UNROLL = 8;

template <int MODE>
__global__ void __launch_bounds__(64) k(const uint32_t *img, const uint4 *img16, unsigned long long *cycles, float *sink,
                                        int n_elems)
{
    const int lane = threadIdx.x, wave = blockIdx.x;
    // every wave its own patch position; positions move a little every iteration (like a GN iteration)
    unsigned s = wave * 2654435761u + 12345u;
    const int x0 = 30 + (s >> 8) % (W - 80), y0 = 30 + (s >> 20) % (H - 80);
    const int px = lane % 21, py = lane / 21;
    float acc = 0.f;
    i32x4 rs;
    {
        const unsigned long long b = (unsigned long long)img;
        rs.x = (int)(unsigned)b;
        rs.y = (int)((unsigned)(b >> 32) & 0xffffu) | (4 << 16);
        rs.z = n_elems;
        rs.w = 4 | (6 << 3) | (5 << 6) | (7 << 9) | (2 << 12) | (10 << 15);
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; it += UNROLL) {
        float v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const int k = it + u;
            const int dx = (k % 5 == 1) - (k % 5 == 2), dy = (k % 5 == 3) - (k % 5 == 4);  // centre, x+-1, y+-1
            const int sh = (k / 5) & 3;                                                     // the patch drifts
            int idx;
            if (MODE == 0) idx = ((wave * 64 + k * 4096) % (n_elems - 64)) + lane;           // coalesced
            else if (MODE == 3) idx = (int)(((unsigned)(lane * 40503u + k * 9973u + s) * 2654435761u) % (unsigned)n_elems);
            else idx = (y0 + py + dy + sh) * W + x0 + px + dx + sh;                          // patch-shaped
            if (MODE == 2) {
                const uint4 q = img16[idx];
                v[u] = __uint_as_float(q.x ^ q.y ^ q.z ^ q.w);
            } else if (MODE == 4) {
                const f32x4 q = buffer_load_format_xyzw(rs, idx, 0, 0, 0);
                v[u] = q.x + q.y + q.z + q.w;
            } else {
                v[u] = __uint_as_float(img[idx]);
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; u++) acc += v[u];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cycles[wave] = t1 - t0;
    if (acc == 12345.678f) sink[0] = acc;
}

template <int MODE>
void run(const char *name, const uint32_t *img, const uint4 *img16, unsigned long long *d_cyc, float *sink, int n_elems)
{
    for (int wpc : {1, 4, 16}) {
        const int waves = 256 * wpc;
        for (int rep = 0; rep < 2; rep++) {
            hipLaunchKernelGGL(k<MODE>, dim3(waves), dim3(64), 0, 0, img, img16, d_cyc, sink, n_elems);
            hipDeviceSynchronize();
        }
        hipEvent_t e0, e1;
        hipEventCreate(&e0), hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(waves), dim3(64), 0, 0, img, img16, d_cyc, sink, n_elems);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> c(waves);
        hipMemcpy(c.data(), d_cyc, waves * 8, hipMemcpyDeviceToHost);
        double sum = 0;
        for (auto v : c) sum += (double)v;
        const double per_wave = sum / waves / ITERS;  // cycles a wave needs per load instruction (latency-overlapped)
        // CU-level throughput: a CU issued wpc * ITERS instructions in (kernel time) ~ per_wave * ITERS cycles
        printf("%-34s %2d waves/CU: %7.1f cycles per load per wave  => one load per %6.1f cycles per CU   (kernel %.1f us)\n",
               name, wpc, per_wave, per_wave / wpc, ms * 1e3);
    }
}

int main()
{
    const int n_elems = W * H;
    uint32_t *img;
    uint4 *img16;
    unsigned long long *d_cyc;
    float *sink;
    hipMalloc(&img, n_elems * 4);
    hipMalloc(&img16, (size_t)n_elems * 16);
    hipMalloc(&d_cyc, 256 * 16 * 8);
    hipMalloc(&sink, 4);
    hipMemset(img, 1, n_elems * 4);
    hipMemset(img16, 1, (size_t)n_elems * 16);
    run<0>("coalesced dword", img, img16, d_cyc, sink, n_elems);
    run<1>("patch gather, dword (4 B/px)", img, img16, d_cyc, sink, n_elems);
    run<2>("patch gather, dwordx4 (16 B/px)", img, img16, d_cyc, sink, n_elems);
    run<4>("patch gather, typed xyzw (4 B/px)", img, img16, d_cyc, sink, n_elems);
    run<3>("random dword", img, img16, d_cyc, sink, n_elems);
    return 0;
}
