// microbench16 -- what a ds_read_b64 costs by address pattern (round 4): the MFMA operand reads of k_track_quad.
// One wave issues 16 x 256 independent ds_read_b64 with per-lane byte addresses taken from the host; throughput in
// cycles per instruction (2.0 = conflict-free by MI355X_MICROARCH.md's table).  Patterns: linear, the quad kernel's A / B
// operand patterns with `ones` at byte 0 / 160 of the 256-B span, and probes of the grouping rule (which lanes conflict).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/microbench16 tools/microbench16.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

__global__ void __launch_bounds__(64) k_read(const int *addr, unsigned long long *out)
{
    __shared__ __attribute__((aligned(256))) double lds[2048];
    for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = (double)i;
    __syncthreads();
    const unsigned a = (unsigned)(size_t)lds + (unsigned)addr[threadIdx.x];
    double v0, v1, v2, v3, v4, v5, v6, v7, acc = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int k = 0; k < 256; k++) {
        asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:32\n ds_read_b64 %2, %8 offset:64\n ds_read_b64 %3, %8 offset:96\n"
                     "ds_read_b64 %4, %8 offset:128\n ds_read_b64 %5, %8 offset:160\n ds_read_b64 %6, %8 offset:192\n ds_read_b64 %7, %8 offset:224\n"
                     "s_waitcnt lgkmcnt(0)\n"
                     : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7)
                     : "v"(a));
        acc += v0 + v7;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    if (acc == 1.2345) out[1] = 1;
}

int main()
{
    int *d_addr;
    unsigned long long *d_out;
    hipMalloc(&d_addr, 64 * sizeof(int));
    hipMalloc(&d_out, 16);
    auto run = [&](const std::string &name, std::vector<int> ad) {
        hipMemcpy(d_addr, ad.data(), 64 * sizeof(int), hipMemcpyHostToDevice);
        unsigned long long best = ~0ull;
        for (int r = 0; r < 5; r++) {
            hipLaunchKernelGGL(k_read, dim3(1), dim3(64), 0, 0, d_addr, d_out);
            unsigned long long o[2];
            hipMemcpy(o, d_out, 16, hipMemcpyDeviceToHost);
            if (o[0] < best) best = o[0];
        }
        printf("%-58s %6.2f cycles per ds_read_b64\n", name.c_str(), (double)best / (256.0 * 8.0));
    };
    auto lanes = [&](auto f) {
        std::vector<int> v(64);
        for (int l = 0; l < 64; l++) v[l] = f(l);
        return v;
    };
    run("linear 8 * lane", lanes([](int l) { return 8 * l; }));
    run("all lanes one address (broadcast)", lanes([](int) { return 0; }));
    run("lanes l and l+32 same bank (256 apart)", lanes([](int l) { return 8 * (l & 31) + 256 * (l >> 5) + 2048 * 0; }));
    run("lanes l and l+16 same bank (256 apart)", lanes([](int l) { return 8 * (l & 15) + 256 * ((l >> 4) & 1) + 128 * (l >> 5); }));
    run("lanes l and l+16 128 B apart", lanes([](int l) { return 8 * (l & 15) + 128 * ((l >> 4) & 1) + 2048 * (l >> 5); }));
    run("lanes l and l+1 same bank (pairs, 256 apart)", lanes([](int l) { return 8 * (l >> 1) + 256 * (l & 1) + 2048; }));
    run("lanes l and l+1 128 B apart", lanes([](int l) { return 8 * (l >> 1) + 128 * (l & 1) + 4096; }));
    // the quad kernel's operand patterns: lane = 16 mk + 4 mq + mi; streams 64 mi + 16 mq + 8 mk, cconst at 192 + 16 mq + 8 mk,
    // ones at O + 8 mk (struct offsets: chunk stream stride 2112, feature stride 528, cconst at 6336 + 272 mq, ones at base)
    auto quad = [&](bool isA, int ones_at) {
        return lanes([=](int l) {
            const int mk = l >> 4, mq = (l >> 2) & 3, mi = l & 3;
            const int stream = 2112 * mi + 528 * mq + 8 * mk, cc = 6336 + 272 * mq + 8 * mk, one = ones_at + 8 * mk;
            if (isA) return mi < 2 ? stream : (mi == 2 ? cc : one);
            return mi < 3 ? stream : cc;
        });
    };
    run("quad A operand, ones at 7424 (= 0 mod 256; round 3)", quad(true, 7424));
    run("quad A operand, ones at 7584 (= 160 mod 256)", quad(true, 7584));
    run("quad A operand, ones at 7552 (= 128 mod 256)", quad(true, 7552));
    run("quad B operand", quad(false, 0));
    // the same with the two middle bits of the lane index exchanged (is the grouping really 0-31 / 32-63?)
    run("quad B operand, lanes permuted (mk <-> mq)", lanes([&](int l) {
            const int mq = l >> 4, mk = (l >> 2) & 3, mi = l & 3;
            const int stream = 2112 * mi + 528 * mq + 8 * mk, cc = 6336 + 272 * mq + 8 * mk;
            return mi < 3 ? stream : cc;
        }));
    for (int stride : {8, 16, 24, 32, 40, 64, 72, 128, 136, 264, 520, 528}) {
        char nm[64];
        snprintf(nm, sizeof nm, "stride %d bytes per lane", stride);
        run(nm, lanes([=](int l) { return (stride * l) % 8192; }));
    }
    return 0;
}
