// microbench13 -- where the hardware puts the waves of a launch shaped like k_track_block at BASELINE configs[1]
// (1000 workgroups x 256 threads, 31 KB of LDS, 128 VGPRs): XCC / SE / CU / SIMD / TG_ID of every wave, read from
// HW_REG_HW_ID and HW_REG_XCC_ID.  Answers: does wave w of a workgroup always sit on SIMD w (then the role "wave 0
// runs the solve and a chain" loads SIMD 0 of every CU), and does TG_ID distinguish the workgroups of one CU?
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/microbench13 tools/microbench13.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>

__global__ void __launch_bounds__(256, 4) k_where(unsigned *out, int spin)
{
    extern __shared__ unsigned char lds[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // stay resident long enough that every workgroup of the launch is placed while the others still run
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin) __builtin_amdgcn_s_sleep(8);
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
        out[2 * w] = hw;
        out[2 * w + 1] = xcc;
    }
    if (spin < 0) lds[threadIdx.x] = 0;
}

int main()
{
    const int n = 1000;
    unsigned *d;
    hipMalloc(&d, n * 4 * 2 * sizeof(unsigned));
    hipMemset(d, 0xff, n * 4 * 2 * sizeof(unsigned));
    hipLaunchKernelGGL(k_where, dim3(n), dim3(256), 31 * 1024, 0, d, 2000);  // 20 us at 100 MHz
    hipDeviceSynchronize();
    std::vector<unsigned> h(n * 4 * 2);
    hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
    // HW_ID (gfx9): wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13] tg_id[19:16]
    int simd_eq_wave = 0, total = 0;
    std::map<unsigned, std::vector<int>> per_cu;       // (xcc, se, sh, cu) -> blocks
    std::map<unsigned, std::map<int, int>> simd_load;  // cu key -> simd -> waves
    int hist[4][4] = {};
    for (int b = 0; b < n; b++) {
        for (int w = 0; w < 4; w++) {
            const unsigned hw = h[2 * (b * 4 + w)], xcc = h[2 * (b * 4 + w) + 1] & 0xf;
            const int simd = (hw >> 4) & 3, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            const unsigned key = (xcc << 12) | (se << 8) | (sh << 4) | cu;
            hist[w][simd]++;
            simd_eq_wave += simd == w;
            total++;
            simd_load[key][simd]++;
            if (w == 0) per_cu[key].push_back(b);
        }
    }
    printf("waves with simd_id == wave index: %d of %d\n", simd_eq_wave, total);
    for (int w = 0; w < 4; w++) printf("wave %d on simd 0..3: %d %d %d %d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    printf("CUs used: %zu\n", per_cu.size());
    std::map<int, int> wg_per_cu;
    for (auto &kv : per_cu) wg_per_cu[(int)kv.second.size()]++;
    for (auto &kv : wg_per_cu) printf("  CUs holding %d workgroups: %d\n", kv.first, kv.second);
    int shown = 0;
    for (auto &kv : per_cu) {
        if (shown++ >= 12) break;
        printf("cu key %05x:", kv.first);
        for (int b : kv.second) {
            printf("  [block %d:", b);
            for (int w = 0; w < 4; w++) {
                const unsigned hw = h[2 * (b * 4 + w)];
                printf(" s%d/w%d/tg%d", (hw >> 4) & 3, hw & 15, (hw >> 16) & 15);
            }
            printf("]");
        }
        printf("\n");
    }
    // distinctness of tg_id & 3 among the workgroups of one CU
    int cus_distinct = 0;
    for (auto &kv : per_cu) {
        int seen = 0, ok = 1;
        for (int b : kv.second) {
            const int tg = (h[2 * (b * 4)] >> 16) & 3;
            if (seen & (1 << tg)) ok = 0;
            seen |= 1 << tg;
        }
        cus_distinct += ok;
    }
    printf("CUs whose workgroups have pairwise distinct (tg_id & 3): %d of %zu\n", cus_distinct, per_cu.size());
    hipFree(d);
    return 0;
}
