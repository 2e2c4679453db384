// microbench7.hip -- is v_mfma_f64_4x4x4f64 a sequential FMA chain over k?  And its lane layout.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_mfma(const double *a, const double *b, const double *c, double *d)
{
    int l = threadIdx.x;
    d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], c[l], 0, 0, 0);
}
// chain of n dependent MFMAs (latency)
__global__ void k_chain(const double *a, const double *b, double *d, unsigned long long *cyc, int n)
{
    int l = threadIdx.x;
    double av = a[l], bv = b[l], acc = 0.0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int k = 0; k < n; k++) acc = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, acc, 0, 0, 0);
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    d[l] = acc;
    if (l == 0) cyc[0] = t1 - t0;
}

int main()
{
    double *da, *db, *dc, *dd;
    unsigned long long *dcyc;
    CHK(hipMalloc(&da, 512)); CHK(hipMalloc(&db, 512)); CHK(hipMalloc(&dc, 512)); CHK(hipMalloc(&dd, 512)); CHK(hipMalloc(&dcyc, 8));
    double ha[64], hb[64], hc[64], hd[64];
    // 1. layout: one-hot A lane la, one-hot B lane lb -> which D lanes light up
    int mapA[64][3], mapB[64][3];  // lane -> (block, row/col, k)
    memset(mapA, -1, sizeof mapA); memset(mapB, -1, sizeof mapB);
    std::vector<int> hits(64 * 64 * 64, 0);
    for (int la = 0; la < 64; la++)
        for (int lb = 0; lb < 64; lb++) {
            for (int i = 0; i < 64; i++) ha[i] = hb[i] = hc[i] = 0;
            ha[la] = 1; hb[lb] = 1;
            hipMemcpy(da, ha, 512, hipMemcpyHostToDevice); hipMemcpy(db, hb, 512, hipMemcpyHostToDevice); hipMemcpy(dc, hc, 512, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, da, db, dc, dd);
            hipMemcpy(hd, dd, 512, hipMemcpyDeviceToHost);
            for (int i = 0; i < 64; i++) if (hd[i] != 0) hits[(la * 64 + lb) * 64 + i] = 1;
        }
    // print for each D lane the list of (la, lb) pairs
    printf("D lane -> contributing (A lane, B lane) pairs:\n");
    for (int i = 0; i < 64; i++) {
        printf("D[%2d]:", i);
        for (int la = 0; la < 64; la++) for (int lb = 0; lb < 64; lb++) if (hits[(la * 64 + lb) * 64 + i]) printf(" (%d,%d)", la, lb);
        printf("\n");
    }
    // 2. exactness: random f32-valued doubles; compare each D with fma chains over its 4 contributing pairs in
    //    every one of the 24 orders
    srand(12345);
    auto rnd = []() { return (double)(float)(((double)rand() / RAND_MAX - 0.5) * 200.0); };
    int perm[24][4], np = 0;
    int idx[4] = {0, 1, 2, 3};
    do { memcpy(perm[np++], idx, sizeof idx); } while (std::next_permutation(idx, idx + 4));
    long match[24] = {0}, total = 0; long match_tree = 0;
    for (int trial = 0; trial < 2000; trial++) {
        for (int i = 0; i < 64; i++) { ha[i] = rnd(); hb[i] = rnd(); hc[i] = rnd() * 1000.0 * rnd(); }
        hipMemcpy(da, ha, 512, hipMemcpyHostToDevice); hipMemcpy(db, hb, 512, hipMemcpyHostToDevice); hipMemcpy(dc, hc, 512, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, da, db, dc, dd);
        hipMemcpy(hd, dd, 512, hipMemcpyDeviceToHost);
        for (int i = 0; i < 64; i++) {
            int pa[4], pb[4], n = 0;
            for (int la = 0; la < 64 && n < 4; la++) for (int lb = 0; lb < 64 && n < 4; lb++) if (hits[(la * 64 + lb) * 64 + i]) { pa[n] = la; pb[n] = lb; n++; }
            if (n != 4) continue;
            total++;
            for (int p = 0; p < 24; p++) {
                double s = hc[i];
                for (int k = 0; k < 4; k++) s = fma(ha[pa[perm[p][k]]], hb[pb[perm[p][k]]], s);
                if (s == hd[i]) match[p]++;
            }
            double t = (ha[pa[0]] * hb[pb[0]] + ha[pa[1]] * hb[pb[1]]) + (ha[pa[2]] * hb[pb[2]] + ha[pa[3]] * hb[pb[3]]) + hc[i];
            if (t == hd[i]) match_tree++;
        }
    }
    printf("exactness over %ld outputs: ", total);
    for (int p = 0; p < 24; p++) printf("order %d%d%d%d: %ld  ", perm[p][0], perm[p][1], perm[p][2], perm[p][3], match[p]);
    printf("\n tree form: %ld\n", match_tree);
    // 3. dependent-chain latency
    for (int i = 0; i < 64; i++) { ha[i] = 1e-3 * (i + 1); hb[i] = 1e-3; }
    hipMemcpy(da, ha, 512, hipMemcpyHostToDevice); hipMemcpy(db, hb, 512, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, 0, da, db, dd, dcyc, 2048);
        unsigned long long c; hipDeviceSynchronize(); hipMemcpy(&c, dcyc, 8, hipMemcpyDeviceToHost);
        printf("dependent v_mfma_f64_4x4x4f64: %.2f cycles each (= %.2f per k-step)\n", (double)c / 2048, (double)c / 2048 / 4);
    }
    return 0;
}
