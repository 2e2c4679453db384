// microbench14.hip -- can an f32 MFMA carry the ordered f32 cost sum (src/patch_match.cpp:294 `cost += e*e`)?
//
// With B = 1.0 every product is exact, so an MFMA whose k-steps are sequentially rounded f32 additions would give
// acc = (((c + a0) + a1) + a2) + a3 -- four terms of the reference's sum per instruction, on the matrix pipe instead of
// the VALU.  This program finds out what v_mfma_f32_16x16x4_f32 / v_mfma_f32_32x32x2_f32 / v_mfma_f32_4x4x1_16B_f32 do:
//   1. which order / shape of additions reproduces D bit for bit (24 sequential orders, pairwise trees, a single
//      rounding of the exact sum), on operands with the dynamic range of the cost sum (a large running sum + small squares);
//   2. denormal operands and results (flushed or kept);
//   3. the latency of a dependent chain.
// hipcc --offload-arch=gfx950 -O2 -o tools/bin/microbench14 tools/microbench14.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// 16x16x4: A(i, k) in lane 16 k + i, B(k, j) in lane 16 k + j, D(i = 4 (l / 16) + r, j = l % 16) in register r of lane l
__global__ void k_16x16x4(const float *a, const float *b, const float *c, float *d)
{
    const int l = threadIdx.x;
    f32x4 cv = {c[4 * l], c[4 * l + 1], c[4 * l + 2], c[4 * l + 3]};
    f32x4 r = __builtin_amdgcn_mfma_f32_16x16x4f32(a[l], b[l], cv, 0, 0, 0);
    for (int k = 0; k < 4; k++) d[4 * l + k] = r[k];
}
// 32x32x2: A(i, k) in lane 32 k + i, B(k, j) in lane 32 k + j, D(i = 8 (r / 4) + 4 (l / 32) + r % 4, j = l % 32)
__global__ void k_32x32x2(const float *a, const float *b, const float *c, float *d)
{
    const int l = threadIdx.x;
    f32x16 cv;
    for (int k = 0; k < 16; k++) cv[k] = c[16 * l + k];
    f32x16 r = __builtin_amdgcn_mfma_f32_32x32x2f32(a[l], b[l], cv, 0, 0, 0);
    for (int k = 0; k < 16; k++) d[16 * l + k] = r[k];
}
// 4x4x1, sixteen blocks: A(b, i) in lane 4 b + i, B(b, j) in lane 4 b + j, D(b, i = r, j = l % 4) in register r of lane l
__global__ void k_4x4x1(const float *a, const float *b, const float *c, float *d)
{
    const int l = threadIdx.x;
    f32x4 cv = {c[4 * l], c[4 * l + 1], c[4 * l + 2], c[4 * l + 3]};
    f32x4 r = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], cv, 0, 0, 0);
    for (int k = 0; k < 4; k++) d[4 * l + k] = r[k];
}
template <int KIND>
__global__ void k_chain(const float *a, float *d, unsigned long long *cyc, int n)
{
    const int l = threadIdx.x;
    const float av = a[l];
    f32x4 acc = {0, 0, 0, 0};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int k = 0; k < n; k++) {
        if (KIND == 0) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, 1.0f, acc, 0, 0, 0);
        else acc = __builtin_amdgcn_mfma_f32_4x4x1f32(av, 1.0f, acc, 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    d[l] = acc[0] + acc[1] + acc[2] + acc[3];
    if (l == 0) cyc[0] = t1 - t0;
}

static float frand(float lo, float hi) { return lo + (hi - lo) * (float)((double)rand() / RAND_MAX); }
// a "square": e * e rounded to f32, |e| up to 255 with a random binary scale (the cost sum's terms)
static float square_term()
{
    float e = frand(-255.0f, 255.0f) * ldexpf(1.0f, -(rand() % 12));
    return e * e;
}

int main()
{
    float *da, *db, *dc, *dd;
    unsigned long long *dcyc;
    CHK(hipMalloc(&da, 256)); CHK(hipMalloc(&db, 256)); CHK(hipMalloc(&dc, 4096)); CHK(hipMalloc(&dd, 4096)); CHK(hipMalloc(&dcyc, 8));
    float ha[64], hb[64], hc[1024], hd[1024];
    srand(2026);
    int perm[24][4], np = 0, idx[4] = {0, 1, 2, 3};
    do { memcpy(perm[np++], idx, sizeof idx); } while (std::next_permutation(idx, idx + 4));

    // ---- 16x16x4, B = 1.0 --------------------------------------------------------------------------------------------
    {
        long total = 0, match[24] = {0}, tree = 0, tree_c_last = 0, once = 0, differ_from_once = 0;
        for (int trial = 0; trial < 4000; trial++) {
            for (int i = 0; i < 64; i++) { ha[i] = square_term(); hb[i] = 1.0f; }
            for (int i = 0; i < 256; i++) hc[i] = (trial & 1) ? square_term() * (float)(1 + rand() % 400) : frand(0.0f, 3.0e7f);
            CHK(hipMemcpy(da, ha, 256, hipMemcpyHostToDevice)); CHK(hipMemcpy(db, hb, 256, hipMemcpyHostToDevice));
            CHK(hipMemcpy(dc, hc, 1024, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(k_16x16x4, dim3(1), dim3(64), 0, 0, da, db, dc, dd);
            CHK(hipMemcpy(hd, dd, 1024, hipMemcpyDeviceToHost));
            for (int l = 0; l < 64; l++)
                for (int r = 0; r < 4; r++) {
                    const int i = 4 * (l / 16) + r;
                    const float c = hc[4 * l + r], got = hd[4 * l + r];
                    float t[4];
                    for (int k = 0; k < 4; k++) t[k] = ha[16 * k + i];
                    total++;
                    for (int p = 0; p < 24; p++) {
                        volatile float s = c;
                        for (int k = 0; k < 4; k++) s = s + t[perm[p][k]];
                        if (s == got) match[p]++;
                    }
                    volatile float p01 = t[0] + t[1], p23 = t[2] + t[3];
                    volatile float tr = p01 + p23;
                    volatile float t1 = c + tr;
                    if (t1 == got) tree_c_last++;
                    volatile float q0 = c + t[0];
                    volatile float q1 = q0 + t[1];
                    volatile float q2 = q1 + p23;
                    if (q2 == got) tree++;
                    const float ex = (float)((double)c + (double)t[0] + (double)t[1] + (double)t[2] + (double)t[3]);
                    if (ex == got) once++;
                    volatile float seq = c;
                    for (int k = 0; k < 4; k++) seq = seq + t[k];
                    if (seq != ex) differ_from_once++;
                }
        }
        printf("v_mfma_f32_16x16x4_f32, B = 1.0, %ld outputs (%ld of them tell a sequential sum from a singly rounded one):\n", total, differ_from_once);
        for (int p = 0; p < 24; p++)
            if (match[p] == total || p == 0) printf("  sequential order %d%d%d%d: %ld\n", perm[p][0], perm[p][1], perm[p][2], perm[p][3], match[p]);
        long best = 0; int bp = 0;
        for (int p = 0; p < 24; p++) if (match[p] > best) best = match[p], bp = p;
        printf("  best sequential order %d%d%d%d: %ld   c + ((t0+t1)+(t2+t3)): %ld   ((c+t0)+t1)+(t2+t3): %ld   exact sum rounded once: %ld\n",
               perm[bp][0], perm[bp][1], perm[bp][2], perm[bp][3], best, tree_c_last, tree, once);
    }
    // ---- 16x16x4, general B: fused or product rounded first? --------------------------------------------------------
    {
        long total = 0, fused = 0, unfused = 0;
        for (int trial = 0; trial < 1000; trial++) {
            for (int i = 0; i < 64; i++) { ha[i] = frand(-200, 200); hb[i] = frand(-200, 200); }
            for (int i = 0; i < 256; i++) hc[i] = frand(-1e5f, 1e5f);
            CHK(hipMemcpy(da, ha, 256, hipMemcpyHostToDevice)); CHK(hipMemcpy(db, hb, 256, hipMemcpyHostToDevice));
            CHK(hipMemcpy(dc, hc, 1024, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(k_16x16x4, dim3(1), dim3(64), 0, 0, da, db, dc, dd);
            CHK(hipMemcpy(hd, dd, 1024, hipMemcpyDeviceToHost));
            for (int l = 0; l < 64; l++)
                for (int r = 0; r < 4; r++) {
                    const int i = 4 * (l / 16) + r, j = l % 16;
                    float s = hc[4 * l + r];
                    volatile float u = hc[4 * l + r];
                    for (int k = 0; k < 4; k++) {
                        s = fmaf(ha[16 * k + i], hb[16 * k + j], s);
                        volatile float pr = ha[16 * k + i] * hb[16 * k + j];
                        u = u + pr;
                    }
                    total++;
                    if (s == hd[4 * l + r]) fused++;
                    if (u == hd[4 * l + r]) unfused++;
                }
        }
        printf("v_mfma_f32_16x16x4_f32, random B, %ld outputs: sequential fmaf chain %ld, product rounded then added %ld\n", total, fused, unfused);
    }
    // ---- 32x32x2, B = 1.0 --------------------------------------------------------------------------------------------
    {
        long total = 0, s01 = 0, s10 = 0, once = 0;
        for (int trial = 0; trial < 1000; trial++) {
            for (int i = 0; i < 64; i++) { ha[i] = square_term(); hb[i] = 1.0f; }
            for (int i = 0; i < 1024; i++) hc[i] = (trial & 1) ? square_term() * (float)(1 + rand() % 400) : frand(0.0f, 3.0e7f);
            CHK(hipMemcpy(da, ha, 256, hipMemcpyHostToDevice)); CHK(hipMemcpy(db, hb, 256, hipMemcpyHostToDevice));
            CHK(hipMemcpy(dc, hc, 4096, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(k_32x32x2, dim3(1), dim3(64), 0, 0, da, db, dc, dd);
            CHK(hipMemcpy(hd, dd, 4096, hipMemcpyDeviceToHost));
            for (int l = 0; l < 64; l++)
                for (int r = 0; r < 16; r++) {
                    const int i = 8 * (r / 4) + 4 * (l / 32) + r % 4;
                    const float c = hc[16 * l + r], got = hd[16 * l + r], t0 = ha[i], t1 = ha[32 + i];
                    volatile float a0 = c + t0; volatile float a1 = a0 + t1;
                    volatile float b0 = c + t1; volatile float b1 = b0 + t0;
                    total++;
                    if (a1 == got) s01++;
                    if (b1 == got) s10++;
                    if ((float)((double)c + (double)t0 + (double)t1) == got) once++;
                }
        }
        printf("v_mfma_f32_32x32x2_f32, B = 1.0, %ld outputs: (c+t0)+t1 %ld, (c+t1)+t0 %ld, exact sum rounded once %ld\n", total, s01, s10, once);
    }
    // ---- 4x4x1 (one k-step: a single addition), incl. denormals ------------------------------------------------------
    {
        long total = 0, ok = 0;
        for (int trial = 0; trial < 1000; trial++) {
            for (int i = 0; i < 64; i++) { ha[i] = square_term(); hb[i] = 1.0f; }
            for (int i = 0; i < 256; i++) hc[i] = (trial & 1) ? square_term() * (float)(1 + rand() % 400) : frand(0.0f, 3.0e7f);
            CHK(hipMemcpy(da, ha, 256, hipMemcpyHostToDevice)); CHK(hipMemcpy(db, hb, 256, hipMemcpyHostToDevice));
            CHK(hipMemcpy(dc, hc, 1024, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(k_4x4x1, dim3(1), dim3(64), 0, 0, da, db, dc, dd);
            CHK(hipMemcpy(hd, dd, 1024, hipMemcpyDeviceToHost));
            for (int l = 0; l < 64; l++)
                for (int r = 0; r < 4; r++) {
                    volatile float s = hc[4 * l + r] + ha[4 * (l / 4) + r];
                    total++;
                    if (s == hd[4 * l + r]) ok++;
                }
        }
        printf("v_mfma_f32_4x4x1_16B_f32, B = 1.0, %ld outputs: c + a %ld\n", total, ok);
    }
    {
        // denormals: a and c denormal (sum denormal), a normal + c = -a + denormal, and NaN / inf operands, 16x16x4 and 4x4x1
        const float dn = 1.0e-41f, mn = 1.17549435e-38f;
        for (int kind = 0; kind < 2; kind++) {
            for (int i = 0; i < 64; i++) { ha[i] = 0.0f; hb[i] = 1.0f; }
            for (int i = 0; i < 256; i++) hc[i] = 0.0f;
            // output (i = 0): 16x16x4: lane 0 reg 0 <- a lanes 0, 16, 32, 48; 4x4x1: lane 0 reg 0 <- a lane 0
            ha[0] = dn; hc[0] = 3.0e-42f;                       // denormal + denormal
            // output i = 1: lane 0 reg 1: a lanes 1 (,17, ...)
            ha[1] = mn; hc[1] = -mn + 0.0f + dn * 0.0f - 0.0f;  // min normal - min normal = 0
            hc[1] = -(mn * 1.5f);                                // -> a denormal result: mn - 1.5 mn = -0.5 mn
            ha[2] = INFINITY; hc[2] = 1.0f;
            ha[3] = NAN; hc[3] = 1.0f;
            CHK(hipMemcpy(da, ha, 256, hipMemcpyHostToDevice)); CHK(hipMemcpy(db, hb, 256, hipMemcpyHostToDevice));
            CHK(hipMemcpy(dc, hc, 1024, hipMemcpyHostToDevice));
            if (kind == 0) hipLaunchKernelGGL(k_16x16x4, dim3(1), dim3(64), 0, 0, da, db, dc, dd);
            else hipLaunchKernelGGL(k_4x4x1, dim3(1), dim3(64), 0, 0, da, db, dc, dd);
            CHK(hipMemcpy(hd, dd, 1024, hipMemcpyDeviceToHost));
            volatile float e0 = dn + 3.0e-42f, e1 = mn - mn * 1.5f;
            printf("%s special operands: denormal + denormal = %a (IEEE %a)   min-normal cancellation = %a (IEEE %a)   inf + 1 = %f   nan + 1 = %f\n",
                   kind == 0 ? "16x16x4" : "4x4x1  ", hd[0], (float)e0, hd[1], (float)e1, hd[2], hd[3]);
        }
    }
    // ---- dependent-chain latency -------------------------------------------------------------------------------------
    for (int i = 0; i < 64; i++) ha[i] = 1e-3f * (i + 1);
    CHK(hipMemcpy(da, ha, 256, hipMemcpyHostToDevice));
    for (int rep = 0; rep < 2; rep++) {
        unsigned long long c;
        hipLaunchKernelGGL(k_chain<0>, dim3(1), dim3(64), 0, 0, da, dd, dcyc, 2048);
        CHK(hipDeviceSynchronize()); CHK(hipMemcpy(&c, dcyc, 8, hipMemcpyDeviceToHost));
        printf("dependent v_mfma_f32_16x16x4_f32: %.2f cycles each   ", (double)c / 2048);
        hipLaunchKernelGGL(k_chain<1>, dim3(1), dim3(64), 0, 0, da, dd, dcyc, 2048);
        CHK(hipDeviceSynchronize()); CHK(hipMemcpy(&c, dcyc, 8, hipMemcpyDeviceToHost));
        printf("dependent v_mfma_f32_4x4x1_16B_f32: %.2f cycles each\n", (double)c / 2048);
    }
    return 0;
}
