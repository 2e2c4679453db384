// pagk_neighbor_kernel.h -- NCC nearest-neighbour search, SURVEY.md section 8 row f3:
// GyroAidedTracker::FindAndSortNearNeighbor (reference src/gyro_aided_tracker.cpp:788-851) over the FREE
// functions GetPixelValue (include/utils.h:32-46) and NCC (src/utils.cpp:110-148).
//
// One 256-thread workgroup per reference keypoint i (the reference's cv::parallel_for_ unit, :912):
//   1. the reference patch, (2h+1)^2 samples in the reference's x-outer / y-inner order, and its mean;
//   2. the candidates: every current keypoint j whose undistorted position lies within the search square
//      around the predicted point (:813-815), compacted IN INDEX ORDER (the order the reference meets them in
//      decides how equal scores are ranked);
//   3. per candidate the current patch (warped by the feature's 2x2 affine A), its mean, and the three sums
//      of the zero-mean correlation -- every float sum as an ordered DPP row chain (pagk_chain_asm.h), i.e. in
//      the reference's summation order; sum (v_ref - mean_ref)^2 does not depend on the candidate and is
//      chained once;
//   4. one lane replays the two-stack insertion of :825-842 and writes the list best first.
//
// The free sampler is NOT PatchMatch::GetPixelValue: its upper clamp is `x > cols` (x == cols passes and
// addresses the first bytes of the next row), and the interpolation is the un-factored four-term sum.
#pragma once
#include "pagk_chain_asm.h"
#include "pagk_device.h"

namespace pagk {

struct NeighborArgs {
    DevLevel ref0, cur0;   // level-0 quad images of the two frames
    int pad_ref, pad_cur;  // min(step - cols, 2) of the level-0 source: what lies behind a row's last pixel
    int half, n, m, cap, level, use_ncc;
    int pairs;             // 1: candidate list of feature i is {i} (free NCC of n point pairs, src/utils.cpp:166-200)
    float radius;          // level * mRadiusForFindNearNeighbor (:811)
    const float *keys_ref, *pt_pred, *affine;
    const uint8_t *status;
    const float *keys_cur, *keys_cur_un;
    int *count;
    int *nbr_idx;
    float *nbr_dist, *nbr_ncc;
};

// GetPixelValue of include/utils.h:32-46 on a quad image.  `pad`: bytes between the end of a row and the next
// row in the SOURCE image (0 = continuous: linear addressing runs into the next row; 1 = one padding byte,
// then the next row; >= 2 = padding only; padding bytes are defined as 0, like every byte past the buffer).
__device__ __forceinline__ float sample_free(const DevLevel &L, int pad, float x, float y)
{
    x = fmaxf(x, 0.0f);  // `if (x < 0) x = 0;` (:35) -- a NaN coordinate is mapped to 0 (undefined in the reference)
    y = fmaxf(y, 0.0f);
    x = (x > L.fcols) ? L.fcols_m1 : x;  // `if (x > img.cols) x = img.cols - 1;` (:37)
    y = (y > L.frows) ? L.frows_m1 : y;
    const int ix = (int)x, iy = (int)y;  // 0 <= ix <= cols, 0 <= iy <= rows
    const float xx = __builtin_amdgcn_fractf(x), yy = __builtin_amdgcn_fractf(y);  // x - floor(x), x >= 0 (:41-42)
    const int npx = L.cols * L.rows;
    float d0 = 0.0f, d1 = 0.0f, d2 = 0.0f, d3 = 0.0f;
    if (ix < L.cols || pad == 0) {
        // inside a row, or a continuous image: &data[iy * step + ix] addressed linearly (:40)
        const int idx = iy * L.cols + ix;
        const uint32_t q = idx < npx ? L.quad[idx] : 0u;
        d0 = (float)(q & 0xffu);
        d1 = (float)((q >> 8) & 0xffu);
        d2 = (float)((q >> 16) & 0xffu);
        d3 = (float)(q >> 24);
    } else if (pad == 1) {
        // ix == cols and one padding byte per row: data[0] and data[step] are padding, data[1] and
        // data[step + 1] are the first pixels of the next two rows
        const int i1 = (iy + 1) * L.cols, i3 = (iy + 2) * L.cols;
        d1 = i1 < npx ? (float)(L.quad[i1] & 0xffu) : 0.0f;
        d3 = i3 < npx ? (float)(L.quad[i3] & 0xffu) : 0.0f;
    }  // else: all four taps are padding
    // (1 - yy) * (1 - xx) * data[0] + (1 - yy) * xx * data[1] + yy * (1 - xx) * data[step] + yy * xx * data[step + 1]  (:43-44)
    return (1 - yy) * (1 - xx) * d0 + (1 - yy) * xx * d1 + yy * (1 - xx) * d2 + yy * xx * d3;
}

// dynamic LDS, in floats: vref | vcur | tnum | tden (PP each) | 32 slack | cand (cap ints) | cdist | cncc | order
__host__ __device__ inline size_t neighbor_lds_bytes(int half, int cap)
{
    const size_t P = (size_t)(2 * half + 1) * (2 * half + 1), PP = (P + 31) / 32 * 32;
    return (4 * PP + 32 + 4 * (size_t)cap) * 4 + 64;
}

template <int NR, int TAIL>
__global__ void __launch_bounds__(256) k_near_neighbors(NeighborArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char nb_lds[];
    const int i = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, row = lane >> 4;
    const int h = a.half, Wd = 2 * h + 1, P = Wd * Wd, PP = (P + 31) / 32 * 32, nfull = P / 32;
    float *vref = reinterpret_cast<float *>(nb_lds), *vcur = vref + PP, *tnum = vcur + PP, *tden = tnum + PP;
    int *cand = reinterpret_cast<int *>(tden + PP + 32);
    float *cdist = reinterpret_cast<float *>(cand + a.cap), *cncc = cdist + a.cap;
    int *order = reinterpret_cast<int *>(cncc + a.cap);
    float *sh = reinterpret_cast<float *>(order + a.cap);  // 16 floats of scratch
    int *shi = reinterpret_cast<int *>(sh + 8);

    if (!a.status[i]) return;                 // :791
    if (!a.pairs && a.count[i] > 0) return;   // :793 neighbours already found with a smaller search region

    // ---- 1. reference patch and its mean (:800-808), sample k = (x, y) = (k / Wd - h, k % Wd - h) -------------
    const float kx = a.keys_ref[2 * i], ky = a.keys_ref[2 * i + 1];
    float A00 = 1, A01 = 0, A10 = 0, A11 = 1;
    if (a.affine) A00 = a.affine[4 * i], A01 = a.affine[4 * i + 1], A10 = a.affine[4 * i + 2], A11 = a.affine[4 * i + 3];
    float vr[NR], wx[NR], wy[NR];
#pragma unroll
    for (int r = 0; r < NR; r++) {
        int k = tid + 256 * r;
        k = k < P ? k : P - 1;
        const int xi = k / Wd - h, yi = k - (k / Wd) * Wd - h;
        vr[r] = sample_free(a.ref0, a.pad_ref, kx + xi, ky + yi);
        if (a.affine) {  // src/utils.cpp:127-128
            wx[r] = A00 * xi + A01 * yi;
            wy[r] = A10 * xi + A11 * yi;
        } else {         // :125  pt_cur.x + x (int -> float)
            wx[r] = (float)xi;
            wy[r] = (float)yi;
        }
        if (tid + 256 * r < P) vref[tid + 256 * r] = vr[r];
    }
    __syncthreads();
    if (wave == 0) {
        const float s = chain_rows_f32<TAIL>(lds_off(vref) + 8u * lr, 128u, nfull);
        if (lane == 0) sh[0] = s;
    }
    __syncthreads();
    const float mean_ref = sh[0] / (float)P;  // :808  float /= size_t
    float dr[NR];
#pragma unroll
    for (int r = 0; r < NR; r++) {
        dr[r] = vr[r] - mean_ref;  // v_ref_dot, src/utils.cpp:140
        if (tid + 256 * r < P) tden[tid + 256 * r] = dr[r] * dr[r];
    }
    __syncthreads();
    if (wave == 0) {
        const float s = chain_rows_f32<TAIL>(lds_off(tden) + 8u * lr, 128u, nfull);  // demoniator1 (:143)
        if (lane == 0) sh[1] = s;
    }

    // ---- 2. candidates in index order (:812-815) -----------------------------------------------------------
    int total = 0;
    const float px = a.pt_pred[2 * i], py = a.pt_pred[2 * i + 1];
    if (a.pairs) {
        if (tid == 0) cand[0] = i;
        total = 1;
    } else {
        for (int base = 0; base < a.m; base += 256) {
            const int j = base + tid;
            bool hit = false;
            if (j < a.m) {
                const float dx = px - a.keys_cur_un[2 * j], dy = py - a.keys_cur_un[2 * j + 1];
                hit = !(fabsf(dx) > a.radius || fabsf(dy) > a.radius);  // the reference `continue`s on this (:814)
            }
            const unsigned long long mask = __ballot(hit);
            const int before = __popcll(mask & ((1ull << lane) - 1ull));
            if (lane == 0) shi[wave] = __popcll(mask);
            __syncthreads();
            int off = total;
            for (int w = 0; w < wave; w++) off += shi[w];
            if (hit && off + before < a.cap) cand[off + before] = j;
            total += shi[0] + shi[1] + shi[2] + shi[3];
            __syncthreads();
        }
    }
    __syncthreads();
    const float den1 = sh[1];
    if (total > a.cap) {  // the list does not fit the caller's arrays: report its true size, compute nothing
        if (tid == 0) a.count[i] = total;
        return;
    }

    // ---- 3. per candidate: NCC(halfPatchSize, vValuesRef, mean_ref, cur, mvKeysCur[j].pt, A), src/utils.cpp:110-148
    for (int c = 0; c < total; c++) {
        const int j = cand[c];
        const float cx = a.keys_cur[2 * j], cy = a.keys_cur[2 * j + 1];
        float vc[NR];
#pragma unroll
        for (int r = 0; r < NR; r++) {
            vc[r] = sample_free(a.cur0, a.pad_cur, cx + wx[r], cy + wy[r]);
            if (tid + 256 * r < P) vcur[tid + 256 * r] = vc[r];
        }
        __syncthreads();
        if (wave == 0) {
            const float s = chain_rows_f32<TAIL>(lds_off(vcur) + 8u * lr, 128u, nfull);  // mean_cur (:132)
            if (lane == 0) sh[2] = s;
        }
        __syncthreads();
        const float mean_cur = sh[2] / (float)P;  // :135
#pragma unroll
        for (int r = 0; r < NR; r++) {
            const float dc = vc[r] - mean_cur;
            if (tid + 256 * r < P) {
                tnum[tid + 256 * r] = dr[r] * dc;  // :142
                tden[tid + 256 * r] = dc * dc;     // :144
            }
        }
        __syncthreads();
        if (wave == 0) {
            const float s = chain_rows_f32<TAIL>(lds_off(row == 1 ? tden : tnum) + 8u * lr, 128u, nfull);
            if (lr == 0 && row < 2) sh[3 + row] = s;
        }
        __syncthreads();
        if (tid == 0) {
            // numerator / std::sqrt(demoniator1 * demoniator2 + 1e-10): float product, double from there (:147)
            cncc[c] = (float)((double)sh[3] / sqrt((double)(den1 * sh[4]) + 1e-10));
            const float dx = px - a.keys_cur_un[2 * j], dy = py - a.keys_cur_un[2 * j + 1];
            cdist[c] = sqrtf(dx * dx + dy * dy);  // :818
        }
    }
    __syncthreads();

    // ---- 4. the two-stack insertion of :825-842, one lane; `order` is stack 1 from bottom to top ---------------
    if (tid == 0) {
        int n1 = 0;
        for (int c = 0; c < total; c++) {
            int p = n1;
            if (a.use_ncc) {
                while (p > 0 && cncc[c] < cncc[order[p - 1]]) p--;  // pops while ncc < top.ncc (:826)
            } else {
                while (p > 0 && cdist[c] > cdist[order[p - 1]]) p--;  // pops while distance > top.distance (:832)
            }
            for (int q = n1; q > p; q--) order[q] = order[q - 1];
            order[p] = c;
            n1++;
        }
        const size_t b = (size_t)i * a.cap;
        for (int k = 0; k < total; k++) {  // popped from the top (:845-848)
            const int c = order[total - 1 - k];
            a.nbr_idx[b + k] = cand[c];
            a.nbr_dist[b + k] = cdist[c];
            a.nbr_ncc[b + k] = cncc[c];
        }
        a.count[i] = total;
    }
}

}  // namespace pagk
