// pagk_kernels.h -- gfx950 kernels of the PatchMatch hot path.
//
//   k_pyr_down      CreatePyramids               src/patch_match.cpp:61-76
//   k_build_quads   tap packing for the sampler  src/patch_match.cpp:399-403
//   k_track_block   the GN loop, one 256-thread workgroup per feature (default)
//   k_track_thread  the GN loop, one thread per feature (reference-shaped cross-check)
//
// Why one workgroup per feature.  H is structurally singular (de_dg is sampled at the patch
// centre, :263, so J3 = c*J4): the 4th LLT pivot is rounding noise, that noise is applied to
// dg/db (:334-337) and enters the convergence test (:343).  Results within 1e-3 px of the CPU
// path therefore need H, b and cost accumulated in the reference's order: (2h+1)^2 sequential
// f64 (f32 for cost) additions per entry per iteration.  That chain is the critical path of a
// feature.  The kernel keeps it as short as the hardware allows: all 256 lanes sample the patch
// and form the EXACT f64 products (24-bit x 24-bit fits 53 bits, so the product and
// multiply-add orders agree), stage them in LDS, and then one 16-lane DPP row per accumulator entry
// walks its stream at the dependent-FMA latency (5.9 cycles per pixel, pagk_chain_asm.h).
#pragma once
#include <type_traits>
#include "pagk_device.h"
#include "pagk_chain_asm.h"
#include "pagk_prio.h"

namespace pagk {

// ---- pyramid -----------------------------------------------------------------------------------
// cv::resize(prev, Size(cols*0.5, rows*0.5)), src/patch_match.cpp:69-70: exact 2x
// decimation of 8UC1 = OpenCV's INTER_AREA fast path, (a+b+c+d+2)>>2.
__global__ void k_pyr_down(const uint8_t *__restrict__ src, int64_t pitch, int dw, int dh,
                           uint8_t *__restrict__ dst)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= dw || y >= dh) return;
    const uint8_t *r0 = src + (int64_t)(2 * y) * pitch + 2 * x;
    const uint8_t *r1 = r0 + pitch;
    dst[(int64_t)y * dw + x] = (uint8_t)((r0[0] + r0[1] + r1[0] + r1[1] + 2) >> 2);
}

// The same cv::resize for a parent with an odd dimension: OpenCV's INTER_LINEAR for 8-bit images with
// 11-bit fixed-point coefficients (restated from memory of OpenCV 3.4 imgproc/resize.cpp -- parity
// unpinned; operation for operation the oracle's pagk_oracle_pyr_down).  One thread per output pixel.
__global__ void k_pyr_down_linear(const uint8_t *__restrict__ src, int64_t pitch, int w, int h, int dw, int dh,
                                  double scale_x, double scale_y, uint8_t *__restrict__ dst)
{
    const int dx = blockIdx.x * blockDim.x + threadIdx.x;
    const int dy = blockIdx.y * blockDim.y + threadIdx.y;
    if (dx >= dw || dy >= dh) return;
    float fy = (float)(((double)dy + 0.5) * scale_y - 0.5);
    int sy = (int)floorf(fy);
    fy -= (float)sy;
    const int b0 = __float2int_rn((1.f - fy) * 2048.f), b1 = __float2int_rn(fy * 2048.f);  // cvRound
    const int y0 = sy < 0 ? 0 : (sy < h ? sy : h - 1);
    const int y1 = sy + 1 < 0 ? 0 : (sy + 1 < h ? sy + 1 : h - 1);
    float fx = (float)(((double)dx + 0.5) * scale_x - 0.5);
    int sx = (int)floorf(fx);
    fx -= (float)sx;
    bool single = false;
    if (sx < 0) fx = 0.f, sx = 0;
    if (sx + 1 >= w) {
        single = true;
        if (sx >= w - 1) fx = 0.f, sx = w - 1;
    }
    const int a0 = __float2int_rn((1.f - fx) * 2048.f), a1 = __float2int_rn(fx * 2048.f);
    const uint8_t *r0 = src + (int64_t)y0 * pitch, *r1 = src + (int64_t)y1 * pitch;
    int S0, S1;
    if (single) {
        S0 = (int)r0[sx] * 2048;
        S1 = (int)r1[sx] * 2048;
    } else {
        S0 = (int)r0[sx] * a0 + (int)r0[sx + 1] * a1;
        S1 = (int)r1[sx] * a0 + (int)r1[sx + 1] * a1;
    }
    dst[(int64_t)dy * dw + dx] = (uint8_t)((((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2);
}

// Packs the taps of GetPixelValue for every pixel (see DevLevel).  `wrap` = the source
// image is continuous (step == cols): data[off+1] of the last column is the next row's
// first pixel.  Otherwise that byte is row padding, defined as 0.  Rows past the image = 0.
__global__ void k_build_quads(const uint8_t *__restrict__ src, int64_t pitch, int cols, int rows, int wrap,
                              uint32_t *__restrict__ quad)
{
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    int r = blockIdx.y * blockDim.y + threadIdx.y;
    if (c >= cols || r >= rows) return;
    auto px = [&](int rr, int cc) -> uint32_t {
        if (cc >= cols) {
            if (!wrap) return 0u;
            cc -= cols;
            rr += 1;
        }
        return rr < rows ? (uint32_t)src[(int64_t)rr * pitch + cc] : 0u;
    };
    uint32_t d0 = px(r, c), d1 = px(r, c + 1), d2 = px(r + 1, c), d3 = px(r + 1, c + 1);
    quad[(int64_t)r * cols + c] = d0 | (d1 << 8) | (d2 << 16) | (d3 << 24);
}

// One launch for the whole pyramid (levels <= 3): every level-l pixel is re-derived from level 0 by
// l nested rounded 2x2 means -- exactly the values the level-by-level cv::resize chain produces
// (:69-70), because each stage rounds to a byte before the next.  Removes the inter-level launch
// dependencies: 1 launch instead of 2L-1 per frame.  Blocks are level-uniform (blockIdx ranges).
struct PyrArgs {
    const uint8_t *src;  // level 0
    int64_t pitch;
    int wrap0;
    int n_levels;
    int cols[4], rows[4];
    int first_block[5];  // blocks [first_block[l], first_block[l+1]) belong to level l
    uint8_t *u8[4];      // u8[0] unused
    uint32_t *quad[4];
};

template <int L>
__device__ __forceinline__ uint32_t pyr_pixel(const uint8_t *__restrict__ src, int64_t pitch, int r, int c)
{
    if constexpr (L == 0) {
        return src[(int64_t)r * pitch + c];
    } else {
        return (pyr_pixel<L - 1>(src, pitch, 2 * r, 2 * c) + pyr_pixel<L - 1>(src, pitch, 2 * r, 2 * c + 1) +
                pyr_pixel<L - 1>(src, pitch, 2 * r + 1, 2 * c) + pyr_pixel<L - 1>(src, pitch, 2 * r + 1, 2 * c + 1) + 2) >> 2;
    }
}

template <int L>
__device__ __forceinline__ void pyr_emit(const PyrArgs &a, int idx)
{
    const int cols = a.cols[L], rows = a.rows[L];
    if (idx >= cols * rows) return;
    const int r = idx / cols, c = idx - r * cols;
    const int wrap = L == 0 ? a.wrap0 : 1;
    auto px = [&](int rr, int cc) -> uint32_t {
        if (cc >= cols) {
            if (!wrap) return 0u;
            cc -= cols;
            rr += 1;
        }
        return rr < rows ? pyr_pixel<L>(a.src, a.pitch, rr, cc) : 0u;
    };
    // the four taps one after the other (a scheduling fence between them at the coarse levels): issued all at
    // once, the 4 x 4^L byte loads of a level-2/3 pixel cost ~40 VGPRs, and these blocks share their register
    // allocation with the tracking workgroups of k_track_block_pyr, whose occupancy they must not lower
    const uint32_t d0 = px(r, c);
    if constexpr (L >= 2) asm volatile("" ::: "memory");
    const uint32_t d1 = px(r, c + 1);
    if constexpr (L >= 2) asm volatile("" ::: "memory");
    const uint32_t d2 = px(r + 1, c);
    if constexpr (L >= 2) asm volatile("" ::: "memory");
    const uint32_t d3 = px(r + 1, c + 1);
    a.quad[L][idx] = d0 | (d1 << 8) | (d2 << 16) | (d3 << 24);
    if constexpr (L > 0) a.u8[L][idx] = (uint8_t)d0;
}

// one 256-thread block of the fused pyramid (b = block index within the pyramid's own range)
__device__ __forceinline__ void pyr_block(const PyrArgs &a, int b, int tid)
{
    if (b < a.first_block[1]) {
        pyr_emit<0>(a, (b - a.first_block[0]) * 256 + tid);
    } else if (b < a.first_block[2]) {
        pyr_emit<1>(a, (b - a.first_block[1]) * 256 + tid);
    } else if (b < a.first_block[3]) {
        pyr_emit<2>(a, (b - a.first_block[2]) * 256 + tid);
    } else if (b < a.first_block[4]) {
        pyr_emit<3>(a, (b - a.first_block[3]) * 256 + tid);
    }
}

__global__ void __launch_bounds__(256) k_pyramid_fused(PyrArgs a)
{
    pyr_block(a, (int)blockIdx.x, (int)threadIdx.x);
}

// The pyramids of k frames as ONE launch (pagk_frame_set_device_batch: the CreatePyramids of k trackers that are stepped
// together, src/patch_match.cpp:61-76 per tracker): a block finds its frame by the frames' first blocks (ascending;
// wave-uniform scalar loads), then runs k_pyramid_fused's body on it -- per frame the same bytes as its own launch.
struct PyrBatchEntry {
    PyrArgs a;
    int block_base, pad_;
};
__global__ void __launch_bounds__(256) k_pyramid_fused_batch(const PyrBatchEntry *__restrict__ frames, int k)
{
    int f = 0;
    for (int j = 1; j < k; j++) f = (int)blockIdx.x >= frames[j].block_base ? j : f;
    pyr_block(frames[f].a, (int)blockIdx.x - frames[f].block_base, (int)threadIdx.x);
}

// ---- GyroPredictFeatures (src/gyro_aided_tracker.cpp:118-185,194-231), one thread per feature ----
struct PredictArgs {
    int n, width, height;
    int single_homography;  // mPredictMethod == SINGLE_HOMOGRAPHY: lambda = 1 (:235)
    float half;  // (float)mHalfPatchSize
    float fx, fy, cx, cy, fx_inv, fy_inv, k1, k2, p1, p2, k3;
    float K[6];            // rows 0 and 1 of mKRKinv
    float r31, r32, r33;   // third row of mRcl
    float inv00, inv01;    // (B B^T)^-1 = [[inv00, inv01], [inv01, inv00]] for the +-h corner matrix
    const float *pt_ref;
    float *pt_un, *pt_dist, *affine;
    uint8_t *status;
    // when non-null: 9 floats on the device, rows 0 and 1 of mKRKinv then the third row of mRcl; they replace
    // K[] and r31..r33 above (the rotation of a captured graph must not be baked into its kernel arguments)
    const float *d_rot;
};

// GyroPredictOnePixel (:194-256): PIXEL_AWARE_PREDICTION (:212-231) or SINGLE_HOMOGRAPHY (:233-253, lambda = 1.0)
__device__ __forceinline__ void predict_one(const PredictArgs &a, float rx, float ry, float &ux, float &uy, float &dxo,
                                            float &dyo)
{
    float x_normal = (rx - a.cx) * a.fx_inv;  // :209-210
    float y_normal = (ry - a.cy) * a.fy_inv;
    float lambda = a.single_homography ? 1.0f : (float)(1.0 / (double)(a.r31 * x_normal + a.r32 * y_normal + a.r33));  // :216 / :235
    float pt_x = (a.K[0] * rx + a.K[1] * ry + a.K[2]) * lambda;                            // :217
    float pt_y = (a.K[3] * rx + a.K[4] * ry + a.K[5]) * lambda;                            // :218
    float x = (pt_x - a.cx) * a.fx_inv;                                                    // :221-222
    float y = (pt_y - a.cy) * a.fy_inv;
    float r2 = x * x + y * y;
    float r4 = r2 * r2;
    float r6 = r4 * r2;
    float xd = x * (1 + a.k1 * r2 + a.k2 * r4 + a.k3 * r6) + 2 * a.p1 * x * y + a.p2 * (r2 + 2 * x * x);
    float yd = y * (1 + a.k1 * r2 + a.k2 * r4 + a.k3 * r6) + a.p1 * (r2 + 2 * y * y) + 2 * a.p2 * x * y;
    ux = pt_x;
    uy = pt_y;
    dxo = a.fx * xd + a.cx;  // :229-230
    dyo = a.fy * yd + a.cy;
}

__global__ void __launch_bounds__(256) k_gyro_predict(PredictArgs a)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    if (a.d_rot) {
        for (int k = 0; k < 6; k++) a.K[k] = a.d_rot[k];
        a.r31 = a.d_rot[6], a.r32 = a.d_rot[7], a.r33 = a.d_rot[8];
    }
    // Initialize() state where the loop `continue`s (:92-95, :131-135)
    a.status[i] = 0;
    a.pt_un[2 * i] = a.pt_un[2 * i + 1] = 0.0f;
    a.pt_dist[2 * i] = a.pt_dist[2 * i + 1] = 0.0f;
    const float rx = a.pt_ref[2 * i], ry = a.pt_ref[2 * i + 1];
    float ux, uy, dxs, dys;
    predict_one(a, rx, ry, ux, uy, dxs, dys);
    const float W = (float)a.width, H = (float)a.height;
    if (ux < 0 || ux >= W || uy < 0 || uy >= H) return;      // :131
    if (dxs < 0 || dxs >= W || dys < 0 || dys >= H) return;  // :134
    a.pt_un[2 * i] = ux;
    a.pt_un[2 * i + 1] = uy;
    a.pt_dist[2 * i] = dxs;
    a.pt_dist[2 * i + 1] = dys;
    a.status[i] = 1;
    if (!a.affine) return;
    // four corners (:148-160), then A = C B^T (B B^T)^-1 (:166-167): float products accumulated in
    // double per Mat product, narrowed after each product (cv::Mat gemm on small CV_32F matrices)
    const float cxs[4] = {-a.half, a.half, -a.half, a.half}, cys[4] = {-a.half, -a.half, a.half, a.half};
    double s00 = 0, s01 = 0, s10 = 0, s11 = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        float cux, cuy, t0, t1;
        predict_one(a, rx + cxs[j], ry + cys[j], cux, cuy, t0, t1);
        const float Cx = cux - ux, Cy = cuy - uy;
        s00 += (double)Cx * cxs[j];
        s01 += (double)Cx * cys[j];
        s10 += (double)Cy * cxs[j];
        s11 += (double)Cy * cys[j];
    }
    const float t00 = (float)s00, t01 = (float)s01, t10 = (float)s10, t11 = (float)s11;
    a.affine[4 * i + 0] = (float)((double)t00 * a.inv00 + (double)t01 * a.inv01);
    a.affine[4 * i + 1] = (float)((double)t00 * a.inv01 + (double)t01 * a.inv00);
    a.affine[4 * i + 2] = (float)((double)t10 * a.inv00 + (double)t11 * a.inv01);
    a.affine[4 * i + 3] = (float)((double)t10 * a.inv01 + (double)t11 * a.inv00);
}

// ---- shared epilogue: SetMatcher + DistortPoints for one feature --------------------------------
// (the arrays a launch writes: the launch's own, or -- batched launches -- one camera stream's)
struct OutPtrs {
    float *pt_un, *pt_dist;
    uint8_t *status;
    double *pix_err, *dist_pred;
    float *ncc;
    int *iters;
    const float *pred;  // mvPtPredictUn, or the reference points when there is no prediction (:384)
};
__device__ __forceinline__ void write_outputs_to(const TrackArgs &a, const OutPtrs &o, int i, float p2x, float p2y, int succ,
                                                 float lastCost, int level0_ran, float ncc, int iters)
{
    o.pt_un[2 * i] = p2x;      // mvPtPredictAfterPatchMatchedUn  (:380)
    o.pt_un[2 * i + 1] = p2y;
    o.status[i] = (uint8_t)(level0_ran ? succ : 0);  // :351, :381 (zero-init when skipped)
    if (o.pix_err) o.pix_err[i] = level0_ran ? sqrt((double)lastCost * a.win_size_inv) : 0.0;  // :352
    if (o.dist_pred) {  // :384-385
        float ddx = o.pred[2 * i] - p2x, ddy = o.pred[2 * i + 1] - p2y;
        o.dist_pred[i] = (double)sqrtf(ddx * ddx + ddy * ddy);
    }
    if (o.pt_dist) {  // :116, :379
        float ox, oy;
        distort_point(a, p2x, p2y, ox, oy);
        o.pt_dist[2 * i] = ox;
        o.pt_dist[2 * i + 1] = oy;
    }
    if (o.ncc) o.ncc[i] = ncc;  // :365 / :95
    if (o.iters) o.iters[i] = iters;
}
__device__ __forceinline__ void write_outputs(const TrackArgs &a, int i, float p2x, float p2y, int succ,
                                              float lastCost, int level0_ran, float ncc, int iters)
{
    const OutPtrs o{a.pt_un, a.pt_dist, a.status, a.pix_err, a.dist_pred, a.ncc, a.iters, a.pt_init ? a.pt_init : a.pt_ref};
    write_outputs_to(a, o, i, p2x, p2y, succ, lastCost, level0_ran, ncc, iters);
}

// PatchMatch::NCC, src/patch_match.cpp:433-469, one thread.  Always on the level-0 images (:358,:361);
// loop order x outer, y inner (:438-439); float sums; the final division promotes to double (:468).
__device__ inline float ncc_serial(const TrackArgs &a, float rx, float ry, float cx, float cy, float A00, float A01,
                                   float A10, float A11)
{
    const int h = a.half;
    const float Pf = (float)((2 * h + 1) * (2 * h + 1));
    const DevLevel &R = a.l1[0], &C = a.l2[0];
    auto cur_at = [&](int x, int y) {
        if (!a.use_affine) return sample<true>(C, cx + x, cy + y);  // warp_mat.empty() (:446-447)
        float wx = A00 * x + A01 * y, wy = A10 * x + A11 * y;       // :449-450
        return sample<true>(C, cx + wx, cy + wy);
    };
    float mean_ref = 0.0f, mean_cur = 0.0f;
    for (int x = -h; x <= h; x++)
        for (int y = -h; y <= h; y++) {
            mean_ref += sample<true>(R, rx + x, ry + y);
            mean_cur += cur_at(x, y);
        }
    mean_ref /= Pf;  // :457  float / size_t
    mean_cur /= Pf;
    float num = 0, d1 = 0, d2 = 0;
    for (int x = -h; x <= h; x++)
        for (int y = -h; y <= h; y++) {
            float vr = sample<true>(R, rx + x, ry + y), vc = cur_at(x, y);
            num += ((vr - mean_ref) * (vc - mean_cur));
            d1 += (vr - mean_ref) * (vr - mean_ref);
            d2 += (vc - mean_cur) * (vc - mean_cur);
        }
    return (float)((double)num / sqrt((double)(d1 * d2) + 1e-10));
}

// ---- one thread per feature: the reference's loop nest, verbatim in shape -----------------------
// PatchMatch::OpticalFlowConsideringIlluminationChange_onePixel, src/patch_match.cpp:167-367,
// with the level loop of OpticalFlowMultiLevel (:98) folded in (features are independent
// across levels).  Slow (scattered gathers, divergent trip counts); kept as an on-device
// cross-check of k_track_block.
__global__ void __launch_bounds__(64) k_track_thread(TrackArgs a)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const float *init = a.has_gyro ? a.pt_init : a.pt_ref;  // :85-89
    float p2x = init[2 * i], p2y = init[2 * i + 1];
    if (!a.status_in[i]) {  // :173
        write_outputs(a, i, p2x, p2y, 0, 0.0f, 0, 0.0f, 0);
        return;
    }
    const int h = a.half;
    float A00 = 1, A01 = 0, A10 = 0, A11 = 1;
    if (a.use_affine) {
        A00 = a.affine[4 * i], A01 = a.affine[4 * i + 1], A10 = a.affine[4 * i + 2], A11 = a.affine[4 * i + 3];
    }
    int succ = 1, iters = 0;
    float lastCost = 0.0f;
    for (int level = a.n_levels - 1; level >= 0; level--) {
        const DevLevel &L1 = a.l1[level], &L2 = a.l2[level];
        float ptx = a.pt_ref[2 * i] * a.scales[level], pty = a.pt_ref[2 * i + 1] * a.scales[level];  // :177
        float nx, ny;
        if (level == a.n_levels - 1) {  // :180
            nx = p2x * a.scales[level];
            ny = p2y * a.scales[level];
        } else {  // :182
            nx = (float)((double)(p2x * 1.0f) / 0.5);
            ny = (float)((double)(p2y * 1.0f) / 0.5);
        }
        float dx = nx - ptx, dy = ny - pty, dg = 0.0f, db = 0.0f, cost = 0.0f;
        lastCost = 0.0f;
        succ = 1;
        float cneg = -sample<true>(L1, ptx, pty);  // :263
        for (int iter = 0; iter < a.iterations; iter++) {
            double H[4][4] = {}, b[4] = {};
            iters++;
            cost = 0;
            for (int y = -h; y <= h; y++)
                for (int x = -h; x <= h; x++) {
                    float wx = (float)x, wy = (float)y;
                    if (a.use_affine) {  // :203-204
                        wx = A00 * x + A01 * y;
                        wy = A10 * x + A11 * y;
                    }
                    Five s = sample5<true>(L2, ptx + dx + wx, pty + dy + wy);
                    float e = s.c + db - (1.0f + dg) * sample<true>(L1, ptx + x, pty + y);  // :252-253
                    float Ix = 0.5f * (s.xp - s.xm);                                           // :259
                    float Iy = 0.5f * (s.yp - s.ym);                                           // :261
                    double J[4] = {(double)Ix, (double)Iy, (double)cneg, 1.0};
                    double ed = (double)e;
#pragma unroll
                    for (int r = 0; r < 4; r++) b[r] += (-J[r]) * ed;  // :293
                    cost += e * e;                                     // :294
#pragma unroll
                    for (int r = 0; r < 4; r++)
#pragma unroll
                        for (int c = 0; c <= r; c++) H[r][c] += J[r] * J[c];  // :296 (lower triangle)
                }
            if (a.penalty) add_penalty(a, dx, dy, H, b, cost);
            double upd[4];
            double unorm = llt4_solve_norm(H, b, upd, a.solver);  // :319
            if (upd[0] != upd[0]) {                     // :322
                succ = 0;
                break;
            }
            if (iter > 0 && cost > lastCost) break;  // :328
            dx = (float)((double)dx + upd[0]);       // :332
            dy = (float)((double)dy + upd[1]);
            if (a.illum) {
                dg = (float)((double)dg + upd[2]);
                db = (float)((double)db + upd[3]);
            }
            lastCost = cost;
            succ = 1;
            if (unorm < 1e-2) break;  // :343
        }
        p2x = ptx + dx;  // :348
        p2y = pty + dy;
    }
    float ncc = 1.0f;  // :365
    if (a.calc_ncc) ncc = ncc_serial(a, a.pt_ref[2 * i], a.pt_ref[2 * i + 1], p2x, p2y, A00, A01, A10, A11);
    write_outputs(a, i, p2x, p2y, succ, lastCost, 1, ncc, iters);
}

// ---- one workgroup per feature -----------------------------------------------------------------
// 256 threads = 4 waves.  Per Gauss-Newton iteration:
//   1. sampling  -- every lane owns NR = ceil(P/256) pixels of the patch; for each it takes the five
//                   img2 samples, forms e, Ix, Iy (f32, :252-262) and stores to LDS the eight f64
//                   streams the accumulator entries need (three squares/cross products of the
//                   gradient, Ix*e, Iy*e, and Ix, Iy, e widened) plus the f32 stream e*e;
//   2. ordered accumulation -- 16 DPP rows (4 per wave), one per entry: H00 H10 H11 b0 | b1 H20 H21
//                   H30 | H31 b2 b3 H22 | cost (f32, wave 3).  Entries with a constant second
//                   factor (c or 1) multiply it in the FMA: fma(Ix, c, H20) == H20 + c*Ix because
//                   the product of two f32-valued doubles is exact.  pagk_chain_asm.h;
//   3. solve     -- one lane: penalty, 4x4 LLT, two triangular solves, norm;
//   4. update    -- every lane applies the same update and takes the same exit (:322-343).
// Dynamic LDS: double stream[8][PP + 1]; float esq[PP]; double cslot[2]; double acc[16];
//              double upd[5]; float cost[2]      with PP = 32 * ceil(P / 32).
//
// MFMA variant (template MFMA = true, 2 waves): for launches with more features than the chip can
// hold at once, throughput matters more than one workgroup's latency.  v_mfma_f64_4x4x4f64 computes, in
// each of its 4 blocks, D = C + sum_k A[:,k] B[k,:] as a SEQUENTIAL chain of FMAs in ascending k
// (measured bit-for-bit on 128000 outputs, tools/microbench7.hip; lane layout: A(q,i,k) in lane
// 16k+4q+i, B(q,k,j) in lane 16k+4q+j, D(q,i,j) in lane 16i+4q+j).  With k = four consecutive patch
// pixels, block 0 fed with A = B = J and block 1 with A = -J, B = (e,0,0,0), one instruction advances all
// sixteen H entries and the four b entries by four pixels in the reference's order -- on the matrix
// pipe, beside the VALU.  A dependent MFMA takes 52 cycles (13 per pixel vs 5.9 for the DPP rows), so
// this form loses on latency and wins on issue slots: one wave instead of three, no product streams.
// Half-size workgroups double the number resident per CU.
constexpr int kBlock = 256;   // DPP variant
constexpr int kStreams = 8;   // DPP variant: XX YX YY XE YE X Y E;  MFMA variant: X Y E

__host__ __device__ inline int track_block_pp(int half)
{
    int P = (2 * half + 1) * (2 * half + 1);
    return (P + 31) / 32 * 32;
}
__host__ __device__ inline size_t track_block_lds_bytes(int half)
{
    size_t PP = (size_t)track_block_pp(half);
    size_t bytes = kStreams * (PP + 1) * 8 + PP * 4 + 2 * 8 + 16 * 8 + 5 * 8 + 2 * 4 + 8;  // streams PP + 1 doubles apart
    bytes += 16 + 16 + 8;  // the pipelined body (pagk_pipe_kernel.h): int flags[4], double h22[2], double pen
#ifdef PAGK_EXPERIMENT_LDS_WINDOW
    bytes += 32 * 32 * 4;  // the staged img2 window (experiment build only, see track_block_body)
#endif
    return bytes;
}
// MFMA variant: three f64 streams of PP + 8 (the +8 staggers the banks of neighbouring arrays),
// X = Ix, Y = Iy, NE = -e, written per iteration, followed by a 16-double constant area
// (4 x c, 4 x 1.0, 4 x 0.0, pad) that the constant operand lanes re-read with stride 0.
__host__ __device__ inline size_t track_mfma_lds_bytes(int half)
{
    size_t PP = (size_t)track_block_pp(half);
    return (3 * (PP + 8) + 16) * 8 + PP * 4 + 2 * 8 + 24 * 8 + 5 * 8 + 2 * 4 + 8;
}

__device__ __forceinline__ uint32_t lds_off(const void *p)
{
    return (uint32_t)(uintptr_t)p;  // low half of a flat LDS address = offset in the LDS aperture
}

}  // namespace pagk
#include "pagk_pipe_kernel.h"
namespace pagk {

// ---- relaxed-order accumulation (EXPERIMENT, pagk_set_kernel(ctx, 4)) -----------------------------
// What the reference's summation order costs: the same kernel with the 441-step ordered chains replaced by
// per-lane strided partial sums and a 16-lane tree.  NOT parity-exact -- H is structurally singular, so a
// different rounding of H and b moves a few percent of the features by more than the 1e-3 px bar (SURVEY.md
// section 0; measured in tests/test_parity_gpu.py::test_relaxed_order_experiment) -- and therefore never
// selected automatically and never the benchmark's `value`.
template <int N>
__device__ __forceinline__ double row_shr_f64(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x110 + N, 0xf, 0xf, false);  // row_shr:N, lanes without a source get 0
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x110 + N, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int N>
__device__ __forceinline__ float row_shr_f32(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x110 + N, 0xf, 0xf, false));
}
// sum of arr[0..P) * s1 over the 16 lanes of a DPP row; the total lands in lane 15 of the row
__device__ __forceinline__ double relaxed_row_f64(const double *arr, int P, int lr, double s1)
{
    double s0 = 0.0, s1a = 0.0, s2 = 0.0, s3 = 0.0;
    int k = lr;
    for (; k + 48 < P; k += 64) {
        s0 = __builtin_fma(arr[k], s1, s0);
        s1a = __builtin_fma(arr[k + 16], s1, s1a);
        s2 = __builtin_fma(arr[k + 32], s1, s2);
        s3 = __builtin_fma(arr[k + 48], s1, s3);
    }
    for (; k < P; k += 16) s0 = __builtin_fma(arr[k], s1, s0);
    double s = (s0 + s1a) + (s2 + s3);
    s += row_shr_f64<8>(s);
    s += row_shr_f64<4>(s);
    s += row_shr_f64<2>(s);
    s += row_shr_f64<1>(s);
    return s;
}
__device__ __forceinline__ float relaxed_row_f32(const float *arr, int P, int lr)
{
    float s0 = 0.0f, s1 = 0.0f;
    int k = lr;
    for (; k + 16 < P; k += 32) {
        s0 += arr[k];
        s1 += arr[k + 16];
    }
    for (; k < P; k += 16) s0 += arr[k];
    float s = s0 + s1;
    s += row_shr_f32<8>(s);
    s += row_shr_f32<4>(s);
    s += row_shr_f32<2>(s);
    s += row_shr_f32<1>(s);
    return s;
}

// `i`: the feature; `resume`: nullptr, or the state a throughput kernel suspended the feature in (k_track_resume).
// LEAN: the launch is known to run without the regularisation penalty and with solver_variant 0 (the reference's
// defaults, BASELINE's configs): both become compile-time facts.  A lone wave pays ~6 cycles per instruction, and the
// generic form spends ~55 of the solve's ~390 on them -- the selects of the five association switches, the Eigen <= 3.2
// reciprocal path computed beside the division, and twenty register copies where the penalty's branch rejoins.
// PIPE: two-round patches (h = 8, 9, 10) run the pipelined iteration of pagk_pipe_kernel.h; false keeps the serial
// phases below (the five-workgroups-per-CU build: at 96 VGPRs the pipelined body spills 32 registers instead of 13).
template <int NR, int TAIL, int WAVES = 4, bool MFMA = false, bool RELAXED = false, bool LEAN = false, bool PIPE = true>
__device__ __forceinline__ void track_block_body(const TrackArgs &a, const int i, const SuspState *resume = nullptr)
{
    static_assert(MFMA ? WAVES == 2 : WAVES == 4, "DPP rows need 4 waves; the MFMA variant is 2 waves");
    static_assert(!(MFMA && RELAXED), "the relaxed-order experiment exists for the 4-wave kernel only");
    constexpr int kBlock = WAVES * 64;
    constexpr int kStreams = MFMA ? 3 : 8;
    constexpr int kAcc = MFMA ? 24 : 16;
    // two-round patches: (NR, TAIL) determine the patch size (h = 8, 9, 10 <-> P mod 32 = 1, 9, 25), so every LDS
    // address below is a compile-time constant (immediate offsets instead of address registers)
    constexpr int HC = (NR == 2 && !MFMA) ? (TAIL == 1 ? 8 : (TAIL == 9 ? 9 : 10)) : 0;
#ifndef PAGK_NO_PIPE
    // ... and their iteration is the pipelined one (pagk_pipe_kernel.h; -DPAGK_NO_PIPE builds the serial phases below
    // for A/B runs: the same bits)
    if constexpr (HC != 0 && !RELAXED && PIPE) {
        track_pipe_body<HC, LEAN>(a, i, resume);
        return;
    }
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int h = HC ? HC : a.half, Wd = 2 * h + 1, P = Wd * Wd, PP = track_block_pp(h);
    const int nfull = P / 32;  // P mod 32 == TAIL

    // array stride in doubles.  DPP variant: PP + 1, so that neighbouring streams start 8 bytes apart modulo the
    // 256-byte bank span: the two rows of a 32-lane LDS group (lanes 0-15 / 16-31 read 16 bytes per lane of two
    // DIFFERENT streams) then use complementary banks.  With a stride of PP (a multiple of 256 bytes) every chain
    // read was a 2-way bank conflict (round 1: 0.82 conflict cycles per LDS instruction).
    const int PS = MFMA ? PP + 8 : PP + 1;
    double *stream = reinterpret_cast<double *>(lds_raw);
    double *cst = stream + (size_t)kStreams * PS;  // MFMA variant: constant area (16 doubles)
    float *esq = reinterpret_cast<float *>(cst + (MFMA ? 16 : 0));
    double *cslot = reinterpret_cast<double *>(esq + PP);
    double *acc = cslot + 2;
    double *sh_upd = acc + kAcc;
    float *sh_cost = reinterpret_cast<float *>(sh_upd + 5);
#ifdef PAGK_EXPERIMENT_LDS_WINDOW
    uint32_t *win = reinterpret_cast<uint32_t *>(sh_cost + 2);  // 32 x 32 quads of img2 around the patch (experiment)
#endif

    const float *init = a.has_gyro ? a.pt_init : a.pt_ref;  // :85-89
    float p2x = init[2 * i], p2y = init[2 * i + 1];
    if (!a.status_in[i]) {  // :173 (block-uniform)
        if (tid == 0) write_outputs(a, i, p2x, p2y, 0, 0.0f, 0, 0.0f, 0);
        return;
    }
    float A00 = 1, A01 = 0, A10 = 0, A11 = 1;
    if (a.use_affine) {
        A00 = a.affine[4 * i], A01 = a.affine[4 * i + 1], A10 = a.affine[4 * i + 2], A11 = a.affine[4 * i + 3];
    }
    const float refx = a.pt_ref[2 * i], refy = a.pt_ref[2 * i + 1];

    // lane -> pixel map, fixed for the whole kernel: p = tid + 256 r, row-major (y outer, :233-234)
    float px[NR], py[NR], wx[NR], wy[NR];
#pragma unroll
    for (int r = 0; r < NR; r++) {
        int p = tid + kBlock * r;
        p = p < P ? p : P - 1;  // lanes past the patch shadow the last pixel (never stored)
        int yy = p / Wd, xx = p - yy * Wd;
        int x = xx - h, y = yy - h;
        px[r] = (float)x;
        py[r] = (float)y;
        if (a.use_affine) {  // :203-204  A(0,0)*x + A(0,1)*y, int -> float
            wx[r] = A00 * (float)x + A01 * (float)y;
            wy[r] = A10 * (float)x + A11 * (float)y;
        } else {
            wx[r] = (float)x;
            wy[r] = (float)y;
        }
    }
    // extent of the warped patch, for the interior (clamp-free) fast path
    const float fh = (float)h;
    const float ext_x = fabsf(A00) * fh + fabsf(A01) * fh + 2.0f;
    const float ext_y = fabsf(A10) * fh + fabsf(A11) * fh + 2.0f;

    // accumulator row of this lane: its LDS stream, block stride.  A DPP row broadcasts ONE data value per step to its
    // sixteen lanes, but the second FMA factor and the accumulator are per lane: entries that read the same stream
    // with different constant factors share a row (H20 = sum Ix*c and H30 = sum Ix*1 in lanes 0 / 1 of the X row,
    // likewise H21 | H31 and b2 | b3).  Eight rows = two waves carry the eleven data chains; the cost chain (f32) is
    // wave 2; H22 = sum c*c depends on the level only and is chained once per level by wave 3.
    //   row:      0    1    2    3    4    5          6          7         | 8-11  | 12-15
    //   stream:   XX   YX   YY   XE   YE   X          Y          E         | esq   | cslot
    //   entry:    H00  H10  H11  b0   b1   H20 | H30  H21 | H31  b2 | b3   | cost  | H22 (first iteration of a level)
    // (the relaxed-order experiment keeps one row per entry: its lanes hold partial sums of ONE entry)
    const int lr = lane & 15;
    const int cid = wave * 4 + (lane >> 4);
    const int stream_of[12] = {0, 1, 2, 3, 4, 5, 6, 5, 6, 7, 7, 0};  // relaxed-order experiment only
    uint32_t row_addr, row_inc;
    if constexpr (RELAXED || MFMA) {
        if (MFMA || cid >= 12) {
            row_addr = lds_off(esq) + 8u * lr;  // cost rows (MFMA variant: wave 1, all four rows)
            row_inc = 128u;
        } else if (cid < 11) {
            row_addr = lds_off(stream + (size_t)stream_of[cid] * PS) + 16u * lr;
            row_inc = 256u;
        } else {
            row_addr = lds_off(cslot);  // every lane re-reads (c, c)
            row_inc = 0u;
        }
    } else {
        if (cid < 8) {
            row_addr = lds_off(stream + (size_t)cid * PS) + 16u * lr;
            row_inc = 256u;
        } else if (cid < 12) {
            row_addr = lds_off(esq) + 8u * lr;
            row_inc = 128u;
        } else {
            row_addr = lds_off(cslot);  // every lane re-reads (c, c)
            row_inc = 0u;
        }
    }
    // MFMA variant, operand roles of this lane: k = pixel within the group of four, q = block, i = row
    // (A) or column (B).  q = 0: A = B = J[i];  q = 1: A = J[i], B = (i == 0 ? -e : 0), i.e. b += J * (-e),
    // the same exact product as the reference's (-J) * e (:293);  q >= 2: zero.  Every operand is a plain
    // LDS read: streams X Y NE advance by 4 doubles per MFMA, the constants c / 1 / 0 sit in cst[] and are
    // re-read in place (per-lane step 0).
    const int mk = lane >> 4, mq = (lane >> 2) & 3, mi = lane & 3;
    // J[i]: i = 0 -> X, 1 -> Y, 2 -> c, 3 -> 1
    const double *j_src = mi < 2 ? stream + (size_t)mi * PS + mk : cst + (mi == 2 ? 0 : 4) + mk;
    const int j_step = mi < 2 ? 4 : 0;
    const double *a_src = mq < 2 ? j_src : cst + 8 + mk;
    const int a_step = mq < 2 ? j_step : 0;
    const double *b_src = mq == 0 ? j_src : ((mq == 1 && mi == 0) ? stream + (size_t)2 * PS + mk : cst + 8 + mk);
    const int b_step = mq == 0 ? j_step : ((mq == 1 && mi == 0) ? 4 : 0);
    if constexpr (MFMA) {
        if (tid < 4) {
            cst[4 + tid] = 1.0;
            cst[8 + tid] = 0.0;
        }
    }

    int succ = 1, iters = 0;
    [[maybe_unused]] constexpr bool kPrioByWork = NR > 1;  // pagk_prio.h
    PAGK_PRIO_DECL
    float lastCost = 0.0f;
#ifdef PAGK_STAMPS
    // diagnostic build only: cycles per phase, summed over iterations, written to a.dbg (a buffer
    // nothing else reads).  [0] level setup, [1] sampling, [2] chains, [3] solve, [4] update, [5] total
    unsigned long long st[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t_begin = __builtin_amdgcn_s_memtime(), t0, t1;
    const unsigned long long rt_begin = __builtin_amdgcn_s_memrealtime();
#define STAMP(k)                                  \
    t1 = __builtin_amdgcn_s_memtime();            \
    st[k] += t1 - t0;                             \
    t0 = t1;
#else
#define STAMP(k)
#endif

    const int level_first = resume ? resume->level : a.n_levels - 1;
    for (int level = level_first; level >= 0; level--) {
#ifdef PAGK_STAMPS
        t0 = __builtin_amdgcn_s_memtime();
#endif
        const DevLevel &L1 = a.l1[level];
        const DevLevel L2 = pin_level(a.l2[level]);  // read by every iteration: scalar registers, not kernel-argument loads
        const float ptx = refx * a.scales[level], pty = refy * a.scales[level];  // :177
        float nx, ny;
        if (level == a.n_levels - 1) {  // :180
            nx = p2x * a.scales[level];
            ny = p2y * a.scales[level];
        } else {  // :182
            nx = (float)((double)(p2x * 1.0f) / 0.5);
            ny = (float)((double)(p2y * 1.0f) / 0.5);
        }
        float dx = nx - ptx, dy = ny - pty, dg = 0.0f, db = 0.0f;  // :186-191
        lastCost = 0.0f;                                            // :193
        succ = 1;                                                   // :194
        int iter_first = 0;
        if (resume && level == level_first) {  // pick the feature up where the throughput kernel left it
            dx = resume->dx, dy = resume->dy, dg = resume->dg, db = resume->db;
            lastCost = resume->lastCost;
            iters = resume->iters;
            iter_first = resume->iter;
        }

        // img1 samples are iteration-invariant: once per level (bit-identical to :253, :263)
        const float cneg = -sample<true>(L1, ptx, pty);
        const double cd = (double)cneg;
        float s1[NR];
#pragma unroll
        for (int r = 0; r < NR; r++) s1[r] = sample<true>(L1, ptx + px[r], pty + py[r]);
        // second FMA factor of this lane (:264  J = (Ix, Iy, de_dg, 1))
        double row_s1;
        if constexpr (RELAXED) {
            switch (cid) {
                case 3: case 4: case 10: row_s1 = -1.0; break;  // b0 b1 b3:  -J * e
                case 5: case 6: case 11: row_s1 = cd; break;     // H20 H21 H22
                case 9: row_s1 = -cd; break;                     // b2
                default: row_s1 = 1.0; break;
            }
        } else {
            switch (cid) {
                case 3: case 4: row_s1 = -1.0; break;                  // b0 b1:  -J * e
                case 5: case 6: row_s1 = lr == 0 ? cd : 1.0; break;    // H20 | H30,  H21 | H31
                case 7: row_s1 = lr == 0 ? -cd : -1.0; break;          // b2 | b3
                default: row_s1 = cid >= 12 ? cd : 1.0; break;         // H22 (wave 3);  H00 H10 H11
            }
        }
        if (tid == 0) {
            cslot[0] = cd;
            cslot[1] = cd;
        }
        if constexpr (MFMA) {
            if (tid < 4) cst[tid] = cd;  // first read after the sampling barrier
        }

        STAMP(0)
        for (int iter = iter_first; iter < a.iterations; iter++) {  // :215
            iters++;
            PAGK_PRIO_TIER  // (pagk_prio.h: a workgroup that is behind its neighbours outranks them)
            // ---- 1. sampling ------------------------------------------------------------------
            const float bx = ptx + dx, by = pty + dy;  // (pt.x + dx), then + wx (:252)
            const float gain = 1.0f + dg;
            // interior test (block-uniform): every tap coordinate of every pixel, +-1 included,
            // lies in [0, cols-1) x [0, rows-1) => clamps are no-ops and can be skipped.
            const bool interior = (bx - ext_x >= 0.0f) & (bx + ext_x < L2.fcols_m1) &
                                  (by - ext_y >= 0.0f) & (by + ext_y < L2.frows_m1);
            // all rounds' gathers are issued before any is consumed (lanes past the patch sample
            // a valid pixel and simply do not store)
            // the gathers of up to two rounds are in flight at a time (a tap costs up to four registers until it is
            // interpolated): NR = 2 issues everything up front, NR = 4 works in two groups
            // The clamp-free and the clamped form are two copies of the whole phase, chosen once: with the choice inside
            // (`interior ? issue<false> : issue<true>` per round) the two forms' loads share destination registers on
            // a control-flow path that cannot happen, and the compiler waits for every load in flight before each
            // round's gathers -- the rounds ran one after the other (profiles/r02_ab_runs.md).
            // Measured (same session, tools/ab_lib.py): 4-wave kernel -2.9 % at 250 features, -2.7 % at 1000, -2.2 % on
            // configs[2]; the 2-wave MFMA kernel (four rounds) +1.5 % (13 more spilled registers), so it keeps the choice
            // per round (mode 2).
            auto sampling = [&](auto mode_tag, auto rounds_tag) {
            constexpr int NRX = decltype(rounds_tag)::value;  // rounds this wave samples (its later rounds lie past the patch)
            constexpr int G = NRX < 2 ? NRX : 2;
            constexpr int MODE = decltype(mode_tag)::value;  // 0: clamp-free, 1: clamped, 2: chosen per round
#pragma unroll
            for (int r0 = 0; r0 < NRX; r0 += G) {
                FiveTaps taps[G];
#pragma unroll
                for (int u = 0; u < G; u++) {
                    const int r = r0 + u < NRX ? r0 + u : NRX - 1;
                    float X = bx + wx[r], Y = by + wy[r];
                    if constexpr (MODE == 2)
                        taps[u] = interior ? sample5_issue<false>(L2, X, Y) : sample5_issue<true>(L2, X, Y);
                    else
                        taps[u] = sample5_issue<MODE == 1>(L2, X, Y);
#ifdef PAGK_STAMPS
                    if (r0 == 0 && u == 0) {
                        STAMP(13)  // first round's coordinates computed and gathers issued
                    }
#endif
                }
#ifdef PAGK_STAMPS
                if (r0 == 0) {
                    STAMP(8)   // coordinates computed, gathers issued
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    STAMP(9)   // gathers returned
                }
#endif
#pragma unroll
                for (int u = 0; u < G; u++) {
                    const int r = r0 + u;
                    if (r >= NRX) continue;
                    const Five s = sample5_finish(taps[u]);
                    const int p = tid + kBlock * r;
                    if (p < P) {
                        float e = s.c + db - gain * s1[r];   // :252-253
                        float Ix = 0.5f * (s.xp - s.xm);     // :259-260
                        float Iy = 0.5f * (s.yp - s.ym);     // :261-262
                        double dIx = (double)Ix, dIy = (double)Iy, de = (double)e;
                        if constexpr (MFMA) {
                            stream[0 * PS + p] = dIx;
                            stream[1 * PS + p] = dIy;
                            stream[2 * PS + p] = -de;
                        } else {
                            stream[0 * PS + p] = dIx * dIx;
                            stream[1 * PS + p] = dIy * dIx;
                            stream[2 * PS + p] = dIy * dIy;
                            stream[3 * PS + p] = dIx * de;
                            stream[4 * PS + p] = dIy * de;
                            stream[5 * PS + p] = dIx;
                            stream[6 * PS + p] = dIy;
                            stream[7 * PS + p] = de;
                        }
                        esq[p] = e * e;  // :294
                    }
                }
            }
            };
#ifdef PAGK_EXPERIMENT_LDS_WINDOW
            // EXPERIMENT (never the product; BASELINE.json's north_star prescribes "pyramid levels staged into LDS tiles"):
            // the 32 x 32 quads of img2 that contain every tap of this iteration are staged in LDS with coalesced row
            // loads, and the five samples per pixel read their quad with ds_read_b32 and unpack the bytes (the typed
            // buffer load's free conversion is not available for LDS).  Same arithmetic, same bits; measured against
            // the product's gathers in profiles/r03_lds_window_experiment.log.
            bool staged = false;
            if constexpr (NR == 2 && !MFMA) {
                if (interior && ext_x <= 14.0f && ext_y <= 14.0f) {   // (block-uniform)
                    const int ox = (int)(bx - ext_x), oy = (int)(by - ext_y);
                    for (int k = tid; k < 32 * 32; k += kBlock) {
                        const int r = oy + (k >> 5), c = ox + (k & 31);
                        win[k] = (r < L2.rows && c < L2.cols) ? L2.quad[(uint32_t)(__mul24(r, L2.cols) + c)] : 0u;
                    }
                    __syncthreads();
                    auto tap = [&](float X, float Y) {
                        const Coord cx = prep_coord<false>(X, L2.fcols, L2.fcols_m1), cy = prep_coord<false>(Y, L2.frows, L2.frows_m1);
                        return bilerp(win[((cy.i - oy) << 5) + (cx.i - ox)], cx, cy);
                    };
#pragma unroll
                    for (int r = 0; r < NR; r++) {
                        const float X = bx + wx[r], Y = by + wy[r];
                        Five sv;
                        sv.c = tap(X, Y), sv.xp = tap(X + 1.0f, Y), sv.xm = tap(X - 1.0f, Y);
                        sv.yp = tap(X, Y + 1.0f), sv.ym = tap(X, Y - 1.0f);
                        const int p = tid + kBlock * r;
                        if (p < P) {
                            float e = sv.c + db - gain * s1[r];
                            float Ix = 0.5f * (sv.xp - sv.xm), Iy = 0.5f * (sv.yp - sv.ym);
                            double dIx = (double)Ix, dIy = (double)Iy, de = (double)e;
                            stream[0 * PS + p] = dIx * dIx, stream[1 * PS + p] = dIy * dIx, stream[2 * PS + p] = dIy * dIy;
                            stream[3 * PS + p] = dIx * de, stream[4 * PS + p] = dIy * de;
                            stream[5 * PS + p] = dIx, stream[6 * PS + p] = dIy, stream[7 * PS + p] = de;
                            esq[p] = e * e;
                        }
                    }
                    staged = true;
                }
            }
            if (staged) {
            } else
#endif
            using AllRounds = std::integral_constant<int, NR>;
            if constexpr (MFMA) {
                sampling(std::integral_constant<int, 2>{}, AllRounds{});
            } else if constexpr (HC != 0) {
                // two-round patches: a wave whose 64 round-1 pixels all lie past the patch (wave 3 at h = 10) skips that
                // round instead of sampling shadow pixels nobody stores (wave-uniform)
                const bool one_round = wave * 64 + kBlock >= P;
                if (one_round) {
                    if (interior)
                        sampling(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
                    else
                        sampling(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
                } else if (interior) {
                    sampling(std::integral_constant<int, 0>{}, AllRounds{});
                } else {
                    sampling(std::integral_constant<int, 1>{}, AllRounds{});
                }
            } else if (interior) {
                sampling(std::integral_constant<int, 0>{}, AllRounds{});
            } else {
                sampling(std::integral_constant<int, 1>{}, AllRounds{});
            }
#ifdef PAGK_STAMPS
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            STAMP(10)  // interpolation, products, LDS stores done in this wave
#endif
            __syncthreads();
            STAMP(1)
            // the dependent chain and the solve are a feature's critical path: while they run, this wave wins
            // issue arbitration over co-resident waves that are sampling (s_setprio; arithmetic untouched).
            // Measured A/B in one session (tools/ab_lib.py): -1 % at 1000 features, -2.4 % at 4000.
            PRIO(PAGK_PRIO_N_CHAIN)
            // ---- 2. ordered accumulation (:284-299) -------------------------------------------
            if constexpr (MFMA) {
                if (wave == 0) {
                    // one dependent MFMA per four pixels; operands prefetched one group of kU ahead
                    constexpr int kU = 4;
                    double d = 0.0;
                    const int M = (P + 3) >> 2, Mfull = P >> 2;  // PP >= 4 * M: reads past P stay inside the arrays
                    double an[kU], bn[kU];
                    const double *pa = a_src, *pb = b_src;
#pragma unroll
                    for (int u = 0; u < kU; u++) {
                        an[u] = *pa;
                        bn[u] = *pb;
                        pa += a_step;
                        pb += b_step;
                    }
                    int m = 0;
                    for (; m + kU <= Mfull; m += kU) {
                        double a0[kU], b0[kU];
#pragma unroll
                        for (int u = 0; u < kU; u++) {
                            a0[u] = an[u];
                            b0[u] = bn[u];
                        }
#pragma unroll
                        for (int u = 0; u < kU; u++) {  // next group (may run past Mfull: loaded, never used)
                            an[u] = *pa;
                            bn[u] = *pb;
                            pa += a_step;
                            pb += b_step;
                        }
#pragma unroll
                        for (int u = 0; u < kU; u++) d = __builtin_amdgcn_mfma_f64_4x4x4f64(a0[u], b0[u], d, 0, 0, 0);
                    }
                    // an/bn hold groups m .. m+kU-1
                    const int rem = Mfull - m;
#pragma unroll
                    for (int u = 0; u < kU; u++)
                        if (u < rem) d = __builtin_amdgcn_mfma_f64_4x4x4f64(an[u], bn[u], d, 0, 0, 0);
                    if (M > Mfull) {
                        // last group: pixels past the patch contribute fma(-0.0, 1.0, d) = d exactly
                        const bool pad = 4 * Mfull + mk >= P;
                        const double av = a_src[(size_t)Mfull * a_step], bv = b_src[(size_t)Mfull * b_step];
                        d = __builtin_amdgcn_mfma_f64_4x4x4f64(pad ? -0.0 : av, pad ? 1.0 : bv, d, 0, 0, 0);
                    }
                    // D(q, i, j) sits in lane 16 i + 4 q + j
                    const int di = lane >> 4, dq = (lane >> 2) & 3, dj = lane & 3;
                    if (dq == 0) acc[di * 4 + dj] = d;       // H
                    if (dq == 1 && dj == 0) acc[16 + di] = d;  // b
                } else {
                    float c = chain_rows_f32<TAIL>(row_addr, row_inc, nfull);
                    if (lane == 0) sh_cost[0] = c;
                }
            } else if constexpr (RELAXED) {
                if (wave < 3) {
                    // H22 = sum of c * c: the row re-reads one constant, so its sum is P * c * c
                    double s = cid < 11 ? relaxed_row_f64(stream + (size_t)stream_of[cid] * PS, P, lr, row_s1)
                                        : (double)P * cd * cd;
                    if (lr == 15) acc[cid] = s;
                } else {
                    float c = relaxed_row_f32(esq, P, lr);
                    if (lane == 15) sh_cost[0] = c;
                }
            } else if (wave < 2) {
                // (two-round patches: the block count is a compile-time constant -> the fully unrolled routine)
                double s;
                if constexpr (HC != 0)
                    s = chain_rows_f64_unrolled<(2 * HC + 1) * (2 * HC + 1) / 32, TAIL>(row_addr, row_s1);
                else
                    s = chain_rows_f64<TAIL>(row_addr, row_inc, nfull, row_s1);
                // slots as the solve reads them: H00 H10 H11 b0 b1 H20 H21 H30 H31 b2 b3 H22
                if (lr == 0) acc[cid == 7 ? 9 : cid] = s;              // lane 0: H00 H10 H11 b0 | b1 H20 H21 b2
                if (lr == 1 && cid >= 5) acc[cid == 7 ? 10 : cid + 2] = s;  // lane 1 of rows X Y E: H30 H31 b3
            } else if (wave == 2) {
                float c;
                if constexpr (HC != 0)
                    c = chain_rows_f32_unrolled<(2 * HC + 1) * (2 * HC + 1) / 32, TAIL>(row_addr);
                else
                    c = chain_rows_f32<TAIL>(row_addr, row_inc, nfull);
                if (lane == 0) sh_cost[0] = c;
            } else if (iter == iter_first) {
                // H22 = the ordered sum of P copies of c*c: the same value in every iteration of this level
                double s = chain_rows_f64<TAIL>(row_addr, row_inc, nfull, row_s1);
                if (lane == 0) acc[11] = s;
            }
            __syncthreads();
            STAMP(2)
            // ---- 3. solve (:302-319) ------------------------------------------------------------
            if (tid < 4) {
                double H[4][4], b[4], upd[4];
                if constexpr (MFMA) {
                    for (int r = 0; r < 4; r++)
                        for (int c = 0; c <= r; c++) H[r][c] = acc[r * 4 + c];
                    for (int r = 0; r < 4; r++) b[r] = acc[16 + r];
                } else {
                    H[0][0] = acc[0];
                    H[1][0] = acc[1];
                    H[1][1] = acc[2];
                    H[2][0] = acc[5];
                    H[2][1] = acc[6];
                    H[2][2] = acc[11];
                    H[3][0] = acc[7];
                    H[3][1] = acc[8];
                    H[3][2] = (double)P * cd;  // sum of c*1.0: every partial sum k*c is exact
                    H[3][3] = (double)P;       // sum of 1.0*1.0
                    b[0] = acc[3];
                    b[1] = acc[4];
                    b[2] = acc[9];
                    b[3] = acc[10];
                }
                float cost = sh_cost[0];
                if constexpr (!LEAN) {
                    if (a.penalty) add_penalty(a, dx, dy, H, b, cost);
                }
                // four lanes share the divides of each Cholesky column (pagk_device.h); all end with the result
                double unorm = llt4_solve_nsq_lanes(H, b, tid, upd, LEAN ? 0u : a.solver);  // update.squaredNorm()
#ifdef PAGK_COUNT_REDO
                if (a.dbg) {
                    OperandRange rg;
                    double xx[4];
                    llt4_solve_nsq_lanes_form<true>(H, b, tid, xx, a.solver, rg);
                    if (tid == 0) {
                        atomicAdd(a.dbg, 1ull);
                        if (!rg.in_range()) {
                            atomicAdd(a.dbg + 1, 1ull);
                            a.dbg[4] = rg.t;
                            for (int r = 0; r < 4; r++) {
                                for (int c = 0; c <= r; c++) a.dbg[8 + r * 4 + c] = __double_as_longlong(H[r][c]);
                                a.dbg[24 + r] = __double_as_longlong(b[r]);
                            }
                        }
                    }
                }
#endif
                if (tid == 0) {
                    sh_upd[0] = upd[0];
                    sh_upd[1] = upd[1];
                    sh_upd[2] = upd[2];
                    sh_upd[3] = upd[3];
                    sh_upd[4] = unorm;
                    sh_cost[1] = cost;
                }
            }
            __syncthreads();
            PRIO(PAGK_PRIO_N_REST)
            STAMP(3)
            // ---- 4. update + termination, identically in every lane (:322-344) -----------------
            // (the compiler sinks these reads behind the exit tests that precede their first use: three LDS round trips;
            // forcing one batch was measured and changes nothing, profiles/r03_ab12_update_reads.log)
            const double u0 = sh_upd[0], u1 = sh_upd[1], u2 = sh_upd[2], u3 = sh_upd[3], unorm = sh_upd[4];
            const float cost = sh_cost[1];
            if (u0 != u0) {  // :322
                succ = 0;
                break;
            }
            if (iter > 0 && cost > lastCost) break;  // :328
            dx = (float)((double)dx + u0);           // :332
            dy = (float)((double)dy + u1);
            if (a.illum) {  // :334-337
                dg = (float)((double)dg + u2);
                db = (float)((double)db + u3);
            }
            lastCost = cost;  // :339
            succ = 1;
            if (unorm < kNormSqConverged) break;  // :343  update.norm() < 1e-2
#ifdef PAGK_STAMPS
            asm volatile("" : "+v"(dx), "+v"(dy), "+v"(dg), "+v"(db));
            STAMP(12)  // update read back, applied, termination tests
#endif
        }
        p2x = ptx + dx;  // :348
        p2y = pty + dy;
        // the next level's first sampling pass overwrites the streams: every lane left the loop
        // after the same barrier, and cslot is rewritten before the next pass's first barrier.
    }
    PAGK_PRIO_RESET
    float ncc = 1.0f;  // :365
    if (a.calc_ncc) {
        // PatchMatch::NCC (:433-469) on the level-0 images at the final point.  Same structure as an
        // iteration: every lane samples its pixels, the f32 sums run in the reference's order (x
        // outer, y inner: sample k of the NCC order is patch pixel (x, y) = (k / Wd - h, k % Wd - h))
        // as DPP row chains.  Five f32 arrays of PP floats reuse the stream region.
        float *vref = reinterpret_cast<float *>(stream), *vcur = vref + PP;
        float *tnum = vcur + PP, *td1 = tnum + PP, *td2 = td1 + PP;
        const DevLevel &R0 = a.l1[0], &C0 = a.l2[0];
        float vr[NR], vc[NR];
        __syncthreads();  // the last iteration's readers of the streams are done
#pragma unroll
        for (int r = 0; r < NR; r++) {
            int k = tid + kBlock * r;
            k = k < P ? k : P - 1;
            int xi = k / Wd - h, yi = k - (k / Wd) * Wd - h;  // x outer, y inner
            vr[r] = sample<true>(R0, refx + xi, refy + yi);   // :440
            if (a.use_affine) {
                float wxx = A00 * xi + A01 * yi, wyy = A10 * xi + A11 * yi;  // :449-450
                vc[r] = sample<true>(C0, p2x + wxx, p2y + wyy);
            } else {
                vc[r] = sample<true>(C0, p2x + xi, p2y + yi);  // :447
            }
            if (tid + kBlock * r < P) {
                vref[tid + kBlock * r] = vr[r];
                vcur[tid + kBlock * r] = vc[r];
            }
        }
        __syncthreads();
        const int row = lane >> 4;
        float *sh_f = reinterpret_cast<float *>(acc);  // 8 floats of scratch
        if (wave == 0) {  // rows 0/1: mean_ref, mean_cur (:441, :453); rows 2/3 shadow row 0
            float m = chain_rows_f32<TAIL>(lds_off(row == 1 ? vcur : vref) + 8u * lr, 128u, nfull);
            if (lr == 0 && row < 2) sh_f[row] = m;
        }
        __syncthreads();
        const float mean_ref = sh_f[0] / (float)P, mean_cur = sh_f[1] / (float)P;  // :457-458
#pragma unroll
        for (int r = 0; r < NR; r++) {
            int k = tid + kBlock * r;
            if (k < P) {
                float dr = vr[r] - mean_ref, dc = vc[r] - mean_cur;
                tnum[k] = dr * dc;  // :463
                td1[k] = dr * dr;   // :464
                td2[k] = dc * dc;   // :465
            }
        }
        __syncthreads();
        if (wave == 0) {
            const float *src = row == 0 ? tnum : (row == 1 ? td1 : td2);
            float v = chain_rows_f32<TAIL>(lds_off(src) + 8u * lr, 128u, nfull);
            if (lr == 0 && row < 3) sh_f[2 + row] = v;
        }
        __syncthreads();
        // numerator / std::sqrt(d1 * d2 + 1e-10): float product, double sum / sqrt / divide (:468)
        ncc = (float)((double)sh_f[2] / sqrt((double)(sh_f[3] * sh_f[4]) + 1e-10));
    }
    if (tid == 0) write_outputs(a, i, p2x, p2y, succ, lastCost, 1, ncc, iters);
    if (tid == 0 && kPrioByWork && !resume) prio_account(a, i, iters, a.n_levels);
#ifdef PAGK_STAMPS
    if (tid == 0 && a.dbg) {
        // resumed features: after the throughput kernel's per-wave records
        unsigned long long *dbgp = a.dbg + (resume ? (size_t)16 * (a.susp_waves > 0 ? a.susp_waves : (a.n + 3) / 4) : 0);
        st[5] = __builtin_amdgcn_s_memtime() - t_begin;
        for (int k = 0; k < 6; k++) dbgp[(size_t)i * 16 + k] = st[k];
        dbgp[(size_t)i * 16 + 6] = (unsigned long long)iters;
        dbgp[(size_t)i * 16 + 7] = rt_begin;  // 100 MHz wall clock, common to all XCDs (s_memtime is per XCD)
        for (int k = 8; k < 11; k++) dbgp[(size_t)i * 16 + k] = st[k];
        dbgp[(size_t)i * 16 + 11] = __builtin_amdgcn_s_memrealtime();
        dbgp[(size_t)i * 16 + 12] = st[12];
        dbgp[(size_t)i * 16 + 13] = st[13];
    }
#endif
#undef STAMP
}

// second argument = waves per SIMD the register allocation must allow: 4 (<= 128 VGPRs).  For the 4-wave kernel that is
// four workgroups per CU, i.e. all 1000 features of BASELINE configs[1] resident at once on 256 CUs (checked against
// the code object by __graft_entry__.build()).
template <int NR, int TAIL, int WAVES = 4, bool MFMA = false, bool RELAXED = false, bool LEAN = false>
__global__ void __launch_bounds__(WAVES * 64, 4) k_track_block(TrackArgs a)
{
    track_block_body<NR, TAIL, WAVES, MFMA, RELAXED, LEAN>(a, (int)blockIdx.x);
}

// The same 4-wave kernel compiled for FIVE waves per SIMD (96 VGPRs, 13 spilled at h = 10): five workgroups per CU
// instead of four.  For launches of several rounds of workgroups (2500 features and more), where the launch is
// throughput-bound: -2.5 % at 3000 features, -3 % at 4000, -3.5 % at 5000; nothing at 2000, and at 1000 (all features
// resident at four per CU) the spills would only cost (profiles/r03_ab23_five_workgroups_per_cu.log).
template <int NR, int TAIL, bool LEAN = false>
__global__ void __launch_bounds__(256, 5) k_track_block5(TrackArgs a)
{
    track_block_body<NR, TAIL, 4, false, false, LEAN, false>(a, (int)blockIdx.x);
}

// The latency kernel as the second pass of a large launch: finishes the features a throughput kernel suspended
// (TrackArgs::iter_budget).  A fixed grid walks the list; *susp_count is read on the device, so the launch is the
// same whatever the count (graph-capturable).
__device__ __forceinline__ SuspState load_susp_state(const TrackArgs &a, int i)
{
    const int *p = reinterpret_cast<const int *>(&a.susp_state[i]);
    SuspState st;
    st.level = ld_agent(p + 0), st.iter = ld_agent(p + 1);
    st.dx = __int_as_float(ld_agent(p + 2)), st.dy = __int_as_float(ld_agent(p + 3));
    st.dg = __int_as_float(ld_agent(p + 4)), st.db = __int_as_float(ld_agent(p + 5));
    st.lastCost = __int_as_float(ld_agent(p + 6)), st.iters = ld_agent(p + 7);
    return st;
}

// The finisher that runs BESIDE the throughput kernel (another stream): each workgroup draws a ticket k and waits for
// list entry k to be published, finishes that feature with the 4-wave body and draws again.  It ends when the
// throughput launch has ended (susp_count[2] == susp_waves) and its entry was not published -- or after susp_polls
// looks, whatever happened: a bounded wait, k_track_resume sweeps up behind it.
template <int NR, int TAIL, bool LEAN = false>
__global__ void __launch_bounds__(256, 4) k_track_resume_live(TrackArgs a)
{
    __shared__ int s_entry;
    const int tid = threadIdx.x;
    int polls = 0;
    for (;;) {
        if (tid == 0) {
            int entry = 0;
            const int ticket = atomicAdd(a.susp_count + 1, 1);
            if (ticket < a.n) {
                for (;;) {
                    entry = __hip_atomic_load(a.susp_list + ticket, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                    if (entry != 0) break;
                    if (ld_agent(a.susp_count + 2) >= a.susp_waves) {  // the producers are gone: a last look
                        entry = __hip_atomic_load(a.susp_list + ticket, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                    if (++polls > a.susp_polls) break;
                    __builtin_amdgcn_s_sleep(64);
                }
            }
            s_entry = entry > 0 ? ticket + 1 : 0;
        }
        __syncthreads();
        const int slot = s_entry - 1;
        if (slot < 0) return;  // (uniform)
        const int i = ld_agent(a.susp_list + slot) - 1;
        const SuspState st = load_susp_state(a, i);
        track_block_body<NR, TAIL, 4, false, false, LEAN>(a, i, &st);
        __syncthreads();  // LDS and s_entry are reused
        if (tid == 0) st_agent(a.susp_list + slot, -(i + 1));
    }
}

// The sweep after both: every published entry that is still waiting (all of them when the live finisher is not used).
// A fixed grid walks the list; the count is read on the device, so the launch is the same whatever it is.
template <int NR, int TAIL, bool LEAN = false>
__global__ void __launch_bounds__(256, 4) k_track_resume(TrackArgs a)
{
    const int count = *a.susp_count;
    for (int b = (int)blockIdx.x; b < count; b += (int)gridDim.x) {
        const int entry = a.susp_list[b];
        if (entry <= 0) continue;  // finished by the live finisher
        const int i = entry - 1;
        const SuspState st = load_susp_state(a, i);
        track_block_body<NR, TAIL, 4, false, false, LEAN>(a, i, &st);
        __syncthreads();  // LDS is reused by the next feature
    }
}

// The 4-wave kernel with the NEXT frame's pyramid built by trailing workgroups of the same launch: blocks
// [0, n) are features, blocks [n, n + pyramid blocks) run k_pyramid_fused's body on another frame slot.  For
// pipelines that already hold frame k+1 while pair (k-1, k) is tracked (replays, or a camera loop that accepts
// one frame of latency): the pyramid then costs no launch of its own and runs in the tracking launch's shadow.
template <int NR, int TAIL, bool LEAN = false>
__global__ void __launch_bounds__(256, 4) k_track_block_pyr(TrackArgs a, PyrArgs pa)
{
    if ((int)blockIdx.x >= a.n) {
        pyr_block(pa, (int)blockIdx.x - a.n, (int)threadIdx.x);
        return;
    }
    track_block_body<NR, TAIL, 4, false, false, LEAN>(a, (int)blockIdx.x);
}

}  // namespace pagk

#include "pagk_wave_kernel.h"
#include "pagk_quad_kernel.h"
#ifdef PAGK_ALL_VARIANTS
#include "pagk_rows_kernel.h"
#endif
#include "pagk_score_kernel.h"
#include "pagk_neighbor_kernel.h"
#include "pagk_selftest_kernel.h"
