/*
 * pagk_oracle.c -- CPU restatement of the reference's PatchMatch hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see pagk_oracle.h).  PARITY UNPINNED (see pagk_oracle.h
 * and oracle/README.md): written from the reference's source text, not checked
 * against reference outputs, because the reference cannot be built here.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (oracle/Makefile).  The reference
 * is built with g++ -O3 for baseline x86-64 (CMakeLists.txt:10-11, no -march), i.e.
 * SSE2 scalar float/double arithmetic, one IEEE rounding per operation, no FMA.
 * -ffp-contract=off reproduces exactly that on any host.
 *
 * All "file:line" citations are relative to /root/reference.
 */
#include "pagk_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------
 * libm log -- src/patch_match.cpp:305 calls std::log(double).  glibc's result is not
 * bit-reproducible across hosts (ifunc FMA / non-FMA variants) nor available on the
 * device, so oracle and product each carry this fixed sequence of IEEE double
 * operations (the classic k*ln2 + log1p(f) reduction with the 7-term minimax series in
 * s = f/(2+f); error < 1 ulp).  Valid for finite x >= 2^-1022; inf/NaN pass through.
 * ---------------------------------------------------------------------------------------------- */
double pagk_oracle_log(double x)
{
    static const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                        Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
                        Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                        Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                        Lg7 = 1.479819860511658591e-01;
    if (!(x < INFINITY)) return x; /* +inf, NaN */
    if (x <= 0.0) return x == 0.0 ? -INFINITY : NAN;
    uint64_t u;
    memcpy(&u, &x, 8);
    int k = (int)(u >> 52) - 1023;
    u = (u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL; /* m in [1,2) */
    double m;
    memcpy(&m, &u, 8);
    if (m > 1.4142135623730951) {
        m = m * 0.5;
        k += 1;
    }
    double f = m - 1.0;
    double s = f / (2.0 + f);
    double z = s * s;
    double w = z * z;
    double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    double R = t2 + t1;
    double hfsq = 0.5 * f * f;
    double dk = (double)k;
    return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
}

/* src/patch_match.cpp:48-51:  mInvLogMaxDist = 1.0 / (std::log(mAlpha * mMaxDistance + 1));
 * float*int -> float, +1 -> float, std::log(float) -> float overload, 1.0/float -> double,
 * stored into a float member (include/patch_match.h:88). */
float pagk_oracle_inv_log_max_dist(float alpha, int32_t max_distance)
{
    float arg = alpha * (float)max_distance + 1;
    float lg = logf(arg);
    return (float)(1.0 / (double)lg);
}

/* ------------------------------------------------------------------------------------------------
 * Images.  A level is a byte buffer addressed linearly the way the reference's sampler
 * addresses cv::Mat::data (src/patch_match.cpp:399-403): taps at +1, +step, +step+1 of a
 * clamped pixel can run past the row end (next row's first byte) and, on the last row,
 * past the buffer.  The reference reads heap garbage there; oracle and product define
 * every byte at offset >= rows*step as 0.
 * ---------------------------------------------------------------------------------------------- */
/* tools/parity_risk.py: the choices this restatement had to make about third-party arithmetic, switchable one at
 * a time so that their effect on the results can be measured (oracle/README.md, "How much each guess matters").
 * 0 = the restatement as documented.  Never set by tests or by the benchmark. */
static uint32_t g_alt = 0;
enum {
    ALT_LOWER_SEQ = 1,   /* lower solve, row 3: (c0 + c1) + c2 instead of c0 + (c1 + c2)           */
    ALT_UPPER_TREE = 2,  /* upper solve, row 0: c0 + (c1 + c2) instead of (c0 + c1) + c2           */
    ALT_NORM_SEQ = 4,    /* norm: ((x0^2 + x1^2) + x2^2) + x3^2 instead of the SSE2 packet shape   */
    ALT_LLT_RECIP = 8,   /* Eigen <= 3.2: A21 *= 1/x instead of A21 /= x                           */
    ALT_PYR_LINEAR = 16, /* exact 2x decimation through the 11-bit fixed-point bilinear path       */
    ALT_PIVOT_TREE = 32  /* 4th pivot: A33 - (a0^2 + (a1^2 + a2^2)) instead of the sequential sum  */
};
void pagk_oracle_set_alternatives(uint32_t flags) { g_alt = flags; }

typedef struct {
    const uint8_t *data;
    int cols, rows;
    int64_t step;
    int64_t size; /* rows*step */
} level_t;

static inline float tap(const level_t *im, int64_t off)
{
    /* uchar -> int -> float, as in  a * data[0]  (:402).  Row padding (columns in
     * [cols, step) of a non-continuous Mat) is defined as 0 as well: the product copies
     * only the `cols` valid bytes of each row to the device. */
    if (off < 0 || off >= im->size) return 0.0f;
    if (im->step != im->cols && (off % im->step) >= im->cols) return 0.0f;
    return (float)(int)im->data[off];
}

/* PatchMatch::GetPixelValue, src/patch_match.cpp:391-406 (the member, not the free
 * function of include/utils.h:32-46). */
static inline float get_pixel_value(const level_t *img, float x, float y)
{
    /* `if (x < 0) x = 0;` written so that a NaN coordinate also becomes 0: int(NaN) is
     * undefined in the reference, and NaN cannot reach this point from finite inputs (a
     * NaN update breaks at :322 before being applied).  Identical for every non-NaN x
     * (-0.0 stays -0.0 in both forms). */
    if (!(x >= 0)) x = 0;
    if (!(y >= 0)) y = 0;
    if (x >= img->cols) x = img->cols - 1;
    if (y >= img->rows) y = img->rows - 1;
    int ix = (int)x;
    int iy = (int)y;
    int64_t off = (int64_t)iy * img->step + ix;
    float xx = x - floorf(x), yy = y - floorf(y);
    float a = 1.0f - xx, b = 1.0f - yy;
    float pixel = b * (a * tap(img, off) + xx * tap(img, off + 1)) +
                  yy * (a * tap(img, off + img->step) + xx * tap(img, off + img->step + 1));
    return pixel;
}

/* cv::resize(src, dst, Size(cols*0.5, rows*0.5)), src/patch_match.cpp:69-70, default
 * INTER_LINEAR.  THIRD PARTY (OpenCV >= 3.4, not vendored): for an exact 2x decimation
 * of 8UC1, resize() switches INTER_LINEAR to the INTER_AREA fast path, whose 8-bit
 * kernel is (s00 + s01 + s10 + s11 + 2) >> 2.  Odd parent sizes take OpenCV's 11-bit
 * fixed-point bilinear path (second half of the function). */
int pagk_oracle_pyr_down(const uint8_t *src, int32_t w, int32_t h, int64_t step, uint8_t *dst)
{
    if (!src || !dst || w < 2 || h < 2 || step < w) return PAGK_E_ARG;
    /* :69  cv::Size(cols * 0.5, rows * 0.5): int * double, truncated */
    const int dw = (int)(w * 0.5), dh = (int)(h * 0.5);
    if (!((w & 1) || (h & 1)) && !(g_alt & 16 /* ALT_PYR_LINEAR */)) { /* scale exactly 2 in x and y: INTER_AREA fast path */
        for (int y = 0; y < dh; y++) {
            const uint8_t *r0 = src + (int64_t)(2 * y) * step, *r1 = r0 + step;
            for (int x = 0; x < dw; x++)
                dst[(int64_t)y * dw + x] =
                    (uint8_t)((r0[2 * x] + r0[2 * x + 1] + r1[2 * x] + r1[2 * x + 1] + 2) >> 2);
        }
        return PAGK_OK;
    }
    /* Any other size: OpenCV's INTER_LINEAR for 8-bit images, 11-bit fixed-point coefficients
     * (imgproc/resize.cpp of OpenCV 3.4, restated FROM MEMORY -- parity unpinned, see oracle/README.md):
     *   scale = 1 / (dsize / (double)ssize);  f = (float)((d + 0.5) * scale - 0.5);  s = cvFloor(f);  f -= s;
     *   s < 0 -> (f, s) = (0, 0);   s >= ssize - 1 -> (f, s) = (0, ssize - 1), single tap;
     *   coefficients saturate_cast<short>((1 - f) * 2048), saturate_cast<short>(f * 2048)  (cvRound: nearest even);
     *   horizontal pass in int: S[s] * a0 + S[s + 1] * a1  (a single tap is S[s] * 2048);
     *   vertical pass (VResizeLinear<uchar, int, short>): ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2,
     *   the second row index clipped to the last row. */
    const double scale_x = 1. / ((double)dw / w), scale_y = 1. / ((double)dh / h);
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = (int)floorf(fy);
        fy -= sy;
        const int b0 = (int)lrintf((1.f - fy) * 2048.f), b1 = (int)lrintf(fy * 2048.f);
        /* rows sy and sy + 1, each clipped into [0, h - 1] */
        int y0 = sy < 0 ? 0 : (sy < h ? sy : h - 1);
        int y1 = sy + 1 < 0 ? 0 : (sy + 1 < h ? sy + 1 : h - 1);
        const uint8_t *r0 = src + (int64_t)y0 * step, *r1 = src + (int64_t)y1 * step;
        for (int dx = 0; dx < dw; dx++) {
            float fx = (float)((dx + 0.5) * scale_x - 0.5);
            int sx = (int)floorf(fx);
            fx -= sx;
            int single = 0;
            if (sx < 0) fx = 0, sx = 0;
            if (sx + 1 >= w) { /* dx >= xmax */
                single = 1;
                if (sx >= w - 1) fx = 0, sx = w - 1;
            }
            const int a0 = (int)lrintf((1.f - fx) * 2048.f), a1 = (int)lrintf(fx * 2048.f);
            int S0, S1;
            if (single) {
                S0 = r0[sx] * 2048;
                S1 = r1[sx] * 2048;
            } else {
                S0 = r0[sx] * a0 + r0[sx + 1] * a1;
                S1 = r1[sx] * a0 + r1[sx + 1] * a1;
            }
            dst[(int64_t)dy * dw + dx] = (uint8_t)((((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2);
        }
    }
    return PAGK_OK;
}

/* ------------------------------------------------------------------------------------------------
 * Eigen::Matrix4d::llt().solve(b) and Vector4d::norm(), src/patch_match.cpp:319,343.
 * THIRD PARTY (Eigen >= 3.1; Ubuntu 18.04 ships 3.3.4).  Restated from Eigen 3.3:
 *   - LLT<Matrix4d,Lower>::compute -> llt_inplace<double,Lower>::unblocked (size < 32):
 *       for k: x = A(k,k) - sum_j<k A(k,j)^2   (sequential, dynamic-size redux)
 *              if (x <= 0) return k            (info = NumericalIssue; matrix left as is)
 *              A(k,k) = x = sqrt(x)
 *              A(i,k) -= sum_j<k A(i,j)*A(k,j) (coefficient-based lazy product, sequential)
 *              A(i,k) /= x                     (3.3: true division, not *1/x)
 *   - solve() does not look at info(): both triangular solves run on whatever is in the
 *     lower triangle, so a failed pivot leaves the ORIGINAL H entries in place from that
 *     column on (for a flat patch: division by H(0,0) = 0 -> NaN, the :322 case).
 *   - the solves are the fully unrolled fixed-size ones (triangular_solver_unroller):
 *       lower: rhs[i] -= sum(L(i,0..i-1) .* rhs[0..i-1]);  rhs[i] /= L(i,i)
 *       upper: rhs[i] -= sum(L(i+1..3,i) .* rhs[i+1..3]);  rhs[i] /= L(i,i)
 *     with the fixed-size redux association: the row access of the lower solve is not
 *     vectorisable (tree: c0 + (c1 + c2)); the column access of the upper solve is
 *     ((c0 + c1) + c2).
 *   - norm(): sqrt of the SSE2-vectorised squaredNorm: (x0^2 + x2^2) + (x1^2 + x3^2).
 * ---------------------------------------------------------------------------------------------- */
static void llt4_inplace(double M[4][4])
{
    for (int k = 0; k < 4; k++) {
        double x = M[k][k];
        if (k > 0) {
            double s = M[k][0] * M[k][0];
            if (k == 3 && (g_alt & ALT_PIVOT_TREE))
                s += M[k][1] * M[k][1] + M[k][2] * M[k][2];
            else
                for (int j = 1; j < k; j++) s += M[k][j] * M[k][j];
            x -= s;
        }
        if (x <= 0.0) return;
        M[k][k] = x = sqrt(x);
        const double rx = 1.0 / x;
        for (int i = k + 1; i < 4; i++) {
            if (k > 0) {
                double s = M[i][0] * M[k][0];
                for (int j = 1; j < k; j++) s += M[i][j] * M[k][j];
                M[i][k] -= s;
            }
            if (g_alt & ALT_LLT_RECIP)
                M[i][k] *= rx;
            else
                M[i][k] /= x;
        }
    }
}

static double llt4_solve_norm(double M[4][4], const double b[4], double x[4])
{
    llt4_inplace(M);
    double r0 = b[0], r1 = b[1], r2 = b[2], r3 = b[3];
    /* L y = b */
    r0 /= M[0][0];
    r1 -= M[1][0] * r0;
    r1 /= M[1][1];
    r2 -= M[2][0] * r0 + M[2][1] * r1;
    r2 /= M[2][2];
    if (g_alt & ALT_LOWER_SEQ)
        r3 -= (M[3][0] * r0 + M[3][1] * r1) + M[3][2] * r2;
    else
        r3 -= M[3][0] * r0 + (M[3][1] * r1 + M[3][2] * r2);
    r3 /= M[3][3];
    /* L^T x = y */
    r3 /= M[3][3];
    r2 -= M[3][2] * r3;
    r2 /= M[2][2];
    r1 -= M[2][1] * r2 + M[3][1] * r3;
    r1 /= M[1][1];
    if (g_alt & ALT_UPPER_TREE)
        r0 -= M[1][0] * r1 + (M[2][0] * r2 + M[3][0] * r3);
    else
        r0 -= (M[1][0] * r1 + M[2][0] * r2) + M[3][0] * r3;
    r0 /= M[0][0];
    x[0] = r0;
    x[1] = r1;
    x[2] = r2;
    x[3] = r3;
    if (g_alt & ALT_NORM_SEQ) return sqrt(((r0 * r0 + r1 * r1) + r2 * r2) + r3 * r3);
    return sqrt((r0 * r0 + r2 * r2) + (r1 * r1 + r3 * r3));
}

double pagk_oracle_llt_solve4(const double *H, const double *b, double *x)
{
    double M[4][4];
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) M[r][c] = H[r * 4 + c];
    return llt4_solve_norm(M, b, x);
}

/* ------------------------------------------------------------------------------------------------
 * The tracker state PatchMatch holds between levels.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    const pagk_params *p;
    int n_levels;
    level_t pyr1[PAGK_MAX_PYRAMIDS], pyr2[PAGK_MAX_PYRAMIDS];
    float scales[PAGK_MAX_PYRAMIDS]; /* mvScales, :66,:73 */
    int n;
    const float *pt_ref_un;  /* mvPtPyr1Un, :84 */
    const float *affine;
    const uint8_t *status_in; /* mvGyroPredictStatus, :58 */
    float *pt2_un;            /* mvPtPyr2Un, :86/:88, updated per level :348 */
    uint8_t *success;         /* mvSuccess */
    double *pix_err;          /* mvPixelErrorsOfPatchMatched */
    float *ncc;               /* mvNcc */
    int32_t *iters;           /* diagnostic */
    float inv_log_max_dist;
    double win_size_inv; /* mWinSizeInv, :57 */
    int level;           /* mLevel */
} pm_state;

/* PatchMatch::NCC, src/patch_match.cpp:433-469.  Loop order x outer, y inner (:438-439);
 * float sums; mean /= size() is float / size_t -> float / float. */
static float ncc_patch(const pm_state *st, int h, const level_t *ref, const level_t *cur, float rx,
                       float ry, float cx, float cy, const float *A)
{
    int P = (2 * h + 1) * (2 * h + 1);
    float *vr = (float *)malloc(sizeof(float) * 2 * (size_t)P), *vc = vr + P;
    float mean_ref = 0.0f, mean_cur = 0.0f;
    int k = 0;
    (void)st;
    for (int x = -h; x <= h; x++)
        for (int y = -h; y <= h; y++) {
            float value_ref = get_pixel_value(ref, rx + x, ry + y);
            mean_ref += value_ref;
            vr[k] = value_ref;
            float value_cur;
            if (!A)
                value_cur = get_pixel_value(cur, cx + x, cy + y);
            else {
                float wx = A[0] * x + A[1] * y;
                float wy = A[2] * x + A[3] * y;
                value_cur = get_pixel_value(cur, cx + wx, cy + wy);
            }
            mean_cur += value_cur;
            vc[k] = value_cur;
            k++;
        }
    mean_ref /= (float)P;
    mean_cur /= (float)P;
    float numerator = 0, den1 = 0, den2 = 0;
    for (int i = 0; i < P; i++) {
        numerator += ((vr[i] - mean_ref) * (vc[i] - mean_cur));
        den1 += (vr[i] - mean_ref) * (vr[i] - mean_ref);
        den2 += (vc[i] - mean_cur) * (vc[i] - mean_cur);
    }
    free(vr);
    /* numerator / std::sqrt(d1*d2 + 1e-10): float*float -> float, + double -> double,
     * sqrt(double), float / double -> double, returned as float (:468). */
    return (float)((double)numerator / sqrt((double)(den1 * den2) + 1e-10));
}

/* PatchMatch::OpticalFlowConsideringIlluminationChange_onePixel, src/patch_match.cpp:167-367 */
static void one_pixel(pm_state *st, int i)
{
    const pagk_params *p = st->p;
    if (!st->status_in[i]) return; /* :173 */

    const int h = p->half_patch;
    const int level = st->level;
    const level_t *img1 = &st->pyr1[level], *img2 = &st->pyr2[level];

    /* :177  Point2f * float */
    float ptx = st->pt_ref_un[2 * i] * st->scales[level];
    float pty = st->pt_ref_un[2 * i + 1] * st->scales[level];
    float nx, ny;
    if (level == st->n_levels - 1) { /* :179-180 */
        nx = st->pt2_un[2 * i] * st->scales[level];
        ny = st->pt2_un[2 * i + 1] * st->scales[level];
    } else { /* :182  Point2f * 1.0f / double(0.5): float mul, then double divide, narrowed */
        nx = (float)((double)(st->pt2_un[2 * i] * 1.0f) / 0.5);
        ny = (float)((double)(st->pt2_un[2 * i + 1] * 1.0f) / 0.5);
    }

    float dx = nx - ptx, dy = ny - pty; /* :186-187 */
    float dg = 0.0f, db = 0.0f;         /* :190-191 */
    float cost = 0.0f, lastCost = 0.0f; /* :193 */
    int succ = 1;                       /* :194 */

    const int W = 2 * h + 1, P = W * W;
    /* :197-209 warp_patch.at(x+h, y+h) = A * (x, y) */
    float *warp = NULL;
    if (p->consider_affine) {
        const float *A = st->affine + 4 * (size_t)i;
        warp = (float *)malloc(sizeof(float) * 2 * (size_t)P);
        for (int x = -h; x <= h; x++)
            for (int y = -h; y <= h; y++) {
                float wx = A[0] * x + A[1] * y;
                float wy = A[2] * x + A[3] * y;
                warp[2 * ((x + h) * W + (y + h))] = wx;
                warp[2 * ((x + h) * W + (y + h)) + 1] = wy;
            }
    }

    double(*vJ)[4] = (double(*)[4])malloc(sizeof(double) * 4 * (size_t)P); /* :226 */
    float *vE = (float *)malloc(sizeof(float) * (size_t)P);                /* :227 */
    int iters_done = 0;

    for (int iter = 0; iter < p->iterations; iter++) { /* :215 */
        double H[4][4], b[4];                          /* :217-218 (forward mode) */
        memset(H, 0, sizeof H);
        memset(b, 0, sizeof b);
        iters_done++;

        int index = 0;
        for (int y = -h; y <= h; y++)       /* :233 */
            for (int x = -h; x <= h; x++) { /* :234 */
                float wx = x, wy = y;       /* :235 */
                if (warp) {                 /* :236-239 */
                    wx = warp[2 * ((x + h) * W + (y + h))];
                    wy = warp[2 * ((x + h) * W + (y + h)) + 1];
                }
                /* :252-253 */
                float error = get_pixel_value(img2, ptx + dx + wx, pty + dy + wy) + db -
                              (1.0f + dg) * get_pixel_value(img1, ptx + x, pty + y);
                /* :259-262  double 0.5 * float difference, narrowed to float */
                float Ix = (float)(0.5 * (double)(get_pixel_value(img2, ptx + dx + wx + 1, pty + dy + wy) -
                                                  get_pixel_value(img2, ptx + dx + wx - 1, pty + dy + wy)));
                float Iy = (float)(0.5 * (double)(get_pixel_value(img2, ptx + dx + wx, pty + dy + wy + 1) -
                                                  get_pixel_value(img2, ptx + dx + wx, pty + dy + wy - 1)));
                float de_dg = -get_pixel_value(img1, ptx, pty); /* :263 (patch CENTRE) */
                vJ[index][0] = Ix;                              /* :264 */
                vJ[index][1] = Iy;
                vJ[index][2] = de_dg;
                vJ[index][3] = 1;
                vE[index] = error; /* :278 */
                index++;
            }

        cost = 0; /* :283 */
        for (int k = 0; k < P; k++) {
            const double *J = vJ[k];
            double e = (double)vE[k];
            for (int r = 0; r < 4; r++) b[r] += (-J[r]) * e; /* :293  b += -J * vE[i] */
            cost += vE[k] * vE[k];                           /* :294  float */
            for (int r = 0; r < 4; r++)                      /* :296  H += J * J^T */
                for (int c = 0; c < 4; c++) H[r][c] += J[r] * J[c];
        }

        if (p->regularization_penalty) { /* :302-314 */
            /* :304  std::sqrt(float) -> float, widened */
            double d = (double)sqrtf(dx * dx + dy * dy);
            /* :305  (mLambda * mInvLogMaxDist) float, * log(double(mAlpha) * d + 1) */
            double e_penalty =
                (double)(p->lambda * st->inv_log_max_dist) * pagk_oracle_log((double)p->alpha * d + 1);
            /* :307-308  float*float*float / double * (float / double) */
            double jpx = (double)(p->lambda * st->inv_log_max_dist * p->alpha) /
                         ((double)p->alpha * d + 1) * ((double)dx / d);
            double jpy = (double)(p->lambda * st->inv_log_max_dist * p->alpha) /
                         ((double)p->alpha * d + 1) * ((double)dy / d);
            double JP[4] = {jpx, jpy, 0, 0}; /* :310 */
            for (int r = 0; r < 4; r++)
                for (int c = 0; c < 4; c++) H[r][c] += JP[r] * JP[c]; /* :311 */
            for (int r = 0; r < 4; r++) b[r] += JP[r] * e_penalty;    /* :312 (PLUS) */
            cost = (float)((double)cost + e_penalty * e_penalty);     /* :313 */
        }

        double update[4];
        double unorm = llt4_solve_norm(H, b, update); /* :319 */

        if (isnan(update[0])) { /* :322-326 */
            succ = 0;
            break;
        }
        if (iter > 0 && cost > lastCost) break; /* :328-329 */

        dx = (float)((double)dx + update[0]); /* :332  float += double */
        dy = (float)((double)dy + update[1]); /* :333 */
        if (p->consider_illumination) {       /* :334-337 */
            dg = (float)((double)dg + update[2]);
            db = (float)((double)db + update[3]);
        }
        lastCost = cost; /* :339 */
        succ = 1;        /* :340 */
        if (unorm < 1e-2) break; /* :343 */
    }
    free(vJ);
    free(vE);
    free(warp);

    st->pt2_un[2 * i] = ptx + dx; /* :348 */
    st->pt2_un[2 * i + 1] = pty + dy;
    if (st->iters) st->iters[i] += iters_done;

    if (level == 0) { /* :350-353 */
        st->success[i] = (uint8_t)succ;
        st->pix_err[i] = sqrt((double)lastCost * st->win_size_inv);
    }

    if (p->calculate_ncc) { /* :356-363 always on the level-0 images */
        st->ncc[i] = ncc_patch(st, h, &st->pyr1[0], &st->pyr2[0], st->pt_ref_un[2 * i],
                               st->pt_ref_un[2 * i + 1], st->pt2_un[2 * i], st->pt2_un[2 * i + 1],
                               p->consider_affine ? st->affine + 4 * (size_t)i : NULL);
    } else {
        st->ncc[i] = 1; /* :365 */
    }
}

/* cv::parallel_for_(cv::Range(0,mN), ...), src/patch_match.cpp:103-109: contiguous
 * stripes of the index range, one per worker. */
typedef struct {
    pm_state *st;
    int begin, end;
} stripe_t;

static void *stripe_main(void *arg)
{
    stripe_t *s = (stripe_t *)arg;
    for (int i = s->begin; i < s->end; i++) one_pixel(s->st, i);
    return NULL;
}

static void parallel_for(pm_state *st, int nthreads)
{
    if (nthreads <= 1 || st->n < 2 * nthreads) {
        stripe_t s = {st, 0, st->n};
        stripe_main(&s);
        return;
    }
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
    stripe_t *sp = (stripe_t *)malloc(sizeof(stripe_t) * (size_t)nthreads);
    /* small stripes interleaved over workers would balance better, but the reference
     * hands each worker one contiguous range; keep that. */
    for (int t = 0; t < nthreads; t++) {
        sp[t].st = st;
        sp[t].begin = (int)((int64_t)st->n * t / nthreads);
        sp[t].end = (int)((int64_t)st->n * (t + 1) / nthreads);
        pthread_create(&th[t], NULL, stripe_main, &sp[t]);
    }
    for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    free(th);
    free(sp);
}

/* DistortVecPoints, src/utils.cpp:49-76 (called from PatchMatch::DistortPoints,
 * src/patch_match.cpp:409-416). */
static void distort_points(const pagk_params *p, int n, const float *un, float *dist)
{
    if (p->dist_coef[0] == 0.0) { /* :410 */
        memcpy(dist, un, sizeof(float) * 2 * (size_t)n);
        return;
    }
    float mfx = p->fx, mfy = p->fy, mcx = p->cx, mcy = p->cy;
    float mfx_inv = (float)(1.0 / (double)mfx), mfy_inv = (float)(1.0 / (double)mfy); /* utils.cpp:53 */
    float K1 = p->dist_coef[0], K2 = p->dist_coef[1], mp1 = p->dist_coef[2], mp2 = p->dist_coef[3];
    float K3 = p->n_dist_coef == 5 ? p->dist_coef[4] : 0;
    for (int i = 0; i < n; i++) {
        float x = (un[2 * i] - mcx) * mfx_inv;
        float y = (un[2 * i + 1] - mcy) * mfy_inv;
        float r2 = x * x + y * y;
        float r4 = r2 * r2;
        float r6 = r4 * r2;
        float x_distort = x * (1 + K1 * r2 + K2 * r4 + K3 * r6) + 2 * mp1 * x * y + mp2 * (r2 + 2 * x * x);
        float y_distort = y * (1 + K1 * r2 + K2 * r4 + K3 * r6) + mp1 * (r2 + 2 * y * y) + 2 * mp2 * x * y;
        dist[2 * i] = mfx * x_distort + mcx;
        dist[2 * i + 1] = mfy * y_distort + mcy;
    }
}

static int check_params(const pagk_params *p)
{
    if (!p) return PAGK_E_ARG;
    if (p->half_patch < 1 || p->half_patch > PAGK_MAX_HALF_PATCH) return PAGK_E_ARG;
    if (p->iterations < 0 || p->pyramids < 1 || p->pyramids > PAGK_MAX_PYRAMIDS) return PAGK_E_ARG;
    if (p->inverse) return PAGK_E_UNSUPPORTED;
    return PAGK_OK;
}

static int track_levels(const pagk_params *p, int n_levels, const level_t *l1, const level_t *l2, int n,
                        const float *pt_ref_un, const float *pt_init_un, const float *affine,
                        const uint8_t *status_in, const pagk_outputs *out, int nthreads)
{
    if (n < 0 || !out || !out->pt_un || !out->status) return PAGK_E_ARG;
    if (n > 0 && (!pt_ref_un || !status_in)) return PAGK_E_ARG;
    if (n > 0 && p->has_gyro_predict_initial && !pt_init_un) return PAGK_E_ARG;
    if (n > 0 && p->consider_affine && !affine) return PAGK_E_ARG;

    pm_state st;
    memset(&st, 0, sizeof st);
    st.p = p;
    st.n_levels = n_levels;
    for (int l = 0; l < n_levels; l++) {
        st.pyr1[l] = l1[l];
        st.pyr2[l] = l2[l];
        /* :66, :73  mvScales[i] = mvScales[i-1] * mPyramidScale (float * double -> float) */
        st.scales[l] = l == 0 ? 1.0f : (float)((double)st.scales[l - 1] * 0.5);
    }
    st.n = n;
    st.pt_ref_un = pt_ref_un;
    st.affine = affine;
    st.status_in = status_in;
    st.inv_log_max_dist =
        p->inv_log_max_dist != 0.0f ? p->inv_log_max_dist : pagk_oracle_inv_log_max_dist(p->alpha, p->max_distance);
    /* :57  1.0f / (2.0f*h + 1.0f) / (2.0f*h + 1.0f) in float, stored as double */
    st.win_size_inv = (double)(1.0f / (2.0f * p->half_patch + 1.0f) / (2.0f * p->half_patch + 1.0f));

    size_t nn = n > 0 ? (size_t)n : 1;
    st.pt2_un = out->pt_un; /* result vector doubles as mvPtPyr2Un */
    st.success = (uint8_t *)calloc(nn, 1);            /* :93 resize -> false */
    st.pix_err = (double *)calloc(nn, sizeof(double)); /* :94 */
    st.ncc = (float *)calloc(nn, sizeof(float));       /* :95 */
    st.iters = out->iters;
    if (st.iters) memset(st.iters, 0, sizeof(int32_t) * (size_t)n);

    for (int i = 0; i < n; i++) { /* :83-90 */
        const float *src = p->has_gyro_predict_initial ? pt_init_un : pt_ref_un;
        st.pt2_un[2 * i] = src[2 * i];
        st.pt2_un[2 * i + 1] = src[2 * i + 1];
    }

    for (int level = n_levels - 1; level >= 0; level--) { /* :98 */
        st.level = level;
        parallel_for(&st, nthreads);
    }

    /* DistortPoints :116, SetMatcher :119 / :370-388 */
    if (out->pt_dist) distort_points(p, n, st.pt2_un, out->pt_dist);
    for (int i = 0; i < n; i++) {
        out->status[i] = st.success[i];
        if (out->pix_err) out->pix_err[i] = st.pix_err[i];
        if (out->dist_pred) {
            /* :384-385  pt_dist = mvPtPredictUn[i] - mvPtPyr2Un[i] (always the tracker's
             * predicted point, also when has_gyro_predict_initial is false);
             * std::sqrt(float) widened to double */
            const float *pred = pt_init_un ? pt_init_un : pt_ref_un;
            float ddx = pred[2 * i] - st.pt2_un[2 * i];
            float ddy = pred[2 * i + 1] - st.pt2_un[2 * i + 1];
            out->dist_pred[i] = (double)sqrtf(ddx * ddx + ddy * ddy);
        }
        if (out->ncc) out->ncc[i] = st.ncc[i];
    }
    free(st.success);
    free(st.pix_err);
    free(st.ncc);
    return PAGK_OK;
}

static int view_to_level(const pagk_image *im, level_t *lv)
{
    if (!im || !im->data || im->width < 1 || im->height < 1 || im->step < im->width) return PAGK_E_ARG;
    lv->data = im->data;
    lv->cols = im->width;
    lv->rows = im->height;
    lv->step = im->step;
    lv->size = (int64_t)im->height * im->step;
    return PAGK_OK;
}

int pagk_oracle_track_pyr(const pagk_params *params, int32_t n_levels, const pagk_image *ref_levels,
                          const pagk_image *cur_levels, int32_t n, const float *pt_ref_un,
                          const float *pt_init_un, const float *affine, const uint8_t *status_in,
                          const pagk_outputs *out, int32_t nthreads)
{
    int rc = check_params(params);
    if (rc) return rc;
    if (n_levels != params->pyramids || !ref_levels || !cur_levels) return PAGK_E_ARG;
    level_t l1[PAGK_MAX_PYRAMIDS], l2[PAGK_MAX_PYRAMIDS];
    for (int l = 0; l < n_levels; l++) {
        if ((rc = view_to_level(&ref_levels[l], &l1[l])) || (rc = view_to_level(&cur_levels[l], &l2[l])))
            return rc;
    }
    return track_levels(params, n_levels, l1, l2, n, pt_ref_un, pt_init_un, affine, status_in, out, nthreads);
}

/* PatchMatch::CreatePyramids, src/patch_match.cpp:61-76 */
static int build_pyramid(const pagk_image *im, int n_levels, level_t *lv, uint8_t **owned)
{
    int rc = view_to_level(im, &lv[0]);
    if (rc) return rc;
    for (int l = 1; l < n_levels; l++) {
        const level_t *pv = &lv[l - 1];
        /* :69  cv::Size(cols * 0.5, rows * 0.5): int*double truncated */
        int dw = (int)(pv->cols * 0.5), dh = (int)(pv->rows * 0.5);
        if (pv->cols < 2 || pv->rows < 2 || dw < 1 || dh < 1) return PAGK_E_ARG;
        owned[l] = (uint8_t *)malloc((size_t)dw * (size_t)dh);
        if (!owned[l]) return PAGK_E_NOMEM;
        pagk_oracle_pyr_down(pv->data, pv->cols, pv->rows, pv->step, owned[l]);
        lv[l].data = owned[l];
        lv[l].cols = dw;
        lv[l].rows = dh;
        lv[l].step = dw;
        lv[l].size = (int64_t)dw * dh;
    }
    return PAGK_OK;
}

int pagk_oracle_track(const pagk_params *params, const pagk_image *ref, const pagk_image *cur, int32_t n,
                      const float *pt_ref_un, const float *pt_init_un, const float *affine,
                      const uint8_t *status_in, const pagk_outputs *out, int32_t nthreads)
{
    int rc = check_params(params);
    if (rc) return rc;
    level_t l1[PAGK_MAX_PYRAMIDS], l2[PAGK_MAX_PYRAMIDS];
    uint8_t *own1[PAGK_MAX_PYRAMIDS] = {0}, *own2[PAGK_MAX_PYRAMIDS] = {0};
    rc = build_pyramid(ref, params->pyramids, l1, own1);
    if (!rc) rc = build_pyramid(cur, params->pyramids, l2, own2);
    if (!rc)
        rc = track_levels(params, params->pyramids, l1, l2, n, pt_ref_un, pt_init_un, affine, status_in, out,
                          nthreads);
    for (int l = 0; l < PAGK_MAX_PYRAMIDS; l++) {
        free(own1[l]);
        free(own2[l]);
    }
    return rc;
}

/* GyroAidedTracker::GyroPredictFeaturesAndOpticalFlowRefined, Step 3,
 * src/gyro_aided_tracker.cpp:289-341 */
int pagk_oracle_post_filter(int32_t n, int32_t half_patch, const uint8_t *status_pm, const double *pix_err,
                            const double *dist_pred, const float *pt_pm, const float *pt_pm_un,
                            uint8_t *status_out, float *pt_predict, float *pt_predict_un)
{
    if (n < 0 || (n > 0 && (!status_pm || !pix_err || !dist_pred || !status_out))) return PAGK_E_ARG;
    double sumPixelError = 0;
    int cnt = 0;
    for (int i = 0; i < n; ++i) /* :297-304 */
        if (status_pm[i]) {
            sumPixelError += pix_err[i];
            cnt++;
        }
    double avgPixelError = sumPixelError / cnt; /* :305  cnt == 0 -> NaN -> th = h */
    double thPixelError = 4.0 * avgPixelError > half_patch ? 4.0 * avgPixelError : half_patch; /* :308 */
    double thDistance = half_patch * 4.0;                                                      /* :312 */
    int n_predict = 0;
    for (int i = 0; i < n; i++) { /* :318-336 */
        if (status_pm[i] && pix_err[i] < thPixelError && dist_pred[i] < thDistance) {
            if (pt_predict && pt_pm) {
                pt_predict[2 * i] = pt_pm[2 * i];
                pt_predict[2 * i + 1] = pt_pm[2 * i + 1];
            }
            if (pt_predict_un && pt_pm_un) {
                pt_predict_un[2 * i] = pt_pm_un[2 * i];
                pt_predict_un[2 * i + 1] = pt_pm_un[2 * i + 1];
            }
            status_out[i] = 1;
            n_predict++;
        } else {
            status_out[i] = 0;
        }
    }
    return n_predict;
}

/* GyroAidedTracker::GyroPredictOnePixel, PIXEL_AWARE_PREDICTION branch,
 * src/gyro_aided_tracker.cpp:194-231 (mNormalizeTable empty: the apps pass cv::Mat()). */
typedef struct {
    float fx, fy, cx, cy, fx_inv, fy_inv, k1, k2, p1, p2, k3;
    const float *KRK; /* mKRKinv, 3x3 row-major */
    float r31, r32, r33;
    int single_homography; /* mPredictMethod == SINGLE_HOMOGRAPHY (:233-253) */
} cam_t;

static void predict_one(const cam_t *c, float rx, float ry, float *ux, float *uy, float *dxo, float *dyo)
{
    float x_normal = (rx - c->cx) * c->fx_inv; /* :209-210 */
    float y_normal = (ry - c->cy) * c->fy_inv;
    /* :216  1.0 / (float expr) -> double, narrowed;  :235  float lambda = 1.0 */
    float lambda = c->single_homography ? 1.0f : (float)(1.0 / (double)(c->r31 * x_normal + c->r32 * y_normal + c->r33));
    float pt_x = (c->KRK[0] * rx + c->KRK[1] * ry + c->KRK[2]) * lambda; /* :217 */
    float pt_y = (c->KRK[3] * rx + c->KRK[4] * ry + c->KRK[5]) * lambda; /* :218 */
    float x = (pt_x - c->cx) * c->fx_inv;                                /* :221-222 */
    float y = (pt_y - c->cy) * c->fy_inv;
    float r2 = x * x + y * y;
    float r4 = r2 * r2;
    float r6 = r4 * r2;
    float x_distort = x * (1 + c->k1 * r2 + c->k2 * r4 + c->k3 * r6) + 2 * c->p1 * x * y + c->p2 * (r2 + 2 * x * x);
    float y_distort = y * (1 + c->k1 * r2 + c->k2 * r4 + c->k3 * r6) + c->p1 * (r2 + 2 * y * y) + 2 * c->p2 * x * y;
    *ux = pt_x;
    *uy = pt_y;
    *dxo = c->fx * x_distort + c->cx; /* :229-230 */
    *dyo = c->fy * y_distort + c->cy;
}

/* GyroAidedTracker::GyroPredictFeatures, src/gyro_aided_tracker.cpp:118-185.
 * THIRD PARTY (OpenCV cv::Mat gemm / inv, :166-167): A = C * B^T * (B * B^T)^-1 with
 * B the +-h corner matrix (:73-78).  Restated as: float products accumulated in double
 * per gemm output element, narrowed to float after each Mat product; the 2x2 inverse
 * of B*B^T = diag(4h^2) through the double determinant. */
int pagk_oracle_gyro_predict(const pagk_params *cam, int32_t width, int32_t height, int32_t half_patch,
                             const float *KRKinv, const float *r3, int32_t n, const float *pt_ref_un,
                             float *pt_predict_un, float *pt_predict, uint8_t *status, float *affine)
{
    if (!cam || !KRKinv || !r3 || n < 0 || (n > 0 && (!pt_ref_un || !pt_predict_un || !pt_predict || !status)))
        return PAGK_E_ARG;
    cam_t c;
    c.fx = cam->fx, c.fy = cam->fy, c.cx = cam->cx, c.cy = cam->cy;
    c.fx_inv = (float)(1.0 / (double)c.fx), c.fy_inv = (float)(1.0 / (double)c.fy); /* :66 */
    c.k1 = cam->dist_coef[0], c.k2 = cam->dist_coef[1], c.p1 = cam->dist_coef[2], c.p2 = cam->dist_coef[3];
    c.k3 = cam->n_dist_coef == 5 ? cam->dist_coef[4] : 0;
    c.KRK = KRKinv;
    c.r31 = r3[0], c.r32 = r3[1], c.r33 = r3[2];
    c.single_homography = cam->predict_method == 2;

    const float hh = (float)half_patch;
    const float cornx[4] = {-hh, hh, -hh, hh}, corny[4] = {-hh, -hh, hh, hh}; /* :73-77 */
    /* (B B^T): entries are sums of 4 products of small integers: exact */
    double bbt00 = 0, bbt01 = 0, bbt11 = 0;
    for (int j = 0; j < 4; j++) {
        bbt00 += (double)cornx[j] * cornx[j];
        bbt01 += (double)cornx[j] * corny[j];
        bbt11 += (double)corny[j] * corny[j];
    }
    float m00 = (float)bbt00, m01 = (float)bbt01, m11 = (float)bbt11;
    double det = (double)m00 * m11 - (double)m01 * m01;
    double dinv = det != 0 ? 1. / det : 0;
    float inv00 = (float)(m11 * dinv), inv11 = (float)(m00 * dinv);
    float inv01 = (float)(-m01 * dinv), inv10 = inv01;

    int n_predict = 0;
    for (int i = 0; i < n; i++) {
        /* Initialize() leaves status 0, predict (0,0) where the loop `continue`s (:92-95,:131-135) */
        status[i] = 0;
        pt_predict_un[2 * i] = pt_predict_un[2 * i + 1] = 0;
        pt_predict[2 * i] = pt_predict[2 * i + 1] = 0;
        float rx = pt_ref_un[2 * i], ry = pt_ref_un[2 * i + 1];
        float ux, uy, dxs, dys;
        predict_one(&c, rx, ry, &ux, &uy, &dxs, &dys);
        if (ux < 0 || ux >= width || uy < 0 || uy >= height) continue;     /* :131 */
        if (dxs < 0 || dxs >= width || dys < 0 || dys >= height) continue; /* :134 */
        pt_predict_un[2 * i] = ux;
        pt_predict_un[2 * i + 1] = uy;
        pt_predict[2 * i] = dxs;
        pt_predict[2 * i + 1] = dys;
        status[i] = 1;
        n_predict++;
        if (!affine) continue;
        float Cx[4], Cy[4];
        for (int j = 0; j < 4; j++) { /* :148-160 */
            float cux, cuy, t0, t1;
            predict_one(&c, rx + cornx[j], ry + corny[j], &cux, &cuy, &t0, &t1);
            Cx[j] = cux - ux;
            Cy[j] = cuy - uy;
        }
        /* matC * B^T (2x4 * 4x2) */
        double s00 = 0, s01 = 0, s10 = 0, s11 = 0;
        for (int j = 0; j < 4; j++) {
            s00 += (double)Cx[j] * cornx[j];
            s01 += (double)Cx[j] * corny[j];
            s10 += (double)Cy[j] * cornx[j];
            s11 += (double)Cy[j] * corny[j];
        }
        float t00 = (float)s00, t01 = (float)s01, t10 = (float)s10, t11 = (float)s11;
        /* (...) * inv  (2x2 * 2x2) */
        affine[4 * i + 0] = (float)((double)t00 * inv00 + (double)t01 * inv10);
        affine[4 * i + 1] = (float)((double)t00 * inv01 + (double)t01 * inv11);
        affine[4 * i + 2] = (float)((double)t10 * inv00 + (double)t11 * inv10);
        affine[4 * i + 3] = (float)((double)t10 * inv01 + (double)t11 * inv11);
    }
    return n_predict;
}

/* ---- geometry validation: the scoring loops next to the path ------------------- */
/* GyroAidedTracker::CheckHomography, src/gyro_aided_tracker.cpp:620-676.  The h's are double, the
 * points float: products and sums run in double (usual arithmetic conversions), each `const float`
 * initialiser narrows once. */
int pagk_oracle_check_homography(const double *H21, const double *H12, int32_t n, const float *pts1,
                                 const float *pts2, float sigma, uint8_t *inliers, float *score_out)
{
    if (n < 0 || !H21 || !H12 || !score_out || (n > 0 && (!pts1 || !pts2 || !inliers))) return PAGK_E_ARG;
    const double h11 = H21[0], h12 = H21[1], h13 = H21[2], h21 = H21[3], h22 = H21[4], h23 = H21[5];
    const double h31 = H21[6], h32 = H21[7], h33 = H21[8];
    const double h11inv = H12[0], h12inv = H12[1], h13inv = H12[2], h21inv = H12[3], h22inv = H12[4];
    const double h23inv = H12[5], h31inv = H12[6], h32inv = H12[7], h33inv = H12[8];
    float score = 0;                                    /* :623 */
    const float th = 5.99;                              /* :624 */
    const float invSigmaSquare = 1.0 / (sigma * sigma); /* :625  float product, double divide, narrowed */
    for (int i = 0; i < n; i++) {
        int bIn = 1;
        const float u1 = pts1[2 * i], v1 = pts1[2 * i + 1], u2 = pts2[2 * i], v2 = pts2[2 * i + 1];
        const float w1in2inv = 1.0 / (h31 * u1 + h32 * v1 + h33); /* :641 */
        const float u1in2 = (h11 * u1 + h12 * v1 + h13) * w1in2inv;
        const float v1in2 = (h21 * u1 + h22 * v1 + h23) * w1in2inv;
        const float squareDist2 = (u2 - u1in2) * (u2 - u1in2) + (v2 - v1in2) * (v2 - v1in2); /* :645 */
        const float chiSquare2 = squareDist2 * invSigmaSquare;
        if (chiSquare2 > th) /* :648  a NaN takes the else branch */
            bIn = 0;
        else
            score += th - chiSquare2;
        const float w2in1inv = 1.0 / (h31inv * u2 + h32inv * v2 + h33inv); /* :657 */
        const float u2in1 = (h11inv * u2 + h12inv * v2 + h13inv) * w2in1inv;
        const float v2in1 = (h21inv * u2 + h22inv * v2 + h23inv) * w2in1inv;
        const float squareDist1 = (u1 - u2in1) * (u1 - u2in1) + (v1 - v2in1) * (v1 - v2in1);
        const float chiSquare1 = squareDist1 * invSigmaSquare;
        if (chiSquare1 > th) /* :664 */
            bIn = 0;
        else
            score += th - chiSquare1;
        inliers[i] = (uint8_t)bIn; /* :671-676 */
    }
    *score_out = score;
    return PAGK_OK;
}

/* GyroAidedTracker::CheckFundamental, src/gyro_aided_tracker.cpp:704-768. */
int pagk_oracle_check_fundamental(const double *F21, int32_t n, const float *pts1, const float *pts2,
                                  float sigma, uint8_t *inliers, float *score_out)
{
    if (n < 0 || !F21 || !score_out || (n > 0 && (!pts1 || !pts2 || !inliers))) return PAGK_E_ARG;
    const double f11 = F21[0], f12 = F21[1], f13 = F21[2], f21 = F21[3], f22 = F21[4], f23 = F21[5];
    const double f31 = F21[6], f32 = F21[7], f33 = F21[8];
    float score = 0;
    const float th = 3.84;      /* :707 */
    const float thScore = 5.99; /* :708 */
    const float invSigmaSquare = 1.0 / (sigma * sigma);
    for (int i = 0; i < n; i++) {
        int bIn = 1;
        const float u1 = pts1[2 * i], v1 = pts1[2 * i + 1], u2 = pts2[2 * i], v2 = pts2[2 * i + 1];
        const float a2 = f11 * u1 + f12 * v1 + f13; /* :725-727  l2 = F21 p1 */
        const float b2 = f21 * u1 + f22 * v1 + f23;
        const float c2 = f31 * u1 + f32 * v1 + f33;
        const float num2 = a2 * u2 + b2 * v2 + c2; /* :730  float */
        const float squareDist2 = num2 * num2 / (a2 * a2 + b2 * b2);
        const float chiSquare2 = squareDist2 * invSigmaSquare;
        if (chiSquare2 > th) /* :734 */
            bIn = 0;
        else
            score += thScore - chiSquare2;
        const float a1 = u2 * f11 + v2 * f21 + f31; /* :743-745  l1 = p2^T F21 */
        const float b1 = u2 * f12 + v2 * f22 + f32;
        const float c1 = u2 * f13 + v2 * f23 + f33;
        const float num1 = a1 * u1 + b1 * v1 + c1;
        const float squareDist1 = num1 * num1 / (a1 * a1 + b1 * b1);
        const float chiSquare1 = squareDist1 * invSigmaSquare;
        if (chiSquare1 > th) /* :752 */
            bIn = 0;
        else
            score += thScore - chiSquare1;
        inliers[i] = (uint8_t)bIn;
    }
    *score_out = score;
    return PAGK_OK;
}

/* src/gyro_aided_tracker.cpp:459-469: `float RH = score_H / (score_F + score_H); if (RH > 0.45)` --
 * the comparison promotes RH to double. */
int pagk_oracle_geometry_select(float score_H, float score_F)
{
    float RH = score_H / (score_F + score_H);
    return RH > 0.45 ? 1 : 0;
}

/* GyroAidedTracker::GeometryValidation, src/gyro_aided_tracker.cpp:429-478 (model fitting excluded:
 * the fitted matrices come in as arguments). */
int pagk_oracle_geometry_validation(const double *H21, const double *H12, const double *F21, int32_t n,
                                    const float *pt_ref_un, const float *pt_predict_un, uint8_t *status,
                                    float sigma, float *track_score)
{
    if (n < 0 || (n > 0 && (!pt_ref_un || !pt_predict_un || !status))) return PAGK_E_ARG;
    int m = 0;
    for (int i = 0; i < n; i++) m += status[i] ? 1 : 0; /* :434-440 */
    if (track_score) *track_score = 0;
    if (m <= 8) return 0; /* :445 */
    int *idx = (int *)malloc(sizeof(int) * (size_t)m);
    float *p1 = (float *)malloc(sizeof(float) * 2 * (size_t)m), *p2 = (float *)malloc(sizeof(float) * 2 * (size_t)m);
    uint8_t *inH = (uint8_t *)malloc((size_t)m), *inF = (uint8_t *)malloc((size_t)m);
    if (!idx || !p1 || !p2 || !inH || !inF) {
        free(idx), free(p1), free(p2), free(inH), free(inF);
        return PAGK_E_NOMEM;
    }
    for (int i = 0, k = 0; i < n; i++)
        if (status[i]) {
            idx[k] = i;
            p1[2 * k] = pt_ref_un[2 * i], p1[2 * k + 1] = pt_ref_un[2 * i + 1];
            p2[2 * k] = pt_predict_un[2 * i], p2[2 * k + 1] = pt_predict_un[2 * i + 1];
            k++;
        }
    float sH = 0, sF = 0;
    int rc = pagk_oracle_check_homography(H21, H12, m, p1, p2, sigma, inH, &sH);
    if (rc == PAGK_OK) rc = pagk_oracle_check_fundamental(F21, m, p1, p2, sigma, inF, &sF);
    int cnt_inlier = 0;
    if (rc == PAGK_OK) {
        const int useH = pagk_oracle_geometry_select(sH, sF);
        const uint8_t *in = useH ? inH : inF;
        if (track_score) *track_score = useH ? sH : sF;
        for (int k = 0; k < m; k++) { /* :472-480 */
            if (!in[k])
                status[idx[k]] = 0;
            else
                cnt_inlier++;
        }
    }
    free(idx), free(p1), free(p2), free(inH), free(inF);
    return rc == PAGK_OK ? cnt_inlier : rc;
}

/* ------------------------------------------------------------------------------------------------
 * NCC nearest-neighbour matching,
 * GyroAidedTracker::FindAndSortNearNeighbor / MatchFeatures (src/gyro_aided_tracker.cpp:788-851,
 * 949-1008) over the FREE functions GetPixelValue (include/utils.h:32-46) and NCC
 * (src/utils.cpp:110-148).  The free sampler is not the member PatchMatch::GetPixelValue: its upper
 * clamp is `>` (x == cols stays), and its formula is the un-factored four-term sum.
 * ---------------------------------------------------------------------------------------------- */
static inline float get_pixel_value_free(const level_t *img, float x, float y)
{
    /* include/utils.h:35-38.  NaN is mapped to 0 like in the member sampler above (int(NaN) is undefined
     * in the reference); identical for every non-NaN coordinate. */
    if (!(x >= 0)) x = 0;
    if (!(y >= 0)) y = 0;
    if (x > img->cols) x = img->cols - 1;
    if (y > img->rows) y = img->rows - 1;
    int ix = (int)x, iy = (int)y;
    int64_t off = (int64_t)iy * img->step + ix; /* :40 */
    float xx = x - floorf(x);
    float yy = y - floorf(y);
    /* :43-44  (1 - yy) * (1 - xx) * data[0] + (1 - yy) * xx * data[1] + yy * (1 - xx) * data[step] + yy * xx * data[step+1]
     * -- left to right: ((1-yy)*(1-xx))*d0, sums in order */
    float pixel = (1 - yy) * (1 - xx) * tap(img, off) + (1 - yy) * xx * tap(img, off + 1) +
                  yy * (1 - xx) * tap(img, off + img->step) + yy * xx * tap(img, off + img->step + 1);
    return pixel;
}

/* NCC(halfPatchSize, vValuesRef, mean_ref, cur, pt_cur, warp_mat), src/utils.cpp:110-148. */
static float ncc_free(int h, const float *v_ref, float mean_ref, const level_t *cur, float cx, float cy, const float *A,
                      float *v_cur /* scratch, P floats */)
{
    const int P = (2 * h + 1) * (2 * h + 1);
    float mean_cur = 0.0f;
    int k = 0;
    for (int x = -h; x <= h; x++) /* :120-121  x outer, y inner */
        for (int y = -h; y <= h; y++) {
            float value_cur;
            if (!A) /* warp_mat.empty() :124 */
                value_cur = get_pixel_value_free(cur, cx + x, cy + y);
            else {
                float wx = A[0] * x + A[1] * y; /* :127-128 */
                float wy = A[2] * x + A[3] * y;
                value_cur = get_pixel_value_free(cur, cx + wx, cy + wy);
            }
            mean_cur += value_cur;
            v_cur[k++] = value_cur;
        }
    mean_cur /= (float)P; /* :135  float /= size_t */
    float numerator = 0, den1 = 0, den2 = 0;
    for (int i = 0; i < P; i++) { /* :139-145 */
        float v_ref_dot = v_ref[i] - mean_ref;
        float v_cur_dot = v_cur[i] - mean_cur;
        numerator += (v_ref_dot * v_cur_dot);
        den1 += v_ref_dot * v_ref_dot;
        den2 += v_cur_dot * v_cur_dot;
    }
    return (float)((double)numerator / sqrt((double)(den1 * den2) + 1e-10)); /* :147 */
}

/* NCC(halfPatchSize, ref, cur, pt_ref, pt_cur, warp_mat), src/utils.cpp:166-200 (the two-image overload). */
float pagk_oracle_ncc_free(const pagk_image *ref, const pagk_image *cur, int32_t half_patch, float rx, float ry,
                           float cx, float cy, const float *A)
{
    level_t lr, lc;
    if (view_to_level(ref, &lr) || view_to_level(cur, &lc) || half_patch < 1) return NAN;
    const int h = half_patch, P = (2 * h + 1) * (2 * h + 1);
    float *vr = (float *)malloc(sizeof(float) * 2 * (size_t)P), *vc = vr + P;
    if (!vr) return NAN;
    float mean_ref = 0.0f, mean_cur = 0.0f;
    int k = 0;
    for (int x = -h; x <= h; x++)
        for (int y = -h; y <= h; y++) {
            float value_ref = get_pixel_value_free(&lr, rx + x, ry + y);
            mean_ref += value_ref;
            vr[k] = value_ref;
            float value_cur;
            if (!A)
                value_cur = get_pixel_value_free(&lc, cx + x, cy + y);
            else {
                float wx = A[0] * x + A[1] * y;
                float wy = A[2] * x + A[3] * y;
                value_cur = get_pixel_value_free(&lc, cx + wx, cy + wy);
            }
            mean_cur += value_cur;
            vc[k] = value_cur;
            k++;
        }
    mean_ref /= (float)P;
    mean_cur /= (float)P;
    float numerator = 0, den1 = 0, den2 = 0;
    for (int i = 0; i < P; i++) { /* :193-197: differences re-evaluated per term */
        numerator += ((vr[i] - mean_ref) * (vc[i] - mean_cur));
        den1 += (vr[i] - mean_ref) * (vr[i] - mean_ref);
        den2 += (vc[i] - mean_cur) * (vc[i] - mean_cur);
    }
    free(vr);
    return (float)((double)numerator / sqrt((double)(den1 * den2) + 1e-10));
}

typedef struct {
    int train;
    float distance, ncc;
} nn_t;

/* GyroAidedTracker::FindAndSortNearNeighbor, src/gyro_aided_tracker.cpp:788-851, for range [0, n). */
int pagk_oracle_find_near_neighbors(const pagk_image *ref, const pagk_image *cur, int32_t half_patch, int32_t n,
                                    const float *keys_ref, const float *pt_predict_un, const uint8_t *status,
                                    const float *affine, int32_t m, const float *keys_cur, const float *keys_cur_un,
                                    int32_t level, float radius_unit, int32_t use_ncc, int32_t cap, int32_t *count,
                                    int32_t *nbr_idx, float *nbr_dist, float *nbr_ncc)
{
    level_t lr, lc;
    if (view_to_level(ref, &lr) || view_to_level(cur, &lc)) return PAGK_E_ARG;
    if (half_patch < 1 || n < 0 || m < 0 || cap < 1 || !count) return PAGK_E_ARG;
    const int h = half_patch, P = (2 * h + 1) * (2 * h + 1);
    float *v_ref = (float *)malloc(sizeof(float) * 2 * (size_t)P), *v_cur = v_ref + P;
    nn_t *st1 = (nn_t *)malloc(sizeof(nn_t) * 2 * (size_t)(m > 0 ? m : 1)), *st2 = st1 + (m > 0 ? m : 1);
    if (!v_ref || !st1) {
        free(v_ref);
        free(st1);
        return PAGK_E_NOMEM;
    }
    int overflow = 0;
    for (int i = 0; i < n; i++) {
        if (!status[i]) continue;     /* :791 */
        if (count[i] > 0) continue;   /* :793  neighbours already found with a smaller search region */
        /* :800-808  reference patch values and mean, x outer / y inner, on mvKeysRef[i].pt */
        float mean_ref = 0.0f;
        int k = 0;
        for (int x = -h; x <= h; x++)
            for (int y = -h; y <= h; y++) {
                float value_ref = get_pixel_value_free(&lr, keys_ref[2 * i] + x, keys_ref[2 * i + 1] + y);
                mean_ref += value_ref;
                v_ref[k++] = value_ref;
            }
        mean_ref /= (float)P;
        const float radius = (float)level * radius_unit; /* :811 */
        int n1 = 0, n2 = 0; /* the two std::stack of :797 as arrays; top = last element */
        for (int j = 0; j < m; j++) {
            float dx = pt_predict_un[2 * i] - keys_cur_un[2 * j], dy = pt_predict_un[2 * i + 1] - keys_cur_un[2 * j + 1];
            if (fabsf(dx) > radius || fabsf(dy) > radius) continue; /* :814 */
            float distance = sqrtf(dx * dx + dy * dy);              /* :818 */
            float ncc = ncc_free(h, v_ref, mean_ref, &lc, keys_cur[2 * j], keys_cur[2 * j + 1],
                                 affine ? affine + 4 * i : NULL, v_cur); /* :821 */
            nn_t match = {j, distance, ncc};
            if (use_ncc) { /* :825-830 */
                while (n1 > 0 && ncc < st1[n1 - 1].ncc) st2[n2++] = st1[--n1];
            } else {       /* :831-836 */
                while (n1 > 0 && distance > st1[n1 - 1].distance) st2[n2++] = st1[--n1];
            }
            st1[n1++] = match;                      /* :838 */
            while (n2 > 0) st1[n1++] = st2[--n2];   /* :839-842 */
        }
        count[i] = n1;
        if (n1 > cap) {
            overflow = 1;
            continue;
        }
        for (int k2 = 0; k2 < n1; k2++) { /* :845-849  popped from the top */
            const nn_t *e = &st1[n1 - 1 - k2];
            nbr_idx[(size_t)i * cap + k2] = e->train;
            nbr_dist[(size_t)i * cap + k2] = e->distance;
            nbr_ncc[(size_t)i * cap + k2] = e->ncc;
        }
    }
    free(v_ref);
    free(st1);
    return overflow ? PAGK_E_ARG : PAGK_OK;
}

/* GyroAidedTracker::MatchFeatures, src/gyro_aided_tracker.cpp:949-1008.  TH_NCC_HIGH = 0.6f,
 * TH_NCC_LOW = 0.3f, TH_RATIO = 0.75f (:7-9).  Returns the number of matches. */
int pagk_oracle_match_features(int32_t n, int32_t cap, const int32_t *count, const int32_t *nbr_idx,
                               const float *nbr_dist, const float *nbr_ncc, int32_t use_ncc, int32_t *match_query,
                               int32_t *match_train, float *match_dist, float *match_ncc)
{
    const float TH_NCC_HIGH = 0.6f, TH_NCC_LOW = 0.3f, TH_RATIO = 0.75f;
    int nm = 0;
    for (int i = 0; i < n; i++) {
        const int c = count[i];
        if (c <= 0) continue; /* :955 */
        const size_t b = (size_t)i * cap;
        if (use_ncc) {
            if (nbr_ncc[b] > TH_NCC_HIGH) { /* :961 */
            } else if (c > 1) {
                if (nbr_ncc[b] < TH_NCC_LOW) continue;               /* :964 */
                if (!(nbr_ncc[b + 1] < nbr_ncc[b] * TH_RATIO)) continue; /* :968-971 */
            } else
                continue;
        } else {
            if (c == 1) { /* :978 */
            } else if (!(nbr_dist[b] < nbr_dist[b + 1] * TH_RATIO))
                continue; /* :982-985 */
        }
        const int train = nbr_idx[b];
        match_query[nm] = i;
        match_train[nm] = train;
        match_dist[nm] = nbr_dist[b];
        match_ncc[nm] = nbr_ncc[b];
        nm++;
    }
    /* second pass: the std::set / erase logic of :991-1005 replayed in order over the candidates chosen above */
    int out = 0;
    int32_t *seen = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nm > 0 ? nm : 1));
    int nseen = 0;
    if (!seen) return PAGK_E_NOMEM;
    for (int k = 0; k < nm; k++) {
        const int train = match_train[k];
        int in_set = 0;
        for (int s = 0; s < nseen && !in_set; s++) in_set = seen[s] == train;
        if (!in_set) { /* :991-994 */
            match_query[out] = match_query[k];
            match_train[out] = train;
            match_dist[out] = match_dist[k];
            match_ncc[out] = match_ncc[k];
            out++;
            seen[nseen++] = train;
        } else { /* :995-1005  erase every earlier match to this keypoint; the index stays in the set */
            int w = 0;
            for (int r = 0; r < out; r++)
                if (match_train[r] != train) {
                    match_query[w] = match_query[r];
                    match_train[w] = match_train[r];
                    match_dist[w] = match_dist[r];
                    match_ncc[w] = match_ncc[r];
                    w++;
                }
            out = w;
        }
    }
    free(seen);
    return out;
}
