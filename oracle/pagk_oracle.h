/*
 * pagk_oracle.h -- CPU oracle for the PatchMatch hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and
 * only as the checker / the timed CPU baseline.  The product (libpagk_hip.so) never
 * links, loads or falls back to this code.
 *
 * PARITY UNPINNED: the reference (C++ on OpenCV + Eigen + glog) cannot be built in
 * this environment and ships no tests or golden vectors, so this restatement could
 * not be checked against reference outputs.  It follows the reference's own source
 * type-for-type (citations on every function); the arithmetic that lives in absent
 * third-party code is restated from the published algorithms and listed in
 * oracle/README.md (cv::resize 2x decimation, Eigen 3.3 fixed-size LLT / triangular
 * solves / norm reduction order, libm log).
 */
#ifndef PAGK_ORACLE_H
#define PAGK_ORACLE_H

#include "../include/pagk.h"

#ifdef __cplusplus
extern "C" {
#endif

/* cv::resize(src, dst, Size(cols*0.5, rows*0.5)) of an 8UC1 image, default INTER_LINEAR
 * (reference src/patch_match.cpp:69-70): the exact-2x case is OpenCV's INTER_AREA fast path, any
 * other (odd) size its 11-bit fixed-point bilinear.  dst is (int)(w*0.5) x (int)(h*0.5), contiguous. */
int pagk_oracle_pyr_down(const uint8_t *src, int32_t w, int32_t h, int64_t step, uint8_t *dst);

/* PatchMatch::OpticalFlowMultiLevel (src/patch_match.cpp:79-142) on host buffers;
 * same arguments as pagk_track.  nthreads stripes features over pthreads the way
 * cv::parallel_for_ does (:103); nthreads <= 0 means one thread. */
int pagk_oracle_track(const pagk_params *params, const pagk_image *ref, const pagk_image *cur,
                      int32_t n, const float *pt_ref_un, const float *pt_init_un,
                      const float *affine, const uint8_t *status_in, const pagk_outputs *out,
                      int32_t nthreads);

/* Same with caller-built pyramids (same arguments as pagk_track_pyr). */
int pagk_oracle_track_pyr(const pagk_params *params, int32_t n_levels,
                          const pagk_image *ref_levels, const pagk_image *cur_levels, int32_t n,
                          const float *pt_ref_un, const float *pt_init_un, const float *affine,
                          const uint8_t *status_in, const pagk_outputs *out, int32_t nthreads);

/* GyroAidedTracker::GyroPredictFeaturesAndOpticalFlowRefined Step 3
 * (src/gyro_aided_tracker.cpp:289-341); same arguments as pagk_post_filter. */
int pagk_oracle_post_filter(int32_t n, int32_t half_patch, const uint8_t *status_pm,
                            const double *pix_err, const double *dist_pred, const float *pt_pm,
                            const float *pt_pm_un, uint8_t *status_out, float *pt_predict,
                            float *pt_predict_un);

/* GyroAidedTracker::GyroPredictFeatures + GyroPredictOnePixel, PIXEL_AWARE_PREDICTION
 * (src/gyro_aided_tracker.cpp:118-185,194-231).  KRKinv: 3x3 row-major float
 * (mKRKinv), r3: third row of Rcl (mr31 mr32 mr33).  Outputs: pt_predict_un,
 * pt_predict (distorted), status (n), affine (n x 4; untouched where status 0). */
int pagk_oracle_gyro_predict(const pagk_params *cam, int32_t width, int32_t height, int32_t half_patch,
                             const float *KRKinv, const float *r3, int32_t n, const float *pt_ref_un,
                             float *pt_predict_un, float *pt_predict, uint8_t *status, float *affine);

/* The shared software log (see oracle/README.md, "libm log"). */
double pagk_oracle_log(double x);
/* mInvLogMaxDist (src/patch_match.cpp:51). */
float pagk_oracle_inv_log_max_dist(float alpha, int32_t max_distance);

/* 4x4 LLT + solve + norm exactly as the GN loop uses them (src/patch_match.cpp:319,343);
 * exposed so the tests can pin the restatement's own corner cases.
 * H: 16 doubles row-major (only the lower triangle is read); returns ||x||. */
double pagk_oracle_llt_solve4(const double *H, const double *b, double *x);

/* GyroAidedTracker::CheckHomography scoring loop (src/gyro_aided_tracker.cpp:620-676): symmetric
 * transfer error of every correspondence under H21 (row-major 3x3 double, what cv::findHomography
 * returned) and H12 = H21.inv() (the caller's cv::Mat::inv -- third-party arithmetic stays outside),
 * chi-square test against 5.99, inlier flags, score accumulated in float in index order.
 * pts1 / pts2: n x 2 float.  Returns PAGK_OK. */
int pagk_oracle_check_homography(const double *H21, const double *H12, int32_t n, const float *pts1,
                                 const float *pts2, float sigma, uint8_t *inliers, float *score);
/* GyroAidedTracker::CheckFundamental scoring loop (src/gyro_aided_tracker.cpp:704-768): point to
 * epipolar line distances under F21, test against 3.84, score against 5.99. */
int pagk_oracle_check_fundamental(const double *F21, int32_t n, const float *pts1, const float *pts2,
                                  float sigma, uint8_t *inliers, float *score);
/* Model choice of GyroAidedTracker::GeometryValidation (src/gyro_aided_tracker.cpp:459-469):
 * 1 = homography (RH > 0.45), 0 = fundamental. */
int pagk_oracle_geometry_select(float score_H, float score_F);
/* GyroAidedTracker::GeometryValidation bookkeeping (src/gyro_aided_tracker.cpp:433-478) around the two
 * scoring loops: compacts the status-true correspondences, skips everything unless more than 8 remain,
 * scores both models, keeps the chosen model's inliers.  status: n flags, updated in place.
 * Returns cnt_inlier (0 when <= 8 correspondences: the reference then leaves mvStatus untouched and
 * returns 0); track_score may be NULL. */
int pagk_oracle_geometry_validation(const double *H21, const double *H12, const double *F21, int32_t n,
                                    const float *pt_ref_un, const float *pt_predict_un, uint8_t *status,
                                    float sigma, float *track_score);

/* tools/parity_risk.py only: switch ONE of the restatement's guesses about third-party arithmetic to its
 * alternative (bit flags, see pagk_oracle.c) to measure how much the results depend on it.  0 restores the
 * documented restatement.  Process-global; never used by tests or the benchmark. */
void pagk_oracle_set_alternatives(uint32_t flags);

/* ---- NCC nearest-neighbour matching ------------------------------------------- */
/* Free NCC(halfPatchSize, ref, cur, pt_ref, pt_cur, warp_mat), src/utils.cpp:166-200, over the free
 * GetPixelValue of include/utils.h:32-46 (NOT PatchMatch::GetPixelValue).  A: 2x2 row-major or NULL (empty Mat). */
float pagk_oracle_ncc_free(const pagk_image *ref, const pagk_image *cur, int32_t half_patch, float rx, float ry,
                           float cx, float cy, const float *A);
/* GyroAidedTracker::FindAndSortNearNeighbor (src/gyro_aided_tracker.cpp:788-851) over [0, n).
 *   keys_ref      n x 2  mvKeysRef[i].pt          pt_predict_un n x 2  mvPtPredictUn[i]
 *   status        n      mvStatus[i]              affine        n x 4  mvAffineDeformationMatrix[i] or NULL (empty)
 *   keys_cur      m x 2  mvKeysCur[j].pt          keys_cur_un   m x 2  mvKeysCurUn[j].pt
 *   level, radius_unit   search radius = level * mRadiusForFindNearNeighbor (:811; the tracker sets 2 * h, :62)
 *   use_ncc              mbNCC (:60 true)
 *   count         n      in/out: size of mvvNearNeighbors[i]; features with count > 0 are skipped (:793)
 *   nbr_*         n x cap  the sorted neighbour lists (trainIdx, distance, ncc), best first
 * Returns PAGK_OK, or PAGK_E_ARG when a list is longer than cap (count[] then still holds the true sizes). */
int pagk_oracle_find_near_neighbors(const pagk_image *ref, const pagk_image *cur, int32_t half_patch, int32_t n,
                                    const float *keys_ref, const float *pt_predict_un, const uint8_t *status,
                                    const float *affine, int32_t m, const float *keys_cur, const float *keys_cur_un,
                                    int32_t level, float radius_unit, int32_t use_ncc, int32_t cap, int32_t *count,
                                    int32_t *nbr_idx, float *nbr_dist, float *nbr_ncc);
/* GyroAidedTracker::MatchFeatures (src/gyro_aided_tracker.cpp:949-1008).  match_*: capacity n.  Returns the
 * number of matches (mvMatches.size()). */
int pagk_oracle_match_features(int32_t n, int32_t cap, const int32_t *count, const int32_t *nbr_idx,
                               const float *nbr_dist, const float *nbr_ncc, int32_t use_ncc, int32_t *match_query,
                               int32_t *match_train, float *match_dist, float *match_ncc);

#ifdef __cplusplus
}
#endif
#endif
