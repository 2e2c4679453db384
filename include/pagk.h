/*
 * pagk.h -- C ABI of the MI355X-native pyramidal patch-based KLT refinement.
 *
 * This is the drop-in boundary for ONE path of the reference tracker: the body of
 * PatchMatch::OpticalFlowMultiLevel() (reference src/patch_match.cpp:79-142) and
 * everything it calls.  Plain pointers and sizes only; no C++/torch types.
 *
 * Every entry point cites the reference interface it replaces.  All buffers are
 * caller-owned; the library owns only its pagk_ctx (device buffers, stream).
 *
 * Conventions
 *   - points are interleaved (x, y) float32 pairs, n of them  (cv::Point2f layout)
 *   - affine is n x 4 float32, row-major 2x2 per feature      (cv::Mat CV_32F 2x2,
 *     reference src/gyro_aided_tracker.cpp:166-168)
 *   - status bytes are 0 / 1                                   (std::vector<uchar>)
 *   - return value: 0 = PAGK_OK, negative = error (never throws, never aborts)
 */
#ifndef PAGK_H
#define PAGK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PAGK_VERSION 303 /* 0.3.3: + pagk_priority_threshold; 0.3.2: + pagk_frame_set_device_batch; 0.3.1: pipelined 4-wave kernel, pagk_track_device_batch, pagk_check_launch, pagk_selftest_repeat_sum */

#define PAGK_MAX_PYRAMIDS 8
#define PAGK_MAX_HALF_PATCH 15 /* (2h+1)^2 <= 961 pixels */

enum {
    PAGK_OK = 0,
    PAGK_E_ARG = -1,         /* null pointer / out-of-range parameter                     */
    PAGK_E_HIP = -2,         /* a HIP runtime call failed (see pagk_last_error)           */
    PAGK_E_NOMEM = -3,       /* device or host allocation failed                          */
    PAGK_E_UNSUPPORTED = -4, /* inverse-compositional mode (reference: "not support yet") */
    PAGK_E_NODEVICE = -5,    /* no HIP device / HIP library could not initialise          */
    PAGK_E_NCCL = -6,        /* an RCCL call of the sharded path failed (see pagk_last_error) */
    PAGK_E_CAPACITY = -7     /* a caller-sized output array is too small (neighbour lists)  */
};

/* 8-bit single-channel image view.  Mirrors the fields of cv::Mat (CV_8UC1) the
 * reference reads: data, cols, rows, step (src/patch_match.cpp:396-403). */
typedef struct pagk_image {
    const uint8_t *data;
    int32_t width;  /* cv::Mat::cols */
    int32_t height; /* cv::Mat::rows */
    int64_t step;   /* cv::Mat::step, bytes per row, >= width */
} pagk_image;

/* Arguments of PatchMatch::PatchMatch (include/patch_match.h:44-49) plus the
 * constants its constructor hard-codes (src/patch_match.cpp:48-57) and the camera
 * model DistortPoints() reads from the tracker (src/patch_match.cpp:409-416,
 * src/utils.cpp:49-76). Fill with pagk_params_default() and override. */
typedef struct pagk_params {
    int32_t half_patch; /* halfPatchSize_  (reference apps: 5; BASELINE: 10)             */
    int32_t iterations; /* iterations_     (reference call site: 10; BASELINE: 30)       */
    int32_t pyramids;   /* pyramids_       (reference call site: 3)                      */
    uint8_t has_gyro_predict_initial; /* bHasGyroPredictInitial_: 0 => pt_init ignored   */
    uint8_t inverse;                  /* bInverse_: must be 0 (PAGK_E_UNSUPPORTED)        */
    uint8_t consider_illumination;    /* bConsiderIllumination_                           */
    uint8_t consider_affine;          /* bConsiderAffineDeformation_                      */
    uint8_t regularization_penalty;   /* bRegularizationPenalty_                          */
    uint8_t calculate_ncc;            /* bCalculateNCC_                                   */
    uint8_t predict_method;           /* pagk_gyro_predict_device only: 0 or 1 = PIXEL_AWARE_PREDICTION, 2 =
                                         SINGLE_HOMOGRAPHY (ePredictMethod, include/gyro_aided_tracker.h:66-69)   */
    uint8_t solver_variant;           /* 0 = Eigen 3.3 with SSE2 packets, as restated in oracle/README.md.  Bits select
                                         the association another Eigen version / build uses in H.llt().solve(b) and
                                         update.norm() (src/patch_match.cpp:319,343):  1 lower solve row 3 as
                                         (c0+c1)+c2;  2 upper solve row 0 as c0+(c1+c2);  4 squaredNorm sequential
                                         (EIGEN_DONT_VECTORIZE);  8 LLT column scaling by the reciprocal (Eigen <= 3.2);
                                         32 4th pivot's squaredNorm as a0+(a1+a2).  Same bits as
                                         pagk_oracle_set_alternatives.  Others: PAGK_E_ARG.                         */
    float lambda;            /* mLambda      = 1.0f  (:48) */
    float alpha;             /* mAlpha       = 0.5f  (:49) */
    int32_t max_distance;    /* mMaxDistance = 25    (:50) */
    float inv_log_max_dist;  /* mInvLogMaxDist (:51); 0 => computed by the library       */
    /* camera model used only by the distortion epilogue */
    float fx, fy, cx, cy;    /* mK(0,0), mK(1,1), mK(0,2), mK(1,2)                        */
    float dist_coef[5];      /* k1 k2 p1 p2 k3                                            */
    int32_t n_dist_coef;     /* 4 or 5 (mDistCoef.total()); k3 ignored unless 5           */
} pagk_params;

/* Outputs == the six vectors PatchMatch::SetMatcher fills on the tracker
 * (src/patch_match.cpp:370-388, include/gyro_aided_tracker.h:214-219) plus an
 * optional diagnostic. Any pointer except pt_un/status may be NULL. */
typedef struct pagk_outputs {
    float *pt_un;      /* n x 2  mvPtPredictAfterPatchMatchedUn                           */
    float *pt_dist;    /* n x 2  mvPtPredictAfterPatchMatched (distorted)                 */
    uint8_t *status;   /* n      mvStatusAfterPatchMatched                                */
    double *pix_err;   /* n      mvPixelErrorsOfPatchMatched                              */
    double *dist_pred; /* n      mvDistanceBetweenPredictedAndPatchMatched                */
    float *ncc;        /* n      mvNccAfterPatchMatched                                   */
    int32_t *iters;    /* n      diagnostic: Gauss-Newton iterations executed, summed
                                 over levels (not in the reference)                       */
} pagk_outputs;

typedef struct pagk_ctx pagk_ctx;

/* ---- library / context ------------------------------------------------------ */
int pagk_version(void);
const char *pagk_strerror(int code);
/* Text of the last HIP failure on this context (empty string if none). */
const char *pagk_last_error(const pagk_ctx *ctx);

/* Defaults == reference call site src/gyro_aided_tracker.cpp:276-282 with
 * eType GYRO_PREDICT_WITH_OPTICAL_FLOW_REFINED_CONSIDER_ILLUMINATION_DEFORMATION
 * (:402-408): h=5, 10 iterations, 3 levels, gyro init, illumination + affine. */
void pagk_params_default(pagk_params *p);
/* mInvLogMaxDist exactly as src/patch_match.cpp:51 computes it. */
float pagk_inv_log_max_dist(float alpha, int32_t max_distance);

/* One context per host thread / GPU (reference: one PatchMatch per tracker, not
 * re-entrant because of the shared mLevel, src/patch_match.cpp:99). */
int pagk_create(pagk_ctx **out, int device);
void pagk_destroy(pagk_ctx *ctx);

/* ---- the hot path, host buffers (drop-in for OpticalFlowMultiLevel) -------- */
/* Replaces PatchMatch::OpticalFlowMultiLevel() src/patch_match.cpp:79-142:
 * CreatePyramids (:61-76) + per-level per-feature GN loop (:167-367) +
 * DistortPoints (:409-416) + SetMatcher (:370-388).  Synchronous.
 *   pt_ref_un  = mvKeysRefUn[i].pt        pt_init_un = mvPtPredictUn[i]
 *   affine     = mvAffineDeformationMatrix[i] (may be NULL iff !consider_affine)
 *   status_in  = mvStatus[i] snapshot (mvGyroPredictStatus, :58)                */
int pagk_track(pagk_ctx *ctx, const pagk_params *params, const pagk_image *ref,
               const pagk_image *cur, int32_t n, const float *pt_ref_un,
               const float *pt_init_un, const float *affine, const uint8_t *status_in,
               const pagk_outputs *out);

/* Same, but with caller-built pyramids (levels[0] = full resolution), so a host
 * that links real OpenCV can hand over cv::resize output (:69-70) verbatim. */
int pagk_track_pyr(pagk_ctx *ctx, const pagk_params *params, int32_t n_levels,
                   const pagk_image *ref_levels, const pagk_image *cur_levels, int32_t n,
                   const float *pt_ref_un, const float *pt_init_un, const float *affine,
                   const uint8_t *status_in, const pagk_outputs *out);

/* ---- the hot path, device-resident (streams of frame pairs, benchmarks) ---- */
/* Upload a frame into one of the context's frame slots and build its pyramid on
 * the device (CreatePyramids :61-76). In a sequence, cur of pair t is ref of pair
 * t+1, so each frame is uploaded once. slot in [0, 4).  Returns after img->data has been read (the caller may
 * reuse the buffer at once, also when it is pinned memory); the pyramid kernels may still be running. */
int pagk_frame_upload(pagk_ctx *ctx, int32_t slot, const pagk_image *img, int32_t pyramids);
/* The same for a frame in PINNED host memory (a camera ring buffer): asynchronous on the context's stream and
 * capturable -- inside pagk_graph_begin / pagk_graph_end the host -> device copy becomes a graph node, so a live loop
 * (Examples/Demo/RealSenseD435i.cpp:199-321: grab, track) replays [copy the frame -> pyramid -> PatchMatch] with one
 * pagk_graph_launch per frame.  The memory must stay pinned, and unchanged until the enqueued work has run. */
int pagk_frame_upload_pinned(pagk_ctx *ctx, int32_t slot, const pagk_image *img, int32_t pyramids);
/* Same for an image that already lives in device memory (d_data: device pointer,
 * rows of `step` bytes). Asynchronous on the context stream. */
int pagk_frame_set_device(pagk_ctx *ctx, int32_t slot, const void *d_data, int32_t width,
                          int32_t height, int64_t step, int32_t pyramids);
/* pagk_frame_set_device for the frames of k contexts that share one device, as ONE launch: PatchMatch::CreatePyramids
 * (src/patch_match.cpp:61-76) of k trackers that are stepped together (BASELINE configs[4], "batched multi-camera ...
 * shared pyramid upload"); the producer side of pagk_track_device_batch.  Frame j (device image d_data[j], width[j] x
 * height[j], rows step[j] bytes apart) goes into slot[j] of ctxs[j]; per frame the same bytes as pagk_frame_set_device.
 * The launch is issued on ctxs[0]'s stream and ordered against the other contexts' streams like pagk_track_device_batch;
 * its frame descriptors follow the same rules inside a capture (issue the call once directly first; at most four batched
 * calls of either kind per capture).  Frames the single-launch kernel does not serve (a parent level with an odd
 * dimension, more than four levels) make the call k per-context launches.  k <= 64. */
int pagk_frame_set_device_batch(pagk_ctx *const *ctxs, int32_t k, const int32_t *slot, const void *const *d_data,
                                const int32_t *width, const int32_t *height, const int64_t *step, int32_t pyramids);
/* Copy one pyramid level of a slot back to the host (tests: pyramid parity). */
int pagk_frame_download_level(pagk_ctx *ctx, int32_t slot, int32_t level, uint8_t *dst,
                              int32_t *width, int32_t *height);

/* Track with every per-feature array in DEVICE memory (all pointers are device
 * pointers; same layouts as pagk_track). Asynchronous on the context stream;
 * call pagk_sync before reading results on the host. */
int pagk_track_device(pagk_ctx *ctx, const pagk_params *params, int32_t slot_ref,
                      int32_t slot_cur, int32_t n, const float *d_pt_ref_un,
                      const float *d_pt_init_un, const float *d_affine,
                      const uint8_t *d_status_in, const pagk_outputs *d_out);
/* pagk_track_device for the pair (slot_ref, slot_cur) AND CreatePyramids (src/patch_match.cpp:61-76) of another
 * frame -- device image d_next, built into slot_next -- in ONE launch: when the 4-wave kernel is the one
 * selected, the pyramid is computed by trailing workgroups of the tracking launch and costs no launch of its
 * own; otherwise it is launched separately.  Results are those of pagk_frame_set_device(slot_next, ...) plus
 * pagk_track_device(...).  For pipelines that hold frame k+1 while pair (k-1, k) is tracked (replays, or a
 * camera loop that accepts one frame of latency).  slot_next must differ from slot_ref and slot_cur. */
int pagk_track_device_fused(pagk_ctx *ctx, const pagk_params *params, int32_t slot_ref, int32_t slot_cur, int32_t n,
                            const float *d_pt_ref_un, const float *d_pt_init_un, const float *d_affine,
                            const uint8_t *d_status_in, const pagk_outputs *d_out, int32_t slot_next,
                            const void *d_next, int32_t width, int32_t height, int64_t step, int32_t pyramids);
/* pagk_track_device for k camera streams that share one device, as ONE launch (BASELINE configs[4]: "batched
 * multi-camera").  The reference builds one PatchMatch per tracker (src/gyro_aided_tracker.cpp:276-283); this is k of those
 * calls at once: stream j is context ctxs[j] (its frame slots slot_ref[j] / slot_cur[j] hold the pyramids), n[j] features
 * in the device arrays d_pt_ref_un[j], d_pt_init_un[j], d_affine[j], d_status_in[j], results to d_out[j]; `params` is
 * common to all (the trackers of one application are configured alike).  Results per stream are bit-identical to its own
 * pagk_track_device.  From 6000 features in total (or pagk_set_kernel(ctxs[0], 7)) the streams are served by one launch
 * of variant 7 whose four-feature groups carry their stream; smaller batches, calculate_ncc and one-level pyramids run
 * as k launches.  The launch is issued on ctxs[0]'s stream: it waits for what the other contexts' streams have enqueued
 * so far, and their later work waits for it (contexts switched to one common stream with pagk_set_stream need neither,
 * and can be captured together: pagk_graph_begin(ctxs[0]) ... pagk_graph_end -- after the same call has been issued once
 * directly, at most four batched calls (of this kind and of pagk_frame_set_device_batch together) per capture: the stream descriptors a captured launch reads are buffers that
 * pagk_graph_begin reserves and the graph owns, so that no later call can rewrite them under a replay; a direct call's are
 * not reused before that launch is over).  Pointer arrays are host arrays of device pointers, read during the call;
 * d_pt_init_un / d_affine may be NULL when the flags do not use them.  k <= 64, all contexts on one device. */
int pagk_track_device_batch(pagk_ctx *const *ctxs, int32_t k, const pagk_params *params, const int32_t *slot_ref,
                            const int32_t *slot_cur, const int32_t *n, const float *const *d_pt_ref_un,
                            const float *const *d_pt_init_un, const float *const *d_affine,
                            const uint8_t *const *d_status_in, const pagk_outputs *d_out);
int pagk_sync(pagk_ctx *ctx);
/* For callers that synchronise the stream themselves (pagk_set_stream: a torch stream, the host application's own) and
 * so never pass through pagk_sync: the error state pagk_sync would have returned, without synchronising.  Call it after
 * your own synchronisation.  PAGK_E_HIP (once) when a wave of a level-by-level launch (kernel 7) gave up its bounded
 * wait -- never expected; such a launch also clears its status array, so nothing stale can pass for a tracked point. */
int pagk_check_launch(pagk_ctx *ctx);
/* Use an external HIP stream (e.g. torch's current stream) instead of the
 * context's own; pass NULL to restore. */
int pagk_set_stream(pagk_ctx *ctx, void *hip_stream);

/* Kernel selection.  0 = automatic (default): by launch size (thresholds measured on MI355X at half_patch 10), one of
 *   - 4-wave workgroup per feature, DPP-row ordered accumulation  (lowest latency; < 6000 features)
 *   - one wavefront per feature, f32 streams + MFMA chain  (= 3; calculate_ncc launches from 6000 features, and 6000 to
 *     6999 features in total when the device is shared)
 *   - four features per wavefront, one pyramid level per wavefront  (= 7; from 6000 features, context alone on the device)
 *   - four features per wavefront  (= 5; from 7000 features in total when pagk_set_concurrency says the device is shared)
 * Selected explicitly only:
 *   - 2-wave workgroup per feature, ordered accumulation as a v_mfma_f64_4x4x4f64 chain  (= 2)
 * 1 = reference-shaped one-thread-per-feature kernel (debug / cross-check).  2 and 3 force a variant
 * (half_patch 5, 7 or 10; other sizes fall back to the 4-wave kernel).  Every variant 0-3 produces
 * bit-identical results.
 * 4 = EXPERIMENT, never chosen automatically: the 4-wave kernel with the reference's summation order
 *     given up (strided partial sums + tree instead of the 441-step ordered chains).  Not parity-exact:
 *     it exists to measure what the ordered accumulation costs.
 * 5 = four features per wavefront (block q of the f64 MFMA, row q of the cost chain and lane = feature solve shared
 *     by four features): a throughput variant for very large launches; bit-identical like 0-3.
 *     A launch of this variant that would end with an exposed tail (0.45 to 1.25 rounds of resident waves, the
 *     context alone on the device, not inside a graph capture) hands the features that have run 20 iterations to the
 *     4-wave latency kernel, which runs beside it on the context's auxiliary stream; results are the same bits.
 * 6 = 5 with the four rows of a wave independent (each row runs its own feature at its own level and takes the next
 *     feature from a work queue when it is done; a resident grid): bit-identical like 0-3.
 * 7 = 5 with one pyramid LEVEL per wavefront: pyramids x ceil(n / 4) wavefronts, each a third (a quarter) of the
 *     lifetime of a whole-feature wavefront, a quad handed from its level to the next through device memory (the
 *     waiting wavefront always waits for one that started earlier; the wait is bounded all the same and a launch in
 *     which one ran out reports PAGK_E_HIP at the next synchronisation).  For launches of about one to two rounds of
 *     resident wavefronts, which otherwise end with most of the device idle.  Bit-identical like 0-3.  With one
 *     pyramid level, or calc_ncc, it is 5. */
int pagk_set_kernel(pagk_ctx *ctx, int32_t which);
/* Variants 2 and 6 no longer win at any launch size and nothing selects them automatically; the product's build leaves
 * them out (pagk_set_kernel returns PAGK_E_UNSUPPORTED for them) and a library compiled with -DPAGK_ALL_VARIANTS carries
 * them for cross-checks.  1 = `which` can be selected in this build. */
int pagk_has_variant(int32_t which);
/* The variant (numbering above; 0 = the 4-wave kernel) the last tracking launch of this context actually used;
 * -1 before the first launch. */
int pagk_last_variant(const pagk_ctx *ctx);

/* Number of features the last tracking launch of this context handed from the throughput kernel to the latency
 * kernel (variant 5, see above); 0 when the launch did not use the hand-over.  Synchronises the context's stream. */
int pagk_last_handover(pagk_ctx *ctx);

/* Diagnostic: the threshold K the next launch of the 4-wave kernels will use for its issue priorities (a workgroup that has used more
 * than K iterations per pyramid level entered outranks its neighbours; csrc/pagk_prio.h -- no arithmetic effect).  4 by default;
 * PAGK_PRIO_K in the environment of pagk_create fixes another value (0 = off) or, as "auto", lets K follow the mean number of
 * iterations per feature and level the context's launches have run.  No counterpart in the reference.  Synchronises the context's stream. */
int pagk_priority_threshold(pagk_ctx *ctx);

/* Concurrency hint for the automatic selection: the caller runs `streams` contexts like this one at the same time on
 * this device (one PatchMatch per camera stream, BASELINE configs[4]: src/patch_match.cpp:79-142 called from several
 * threads).  The launch-size thresholds above are then applied to streams * n: a launch that would get a latency
 * variant on an empty device gets the throughput variant when the device is shared -- measured with eight concurrent
 * 1280x720 x 4000 streams: 26-29 instead of 17.6 Mfeat/s in aggregate (profiles/r02_ab_runs.md).  Results do not
 * depend on the variant.  streams = 1 (default) .. 64. */
int pagk_set_concurrency(pagk_ctx *ctx, int32_t streams);

/* Milliseconds spent in the tracking kernel(s) of the last pagk_track*_ call,
 * measured with HIP events on the stream the kernels ran on. Synchronises. */
int pagk_last_kernel_ms(pagk_ctx *ctx, float *track_ms, float *pyramid_ms);

/* ---- producer / consumer rows next to the path ----------------------------- */
/* GyroAidedTracker::GyroPredictFeatures + GyroPredictOnePixel (src/gyro_aided_tracker.cpp:118-185,194-256), both
 * prediction methods (params->predict_method: PIXEL_AWARE_PREDICTION :212-231, or SINGLE_HOMOGRAPHY :233-253, the
 * same with lambda = 1), on the device: predicted point (un-distorted and
 * distorted), border status and the 2x2 affine A = C B^T (B B^T)^-1 from the four predicted patch corners.
 * Produces exactly the arrays pagk_track_device consumes, so prediction -> tracking needs no host
 * round trip.  All pointers are device pointers; camera model from params (fx fy cx cy dist_coef);
 * KRKinv = mK * mRcl * mK^-1 (3x3 row-major, :518), r3 = third row of mRcl.  Where a prediction leaves
 * the image the outputs keep the tracker's initial state: status 0, points (0,0), affine untouched.
 * d_affine may be NULL.  Asynchronous on the context stream. */
int pagk_gyro_predict_device(pagk_ctx *ctx, const pagk_params *params, int32_t width, int32_t height,
                             const float *KRKinv, const float *r3, int32_t n, const float *d_pt_ref_un,
                             float *d_pt_predict_un, float *d_pt_predict, uint8_t *d_status, float *d_affine);
/* The same with the rotation in DEVICE memory: d_rot = 9 floats, rows 0 and 1 of KRKinv followed by r3.
 * For graph capture: kernel arguments are frozen into a captured graph, device memory is not, so a graph
 * holding upload -> pyramid -> prediction -> tracking replays every frame with that frame's rotation. */
int pagk_gyro_predict_device_rot(pagk_ctx *ctx, const pagk_params *params, int32_t width, int32_t height,
                                 const float *d_rot, int32_t n, const float *d_pt_ref_un, float *d_pt_predict_un,
                                 float *d_pt_predict, uint8_t *d_status, float *d_affine);

/* Tracker-side post-filter, GyroAidedTracker::GyroPredictFeaturesAndOpticalFlowRefined
 * Step 3 (src/gyro_aided_tracker.cpp:289-341): thresholds from the mean pixel
 * error, final inlier mask, survivors' points copied into pt_predict(_un).
 * Host-side; returns the number of survivors (>= 0) or a negative error. */
int pagk_post_filter(int32_t n, int32_t half_patch, const uint8_t *status_pm,
                     const double *pix_err, const double *dist_pred, const float *pt_pm,
                     const float *pt_pm_un, uint8_t *status_out, float *pt_predict,
                     float *pt_predict_un);

/* Diagnostics (never on the tracking path): the arithmetic of H.llt().solve(b) / update.norm()
 * (src/patch_match.cpp:319,343) on the caller's operands, so that a host can check on its own device -- and, with
 * Eigen at hand, against its own Eigen -- what pagk_params::solver_variant selects.
 * pagk_selftest_divide: per item i, q_plain[i] = num[i] / den[i] (the compiler's correctly rounded division),
 * q_prepared[i] = the same quotient through the prepared-denominator form the kernels use, root[i] = sqrt(num[i]),
 * root_lean[i] = the same root through the solve's ten-instruction form (plain where num[i] is outside its range).
 * pagk_selftest_solve: per 4x4 system (H row-major, lower triangle read; b) the update x and its norm from the
 * one-lane form (x_serial, norm_serial = sqrt of the squared norm) and from the four-lane form (x_lanes, nsq_lanes =
 * the squared norm the kernels compare with the threshold equivalent to `norm < 1e-2`).  Host pointers. */
int pagk_selftest_divide(pagk_ctx *ctx, int32_t n, const double *num, const double *den, double *q_plain,
                         double *q_prepared, double *root, double *root_lean);
int pagk_selftest_solve(pagk_ctx *ctx, int32_t n, const double *H, const double *b, uint32_t solver_variant,
                        double *x_serial, double *norm_serial, double *x_lanes, double *nsq_lanes);
/* pagk_selftest_repeat_sum: H(2,2) of src/patch_match.cpp:296 is the ordered sum of `count` = (2h+1)^2 copies of c * c
 * (J[2] = de_dg = c is constant over the patch, :263).  The pipelined 4-wave kernel (half_patch 8, 9, 10) computes it in
 * closed form -- binade by binade, ~100 instructions -- instead of as a `count`-step chain; per item i this returns the
 * closed form (closed[i]) beside the loop s = fma(c, c, s) (loop[i]) for the caller's c[i].  57 < count <= 480.
 * Host pointers. */
int pagk_selftest_repeat_sum(pagk_ctx *ctx, int32_t n, const float *c, int32_t count, double *closed, double *loop);

/* hipGraph capture of the per-frame work (BASELINE configs[4], "hipGraph-captured iterate").  A camera
 * stream issues the same launches on the same device pointers every frame; between pagk_graph_begin and
 * pagk_graph_end the *_device entry points (pagk_frame_set_device, pagk_gyro_predict_device[_rot],
 * pagk_track_device, pagk_geometry_scores_device) are recorded on the context stream instead of executed,
 * pagk_graph_launch replays them with one hipGraphLaunch.  Rules: run the same calls once before capturing
 * (nothing may allocate during capture); host-buffer and synchronising entry points return PAGK_E_ARG while
 * capturing; the context stream must not be the legacy default stream; the kernel timers
 * (pagk_last_kernel_ms) do not see replays.  Up to 8 graphs per context.  The reference has no counterpart:
 * its per-frame loop is Examples/Demo/RealSenseD435i.cpp:199-321.
 * A captured pagk_track_device that hands its stragglers to the latency kernel (kernels 5 / 7 on large launches) is
 * replayed in SEGMENTS: HIP replays the parallel branches of one graph one after the other, and that kernel has to run
 * beside the throughput kernel, so the capture is closed in front of it, the finisher becomes a plain launch on the
 * context's auxiliary stream, and the capture reopens -- pagk_graph_launch then issues graph, finisher, graph, graph.
 * Same results, and the replayed step is as fast as the direct one (BASELINE configs[3]: 0.57 ms either way).  Such a
 * capture must not have forked other streams into itself at that point. */
int pagk_graph_begin(pagk_ctx *ctx);
int pagk_graph_end(pagk_ctx *ctx, int32_t *graph_id);
int pagk_graph_launch(pagk_ctx *ctx, int32_t graph_id);
int pagk_graph_destroy(pagk_ctx *ctx, int32_t graph_id);

/* Geometry validation, the consumer after the post-filter:
 * the per-correspondence scoring loops of GyroAidedTracker::CheckHomography
 * (src/gyro_aided_tracker.cpp:620-676) and ::CheckFundamental (:704-768).  The RANSAC fits in front
 * of them (cv::findHomography / cv::findFundamentalMat, :596, :699) and H21.inv() (:597) are
 * third-party and stay with the caller, who passes the fitted 3x3 matrices (row-major double, the
 * layout of a CV_64F cv::Mat).  One launch scores both models (the reference runs them on two
 * threads, :455-460); inlier flags and the float scores, accumulated in index order, are
 * bit-identical to the reference loops.  pts1 / pts2: n x 2 float (mvKeysRefUn[i].pt, mvPtPredictUn[i]
 * of the status-true features, :434-440).
 *   pagk_geometry_scores_device: device pointers (d_scores: 2 floats, [0] = H, [1] = F),
 *                                asynchronous on the context stream;
 *   pagk_geometry_scores:        host buffers, synchronous. */
int pagk_geometry_scores_device(pagk_ctx *ctx, const double *H21, const double *H12, const double *F21,
                                int32_t n, const float *d_pts1, const float *d_pts2, float sigma,
                                uint8_t *d_inliers_H, uint8_t *d_inliers_F, float *d_scores);
int pagk_geometry_scores(pagk_ctx *ctx, const double *H21, const double *H12, const double *F21, int32_t n,
                         const float *pts1, const float *pts2, float sigma, uint8_t *inliers_H,
                         uint8_t *inliers_F, float *score_H, float *score_F);
/* Model choice of GeometryValidation (:462-470): 1 = homography (RH = score_H / (score_F + score_H)
 * > 0.45), 0 = fundamental. */
int pagk_geometry_select(float score_H, float score_F);
/* GyroAidedTracker::GeometryValidation (:429-480) around the fits: compacts the status-true
 * correspondences (:434-440), does nothing unless more than 8 remain (:445), scores both models on the
 * device, clears the status of the chosen model's outliers (:472-480).  status: n flags, updated in
 * place; track_score (may be NULL) receives the chosen model's score.  Host buffers, synchronous.
 * Returns cnt_inlier (>= 0; 0 when nothing was validated) or a negative error. */
int pagk_geometry_validation(pagk_ctx *ctx, const double *H21, const double *H12, const double *F21,
                             int32_t n, const float *pt_ref_un, const float *pt_predict_un,
                             uint8_t *status, float sigma, float *track_score);

/* ---- NCC nearest-neighbour matching ------------------------------ */
/* GyroAidedTracker::FindAndSortNearNeighbor (src/gyro_aided_tracker.cpp:788-851) for all n reference keypoints
 * (the reference's cv::parallel_for_ over [0, mN), :912): for every feature with status 1 and no neighbours yet,
 * the current keypoints j whose undistorted position lies within level * radius_unit of the predicted point in
 * both coordinates (:811-815), each scored with the free NCC (src/utils.cpp:110-148) of the reference patch
 * around mvKeysRef[i].pt against the patch around mvKeysCur[j].pt warped by the feature's affine A -- sampled
 * with the free GetPixelValue of include/utils.h:32-46 (`>` clamp, four-term formula), not with
 * PatchMatch::GetPixelValue -- and sorted by the two-stack insertion of :825-842 (best first; equal keys: the
 * later index first).
 *   keys_ref      n x 2  mvKeysRef[i].pt           pt_predict_un  n x 2  mvPtPredictUn[i]
 *   status        n      mvStatus[i]               affine         n x 4  mvAffineDeformationMatrix[i]; NULL = empty Mat
 *   keys_cur      m x 2  mvKeysCur[j].pt           keys_cur_un    m x 2  mvKeysCurUn[j].pt
 *   level                1 or 2 (:912, :922)       radius_unit           mRadiusForFindNearNeighbor (= 2 * h, :62)
 *   use_ncc              mbNCC (:60: true); 0 sorts by distance, nearest first
 *   count         n      IN/OUT: mvvNearNeighbors[i].size(); features with count > 0 are skipped (:793), so a
 *                        second call with level 2 only fills the features the first left empty.  Zero it first.
 *   nbr_idx/dist/ncc  n x cap  sMatch::trainIdx / distance / ncc of the sorted lists (queryIdx = i, level = level)
 * pagk_near_neighbors_device: frames in slots (level 0 is used), device pointers, asynchronous on the context
 * stream; a list longer than cap leaves only its true size in d_count[i].  pagk_find_near_neighbors: host
 * buffers, synchronous; returns PAGK_E_CAPACITY when a list did not fit (count[] then holds the sizes needed). */
int pagk_near_neighbors_device(pagk_ctx *ctx, int32_t slot_ref, int32_t slot_cur, int32_t half_patch, int32_t n,
                               const float *d_keys_ref, const float *d_pt_predict_un, const uint8_t *d_status,
                               const float *d_affine, int32_t m, const float *d_keys_cur, const float *d_keys_cur_un,
                               int32_t level, float radius_unit, int32_t use_ncc, int32_t cap, int32_t *d_count,
                               int32_t *d_nbr_idx, float *d_nbr_dist, float *d_nbr_ncc);
int pagk_find_near_neighbors(pagk_ctx *ctx, const pagk_image *ref, const pagk_image *cur, int32_t half_patch, int32_t n,
                             const float *keys_ref, const float *pt_predict_un, const uint8_t *status,
                             const float *affine, int32_t m, const float *keys_cur, const float *keys_cur_un,
                             int32_t level, float radius_unit, int32_t use_ncc, int32_t cap, int32_t *count,
                             int32_t *nbr_idx, float *nbr_dist, float *nbr_ncc);
/* The free NCC(halfPatchSize, ref, cur, pt_ref, pt_cur, warp_mat) of src/utils.cpp:166-200 for n point pairs
 * (affine: n x 4 or NULL = empty warp_mat).  Host buffers, synchronous. */
int pagk_ncc_free(pagk_ctx *ctx, const pagk_image *ref, const pagk_image *cur, int32_t half_patch, int32_t n,
                  const float *pt_ref, const float *pt_cur, const float *affine, float *ncc);
/* GyroAidedTracker::MatchFeatures (src/gyro_aided_tracker.cpp:949-1008) on the lists above: thresholds
 * TH_NCC_HIGH 0.6, TH_NCC_LOW 0.3, TH_RATIO 0.75 (:7-9), one match per current keypoint -- a keypoint claimed
 * twice loses every match and stays banned (:991-1005).  Host-side (sequential by nature).  match_*: capacity n
 * (match_dist / match_ncc may be NULL).  Returns mvMatches.size() or a negative error. */
int pagk_match_features(int32_t n, int32_t cap, const int32_t *count, const int32_t *nbr_idx, const float *nbr_dist,
                        const float *nbr_ncc, int32_t use_ncc, int32_t *match_query, int32_t *match_train,
                        float *match_dist, float *match_ncc);

/* ---- the path sharded over the GPUs of one node ---------------------- */
/* Features are independent units (the cv::parallel_for_ of src/patch_match.cpp:103), so the path shards by
 * contiguous index blocks of ceil(n / G) features: rank r owns [r * ceil(n/G), min(n, (r+1) * ceil(n/G))).  Every
 * GPU holds both pyramids; the only exchange is ONE all-gather (RCCL ncclAllGather over xGMI) of the packed
 * per-rank result slice, after which every rank holds every result and the tracker's global post-filter
 * (src/gyro_aided_tracker.cpp:289-341) runs on them in index order.  The library owns the RCCL communicator
 * (loaded with dlopen on first use); failures of RCCL calls return PAGK_E_NCCL. */
typedef struct pagk_multi pagk_multi;
/* One process driving n_devices GPUs (a C++ GyroAidedTracker host): one context and one RCCL rank per device
 * (ncclCommInitAll).  n_devices = 1 is a valid group. */
int pagk_multi_create(pagk_multi **out, const int32_t *devices, int32_t n_devices);
/* One process per GPU (e.g. under torchrun): rank 0 calls pagk_multi_unique_id, the host application hands the 128
 * bytes to every rank through its own channel, each rank joins with its device (ncclCommInitRank). */
int pagk_multi_unique_id(uint8_t id[128]);
int pagk_multi_create_rank(pagk_multi **out, const uint8_t id[128], int32_t rank, int32_t world, int32_t device);
void pagk_multi_destroy(pagk_multi *pm);
int32_t pagk_multi_world(const pagk_multi *pm);  /* ranks of the group                        */
int32_t pagk_multi_local(const pagk_multi *pm);  /* ranks driven by this process              */
int32_t pagk_multi_comm_count(const pagk_multi *pm); /* ncclCommCount of the group's communicator: the ranks RCCL itself
                                                        reports (a benchmark line's proof that the gather spans N GPUs);
                                                        negative when unavailable */
pagk_ctx *pagk_multi_ctx(pagk_multi *pm, int32_t local_index);  /* owned by the group; do not pagk_destroy */
const char *pagk_multi_last_error(const pagk_multi *pm);
/* The partition and the layout of a rank's packed result slice for m = ceil(n / G) features: seven SoA blocks in
 * SetMatcher order (pt_un, pt_dist, status, pix_err, dist_pred, ncc, iters), each padded to 8 bytes; returns the
 * slice size in bytes. */
void pagk_shard_range(int32_t n, int32_t rank, int32_t world, int32_t *lo, int32_t *hi);
size_t pagk_shard_layout(int32_t m, size_t offsets[7]);
/* The exchange: every local member k contributes `bytes` bytes at d_send[k] and receives world * bytes at
 * d_recv[k], in rank order, on its context's stream (hip_streams: NULL, or one stream per local member).
 * Asynchronous; ordered on the stream after the tracking launch that produced the slice. */
int pagk_multi_allgather(pagk_multi *pm, const void *const *d_send, void *const *d_recv, size_t bytes,
                         void *const *hip_streams);
/* pagk_track with the feature loop split over the group (single-process groups): same arguments, same results
 * bit for bit.  Every GPU uploads both frames and builds both pyramids, tracks its block, the packed slices are
 * all-gathered, the host reads the gathered result from member 0.  Synchronous. */
int pagk_track_sharded(pagk_multi *pm, const pagk_params *params, const pagk_image *ref, const pagk_image *cur,
                       int32_t n, const float *pt_ref_un, const float *pt_init_un, const float *affine,
                       const uint8_t *status_in, const pagk_outputs *out);

#ifdef __cplusplus
}
#endif
#endif /* PAGK_H */
