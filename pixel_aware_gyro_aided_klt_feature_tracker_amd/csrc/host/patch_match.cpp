// patch_match.cpp -- PatchMatch over the C ABI.  Mirrors the reference's interface
// (include/patch_match.h:41-103, src/patch_match.cpp:33-59,79-142,370-388); the arithmetic lives in
// libpagk_hip.so.
#include "patch_match.h"

#include <stdexcept>
#include <string>

#include "gyro_aided_tracker.h"

namespace {
// One context per host thread (the reference builds a PatchMatch per frame pair on whatever thread calls it, and a
// PatchMatch is not re-entrant).  The holder's destructor runs at thread exit, so a worker thread that tracked a
// few frames does not leak its HIP stream, device slots and pinned buffers; ReleaseContext() frees it earlier.
struct CtxHolder {
    pagk_ctx *ctx = nullptr;
    ~CtxHolder()
    {
        if (ctx) pagk_destroy(ctx);
    }
};
thread_local CtxHolder g_holder;
}  // namespace

pagk_ctx *PatchMatch::Context(int device)
{
    if (!g_holder.ctx) {
        int rc = pagk_create(&g_holder.ctx, device);
        if (rc != PAGK_OK) {
            g_holder.ctx = nullptr;
            throw std::runtime_error(std::string("PatchMatch: pagk_create failed: ") + pagk_strerror(rc) +
                                     " (the HIP path is the only implementation; there is no CPU fallback)");
        }
    }
    return g_holder.ctx;
}

void PatchMatch::ReleaseContext()
{
    if (g_holder.ctx) pagk_destroy(g_holder.ctx);
    g_holder.ctx = nullptr;
}

// reference src/patch_match.cpp:33-59
PatchMatch::PatchMatch(GyroAidedTracker *pMatcher_, int halfPatchSize_, int iterations_, int pyramids_,
                       bool bHasGyroPredictInitial_, bool bInverse_, bool bConsiderIllumination_,
                       bool bConsiderAffineDeformation_, bool bRegularizationPenalty_, bool bCalculateNCC_)
    : mpMatcher(pMatcher_), mN((int)pMatcher_->mvKeysRef.size()), mHalfPatchSize(halfPatchSize_),
      mIterations(iterations_), mPyramids(pyramids_), mbHasGyroPredictInitial(bHasGyroPredictInitial_),
      mbInverse(bInverse_), mbConsiderIllumination(bConsiderIllumination_),
      mbConsiderAffineDeformation(bConsiderAffineDeformation_), mbRegularizationPenalty(bRegularizationPenalty_),
      mbCalculateNCC(bCalculateNCC_),
      mvGyroPredictStatus(pMatcher_->mvStatus.begin(), pMatcher_->mvStatus.end())  // :58
{
}

void PatchMatch::OpticalFlowMultiLevel()
{
    GyroAidedTracker &T = *mpMatcher;
    pagk_params p;
    pagk_params_default(&p);  // mLambda, mAlpha, mMaxDistance of :48-50
    p.half_patch = mHalfPatchSize;
    p.iterations = mIterations;
    p.pyramids = mPyramids;
    p.has_gyro_predict_initial = mbHasGyroPredictInitial;
    p.inverse = mbInverse;
    p.consider_illumination = mbConsiderIllumination;
    p.consider_affine = mbConsiderAffineDeformation;
    p.regularization_penalty = mbRegularizationPenalty;
    p.calculate_ncc = mbCalculateNCC;
    p.fx = T.mK.at<float>(0, 0), p.fy = T.mK.at<float>(1, 1);
    p.cx = T.mK.at<float>(0, 2), p.cy = T.mK.at<float>(1, 2);
    p.n_dist_coef = (int)T.mDistCoef.total();
    for (int k = 0; k < p.n_dist_coef && k < 5; k++) p.dist_coef[k] = T.mDistCoef.at<float>(k);

    const cv::Mat &r = T.mImgGrayRef, &c = T.mImgGrayCur;
    pagk_image ref{r.data, r.cols, r.rows, (int64_t)r.step}, cur{c.data, c.cols, c.rows, (int64_t)c.step};

    const int n = mN;
    std::vector<float> ptRef(2 * (size_t)n + 2), ptInit(2 * (size_t)n + 2), aff(4 * (size_t)n + 4, 0.f);
    for (int i = 0; i < n; i++) {  // :83-90 reads mvKeysRefUn[i].pt and mvPtPredictUn[i]
        ptRef[2 * i] = T.mvKeysRefUn[i].pt.x, ptRef[2 * i + 1] = T.mvKeysRefUn[i].pt.y;
        ptInit[2 * i] = T.mvPtPredictUn[i].x, ptInit[2 * i + 1] = T.mvPtPredictUn[i].y;
        const cv::Mat &A = T.mvAffineDeformationMatrix[i];  // empty where GyroPredictFeatures `continue`d
        if (!A.empty())
            for (int k = 0; k < 4; k++) aff[4 * i + k] = A.at<float>(k / 2, k % 2);
    }
    // SetMatcher (:372-377)
    T.mvPtPredictAfterPatchMatched.resize(n);
    T.mvPtPredictAfterPatchMatchedUn.resize(n);
    T.mvStatusAfterPatchMatched.resize(n);
    T.mvPixelErrorsOfPatchMatched.resize(n);
    T.mvDistanceBetweenPredictedAndPatchMatched.resize(n);
    T.mvNccAfterPatchMatched.resize(n);
    static_assert(sizeof(cv::Point2f) == 2 * sizeof(float), "Point2f must be two packed floats");
    std::vector<float> dummy2(2);
    std::vector<cv::uchar> dummy1(1);
    pagk_outputs out{};
    out.pt_un = n ? reinterpret_cast<float *>(T.mvPtPredictAfterPatchMatchedUn.data()) : dummy2.data();
    out.pt_dist = n ? reinterpret_cast<float *>(T.mvPtPredictAfterPatchMatched.data()) : nullptr;
    out.status = n ? T.mvStatusAfterPatchMatched.data() : dummy1.data();
    out.pix_err = n ? T.mvPixelErrorsOfPatchMatched.data() : nullptr;
    out.dist_pred = n ? T.mvDistanceBetweenPredictedAndPatchMatched.data() : nullptr;
    out.ncc = n ? T.mvNccAfterPatchMatched.data() : nullptr;
    out.iters = nullptr;
    std::vector<cv::uchar> st(mvGyroPredictStatus);
    st.resize((size_t)n + 1);
    int rc = pagk_track(Context(), &p, &ref, &cur, n, ptRef.data(), ptInit.data(), aff.data(), st.data(), &out);
    if (rc != PAGK_OK)
        throw std::runtime_error(std::string("PatchMatch::OpticalFlowMultiLevel: pagk_track: ") + pagk_strerror(rc) +
                                 " " + pagk_last_error(Context()));
}
