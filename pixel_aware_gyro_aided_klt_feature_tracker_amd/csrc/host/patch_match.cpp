// patch_match.cpp -- PatchMatch over the C ABI.  Mirrors the reference's interface
// (include/patch_match.h:41-103, src/patch_match.cpp:33-59,79-142,370-388); the arithmetic lives in
// libpagk_hip.so.
#include "patch_match.h"

#include <cmath>
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
      mbCalculateNCC(bCalculateNCC_), mPyramidScale(0.5),  // :54
      mLevel(pyramids_ - 1),
      mvGyroPredictStatus(pMatcher_->mvStatus.begin(), pMatcher_->mvStatus.end())  // :58
{
}

pagk_params PatchMatch::MakeParams() const
{
    const GyroAidedTracker &T = *mpMatcher;
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
    return p;
}

// :83-95
void PatchMatch::InitPoints()
{
    const GyroAidedTracker &T = *mpMatcher;
    mvPtPyr1Un.resize(mN);
    mvPtPyr2Un.resize(mN);
    for (int i = 0; i < mN; i++) {
        mvPtPyr1Un[i] = T.mvKeysRefUn[i].pt;
        mvPtPyr2Un[i] = mbHasGyroPredictInitial ? T.mvPtPredictUn[i] : T.mvKeysRefUn[i].pt;
    }
    mvSuccess.assign(mN, false);
    mvPixelErrorsOfPatchMatched.assign(mN, 0.0);
    mvNcc.assign(mN, 0.0f);
}

void PatchMatch::CreatePyramids()
{
    GyroAidedTracker &T = *mpMatcher;
    pagk_ctx *ctx = Context();
    const cv::Mat *src[2] = {&T.mImgGrayRef, &T.mImgGrayCur};
    std::vector<cv::Mat> *dst[2] = {&mvImgPyr1, &mvImgPyr2};
    for (int k = 0; k < 2; k++) {
        const cv::Mat &m = *src[k];
        pagk_image im{m.data, m.cols, m.rows, (int64_t)m.step};
        int rc = pagk_frame_upload(ctx, k, &im, mPyramids);
        if (rc != PAGK_OK)
            throw std::runtime_error(std::string("PatchMatch::CreatePyramids: pagk_frame_upload: ") + pagk_strerror(rc) + " " +
                                     pagk_last_error(ctx));
        dst[k]->assign(1, m);  // :64-65 level 0 shares the tracker's image
        int w = m.cols, h = m.rows;
        for (int l = 1; l < mPyramids; l++) {
            w = (int)(w * mPyramidScale);  // :69  cv::Size(cols * 0.5, rows * 0.5)
            h = (int)(h * mPyramidScale);
            cv::Mat lv(h, w, cv::CV_8UC1);
            int32_t gw = 0, gh = 0;
            rc = pagk_frame_download_level(ctx, k, l, lv.data, &gw, &gh);
            if (rc != PAGK_OK || gw != w || gh != h)
                throw std::runtime_error(std::string("PatchMatch::CreatePyramids: pagk_frame_download_level: ") + pagk_strerror(rc));
            dst[k]->push_back(lv);
        }
    }
    mvScales.assign(1, 1.0f);  // :66
    for (int l = 1; l < mPyramids; l++) mvScales.push_back((float)(mvScales[l - 1] * mPyramidScale));  // :73
}

void PatchMatch::OpticalFlowConsideringIlluminationChange_onePixel(const int i, const bool bConsiderIllumination,
                                                                   const bool bConsiderAffineDeformation,
                                                                   const bool bRegularizationPenalty)
{
    if ((int)mvImgPyr1.size() != mPyramids) CreatePyramids();
    if ((int)mvPtPyr2Un.size() != mN) InitPoints();
    if (i < 0 || i >= mN || mLevel < 0 || mLevel >= mPyramids) throw std::runtime_error("PatchMatch: feature or level out of range");
    if (!mvGyroPredictStatus[i]) return;  // :173
    GyroAidedTracker &T = *mpMatcher;
    const float scale = mvScales[mLevel];
    // the level's coordinates, formed exactly as :177-182 forms them; the single-level launch then multiplies by its own
    // scale 1.0f, which changes nothing
    const cv::Point2f pt(mvPtPyr1Un[i].x * scale, mvPtPyr1Un[i].y * scale);
    cv::Point2f next;
    if (mLevel == mPyramids - 1)
        next = cv::Point2f(mvPtPyr2Un[i].x * scale, mvPtPyr2Un[i].y * scale);
    else
        next = cv::Point2f((float)((double)(mvPtPyr2Un[i].x * 1.0f) / mPyramidScale), (float)((double)(mvPtPyr2Un[i].y * 1.0f) / mPyramidScale));
    pagk_params p = MakeParams();
    p.pyramids = 1;
    p.has_gyro_predict_initial = 1;  // (`next` is the level's initial point whatever the tracker predicted)
    p.consider_illumination = bConsiderIllumination;
    p.consider_affine = bConsiderAffineDeformation;
    p.regularization_penalty = bRegularizationPenalty;
    p.calculate_ncc = 0;  // :356-366 scores on the level-0 images at every level: done below with NCC()
    float aff[4] = {1, 0, 0, 1};
    const cv::Mat &A = T.mvAffineDeformationMatrix[i];
    if (!A.empty())
        for (int k = 0; k < 4; k++) aff[k] = A.at<float>(k / 2, k % 2);
    const cv::Mat &r = mvImgPyr1[mLevel], &c = mvImgPyr2[mLevel];
    pagk_image lr{r.data, r.cols, r.rows, (int64_t)r.step}, lc{c.data, c.cols, c.rows, (int64_t)c.step};
    float pin[2] = {pt.x, pt.y}, pnext[2] = {next.x, next.y}, pout[2] = {0, 0};
    cv::uchar one = 1, st = 0;
    double err = 0;
    pagk_outputs out{};
    out.pt_un = pout;
    out.status = &st;
    out.pix_err = &err;
    int rc = pagk_track_pyr(Context(), &p, 1, &lr, &lc, 1, pin, pnext, aff, &one, &out);
    if (rc != PAGK_OK)
        throw std::runtime_error(std::string("PatchMatch::..._onePixel: pagk_track_pyr: ") + pagk_strerror(rc) + " " +
                                 pagk_last_error(Context()));
    mvPtPyr2Un[i] = cv::Point2f(pout[0], pout[1]);  // :348
    if (mLevel == 0) {                               // :350-353
        mvSuccess[i] = st != 0;
        mvPixelErrorsOfPatchMatched[i] = err;
    }
    if (mbCalculateNCC)  // :356-366
        mvNcc[i] = NCC(mHalfPatchSize, mvImgPyr1[0], mvImgPyr2[0], mvPtPyr1Un[i], mvPtPyr2Un[i],
                       bConsiderAffineDeformation ? A : cv::Mat());
    else
        mvNcc[i] = 1;
}

// :409-416 -> src/utils.cpp:49-76
void PatchMatch::DistortPoints()
{
    GyroAidedTracker &T = *mpMatcher;
    if (T.mDistCoef.at<float>(0) == 0.0) {
        mvPtPyr2 = mvPtPyr2Un;
        return;
    }
    const float fx = T.mK.at<float>(0, 0), fy = T.mK.at<float>(1, 1), cx = T.mK.at<float>(0, 2), cy = T.mK.at<float>(1, 2);
    const float fx_inv = 1.0 / fx, fy_inv = 1.0 / fy;
    const float k1 = T.mDistCoef.at<float>(0), k2 = T.mDistCoef.at<float>(1), p1 = T.mDistCoef.at<float>(2),
                p2 = T.mDistCoef.at<float>(3), k3 = T.mDistCoef.total() == 5 ? T.mDistCoef.at<float>(4) : 0;
    mvPtPyr2.resize(mvPtPyr2Un.size());
    for (size_t i = 0; i < mvPtPyr2Un.size(); i++) {
        const float x = (mvPtPyr2Un[i].x - cx) * fx_inv, y = (mvPtPyr2Un[i].y - cy) * fy_inv;
        const float r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
        const float xd = x * (1 + k1 * r2 + k2 * r4 + k3 * r6) + 2 * p1 * x * y + p2 * (r2 + 2 * x * x);
        const float yd = y * (1 + k1 * r2 + k2 * r4 + k3 * r6) + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y;
        mvPtPyr2[i] = cv::Point2f(fx * xd + cx, fy * yd + cy);
    }
}

// :370-388
void PatchMatch::SetMatcher()
{
    GyroAidedTracker &T = *mpMatcher;
    T.mvPtPredictAfterPatchMatched.resize(mN);
    T.mvPtPredictAfterPatchMatchedUn.resize(mN);
    T.mvStatusAfterPatchMatched.resize(mN);
    T.mvPixelErrorsOfPatchMatched.resize(mN);
    T.mvDistanceBetweenPredictedAndPatchMatched.resize(mN);
    T.mvNccAfterPatchMatched.resize(mN);
    for (int i = 0; i < mN; i++) {
        T.mvPtPredictAfterPatchMatched[i] = mvPtPyr2[i];
        T.mvPtPredictAfterPatchMatchedUn[i] = mvPtPyr2Un[i];
        T.mvStatusAfterPatchMatched[i] = mvSuccess[i];
        T.mvPixelErrorsOfPatchMatched[i] = mvPixelErrorsOfPatchMatched[i];
        const cv::Point2f d = T.mvPtPredictUn[i] - mvPtPyr2Un[i];
        T.mvDistanceBetweenPredictedAndPatchMatched[i] = std::sqrt(d.x * d.x + d.y * d.y);  // float sqrt, stored as double
        T.mvNccAfterPatchMatched[i] = mvNcc[i];
    }
}

// :433-469
float PatchMatch::NCC(int halfPathSize, const cv::Mat &ref, const cv::Mat &cur, const cv::Point2f &pt_ref,
                      const cv::Point2f &pt_cur, const cv::Mat &warp_mat)
{
    std::vector<float> vr, vc;
    float mean_ref = 0.0f, mean_cur = 0.0f;
    for (int x = -halfPathSize; x <= halfPathSize; x++)
        for (int y = -halfPathSize; y <= halfPathSize; y++) {
            const float a = GetPixelValue(ref, pt_ref.x + x, pt_ref.y + y);
            float b;
            if (warp_mat.empty()) {
                b = GetPixelValue(cur, pt_cur.x + x, pt_cur.y + y);
            } else {
                const float wx = warp_mat.at<float>(0, 0) * x + warp_mat.at<float>(0, 1) * y;
                const float wy = warp_mat.at<float>(1, 0) * x + warp_mat.at<float>(1, 1) * y;
                b = GetPixelValue(cur, pt_cur.x + wx, pt_cur.y + wy);
            }
            mean_ref += a;
            mean_cur += b;
            vr.push_back(a);
            vc.push_back(b);
        }
    mean_ref /= vr.size();
    mean_cur /= vc.size();
    float num = 0, d1 = 0, d2 = 0;
    for (size_t k = 0; k < vr.size(); k++) {
        num += ((vr[k] - mean_ref) * (vc[k] - mean_cur));
        d1 += (vr[k] - mean_ref) * (vr[k] - mean_ref);
        d2 += (vc[k] - mean_cur) * (vc[k] - mean_cur);
    }
    return num / std::sqrt(d1 * d2 + 1e-10);
}

void PatchMatch::OpticalFlowMultiLevel()
{
    GyroAidedTracker &T = *mpMatcher;
    const pagk_params p = MakeParams();
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
    // ... and the object's own state as the reference leaves it (:115-116, :348-366)
    mvPtPyr1Un.resize(n);
    for (int i = 0; i < n; i++) mvPtPyr1Un[i] = T.mvKeysRefUn[i].pt;
    mvPtPyr2Un.assign(T.mvPtPredictAfterPatchMatchedUn.begin(), T.mvPtPredictAfterPatchMatchedUn.end());
    mvPtPyr2.assign(T.mvPtPredictAfterPatchMatched.begin(), T.mvPtPredictAfterPatchMatched.end());
    mvSuccess.assign(n, false);
    for (int i = 0; i < n; i++) mvSuccess[i] = T.mvStatusAfterPatchMatched[i] != 0;
    mvPixelErrorsOfPatchMatched.assign(T.mvPixelErrorsOfPatchMatched.begin(), T.mvPixelErrorsOfPatchMatched.end());
    mvNcc.assign(T.mvNccAfterPatchMatched.begin(), T.mvNccAfterPatchMatched.end());
    mLevel = 0;
}
