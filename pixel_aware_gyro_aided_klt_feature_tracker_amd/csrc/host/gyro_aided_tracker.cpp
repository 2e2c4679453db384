// gyro_aided_tracker.cpp -- host side of the hot path: producer of PatchMatch's inputs and consumer
// of its outputs, with the reference's names and arithmetic (src/gyro_aided_tracker.cpp:11-95,118-426,
// 511-587).  float32 throughout, double where the reference's expressions promote.
#include "gyro_aided_tracker.h"

#include <chrono>
#include <cmath>
#include <stdexcept>

#include "pagk.h"
#include "patch_match.h"

// reference src/gyro_aided_tracker.cpp:11-28.  NB: like the reference, mvKeysRefUn binds to
// vKeysRef_ (not vKeysUnRef_), :21.
GyroAidedTracker::GyroAidedTracker(double t, double t_ref, const cv::Mat &imgGrayRef_, const cv::Mat &imgGrayCur_,
                                   const std::vector<cv::KeyPoint> &vKeysRef_,
                                   const std::vector<cv::KeyPoint> &vKeysCur_,
                                   const std::vector<cv::KeyPoint> &vKeysUnRef_,
                                   const std::vector<cv::KeyPoint> &vKeysUnCur_,
                                   const std::vector<IMU::Point> &vImuFromLastFrame, const cv::Point3f &bias_,
                                   cv::Mat K_, cv::Mat DistCoef_, const cv::Mat &normalizeTable_, eType type_,
                                   ePredictMethod predictMethod_, std::string saveFolderPath, int halfPatchSize_)
    : mTimeStamp(t), mTimeStampRef(t_ref), mImgGrayRef(imgGrayRef_), mImgGrayCur(imgGrayCur_), mvKeysRef(vKeysRef_),
      mvKeysRefUn(vKeysRef_), mvKeysCur(vKeysCur_), mvKeysCurUn(vKeysUnCur_), mvImuFromLastFrame(vImuFromLastFrame),
      mHalfPatchSize(halfPatchSize_), mBias(bias_), mK(K_), mDistCoef(DistCoef_), mWidth(imgGrayCur_.cols),
      mHeight(imgGrayCur_.rows), mNormalizeTable(normalizeTable_), mType(type_), mPredictMethod(predictMethod_)
{
    (void)vKeysUnRef_;
    (void)saveFolderPath;  // result logging is out of scope
    mRbc = cv::Mat::eye(3, 3, cv::CV_32F);
    Initialize();
}

// reference :51-95
void GyroAidedTracker::Initialize()
{
    mHalfPatchSize = mHalfPatchSize == 0 ? 5 : mHalfPatchSize;  // :61
    mfx = mK.at<float>(0, 0), mfy = mK.at<float>(1, 1);
    mcx = mK.at<float>(0, 2), mcy = mK.at<float>(1, 2);
    mfx_inv = (float)(1.0 / mfx), mfy_inv = (float)(1.0 / mfy);  // :66
    mk1 = mDistCoef.at<float>(0), mk2 = mDistCoef.at<float>(1);
    mp1 = mDistCoef.at<float>(2), mp2 = mDistCoef.at<float>(3);
    mk3 = mDistCoef.total() == 5 ? mDistCoef.at<float>(4) : 0;
    mvPatchCorners.resize(4);
    const float h = (float)mHalfPatchSize;
    mvPatchCorners[0] = cv::Point2f(-h, -h);  // top left      :73-77
    mvPatchCorners[1] = cv::Point2f(h, -h);   // top right
    mvPatchCorners[2] = cv::Point2f(-h, h);   // bottom left
    mvPatchCorners[3] = cv::Point2f(h, h);    // bottom right
    mN = (int)mvKeysRef.size();
    mvPtPredict.assign(mN, cv::Point2f(0, 0));
    mvPtPredictUn.assign(mN, cv::Point2f(0, 0));
    mvFlowsPredictUn.assign(mN, cv::Point2f(0, 0));
    mvStatus.assign(mN, 0);
    mvError.resize(mN);
    mvvPtPredictCorners.resize(mN);
    mvvPtPredictCornersUn.resize(mN);
    mvvFlowsPredictCorners.resize(mN);
    mvAffineDeformationMatrix.assign(mN, cv::Mat());
}

static cv::Mat inv3(const cv::Mat &m)
{
    double a[3][3];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) a[r][c] = m.at<float>(r, c);
    double det = a[0][0] * (a[1][1] * a[2][2] - a[1][2] * a[2][1]) - a[0][1] * (a[1][0] * a[2][2] - a[1][2] * a[2][0]) +
                 a[0][2] * (a[1][0] * a[2][1] - a[1][1] * a[2][0]);
    cv::Mat o(3, 3, cv::CV_32F);
    double d = det != 0 ? 1.0 / det : 0;
    o.at<float>(0, 0) = (float)((a[1][1] * a[2][2] - a[1][2] * a[2][1]) * d);
    o.at<float>(0, 1) = (float)((a[0][2] * a[2][1] - a[0][1] * a[2][2]) * d);
    o.at<float>(0, 2) = (float)((a[0][1] * a[1][2] - a[0][2] * a[1][1]) * d);
    o.at<float>(1, 0) = (float)((a[1][2] * a[2][0] - a[1][0] * a[2][2]) * d);
    o.at<float>(1, 1) = (float)((a[0][0] * a[2][2] - a[0][2] * a[2][0]) * d);
    o.at<float>(1, 2) = (float)((a[0][2] * a[1][0] - a[0][0] * a[1][2]) * d);
    o.at<float>(2, 0) = (float)((a[1][0] * a[2][1] - a[1][1] * a[2][0]) * d);
    o.at<float>(2, 1) = (float)((a[0][1] * a[2][0] - a[0][0] * a[2][1]) * d);
    o.at<float>(2, 2) = (float)((a[0][0] * a[1][1] - a[0][1] * a[1][0]) * d);
    return o;
}

// reference :511-519
void GyroAidedTracker::SetRcl(const cv::Mat Rcl_)
{
    mRcl = Rcl_.clone();
    mr11 = mRcl.at<float>(0, 0), mr12 = mRcl.at<float>(0, 1), mr13 = mRcl.at<float>(0, 2);
    mr21 = mRcl.at<float>(1, 0), mr22 = mRcl.at<float>(1, 1), mr23 = mRcl.at<float>(1, 2);
    mr31 = mRcl.at<float>(2, 0), mr32 = mRcl.at<float>(2, 1), mr33 = mRcl.at<float>(2, 2);
    mKRKinv = mK * mRcl * inv3(mK);
}

// reference :521-562
void GyroAidedTracker::IntegrateGyroMeasurements()
{
    cv::Mat dR = cv::Mat::eye(3, 3, cv::CV_32F);
    const int n = (int)mvImuFromLastFrame.size() - 1;
    for (int i = 0; i < n; i++) {
        float tstep = 0;
        cv::Point3f angVel;
        const IMU::Point &p0 = mvImuFromLastFrame[i], &p1 = mvImuFromLastFrame[i + 1];
        if ((i == 0) && (i < (n - 1))) {
            float tab = (float)(p1.t - p0.t);
            float tini = (float)(p0.t - mTimeStampRef);
            angVel = (p0.w + p1.w - (p1.w - p0.w) * (tini / tab)) * 0.5f;
            tstep = (float)(p1.t - mTimeStampRef);
        } else if (i < (n - 1)) {
            angVel = (p0.w + p1.w) * 0.5f;
            tstep = (float)(p1.t - p0.t);
        } else if ((i > 0) && (i == (n - 1))) {
            float tab = (float)(p1.t - p0.t);
            float tend = (float)(p1.t - mTimeStamp);
            angVel = (p0.w + p1.w - (p1.w - p0.w) * (tend / tab)) * 0.5f;
            tstep = (float)(mTimeStamp - p0.t);
        } else if ((i == 0) && (i == (n - 1))) {
            angVel = p0.w;
            tstep = (float)(mTimeStamp - mTimeStampRef);
        }
        dR = dR * IntegrateOneGyroMeasurement(angVel, tstep);
    }
    SetRcl(mRbc.t() * dR.t() * mRbc);  // :560
}

// reference :564-587
cv::Mat GyroAidedTracker::IntegrateOneGyroMeasurement(cv::Point3f &gyro, double dt)
{
    const float x = (float)((gyro.x - mBias.x) * dt);
    const float y = (float)((gyro.y - mBias.y) * dt);
    const float z = (float)((gyro.z - mBias.z) * dt);
    const float d2 = x * x + y * y + z * z;
    const float d = std::sqrt(d2);
    cv::Mat W(3, 3, cv::CV_32F);
    const float w[9] = {0, -z, y, z, 0, -x, -y, x, 0};
    for (int k = 0; k < 9; k++) W.at<float>(k / 3, k % 3) = w[k];
    cv::Mat R = cv::Mat::eye(3, 3, cv::CV_32F);
    if (d < 1e-4) {
        for (int k = 0; k < 9; k++) R.at<float>(k / 3, k % 3) += w[k];
    } else {
        cv::Mat WW = W * W;
        const double s = std::sin(d), c = 1.0f - std::cos(d);
        for (int k = 0; k < 9; k++)
            R.at<float>(k / 3, k % 3) =
                (float)((double)R.at<float>(k / 3, k % 3) + (double)w[k] * s / d + (double)WW.at<float>(k / 3, k % 3) * c / d2);
    }
    return R;
}

// reference :194-231 (PIXEL_AWARE_PREDICTION) and :233-253 (SINGLE_HOMOGRAPHY: lambda = 1)
void GyroAidedTracker::GyroPredictOnePixel(cv::Point2f &pt_ref, cv::Point2f &pt_predict,
                                           cv::Point2f &pt_predict_distort, cv::Point2f &flow)
{
    float x_normal = (pt_ref.x - mcx) * mfx_inv;
    float y_normal = (pt_ref.y - mcy) * mfy_inv;
    float lambda = mPredictMethod == PIXEL_AWARE_PREDICTION
                       ? (float)(1.0 / (mr31 * x_normal + mr32 * y_normal + mr33))
                       : 1.0f;
    float pt_x = (mKRKinv.at<float>(0, 0) * pt_ref.x + mKRKinv.at<float>(0, 1) * pt_ref.y + mKRKinv.at<float>(0, 2)) * lambda;
    float pt_y = (mKRKinv.at<float>(1, 0) * pt_ref.x + mKRKinv.at<float>(1, 1) * pt_ref.y + mKRKinv.at<float>(1, 2)) * lambda;
    pt_predict = cv::Point2f(pt_x, pt_y);
    float x = (pt_predict.x - mcx) * mfx_inv;
    float y = (pt_predict.y - mcy) * mfy_inv;
    float r2 = x * x + y * y;
    float r4 = r2 * r2;
    float r6 = r4 * r2;
    float x_distort = x * (1 + mk1 * r2 + mk2 * r4 + mk3 * r6) + 2 * mp1 * x * y + mp2 * (r2 + 2 * x * x);
    float y_distort = y * (1 + mk1 * r2 + mk2 * r4 + mk3 * r6) + mp1 * (r2 + 2 * y * y) + 2 * mp2 * x * y;
    pt_predict_distort = cv::Point2f(mfx * x_distort + mcx, mfy * y_distort + mcy);
    flow = pt_predict - pt_ref;
}

// reference :118-185
int GyroAidedTracker::GyroPredictFeatures()
{
    auto t1 = std::chrono::steady_clock::now();
    const float hh = (float)mHalfPatchSize;
    // (B B^T)^-1 for B = [+-h corner matrix]: diag(1/(4h^2)) through the double determinant (cv::Mat::inv, 2x2)
    const float m00 = 4 * hh * hh;
    const double det = (double)m00 * m00;
    const double dinv = det != 0 ? 1. / det : 0;
    const float inv00 = (float)(m00 * dinv), inv01 = (float)(-0.0f * dinv);
    for (int i = 0; i < mN; i++) {
        cv::Point2f pt_ref_un = mvKeysRefUn[i].pt, pu, pd, flow;
        GyroPredictOnePixel(pt_ref_un, pu, pd, flow);
        if (pu.x < 0 || pu.x >= mWidth || pu.y < 0 || pu.y >= mHeight) continue;  // :131
        if (pd.x < 0 || pd.x >= mWidth || pd.y < 0 || pd.y >= mHeight) continue;  // :134
        mvPtPredictUn[i] = pu;
        mvPtPredict[i] = pd;
        mvStatus[i] = true;
        mvFlowsPredictUn[i] = flow;
        std::vector<cv::Point2f> corners, cornersUn, vecC;
        for (size_t j = 0; j < mvPatchCorners.size(); j++) {  // :148-160
            cv::Point2f c_un(mvKeysRefUn[i].pt + mvPatchCorners[j]), cu, cd, cf;
            GyroPredictOnePixel(c_un, cu, cd, cf);
            cornersUn.push_back(cu);
            corners.push_back(cd);
            vecC.push_back(cu - pu);
        }
        mvvPtPredictCorners[i] = corners;
        mvvPtPredictCornersUn[i] = cornersUn;
        mvvFlowsPredictCorners[i] = vecC;
        // A = C * B^T * (B * B^T)^-1, :166-167 (float products, double accumulation per Mat product)
        double s00 = 0, s01 = 0, s10 = 0, s11 = 0;
        for (int j = 0; j < 4; j++) {
            s00 += (double)vecC[j].x * mvPatchCorners[j].x;
            s01 += (double)vecC[j].x * mvPatchCorners[j].y;
            s10 += (double)vecC[j].y * mvPatchCorners[j].x;
            s11 += (double)vecC[j].y * mvPatchCorners[j].y;
        }
        const float t00 = (float)s00, t01 = (float)s01, t10 = (float)s10, t11 = (float)s11;
        cv::Mat A(2, 2, cv::CV_32F);
        A.at<float>(0, 0) = (float)((double)t00 * inv00 + (double)t01 * inv01);
        A.at<float>(0, 1) = (float)((double)t00 * inv01 + (double)t01 * inv00);
        A.at<float>(1, 0) = (float)((double)t10 * inv00 + (double)t11 * inv01);
        A.at<float>(1, 1) = (float)((double)t10 * inv01 + (double)t11 * inv00);
        mvAffineDeformationMatrix[i] = A;
    }
    int n_predict = 0;
    for (auto s : mvStatus) n_predict += s ? 1 : 0;
    mTimeCostGyroPredict = std::chrono::duration<float>(std::chrono::steady_clock::now() - t1).count();
    mvPtGyroPredict = mvPtPredict;
    mvPtGyroPredictUn = mvPtPredictUn;
    return n_predict;
}

// reference :258-342
int GyroAidedTracker::GyroPredictFeaturesAndOpticalFlowRefined()
{
    if (mbHasGyroPredictInitial)
        GyroPredictFeatures();
    else {
        for (int i = 0; i < mN; i++) {  // :264-270
            mvPtPredictUn[i] = mvKeysRefUn[i].pt;
            mvPtPredict[i] = mvKeysRef[i].pt;
            mvStatus[i] = true;
            mvFlowsPredictUn[i] = cv::Point2f(0, 0);
            mvAffineDeformationMatrix[i] = cv::Mat::eye(2, 2, cv::CV_32F);
        }
    }
    auto t1 = std::chrono::steady_clock::now();
    const bool inverse = false;  // :278
    PatchMatch patchMatch(this, mHalfPatchSize, mIterations, mPyramids, mbHasGyroPredictInitial, inverse,
                          mbConsiderIllumination, mbConsiderAffineDeformation, mbRegularizationPenalty);
    patchMatch.OpticalFlowMultiLevel();
    mTimeCostOptFlow = std::chrono::duration<float>(std::chrono::steady_clock::now() - t1).count();

    auto t3 = std::chrono::steady_clock::now();
    // Step 3 (:289-341): thresholds from the mean pixel error, final mask, survivors copied back
    std::vector<cv::uchar> status(mN ? mN : 1);
    int n_predict = pagk_post_filter(mN, mHalfPatchSize, mvStatusAfterPatchMatched.data(),
                                     mvPixelErrorsOfPatchMatched.data(), mvDistanceBetweenPredictedAndPatchMatched.data(),
                                     reinterpret_cast<const float *>(mvPtPredictAfterPatchMatched.data()),
                                     reinterpret_cast<const float *>(mvPtPredictAfterPatchMatchedUn.data()),
                                     status.data(), reinterpret_cast<float *>(mvPtPredict.data()),
                                     reinterpret_cast<float *>(mvPtPredictUn.data()));
    if (n_predict < 0) throw std::runtime_error("pagk_post_filter failed");
    for (int i = 0; i < mN; i++) mvStatus[i] = status[i];
    mTimeCostOptFlowResultFilterOut = std::chrono::duration<float>(std::chrono::steady_clock::now() - t3).count();
    return n_predict;
}

// reference :344-426
int GyroAidedTracker::TrackFeatures()
{
    IntegrateGyroMeasurements();
    int n_predict = -1;
    if (mType == OPENCV_OPTICAL_FLOW_PYR_LK) {
        // cv::calcOpticalFlowPyrLK (:353-380) is the OpenCV baseline the paper compares against; it
        // is third-party code outside the hot path and is not provided here.
        return -1;
    } else if (mType == GYRO_PREDICT) {
        n_predict = GyroPredictFeatures();
    } else {
        switch (mType) {  // :384-414
            case IMAGE_ONLY_OPTICAL_FLOW_CONSIDER_ILLUMINATION:
                mbHasGyroPredictInitial = false, mbConsiderIllumination = true, mbConsiderAffineDeformation = true,
                mbRegularizationPenalty = false;
                break;
            case GYRO_PREDICT_WITH_OPTICAL_FLOW_REFINED:
                mbHasGyroPredictInitial = true, mbConsiderIllumination = false, mbConsiderAffineDeformation = false,
                mbRegularizationPenalty = false;
                break;
            case GYRO_PREDICT_WITH_OPTICAL_FLOW_REFINED_CONSIDER_ILLUMINATION:
                mbHasGyroPredictInitial = true, mbConsiderIllumination = true, mbConsiderAffineDeformation = false,
                mbRegularizationPenalty = false;
                break;
            case GYRO_PREDICT_WITH_OPTICAL_FLOW_REFINED_CONSIDER_ILLUMINATION_DEFORMATION:
                mbHasGyroPredictInitial = true, mbConsiderIllumination = true, mbConsiderAffineDeformation = true,
                mbRegularizationPenalty = false;
                break;
            case GYRO_PREDICT_WITH_OPTICAL_FLOW_REFINED_CONSIDER_ILLUMINATION_DEFORMATION_REGULAR:
                mbHasGyroPredictInitial = true, mbConsiderIllumination = true, mbConsiderAffineDeformation = true,
                mbRegularizationPenalty = true;
                break;
            default:
                return -1;  // :415-418 "Unsupport type"
        }
        n_predict = GyroPredictFeaturesAndOpticalFlowRefined();
    }
    return n_predict;
}

// ---- Step 2: geometry validation (reference :429-480) ----------------------------------------------
namespace {
GyroAidedTracker::ModelFitter &model_fitter()
{
    static GyroAidedTracker::ModelFitter f;
    return f;
}
}  // namespace

void GyroAidedTracker::SetModelFitter(ModelFitter fitter) { model_fitter() = std::move(fitter); }

int GyroAidedTracker::GeometryValidation()
{
    if (!model_fitter())
        throw std::runtime_error("GyroAidedTracker::GeometryValidation(): no model fitter installed "
                                 "(cv::findHomography / cv::findFundamentalMat are the application's)");
    std::vector<cv::Point2f> vPts1, vPts2;  // :433-440
    for (size_t i = 0, iend = mvKeysRefUn.size(); i < iend; i++)
        if (mvStatus[i]) {
            vPts1.push_back(mvKeysRefUn[i].pt);
            vPts2.push_back(mvPtPredictUn[i]);
        }
    double H21[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, H12[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, F21[9] = {0};
    if (vPts1.size() > 8 && !model_fitter()(vPts1, vPts2, H21, H12, F21))  // :445, :596, :699
        throw std::runtime_error("GyroAidedTracker::GeometryValidation(): model fit failed");
    return GeometryValidation(H21, H12, F21);
}

int GyroAidedTracker::GeometryValidation(const double *H21, const double *H12, const double *F21, float sigma)
{
    auto t0 = std::chrono::steady_clock::now();
    std::vector<float> p1(2 * (size_t)mN + 2), p2(2 * (size_t)mN + 2);
    std::vector<cv::uchar> status(mN ? mN : 1);
    for (int i = 0; i < mN; i++) {
        p1[2 * i] = mvKeysRefUn[i].pt.x, p1[2 * i + 1] = mvKeysRefUn[i].pt.y;
        p2[2 * i] = mvPtPredictUn[i].x, p2[2 * i + 1] = mvPtPredictUn[i].y;
        status[i] = mvStatus[i];
    }
    float score = 0;
    int cnt_inlier = pagk_geometry_validation(PatchMatch::Context(), H21, H12, F21, mN, p1.data(), p2.data(),
                                              status.data(), sigma, &score);
    if (cnt_inlier < 0)
        throw std::runtime_error(std::string("pagk_geometry_validation failed: ") + pagk_strerror(cnt_inlier));
    for (int i = 0; i < mN; i++) mvStatus[i] = status[i];  // :472-476  outliers marked
    mTrackScore = score;
    mTimeCostGeometryValidation = std::chrono::duration<float>(std::chrono::steady_clock::now() - t0).count();
    mTImeCostTotalFeatureTrack =
        mTimeCostGyroPredict + mTimeCostOptFlow + mTimeCostOptFlowResultFilterOut + mTimeCostGeometryValidation;  // :483
    return cnt_inlier;
}
