// gyro_aided_tracker.h -- the hot-path side of the reference's GyroAidedTracker
// (include/gyro_aided_tracker.h:47-260): data constructor, TrackFeatures() type dispatch,
// GyroPredictFeatures(), GyroPredictFeaturesAndOpticalFlowRefined() and the public result vectors
// PatchMatch reads and writes, and GeometryValidation() around externally fitted models (the RANSAC fits
// are cv::findHomography / cv::findFundamentalMat: third-party, supplied by the application).  The unused
// matchers, display and logging are out of scope and are not declared.
#pragma once
#include <functional>
#include <string>
#include <vector>

#include "cvlite.h"

namespace IMU {
// reference include/imu_types.h:93-107
class Point {
public:
    Point() {}
    Point(const float &acc_x, const float &acc_y, const float &acc_z, const float &ang_vel_x,
          const float &ang_vel_y, const float &ang_vel_z, const double &timestamp)
        : a(acc_x, acc_y, acc_z), w(ang_vel_x, ang_vel_y, ang_vel_z), t(timestamp) {}
    cv::Point3f a, w;
    double t = 0;
};
}  // namespace IMU

class GyroAidedTracker {
public:
    enum eType {  // reference include/gyro_aided_tracker.h:55-63
        OPENCV_OPTICAL_FLOW_PYR_LK = 0,
        GYRO_PREDICT = 1,
        GYRO_PREDICT_WITH_OPTICAL_FLOW_REFINED = 2,
        GYRO_PREDICT_WITH_OPTICAL_FLOW_REFINED_CONSIDER_ILLUMINATION = 3,
        GYRO_PREDICT_WITH_OPTICAL_FLOW_REFINED_CONSIDER_ILLUMINATION_DEFORMATION = 4,
        GYRO_PREDICT_WITH_OPTICAL_FLOW_REFINED_CONSIDER_ILLUMINATION_DEFORMATION_REGULAR = 6,
        IMAGE_ONLY_OPTICAL_FLOW_CONSIDER_ILLUMINATION = 5
    };
    enum ePredictMethod { PIXEL_AWARE_PREDICTION = 1, SINGLE_HOMOGRAPHY = 2 };

    // reference include/gyro_aided_tracker.h:109-117
    GyroAidedTracker(double t, double t_ref, const cv::Mat &imgGrayRef_, const cv::Mat &imgGrayCur_,
                     const std::vector<cv::KeyPoint> &vKeysRef_, const std::vector<cv::KeyPoint> &vKeysCur_,
                     const std::vector<cv::KeyPoint> &vKeysUnRef_, const std::vector<cv::KeyPoint> &vKeysUnCur_,
                     const std::vector<IMU::Point> &vImuFromLastFrame, const cv::Point3f &bias_, cv::Mat K_,
                     cv::Mat DistCoef_, const cv::Mat &normalizeTable_,
                     eType type_ = GYRO_PREDICT_WITH_OPTICAL_FLOW_REFINED_CONSIDER_ILLUMINATION_DEFORMATION,
                     ePredictMethod predictMethod_ = PIXEL_AWARE_PREDICTION, std::string saveFolderPath = "",
                     int halfPatchSize_ = 5);

    // reference include/gyro_aided_tracker.h:119-126, src/gyro_aided_tracker.cpp:30-49: the constructor BOTH reference
    // apps use (Examples/Demo/RealSenseD435i.cpp:244-254, Examples/ROS/.../feature_tracker.cpp).  Frame and IMU::Calib
    // stay the host application's own types (out of scope), so the constructor is a template
    // over them; any types with the fields the reference's initialiser list reads bind here:
    //   FrameT: mTimeStamp, mGray, mvKeys, mvKeysUn, mvImuFromLastFrame, mpCameraParams->{mK, mDistCoef, width, height}
    //   CalibT: Tbc (4x4 CV_32F; mRbc = its top-left 3x3, Tbc.colRange(0,3).rowRange(0,3) in the reference)
    template <class FrameT, class CalibT>
    GyroAidedTracker(const FrameT &pFrameRef, const FrameT &pFrameCur, const CalibT &imuCalib, const cv::Point3f &biasg_,
                     const cv::Mat &normalizeTable_,
                     eType type_ = GYRO_PREDICT_WITH_OPTICAL_FLOW_REFINED_CONSIDER_ILLUMINATION_DEFORMATION,
                     ePredictMethod predictMethod_ = PIXEL_AWARE_PREDICTION, std::string saveFolderPath = "",
                     int halfPatchSize_ = 5)
        : mTimeStamp(pFrameCur.mTimeStamp), mTimeStampRef(pFrameRef.mTimeStamp), mImgGrayRef(pFrameRef.mGray),
          mImgGrayCur(pFrameCur.mGray), mvKeysRef(pFrameRef.mvKeys), mvKeysRefUn(pFrameRef.mvKeysUn),
          mvKeysCur(pFrameCur.mvKeys), mvKeysCurUn(pFrameCur.mvKeysUn), mvImuFromLastFrame(pFrameCur.mvImuFromLastFrame),
          mHalfPatchSize(halfPatchSize_), mRbc(TopLeft3x3(imuCalib.Tbc)), mBias(biasg_), mK(pFrameCur.mpCameraParams->mK),
          mDistCoef(pFrameCur.mpCameraParams->mDistCoef), mWidth(pFrameCur.mpCameraParams->width),
          mHeight(pFrameCur.mpCameraParams->height), mNormalizeTable(normalizeTable_), mType(type_),
          mPredictMethod(predictMethod_)
    {
        (void)saveFolderPath;  // result logging is out of scope
        Initialize();
    }

    // reference include/gyro_aided_tracker.h:130, src/gyro_aided_tracker.cpp:97-111: the tracker's results handed back to
    // the application's Frame (fields of include/frame.h that the demo loop reads, RealSenseD435i.cpp:255-300)
    template <class FrameT>
    void SetBackToFrame(FrameT &pFrame)
    {
        pFrame.mvPtGyroPredictUn = std::vector<cv::Point2f>(mvPtGyroPredictUn.begin(), mvPtGyroPredictUn.end());
        pFrame.mvPtPredict = std::vector<cv::Point2f>(mvPtPredict.begin(), mvPtPredict.end());
        pFrame.mvPtPredictUn = std::vector<cv::Point2f>(mvPtPredictUn.begin(), mvPtPredictUn.end());
        pFrame.mvStatus = std::vector<cv::uchar>(mvStatus.begin(), mvStatus.end());
        pFrame.mvNcc = std::vector<float>(mvNccAfterPatchMatched.begin(), mvNccAfterPatchMatched.end());
        pFrame.mvvFlowsPredictCorners.resize(mvvFlowsPredictCorners.size());
        for (size_t i = 0, iend = mvvFlowsPredictCorners.size(); i < iend; i++)
            pFrame.mvvFlowsPredictCorners[i] =
                std::vector<cv::Point2f>(mvvFlowsPredictCorners[i].begin(), mvvFlowsPredictCorners[i].end());
        pFrame.mRcl = mRcl.clone();
    }

    static cv::Mat TopLeft3x3(const cv::Mat &T)
    {
        cv::Mat R(3, 3, cv::CV_32F);
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) R.at<float>(r, c) = T.at<float>(r, c);
        return R;
    }

    void Initialize();
    void SetRegularizationPenalty(bool flag) { mbRegularizationPenalty = flag; }
    int TrackFeatures();
    void SetRcl(const cv::Mat Rcl_);
    cv::Mat GetRcl() { return mRcl.clone(); }
    void SetType(eType type_) { mType = type_; }
    void SetRbc(const cv::Mat &Rbc) { mRbc = Rbc.clone(); }  // ctor #1 of the reference leaves mRbc unset
    // PatchMatch parameters the reference hard-codes at the call site (src/gyro_aided_tracker.cpp:276-277)
    void SetPatchMatchParams(int iterations, int pyramids) { mIterations = iterations, mPyramids = pyramids; }

    int GyroPredictFeatures();
    int GyroPredictFeaturesAndOpticalFlowRefined();

    // Step 2 of the tracker, reference include/gyro_aided_tracker.h:134 / src/gyro_aided_tracker.cpp:429-480.
    // The reference fits H21 and F21 inside CheckHomography / CheckFundamental with OpenCV's RANSAC
    // (:596, :699) and inverts H21 with cv::Mat::inv (:597); here the application supplies those three
    // 3x3 row-major double matrices -- through a fitter installed once, or per call -- and the scoring
    // loops, the model choice and the outlier marking run behind pagk_geometry_validation.
    using ModelFitter = std::function<bool(const std::vector<cv::Point2f> &vPts1, const std::vector<cv::Point2f> &vPts2,
                                           double H21[9], double H12[9], double F21[9])>;
    static void SetModelFitter(ModelFitter fitter);
    int GeometryValidation();  // reference signature; needs a fitter (throws std::runtime_error without one)
    int GeometryValidation(const double *H21, const double *H12, const double *F21, float sigma = 1.0f);
    float mTrackScore = 0;  // `track_score` of the reference's log line (:447, :465-470)
    void IntegrateGyroMeasurements();
    cv::Mat IntegrateOneGyroMeasurement(cv::Point3f &gyro, double dt);
    void GyroPredictOnePixel(cv::Point2f &pt_ref, cv::Point2f &pt_predict, cv::Point2f &pt_predict_distort,
                             cv::Point2f &flow);

public:  // data members keep the reference's names (include/gyro_aided_tracker.h:173-259)
    double mTimeStamp, mTimeStampRef;
    const cv::Mat &mImgGrayRef;
    const cv::Mat &mImgGrayCur;
    const std::vector<cv::KeyPoint> &mvKeysRef, &mvKeysRefUn, &mvKeysCur, &mvKeysCurUn;
    const std::vector<IMU::Point> &mvImuFromLastFrame;

    std::vector<cv::Point2f> mvPtPredict, mvPtPredictUn, mvPtGyroPredict, mvPtGyroPredictUn;
    std::vector<std::vector<cv::Point2f>> mvvPtPredictCorners, mvvPtPredictCornersUn, mvvFlowsPredictCorners;
    std::vector<cv::uchar> mvStatus;
    std::vector<float> mvError;

    int mHalfPatchSize;
    std::vector<cv::Point2f> mvPatchCorners;
    std::vector<cv::Mat> mvAffineDeformationMatrix;

    std::vector<cv::Point2f> mvPtPredictAfterPatchMatched, mvPtPredictAfterPatchMatchedUn;
    std::vector<cv::uchar> mvStatusAfterPatchMatched;
    std::vector<double> mvPixelErrorsOfPatchMatched, mvDistanceBetweenPredictedAndPatchMatched;
    std::vector<float> mvNccAfterPatchMatched;
    std::vector<cv::Point2f> mvFlowsPredictUn;

    float mTimeCostGyroPredict = 0, mTimeCostOptFlow = 0, mTimeCostOptFlowResultFilterOut = 0,
          mTimeCostGeometryValidation = 0, mTImeCostTotalFeatureTrack = 0;

    cv::Mat mRbc, mRcl;
    float mr11, mr12, mr13, mr21, mr22, mr23, mr31, mr32, mr33;
    cv::Point3f mBias;
    cv::Mat mK, mKRKinv;
    float mfx, mfy, mcx, mcy, mfx_inv, mfy_inv;
    cv::Mat mDistCoef;
    float mk1, mk2, mp1, mp2, mk3;
    int mWidth, mHeight, mN;
    const cv::Mat &mNormalizeTable;

    bool mbHasGyroPredictInitial = true, mbConsiderIllumination = true, mbConsiderAffineDeformation = false,
         mbRegularizationPenalty = false;
    eType mType;
    ePredictMethod mPredictMethod;
    int mIterations = 10, mPyramids = 3;
};
