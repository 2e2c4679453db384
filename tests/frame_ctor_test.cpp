// The Frame-based constructor of GyroAidedTracker (reference include/gyro_aided_tracker.h:119-126) and
// SetBackToFrame (:130) through application-side stand-ins for Frame / CameraParams / IMU::Calib that carry exactly
// the fields the reference reads (src/gyro_aided_tracker.cpp:39-45, 97-111).  Pure host code: eType GYRO_PREDICT.
// Exit code 0 = the Frame-based tracker produced what the data-constructor tracker produced and SetBackToFrame
// copied it.  Built and run by tests/test_host_shell.py.
#include <cstdio>
#include <memory>
#include <vector>

#include "gyro_aided_tracker.h"

struct CameraParams {  // reference include/frame.h (camera block)
    cv::Mat mK, mDistCoef;
    int width = 0, height = 0;
};
struct Frame {  // the subset of reference include/frame.h the tracker touches
    double mTimeStamp = 0;
    cv::Mat mGray;
    std::vector<cv::KeyPoint> mvKeys, mvKeysUn;
    std::vector<IMU::Point> mvImuFromLastFrame;
    std::shared_ptr<CameraParams> mpCameraParams;
    // written by SetBackToFrame
    std::vector<cv::Point2f> mvPtGyroPredictUn, mvPtPredict, mvPtPredictUn;
    std::vector<cv::uchar> mvStatus;
    std::vector<float> mvNcc;
    std::vector<std::vector<cv::Point2f>> mvvFlowsPredictCorners;
    cv::Mat mRcl;
};
struct Calib {  // IMU::Calib, reference include/imu_types.h
    cv::Mat Tbc;
};

static bool same(const std::vector<cv::Point2f> &a, const std::vector<cv::Point2f> &b)
{
    if (a.size() != b.size()) return false;
    for (size_t i = 0; i < a.size(); i++)
        if (a[i].x != b[i].x || a[i].y != b[i].y) return false;
    return true;
}

int main()
{
    const int W = 320, H = 240, N = 120;
    std::vector<unsigned char> pix((size_t)W * H);
    for (size_t i = 0; i < pix.size(); i++) pix[i] = (unsigned char)((i * 2654435761u) >> 24);
    auto cam = std::make_shared<CameraParams>();
    cam->mK = cv::Mat::eye(3, 3, cv::CV_32F);
    cam->mK.at<float>(0, 0) = 300.f, cam->mK.at<float>(1, 1) = 305.f, cam->mK.at<float>(0, 2) = 160.f, cam->mK.at<float>(1, 2) = 118.f;
    cam->mDistCoef = cv::Mat(1, 4, cv::CV_32F);
    cam->mDistCoef.at<float>(0) = -0.05f, cam->mDistCoef.at<float>(1) = 0.01f, cam->mDistCoef.at<float>(2) = 0.001f,
    cam->mDistCoef.at<float>(3) = -0.0005f;
    cam->width = W, cam->height = H;
    Frame ref, cur;
    ref.mTimeStamp = 1.0, cur.mTimeStamp = 1.05;
    ref.mGray = cv::Mat(H, W, cv::CV_8UC1, pix.data());
    cur.mGray = cv::Mat(H, W, cv::CV_8UC1, pix.data());
    ref.mpCameraParams = cur.mpCameraParams = cam;
    for (int i = 0; i < N; i++) {
        const float x = 10.f + (float)((i * 37) % 300), y = 8.f + (float)((i * 53) % 224);
        ref.mvKeys.push_back(cv::KeyPoint(x + 0.25f, y - 0.5f));   // "distorted"
        ref.mvKeysUn.push_back(cv::KeyPoint(x, y));                // undistorted: the ones the prediction must use
    }
    cur.mvKeys = ref.mvKeys, cur.mvKeysUn = ref.mvKeysUn;
    // 200 Hz gyro burst over the 50 ms between the frames
    for (int k = 0; k <= 10; k++) cur.mvImuFromLastFrame.push_back(IMU::Point(0, 0, 9.8f, 0.3f, -0.5f, 1.1f, 1.0 + 0.005 * k));
    Calib calib;
    calib.Tbc = cv::Mat::eye(4, 4, cv::CV_32F);
    calib.Tbc.at<float>(0, 3) = 0.01f;  // a translation the 3x3 rotation block must not pick up
    cv::Mat table;
    const cv::Point3f bias(0.001f, -0.002f, 0.0005f);

    GyroAidedTracker a(ref, cur, calib, bias, table, GyroAidedTracker::GYRO_PREDICT, GyroAidedTracker::PIXEL_AWARE_PREDICTION, "", 5);
    if (a.mTimeStamp != 1.05 || a.mTimeStampRef != 1.0 || a.mWidth != W || a.mHeight != H || a.mN != N) return 10;
    if (&a.mvKeysRefUn != &ref.mvKeysUn || &a.mvKeysRef != &ref.mvKeys || &a.mvKeysCurUn != &cur.mvKeysUn) return 11;  // :39-40
    if (a.mRbc.rows != 3 || a.mRbc.cols != 3 || a.mRbc.at<float>(0, 0) != 1.f || a.mRbc.at<float>(0, 2) != 0.f) return 12;
    const int na = a.TrackFeatures();

    // the data constructor binds mvKeysRefUn to vKeysRef_ (reference :21), so hand it the undistorted keypoints there
    GyroAidedTracker b(1.05, 1.0, ref.mGray, cur.mGray, ref.mvKeysUn, cur.mvKeys, ref.mvKeysUn, cur.mvKeysUn,
                       cur.mvImuFromLastFrame, bias, cam->mK, cam->mDistCoef, table, GyroAidedTracker::GYRO_PREDICT,
                       GyroAidedTracker::PIXEL_AWARE_PREDICTION, "", 5);
    const int nb = b.TrackFeatures();
    if (na != nb || na <= 0 || na >= N + 1) return 20;
    if (!same(a.mvPtPredictUn, b.mvPtPredictUn) || !same(a.mvPtPredict, b.mvPtPredict) || a.mvStatus != b.mvStatus) return 21;

    a.SetBackToFrame(cur);
    if (!same(cur.mvPtPredictUn, a.mvPtPredictUn) || !same(cur.mvPtPredict, a.mvPtPredict) ||
        !same(cur.mvPtGyroPredictUn, a.mvPtGyroPredictUn) || cur.mvStatus != a.mvStatus)
        return 30;
    if (cur.mvvFlowsPredictCorners.size() != (size_t)N || cur.mRcl.rows != 3) return 31;
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++)
            if (cur.mRcl.at<float>(r, c) != a.mRcl.at<float>(r, c)) return 32;
    cur.mRcl.at<float>(0, 0) = 7.f;  // a clone, not a view
    if (a.mRcl.at<float>(0, 0) == 7.f) return 33;
    std::printf("frame ctor ok: %d of %d predicted\n", na, N);
    return 0;
}
