// The constructor BOTH reference apps use (Examples/Demo/RealSenseD435i.cpp:244-254; reference
// include/gyro_aided_tracker.h:119-126, src/gyro_aided_tracker.cpp:30-49) driving the hot path on the GPU:
//   Frame-based ctor -> TrackFeatures() [GYRO_PREDICT_WITH_OPTICAL_FLOW_REFINED_CONSIDER_ILLUMINATION_DEFORMATION]
//   -> SetBackToFrame(), against the data constructor (#1) in the same process, and -- through the file this writes --
// against the oracle chain in tests/test_host_shell.py.  Frame / CameraParams / IMU::Calib are application-side
// stand-ins with exactly the fields the reference reads.
//   usage: frame_ctor_gpu_test <in.bin> <out.bin>
//   in : int32 W H N | u8 ref[W*H] | u8 cur[W*H] | f32 keys[N*2] | f32 fx fy cx cy | f32 dist[4] | f32 gyro[3] | f32 dt
//   out: int32 N | prediction (u8 status[N], f32 pt_un[N*2], f32 affine[N*4]) | refined (u8 status_pm[N], f32 pt_pm_un[N*2],
//        f64 pix_err[N], f64 dist_pred[N], u8 status[N], f32 pt_predict_un[N*2], int32 survivors)
#include <cstdio>
#include <cstring>
#include <memory>
#include <vector>

#include "gyro_aided_tracker.h"

struct CameraParams {
    cv::Mat mK, mDistCoef;
    int width = 0, height = 0;
};
struct Frame {
    double mTimeStamp = 0;
    cv::Mat mGray;
    std::vector<cv::KeyPoint> mvKeys, mvKeysUn;
    std::vector<IMU::Point> mvImuFromLastFrame;
    std::shared_ptr<CameraParams> mpCameraParams;
    std::vector<cv::Point2f> mvPtGyroPredictUn, mvPtPredict, mvPtPredictUn;
    std::vector<cv::uchar> mvStatus;
    std::vector<float> mvNcc;
    std::vector<std::vector<cv::Point2f>> mvvFlowsPredictCorners;
    cv::Mat mRcl;
};
struct Calib {
    cv::Mat Tbc;
};

template <class T>
static bool rd(FILE *f, T *p, size_t n) { return fread(p, sizeof(T), n, f) == n; }
template <class T>
static void wr(FILE *f, const T *p, size_t n) { fwrite(p, sizeof(T), n, f); }
static bool same(const std::vector<cv::Point2f> &a, const std::vector<cv::Point2f> &b)
{
    if (a.size() != b.size()) return false;
    for (size_t i = 0; i < a.size(); i++)
        if (memcmp(&a[i].x, &b[i].x, 4) || memcmp(&a[i].y, &b[i].y, 4)) return false;
    return true;
}

int main(int argc, char **argv)
{
    if (argc != 3) return 2;
    FILE *fi = fopen(argv[1], "rb");
    if (!fi) return 3;
    int hdr[3];
    if (!rd(fi, hdr, 3)) return 4;
    const int W = hdr[0], H = hdr[1], N = hdr[2];
    std::vector<unsigned char> ref_px((size_t)W * H), cur_px((size_t)W * H);
    std::vector<float> keys((size_t)N * 2);
    float kk[4], dc[4], gyro[3], dt;
    if (!rd(fi, ref_px.data(), ref_px.size()) || !rd(fi, cur_px.data(), cur_px.size()) || !rd(fi, keys.data(), keys.size()) ||
        !rd(fi, kk, 4) || !rd(fi, dc, 4) || !rd(fi, gyro, 3) || !rd(fi, &dt, 1))
        return 5;
    fclose(fi);

    auto cam = std::make_shared<CameraParams>();
    cam->mK = cv::Mat::eye(3, 3, cv::CV_32F);
    cam->mK.at<float>(0, 0) = kk[0], cam->mK.at<float>(1, 1) = kk[1], cam->mK.at<float>(0, 2) = kk[2], cam->mK.at<float>(1, 2) = kk[3];
    cam->mDistCoef = cv::Mat(1, 4, cv::CV_32F);
    for (int k = 0; k < 4; k++) cam->mDistCoef.at<float>(k) = dc[k];
    cam->width = W, cam->height = H;
    Frame ref, cur;
    ref.mTimeStamp = 1.0, cur.mTimeStamp = 1.0 + dt;
    ref.mGray = cv::Mat(H, W, cv::CV_8UC1, ref_px.data());
    cur.mGray = cv::Mat(H, W, cv::CV_8UC1, cur_px.data());
    ref.mpCameraParams = cur.mpCameraParams = cam;
    for (int i = 0; i < N; i++) {
        ref.mvKeys.push_back(cv::KeyPoint(keys[2 * i], keys[2 * i + 1]));
        ref.mvKeysUn.push_back(cv::KeyPoint(keys[2 * i], keys[2 * i + 1]));
    }
    cur.mvKeys = ref.mvKeys, cur.mvKeysUn = ref.mvKeysUn;
    for (int k = 0; k <= 10; k++)  // the gyro burst between the two frames
        cur.mvImuFromLastFrame.push_back(IMU::Point(0, 0, 9.8f, gyro[0], gyro[1], gyro[2], 1.0 + dt * k / 10.0));
    Calib calib;
    calib.Tbc = cv::Mat::eye(4, 4, cv::CV_32F);
    cv::Mat table;
    const cv::Point3f bias(0.f, 0.f, 0.f);
    const auto REFINE = GyroAidedTracker::GYRO_PREDICT_WITH_OPTICAL_FLOW_REFINED_CONSIDER_ILLUMINATION_DEFORMATION;

    // the prediction alone (what the oracle chain is fed with)
    GyroAidedTracker pr(ref, cur, calib, bias, table, GyroAidedTracker::GYRO_PREDICT, GyroAidedTracker::PIXEL_AWARE_PREDICTION, "", 5);
    pr.TrackFeatures();

    // Frame-based constructor, the reference apps' call (RealSenseD435i.cpp:244-254): default type = the refined one
    GyroAidedTracker a(ref, cur, calib, bias, table, REFINE, GyroAidedTracker::PIXEL_AWARE_PREDICTION, "", 5);
    const int na = a.TrackFeatures();
    a.SetBackToFrame(cur);

    // the data constructor on the same inputs
    GyroAidedTracker b(1.0 + dt, 1.0, ref.mGray, cur.mGray, ref.mvKeysUn, cur.mvKeys, ref.mvKeysUn, cur.mvKeysUn,
                       cur.mvImuFromLastFrame, bias, cam->mK, cam->mDistCoef, table, REFINE,
                       GyroAidedTracker::PIXEL_AWARE_PREDICTION, "", 5);
    b.SetRbc(a.mRbc);
    const int nb = b.TrackFeatures();
    if (na != nb) return 20;
    if (!same(a.mvPtPredictUn, b.mvPtPredictUn) || !same(a.mvPtPredict, b.mvPtPredict) || a.mvStatus != b.mvStatus) return 21;
    if (!same(a.mvPtPredictAfterPatchMatchedUn, b.mvPtPredictAfterPatchMatchedUn) ||
        a.mvStatusAfterPatchMatched != b.mvStatusAfterPatchMatched ||
        memcmp(a.mvPixelErrorsOfPatchMatched.data(), b.mvPixelErrorsOfPatchMatched.data(), (size_t)N * 8))
        return 22;
    if (!same(cur.mvPtPredictUn, a.mvPtPredictUn) || !same(cur.mvPtPredict, a.mvPtPredict) ||
        !same(cur.mvPtGyroPredictUn, a.mvPtGyroPredictUn) || cur.mvStatus != a.mvStatus || cur.mRcl.rows != 3)
        return 30;

    FILE *fo = fopen(argv[2], "wb");
    if (!fo) return 6;
    wr(fo, &N, 1);
    wr(fo, pr.mvStatus.data(), (size_t)N);
    for (int i = 0; i < N; i++) wr(fo, &pr.mvPtPredictUn[i].x, 1), wr(fo, &pr.mvPtPredictUn[i].y, 1);
    for (int i = 0; i < N; i++) {
        const cv::Mat &A = pr.mvAffineDeformationMatrix[i];
        float av[4] = {1, 0, 0, 1};
        if (A.rows == 2 && A.cols == 2) av[0] = A.at<float>(0, 0), av[1] = A.at<float>(0, 1), av[2] = A.at<float>(1, 0), av[3] = A.at<float>(1, 1);
        wr(fo, av, 4);
    }
    wr(fo, a.mvStatusAfterPatchMatched.data(), (size_t)N);
    for (int i = 0; i < N; i++) wr(fo, &a.mvPtPredictAfterPatchMatchedUn[i].x, 1), wr(fo, &a.mvPtPredictAfterPatchMatchedUn[i].y, 1);
    wr(fo, a.mvPixelErrorsOfPatchMatched.data(), (size_t)N);
    wr(fo, a.mvDistanceBetweenPredictedAndPatchMatched.data(), (size_t)N);
    wr(fo, cur.mvStatus.data(), (size_t)N);   // through SetBackToFrame
    for (int i = 0; i < N; i++) wr(fo, &cur.mvPtPredictUn[i].x, 1), wr(fo, &cur.mvPtPredictUn[i].y, 1);
    wr(fo, &na, 1);
    fclose(fo);
    std::printf("frame ctor on the GPU ok: %d of %d survive\n", na, N);
    return 0;
}
