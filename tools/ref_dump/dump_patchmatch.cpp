// dump_patchmatch.cpp -- the pinning kit's reference side.  NOT built in this repository (the reference needs
// OpenCV >= 3.4, Eigen3 and glog, none of which exist in the image this repository is developed in): a maintainer
// with a working checkout of weibohuang0314/pixel_aware_gyro_aided_klt_feature_tracker builds it AGAINST THE REAL
// REFERENCE (CMake snippet: INTEGRATION.md section 8) and runs it over the committed golden inputs:
//
//     python tools/ref_dump/export_cases.py  out/                 # tests/golden/*.npz -> out/<case>.in
//     ./dump_patchmatch out/<case>.in out/<case>.ref   (for each case)
//     python tools/ref_dump/import_ref.py out/ tests/golden/ref/  # -> tests/golden/ref/<case>.npz
//     python -m pytest tests/test_reference_pin.py                # oracle and HIP against the REFERENCE's outputs
//
// What it runs is the path itself: PatchMatch::OpticalFlowMultiLevel() (reference src/patch_match.cpp:79-142) on a
// GyroAidedTracker (ctor #1, include/gyro_aided_tracker.h:109-117) whose public prediction state -- mvPtPredictUn,
// mvStatus, mvAffineDeformationMatrix -- is filled from the case file, exactly what GyroPredictFeatures() leaves
// there (src/gyro_aided_tracker.cpp:118-185) before :276-283 constructs PatchMatch.  It also replays CreatePyramids'
// cv::resize chain (:61-76) and dumps every level, which pins the one OpenCV routine on the path.
//
// Case file (<case>.in, little endian, written by export_cases.py):
//   char magic[8] = "PAGKIN1\0"; int32 W, H, N, half_patch, iterations, pyramids;
//   int32 has_gyro, illumination, affine, penalty, ncc; float cam[8] = fx fy cx cy k1 k2 p1 p2;
//   uint8 ref[H*W]; uint8 cur[H*W]; float pt_ref[N*2]; float pt_init[N*2]; float affine[N*4]; uint8 status_in[N];
// Result file (<case>.ref):
//   char magic[8] = "PAGKREF1"; int32 N, pyramids;
//   float pt_un[N*2]; float pt_dist[N*2]; uint8 status[N]; double pix_err[N]; double dist_pred[N]; float ncc[N];
//   for level 1 .. pyramids-1: int32 w, h; uint8 ref_level[h*w]; uint8 cur_level[h*w];
//   then a text tail: the Eigen and OpenCV versions this binary was built with.
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>

#include <Eigen/Core>
#include <opencv2/core/core.hpp>
#include <opencv2/core/version.hpp>
#include <opencv2/imgproc/imgproc.hpp>

#include "gyro_aided_tracker.h"   // the reference's include/
#include "patch_match.h"

template <class T>
static bool rd(FILE *f, T *p, size_t n) { return fread(p, sizeof(T), n, f) == n; }
template <class T>
static void wr(FILE *f, const T *p, size_t n) { fwrite(p, sizeof(T), n, f); }

int main(int argc, char **argv)
{
    if (argc != 3) {
        fprintf(stderr, "usage: %s <case.in> <case.ref>\n", argv[0]);
        return 2;
    }
    FILE *fi = fopen(argv[1], "rb");
    char magic[8];
    int32_t hd[11];
    float cam[8];
    if (!fi || !rd(fi, magic, 8) || memcmp(magic, "PAGKIN1", 8) != 0 || !rd(fi, hd, 11) || !rd(fi, cam, 8)) return 3;
    const int W = hd[0], H = hd[1], N = hd[2], half = hd[3], iters = hd[4], L = hd[5];
    const bool has_gyro = hd[6], illum = hd[7], affine_on = hd[8], penalty = hd[9], ncc_on = hd[10];
    cv::Mat ref(H, W, CV_8UC1), cur(H, W, CV_8UC1);   // continuous, step == W: the layout the golden cases assume
    std::vector<float> pt_ref((size_t)N * 2), pt_init((size_t)N * 2), aff((size_t)N * 4);
    std::vector<uint8_t> status_in((size_t)N);
    if (!rd(fi, ref.data, (size_t)W * H) || !rd(fi, cur.data, (size_t)W * H) || !rd(fi, pt_ref.data(), pt_ref.size()) ||
        !rd(fi, pt_init.data(), pt_init.size()) || !rd(fi, aff.data(), aff.size()) || !rd(fi, status_in.data(), status_in.size()))
        return 4;
    fclose(fi);

    std::vector<cv::KeyPoint> keys_ref, keys_cur;
    for (int i = 0; i < N; i++) keys_ref.push_back(cv::KeyPoint(pt_ref[2 * i], pt_ref[2 * i + 1], 1.f));
    std::vector<IMU::Point> imu;   // GyroPredictFeatures() is not run: the prediction comes from the case file
    const cv::Point3f bias(0.f, 0.f, 0.f);
    cv::Mat K = cv::Mat::eye(3, 3, CV_32F);
    K.at<float>(0, 0) = cam[0], K.at<float>(1, 1) = cam[1], K.at<float>(0, 2) = cam[2], K.at<float>(1, 2) = cam[3];
    cv::Mat dist(4, 1, CV_32F);
    for (int k = 0; k < 4; k++) dist.at<float>(k) = cam[4 + k];
    cv::Mat table;
    // ctor #1 binds mvKeysRefUn to its vKeysRef_ argument (src/gyro_aided_tracker.cpp:21): pass the undistorted points there
    GyroAidedTracker tracker(0.05, 0.0, ref, cur, keys_ref, keys_cur, keys_ref, keys_cur, imu, bias, K, dist, table,
                             GyroAidedTracker::GYRO_PREDICT_WITH_OPTICAL_FLOW_REFINED_CONSIDER_ILLUMINATION_DEFORMATION,
                             GyroAidedTracker::PIXEL_AWARE_PREDICTION, "", half);
    // the state GyroPredictFeatures() leaves behind (:118-185)
    tracker.mvPtPredictUn.resize(N);
    tracker.mvPtPredict.resize(N);
    tracker.mvStatus.resize(N);
    tracker.mvAffineDeformationMatrix.resize(N);
    for (int i = 0; i < N; i++) {
        tracker.mvPtPredictUn[i] = cv::Point2f(pt_init[2 * i], pt_init[2 * i + 1]);
        tracker.mvPtPredict[i] = tracker.mvPtPredictUn[i];
        tracker.mvStatus[i] = status_in[i];
        cv::Mat A(2, 2, CV_32F);
        A.at<float>(0, 0) = aff[4 * i], A.at<float>(0, 1) = aff[4 * i + 1], A.at<float>(1, 0) = aff[4 * i + 2], A.at<float>(1, 1) = aff[4 * i + 3];
        tracker.mvAffineDeformationMatrix[i] = A;
    }

    // src/gyro_aided_tracker.cpp:276-283, with the case's flags
    PatchMatch pm(&tracker, half, iters, L, has_gyro, /*bInverse_=*/false, illum, affine_on, penalty, ncc_on);
    pm.OpticalFlowMultiLevel();

    FILE *fo = fopen(argv[2], "wb");
    if (!fo) return 5;
    const int32_t out_hd[2] = {N, L};
    wr(fo, "PAGKREF1", 8);
    wr(fo, out_hd, 2);
    for (int i = 0; i < N; i++) wr(fo, &tracker.mvPtPredictAfterPatchMatchedUn[i].x, 1), wr(fo, &tracker.mvPtPredictAfterPatchMatchedUn[i].y, 1);
    for (int i = 0; i < N; i++) wr(fo, &tracker.mvPtPredictAfterPatchMatched[i].x, 1), wr(fo, &tracker.mvPtPredictAfterPatchMatched[i].y, 1);
    wr(fo, tracker.mvStatusAfterPatchMatched.data(), (size_t)N);
    wr(fo, tracker.mvPixelErrorsOfPatchMatched.data(), (size_t)N);
    wr(fo, tracker.mvDistanceBetweenPredictedAndPatchMatched.data(), (size_t)N);
    wr(fo, tracker.mvNccAfterPatchMatched.data(), (size_t)N);
    // CreatePyramids (:61-76), replayed
    cv::Mat p1 = ref, p2 = cur;
    for (int l = 1; l < L; l++) {
        cv::Mat q1, q2;
        cv::resize(p1, q1, cv::Size(p1.cols * 0.5, p1.rows * 0.5));
        cv::resize(p2, q2, cv::Size(p2.cols * 0.5, p2.rows * 0.5));
        const int32_t wh[2] = {q1.cols, q1.rows};
        wr(fo, wh, 2);
        for (int r = 0; r < q1.rows; r++) wr(fo, q1.ptr<uint8_t>(r), (size_t)q1.cols);
        for (int r = 0; r < q2.rows; r++) wr(fo, q2.ptr<uint8_t>(r), (size_t)q2.cols);
        p1 = q1, p2 = q2;
    }
    fprintf(fo, "\nEigen %d.%d.%d; OpenCV %s; EIGEN_VECTORIZE %s\n", EIGEN_WORLD_VERSION, EIGEN_MAJOR_VERSION, EIGEN_MINOR_VERSION,
            CV_VERSION,
#ifdef EIGEN_VECTORIZE
            "on"
#else
            "off"
#endif
    );
    fclose(fo);
    return 0;
}
