// Every public method of PatchMatch (reference include/patch_match.h:51-69) through the API shell on the GPU:
//   (A) OpticalFlowMultiLevel()                                               -- the reference's one call site
//   (B) CreatePyramids(); per level, top first: ..._onePixel(i, ...) for every i; DistortPoints(); SetMatcher()
//                                                                             -- what that call does inside (:79-142)
// (B) must leave the tracker's six result vectors bit-identical to (A); GetPixelValue and NCC are the host-side
// member functions and are checked against the library's own NCC (bCalculateNCC_ = true) and against bytes.
//   usage: patch_match_methods_gpu_test <in.bin>
//   in : int32 W H N | u8 ref[W*H] | u8 cur[W*H] | f32 keys[N*2] | f32 fx fy cx cy | f32 dist[4] | f32 gyro[3] | f32 dt
#include <cstdio>
#include <cstring>
#include <vector>

#include "gyro_aided_tracker.h"
#include "patch_match.h"

template <class T>
static bool rd(FILE *f, T *p, size_t n) { return fread(p, sizeof(T), n, f) == n; }
static bool same(const std::vector<cv::Point2f> &a, const std::vector<cv::Point2f> &b)
{
    if (a.size() != b.size()) return false;
    for (size_t i = 0; i < a.size(); i++)
        if (memcmp(&a[i].x, &b[i].x, 4) || memcmp(&a[i].y, &b[i].y, 4)) return false;
    return true;
}

int main(int argc, char **argv)
{
    if (argc != 2) return 2;
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
    cv::Mat K = cv::Mat::eye(3, 3, cv::CV_32F), D(1, 4, cv::CV_32F);
    K.at<float>(0, 0) = kk[0], K.at<float>(1, 1) = kk[1], K.at<float>(0, 2) = kk[2], K.at<float>(1, 2) = kk[3];
    for (int k = 0; k < 4; k++) D.at<float>(k) = dc[k];
    cv::Mat ref(H, W, cv::CV_8UC1, ref_px.data()), cur(H, W, cv::CV_8UC1, cur_px.data());
    std::vector<cv::KeyPoint> kp;
    for (int i = 0; i < N; i++) kp.push_back(cv::KeyPoint(keys[2 * i], keys[2 * i + 1]));
    std::vector<IMU::Point> imu;
    for (int k = 0; k <= 10; k++) imu.push_back(IMU::Point(0, 0, 9.8f, gyro[0], gyro[1], gyro[2], 1.0 + dt * k / 10.0));
    cv::Mat table;
    const cv::Point3f bias(0.f, 0.f, 0.f);
    const int h = 5, iterations = 10, pyramids = 3;

    // a tracker that has run the gyro prediction: the state PatchMatch reads (mvStatus, mvPtPredictUn, the affine matrices)
    GyroAidedTracker T(1.0 + dt, 1.0, ref, cur, kp, kp, kp, kp, imu, bias, K, D, table, GyroAidedTracker::GYRO_PREDICT,
                       GyroAidedTracker::PIXEL_AWARE_PREDICTION, "", h);
    T.TrackFeatures();

    for (int ncc = 0; ncc < 2; ncc++) {
        // (A) the one call
        PatchMatch a(&T, h, iterations, pyramids, true, false, true, true, false, ncc != 0);
        a.OpticalFlowMultiLevel();
        const auto ptA = T.mvPtPredictAfterPatchMatched, ptUnA = T.mvPtPredictAfterPatchMatchedUn;
        const auto stA = T.mvStatusAfterPatchMatched;
        const auto errA = T.mvPixelErrorsOfPatchMatched, distA = T.mvDistanceBetweenPredictedAndPatchMatched;
        const auto nccA = T.mvNccAfterPatchMatched;
        // (B) its parts, one public method at a time
        PatchMatch b(&T, h, iterations, pyramids, true, false, true, true, false, ncc != 0);
        b.CreatePyramids();
        for (int level = pyramids - 1; level >= 0; level--) {
            b.SetLevel(level);
            for (int i = 0; i < N; i++) b.OpticalFlowConsideringIlluminationChange_onePixel(i, true, true, false);
        }
        b.DistortPoints();
        b.SetMatcher();
        if (!same(ptUnA, T.mvPtPredictAfterPatchMatchedUn)) return 20 + ncc;
        if (!same(ptA, T.mvPtPredictAfterPatchMatched)) return 22 + ncc;
        if (stA != T.mvStatusAfterPatchMatched) return 24 + ncc;
        if (memcmp(errA.data(), T.mvPixelErrorsOfPatchMatched.data(), (size_t)N * 8)) return 26 + ncc;
        if (memcmp(distA.data(), T.mvDistanceBetweenPredictedAndPatchMatched.data(), (size_t)N * 8)) return 28 + ncc;
        // the score: the library computes PatchMatch::NCC on the device (bCalculateNCC_), (B) with the host member function
        if (memcmp(nccA.data(), T.mvNccAfterPatchMatched.data(), (size_t)N * 4)) return 30 + ncc;
        int tracked = 0;
        for (int i = 0; i < N; i++) tracked += stA[i] != 0;
        std::printf("bCalculateNCC %d: %d of %d tracked; (A) == (B) in all six vectors\n", ncc, tracked, N);
    }
    // GetPixelValue on bytes: integer coordinates return the pixel, the last column / row pair with 0 past the buffer
    PatchMatch g(&T, h, iterations, pyramids, true, false, true, true, false, false);
    if (g.GetPixelValue(ref, 10.0f, 20.0f) != (float)ref_px[20 * (size_t)W + 10]) return 40;
    if (g.GetPixelValue(ref, 10.5f, 20.0f) != 0.5f * ref_px[20 * (size_t)W + 10] + 0.5f * ref_px[20 * (size_t)W + 11]) return 41;
    if (g.GetPixelValue(ref, -3.0f, -7.0f) != (float)ref_px[0]) return 42;                                  // clamps (:394-395)
    if (g.GetPixelValue(ref, (float)W + 5, (float)H + 5) != (float)ref_px[(size_t)W * H - 1]) return 43;      // (:396-397)
    // NCC of a patch with itself is 1 (up to the 1e-10 in the denominator)
    const float self = g.NCC(h, ref, ref, cv::Point2f(100.25f, 80.5f), cv::Point2f(100.25f, 80.5f), cv::Mat());
    if (!(self > 0.9999f && self <= 1.0f)) return 44;
    PatchMatch::ReleaseContext();
    std::printf("every public PatchMatch method ok\n");
    return 0;
}
