// track_sequence.cpp -- the reference's per-frame loop (Examples/Demo/RealSenseD435i.cpp:199-321) on the
// MI355X path: for every new frame build a GyroAidedTracker over (last frame, current frame), call
// TrackFeatures(), and carry the surviving points forward as the next reference keypoints.
//
// Frames come from stdin-free synthetic input: a binary file written by tests/test_host_shell.py
//   int32 n_frames, width, height, n_keypoints; float K[9], dist[4];
//   n_frames x (width*height) uint8 images; n_keypoints x 2 float keypoints of frame 0;
//   (n_frames-1) x 9 float Rcl (camera rotation last -> current, what the gyro integration yields)
// Output (stdout): per frame pair "pair k tracked m" and a final line "survivors s checksum c".
//
// Build: g++ -std=c++17 -I include -I <pkg>/csrc/host examples/track_sequence.cpp -L <pkg> -l:libpagk_tracker.so \
//        -l:libpagk_hip.so -Wl,-rpath,<pkg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "gyro_aided_tracker.h"
#include "patch_match.h"

int main(int argc, char **argv)
{
    if (argc < 2) {
        std::fprintf(stderr, "usage: %s sequence.bin [half_patch iterations pyramids]\n", argv[0]);
        return 2;
    }
    const int half = argc > 2 ? std::atoi(argv[2]) : 5, iters = argc > 3 ? std::atoi(argv[3]) : 10,
              pyr = argc > 4 ? std::atoi(argv[4]) : 3;
    FILE *f = std::fopen(argv[1], "rb");
    if (!f) return 3;
    int32_t hdr[4];
    float K[9], dist[4];
    if (std::fread(hdr, 4, 4, f) != 4 || std::fread(K, 4, 9, f) != 9 || std::fread(dist, 4, 4, f) != 4) return 4;
    const int nf = hdr[0], w = hdr[1], h = hdr[2], nk = hdr[3];
    std::vector<std::vector<unsigned char>> img(nf, std::vector<unsigned char>((size_t)w * h));
    for (auto &im : img)
        if (std::fread(im.data(), 1, im.size(), f) != im.size()) return 5;
    std::vector<float> kp((size_t)nk * 2), Rs((size_t)(nf - 1) * 9);
    if (std::fread(kp.data(), 4, kp.size(), f) != kp.size() || std::fread(Rs.data(), 4, Rs.size(), f) != Rs.size()) return 6;
    std::fclose(f);

    cv::Mat Km(3, 3, cv::CV_32F), Dm(1, 4, cv::CV_32F), table;
    for (int k = 0; k < 9; k++) Km.at<float>(k / 3, k % 3) = K[k];
    for (int k = 0; k < 4; k++) Dm.at<float>(k) = dist[k];

    std::vector<cv::KeyPoint> keysLast(nk), none;
    for (int i = 0; i < nk; i++) keysLast[i].pt = cv::Point2f(kp[2 * i], kp[2 * i + 1]);
    std::vector<IMU::Point> noImu;
    double checksum = 0;
    for (int k = 1; k < nf; k++) {
        cv::Mat last(h, w, cv::CV_8UC1, img[k - 1].data()), cur(h, w, cv::CV_8UC1, img[k].data());
        // RealSenseD435i.cpp:244-247: type 4 (illumination + deformation), pixel-aware prediction
        GyroAidedTracker tracker(k * 0.05, (k - 1) * 0.05, last, cur, keysLast, none, keysLast, none, noImu,
                                 cv::Point3f(0, 0, 0), Km, Dm, table,
                                 GyroAidedTracker::GYRO_PREDICT_WITH_OPTICAL_FLOW_REFINED_CONSIDER_ILLUMINATION_DEFORMATION,
                                 GyroAidedTracker::PIXEL_AWARE_PREDICTION, "", half);
        tracker.SetPatchMatchParams(iters, pyr);
        cv::Mat R(3, 3, cv::CV_32F);
        for (int j = 0; j < 9; j++) R.at<float>(j / 3, j % 3) = Rs[(size_t)(k - 1) * 9 + j];
        tracker.SetRcl(R);  // what IntegrateGyroMeasurements() would set from the IMU samples
        tracker.mbHasGyroPredictInitial = true, tracker.mbConsiderIllumination = true;
        tracker.mbConsiderAffineDeformation = true, tracker.mbRegularizationPenalty = false;
        const int tracked = tracker.GyroPredictFeaturesAndOpticalFlowRefined();  // :251 TrackFeatures()
        std::printf("pair %d tracked %d of %zu\n", k, tracked, keysLast.size());
        // RealSenseD435i.cpp:254-258: tracked points become the next frame's keypoints
        std::vector<cv::KeyPoint> next;
        for (size_t i = 0; i < keysLast.size(); i++)
            if (tracker.mvStatus[i]) {
                next.emplace_back(tracker.mvPtPredictUn[i].x, tracker.mvPtPredictUn[i].y);
                checksum += tracker.mvPtPredictUn[i].x + 2.0 * tracker.mvPtPredictUn[i].y;
            }
        keysLast.swap(next);
        if (keysLast.empty()) break;
    }
    std::printf("survivors %zu checksum %.6f\n", keysLast.size(), checksum);
    PatchMatch::ReleaseContext();
    return 0;
}
