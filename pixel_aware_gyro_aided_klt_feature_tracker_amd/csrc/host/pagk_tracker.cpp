// pagk_tracker.cpp -- C entry point over the C++ API shell, so that tests (and hosts in other
// languages) can drive GyroAidedTracker::TrackFeatures() end to end: libpagk_tracker.so.
#include <cstring>
#include <exception>
#include <string>
#include <vector>

#include "gyro_aided_tracker.h"
#include "patch_match.h"
#include "sequence_io.h"

static thread_local std::string g_err;

extern "C" {

const char *pagk_tracker_last_error(void) { return g_err.c_str(); }

// One frame pair through the reference's call stack (apps: RealSenseD435i.cpp:244-251):
// GyroAidedTracker(ctor #1) -> SetRcl(Rcl) or gyro integration -> TrackFeatures().
//   Rcl: 9 floats row-major, or NULL to integrate `imu` (n_imu x 7: ax ay az wx wy wz t) like the reference.
// Outputs (n each): final mvStatus, mvPtPredictUn, mvPtPredict, and PatchMatch's raw vectors.
// Returns the number of tracked features (TrackFeatures' return value) or -100 on an exception.
int pagk_tracker_track_features(const unsigned char *img_ref, const unsigned char *img_cur, int width, int height,
                                long step, int n, const float *keys_ref /*n x 2*/, const float *K /*3x3*/,
                                const float *dist /*4*/, int type, int half_patch, int iterations, int pyramids,
                                const float *Rcl, const double *imu, int n_imu, double t_ref, double t_cur,
                                unsigned char *status_out, float *pt_predict_un, float *pt_predict,
                                unsigned char *status_pm, float *pt_pm_un, double *pix_err, double *dist_pred,
                                float *affine_out /*n x 4 or NULL*/)
{
    try {
        cv::Mat ref(height, width, cv::CV_8UC1, const_cast<unsigned char *>(img_ref), (size_t)step);
        cv::Mat cur(height, width, cv::CV_8UC1, const_cast<unsigned char *>(img_cur), (size_t)step);
        std::vector<cv::KeyPoint> keys(n), none;
        for (int i = 0; i < n; i++) keys[i].pt = cv::Point2f(keys_ref[2 * i], keys_ref[2 * i + 1]);
        cv::Mat Km(3, 3, cv::CV_32F), Dm(1, 4, cv::CV_32F), table;
        for (int k = 0; k < 9; k++) Km.at<float>(k / 3, k % 3) = K[k];
        for (int k = 0; k < 4; k++) Dm.at<float>(k) = dist[k];
        std::vector<IMU::Point> vimu;
        for (int k = 0; k < n_imu; k++)
            vimu.emplace_back((float)imu[7 * k], (float)imu[7 * k + 1], (float)imu[7 * k + 2], (float)imu[7 * k + 3],
                              (float)imu[7 * k + 4], (float)imu[7 * k + 5], imu[7 * k + 6]);
        GyroAidedTracker trk(t_cur, t_ref, ref, cur, keys, none, keys, none, vimu, cv::Point3f(0, 0, 0), Km, Dm, table,
                             (GyroAidedTracker::eType)type, GyroAidedTracker::PIXEL_AWARE_PREDICTION, "", half_patch);
        trk.SetPatchMatchParams(iterations, pyramids);
        int ret;
        if (Rcl) {
            cv::Mat R(3, 3, cv::CV_32F);
            for (int k = 0; k < 9; k++) R.at<float>(k / 3, k % 3) = Rcl[k];
            // TrackFeatures() starts with IntegrateGyroMeasurements(); with no IMU samples that
            // leaves Rcl = I, so set the rotation afterwards through the same dispatch by hand.
            trk.SetRcl(R);
            if (type == GyroAidedTracker::GYRO_PREDICT)
                ret = trk.GyroPredictFeatures();
            else {
                // flags as TrackFeatures sets them (:384-414)
                trk.mbHasGyroPredictInitial = type != GyroAidedTracker::IMAGE_ONLY_OPTICAL_FLOW_CONSIDER_ILLUMINATION;
                trk.mbConsiderIllumination = type != GyroAidedTracker::GYRO_PREDICT_WITH_OPTICAL_FLOW_REFINED;
                trk.mbConsiderAffineDeformation =
                    type == GyroAidedTracker::IMAGE_ONLY_OPTICAL_FLOW_CONSIDER_ILLUMINATION || type == 4 || type == 6;
                trk.mbRegularizationPenalty = type == 6;
                ret = trk.GyroPredictFeaturesAndOpticalFlowRefined();
            }
        } else {
            ret = trk.TrackFeatures();
        }
        for (int i = 0; i < n; i++) {
            status_out[i] = trk.mvStatus[i];
            pt_predict_un[2 * i] = trk.mvPtPredictUn[i].x, pt_predict_un[2 * i + 1] = trk.mvPtPredictUn[i].y;
            pt_predict[2 * i] = trk.mvPtPredict[i].x, pt_predict[2 * i + 1] = trk.mvPtPredict[i].y;
            if ((int)trk.mvStatusAfterPatchMatched.size() == n) {
                status_pm[i] = trk.mvStatusAfterPatchMatched[i];
                pt_pm_un[2 * i] = trk.mvPtPredictAfterPatchMatchedUn[i].x;
                pt_pm_un[2 * i + 1] = trk.mvPtPredictAfterPatchMatchedUn[i].y;
                pix_err[i] = trk.mvPixelErrorsOfPatchMatched[i];
                dist_pred[i] = trk.mvDistanceBetweenPredictedAndPatchMatched[i];
            }
            if (affine_out) {
                const cv::Mat &A = trk.mvAffineDeformationMatrix[i];
                for (int k = 0; k < 4; k++) affine_out[4 * i + k] = A.empty() ? 0.f : A.at<float>(k / 2, k % 2);
            }
        }
        return ret;
    } catch (const std::exception &e) {
        g_err = e.what();
        return -100;
    }
}

// GyroAidedTracker::GeometryValidation (reference :429-480) on a tracker whose state is given by the
// arrays: keys_ref_un / pt_predict_un (n x 2), status (n, updated in place).  H21, H12, F21: the fitted
// models, 3x3 row-major double.  Returns cnt_inlier or -100 on an exception; *track_score as the reference logs it.
int pagk_tracker_geometry_validation(int n, const float *keys_ref_un, const float *pt_predict_un, unsigned char *status,
                                     const double *H21, const double *H12, const double *F21, float *track_score)
{
    try {
        static unsigned char px = 0;
        cv::Mat img(1, 1, cv::CV_8UC1, &px, 1);
        std::vector<cv::KeyPoint> keys(n), none;
        for (int i = 0; i < n; i++) keys[i].pt = cv::Point2f(keys_ref_un[2 * i], keys_ref_un[2 * i + 1]);
        cv::Mat Km = cv::Mat::eye(3, 3, cv::CV_32F), Dm(1, 4, cv::CV_32F), table;
        for (int k = 0; k < 4; k++) Dm.at<float>(k) = 0;
        std::vector<IMU::Point> vimu;
        GyroAidedTracker trk(0, 0, img, img, keys, none, keys, none, vimu, cv::Point3f(0, 0, 0), Km, Dm, table);
        for (int i = 0; i < n; i++) {
            trk.mvStatus[i] = status[i];
            trk.mvPtPredictUn[i] = cv::Point2f(pt_predict_un[2 * i], pt_predict_un[2 * i + 1]);
        }
        // through the installed-fitter form, the way an application with OpenCV would wire it
        GyroAidedTracker::SetModelFitter([&](const std::vector<cv::Point2f> &, const std::vector<cv::Point2f> &,
                                             double h21[9], double h12[9], double f21[9]) {
            for (int k = 0; k < 9; k++) h21[k] = H21[k], h12[k] = H12[k], f21[k] = F21[k];
            return true;
        });
        int ret = trk.GeometryValidation();
        GyroAidedTracker::SetModelFitter(nullptr);
        for (int i = 0; i < n; i++) status[i] = trk.mvStatus[i];
        if (track_score) *track_score = trk.mTrackScore;
        return ret;
    } catch (const std::exception &e) {
        GyroAidedTracker::SetModelFitter(nullptr);
        g_err = e.what();
        return -100;
    }
}

// ---- sequence formats (sequence_io.h), flat C views for tests and other languages ------------------
// Each returns the number of records (written up to `cap`), or -1 if the file cannot be opened.
int pagk_seq_load_keypoints(const char *path, float *xy /*cap x 2*/, int cap)
{
    std::vector<cv::Point2f> pts;
    if (!pagk_seq::LoadDetectedKeypoints(path, pts)) return -1;
    for (int i = 0; i < (int)pts.size() && i < cap; i++) xy[2 * i] = pts[i].x, xy[2 * i + 1] = pts[i].y;
    return (int)pts.size();
}
// times: cap doubles; names: cap x name_len chars, NUL-terminated
int pagk_seq_load_correspondences(const char *path, double *times, char *names, int name_len, int cap)
{
    std::vector<std::pair<double, std::string>> v;
    if (!pagk_seq::LoadCorrespondences(path, v)) return -1;
    for (int i = 0; i < (int)v.size() && i < cap; i++) {
        times[i] = v[i].first;
        std::strncpy(names + (size_t)i * name_len, v[i].second.c_str(), (size_t)name_len - 1);
        names[(size_t)i * name_len + name_len - 1] = 0;
    }
    return (int)v.size();
}
int pagk_seq_find_time(const double *times, int n, double t)
{
    std::vector<std::pair<double, std::string>> v((size_t)n);
    for (int i = 0; i < n; i++) v[i].first = times[i];
    return pagk_seq::FindTimeCorrespondenIndex(v, t);
}
int pagk_seq_parse_image_line(const char *line, double *time_s)
{
    return pagk_seq::ParseImageListLine(line, *time_s) ? 1 : 0;
}
// out: cap x 7 doubles (ax ay az wx wy wz t), the layout pagk_tracker_track_features takes
int pagk_seq_load_imu(const char *path, double *out, int cap)
{
    std::vector<IMU::Point> v;
    if (!pagk_seq::LoadImu(path, v)) return -1;
    for (int i = 0; i < (int)v.size() && i < cap; i++) {
        double *o = out + 7 * (size_t)i;
        o[0] = v[i].a.x, o[1] = v[i].a.y, o[2] = v[i].a.z, o[3] = v[i].w.x, o[4] = v[i].w.y, o[5] = v[i].w.z, o[6] = v[i].t;
    }
    return (int)v.size();
}
// The demo's IMU window over a frame-time list: counts[k] = samples handed to the tracker for the pair
// (times[k-1], times[k]), first[k] = index of the first of them in the IMU file (-1 if none).
int pagk_seq_imu_windows(const char *imu_path, const double *frame_times, int n_frames, int *first, int *counts)
{
    std::vector<IMU::Point> all;
    if (!pagk_seq::LoadImu(imu_path, all)) return -1;
    std::vector<double> stamps;
    for (auto &p : all) stamps.push_back(p.t);
    pagk_seq::ImuWindow win(all);
    double t_prev = 0;
    for (int k = 0; k < n_frames; k++) {
        std::vector<IMU::Point> w = win.Next(t_prev, frame_times[k]);
        counts[k] = (int)w.size();
        first[k] = -1;
        if (!w.empty())
            for (size_t j = 0; j < stamps.size(); j++)
                if (stamps[j] == w[0].t) {
                    first[k] = (int)j;
                    break;
                }
        t_prev = frame_times[k];
    }
    return (int)all.size();
}

void pagk_tracker_release(void) { PatchMatch::ReleaseContext(); }
}
