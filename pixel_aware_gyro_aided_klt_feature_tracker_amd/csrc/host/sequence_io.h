// sequence_io.h -- the text formats the reference's demo reads, so that a
// recorded sequence can drive the tracker:
//   <stamp_ns>.txt       SuperPoint keypoints of one frame, lines "idx, x, y"   (src/frame.cpp:222-240)
//   corresponds.txt      lines "<t_seconds>, <stamp_ns>"                        (Examples/Demo/RealSenseD435i.cpp:167-182)
//   image_file_list.txt  one image path per line, time = file stem in ns * 1e-9 (Examples/Demo/RealSenseD435i.cpp:74-100)
//   imu.txt              lines "<stamp_ns> ax ay az wx wy wz"                   (Examples/Demo/RealSenseD435i.cpp:102-141)
// Parsing follows the reference statement for statement (atof / stol, the same split points) so that the
// same files give the same numbers.  Image decoding (cv::imread) is the application's.
#pragma once
#include <string>
#include <utility>
#include <vector>

#include "cvlite.h"
#include "gyro_aided_tracker.h"  // IMU::Point

namespace pagk_seq {

// Frame::LoadDetectedKeypointFromFile, the parsing half (src/frame.cpp:224-240): every line is split at
// ',', fields go through atof, the point is (field 1, field 2).  Returns false if the file cannot be opened
// (the reference prints and returns).  A line with fewer than three fields is skipped (the reference would
// index past its vector there).
bool LoadDetectedKeypoints(const std::string &path, std::vector<cv::Point2f> &pts);

// Examples/Demo/RealSenseD435i.cpp:167-182: t1 = atof(text before the first ','), name = text after ", ".
bool LoadCorrespondences(const std::string &path, std::vector<std::pair<double, std::string>> &out);

// One line of image_file_list.txt (getNextFrame, Examples/Demo/RealSenseD435i.cpp:89-94): time = stol(text
// between the last '/' and ".png") * 1e-9.  Returns false when the line has no ".png" stem to parse.
bool ParseImageListLine(const std::string &line, double &time_s);
bool LoadImageList(const std::string &path, std::vector<std::pair<double, std::string>> &out);

// getNextIMU (Examples/Demo/RealSenseD435i.cpp:117-131): "stamp ax ay az wx wy wz", t = stol(stamp) * 1e-9,
// the six values read as double and narrowed to the float members of IMU::Point.
bool ParseImuLine(const std::string &line, IMU::Point &imu);
bool LoadImu(const std::string &path, std::vector<IMU::Point> &out);

// findTimeCorrespondenIndex (include/common.h:105-114): first entry within 0.1 ms of t, or -1.
int FindTimeCorrespondenIndex(const std::vector<std::pair<double, std::string>> &v, double t);

// The demo's gyro window as the streaming loop runs it (Examples/Demo/RealSenseD435i.cpp:196-197, 207-218):
// `last_imu` starts as the first sample; for each frame pair the reader first skips samples older than
// t_prev - delay, then collects samples while last_imu.t < t_cur - delay.  The first frame (t_prev == 0)
// collects nothing.  Feed the frames in order.
class ImuWindow {
public:
    explicit ImuWindow(std::vector<IMU::Point> all, double delay = 0.0);  // MANUALLY_ADD_TIME_DELAY = 0 (:44)
    std::vector<IMU::Point> Next(double t_prev, double t_cur);

private:
    bool getNext(IMU::Point &p);
    std::vector<IMU::Point> all_;
    size_t pos_ = 0;
    IMU::Point last_;
    bool valid_ = true;
    double delay_;
};

}  // namespace pagk_seq
