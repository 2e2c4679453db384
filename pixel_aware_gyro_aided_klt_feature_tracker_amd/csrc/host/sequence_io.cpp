// sequence_io.cpp -- see sequence_io.h.  file:line citations are relative to the reference checkout.
#include "sequence_io.h"

#include <cmath>
#include <cstdlib>
#include <fstream>
#include <sstream>

namespace pagk_seq {

bool LoadDetectedKeypoints(const std::string &path, std::vector<cv::Point2f> &pts)
{
    pts.clear();
    std::ifstream fin(path.c_str());
    if (!fin.is_open()) return false;  // src/frame.cpp:227-229
    std::string line;
    while (getline(fin, line)) {  // :232-240
        std::istringstream sin(line);
        std::vector<double> data;
        std::string field;
        while (getline(sin, field, ',')) data.push_back(std::atof(field.c_str()));
        if (data.size() < 3) continue;
        pts.push_back(cv::Point2f((float)data[1], (float)data[2]));  // cv::Point2f(double, double) narrows
    }
    return true;
}

bool LoadCorrespondences(const std::string &path, std::vector<std::pair<double, std::string>> &out)
{
    out.clear();
    std::ifstream fin(path.c_str());
    if (!fin.is_open()) return false;  // Examples/Demo/RealSenseD435i.cpp:170-171
    std::string line;
    while (getline(fin, line)) {  // :174-178
        std::string::size_type p_dot = line.find(",");
        if (p_dot == std::string::npos || p_dot + 2 > line.size()) continue;
        std::string t1_str = line.substr(0, p_dot), t2_str = line.substr(p_dot + 2, line.size() - p_dot);
        // beyond the reference: a CR left by a CRLF file would otherwise end up in the file name
        while (!t2_str.empty() && (t2_str.back() == '\r' || t2_str.back() == '\n')) t2_str.pop_back();
        out.push_back(std::make_pair(std::atof(t1_str.c_str()), t2_str));
    }
    return true;
}

int FindTimeCorrespondenIndex(const std::vector<std::pair<double, std::string>> &v, double t)
{
    for (size_t i = 0; i < v.size(); i++)  // include/common.h:107-112
        if (std::abs(t - v[i].first) < 0.0001) return (int)i;
    return -1;
}

bool ParseImageListLine(const std::string &line, double &time_s)
{
    // Examples/Demo/RealSenseD435i.cpp:92-93
    std::string::size_type pos1 = line.rfind("/"), pos2 = line.rfind(".png");
    if (pos2 == std::string::npos) return false;
    const std::string::size_type start = pos1 == std::string::npos ? 0 : pos1 + 1;
    if (pos2 <= start) return false;
    const std::string stem = line.substr(start, pos2 - start);
    char *end = nullptr;
    const long ns = std::strtol(stem.c_str(), &end, 10);  // std::stol
    if (end == stem.c_str()) return false;
    time_s = ns * 1e-9;
    return true;
}

bool LoadImageList(const std::string &path, std::vector<std::pair<double, std::string>> &out)
{
    out.clear();
    std::ifstream fin(path.c_str());
    if (!fin.is_open()) return false;
    std::string line;
    while (getline(fin, line)) {
        while (!line.empty() && (line.back() == '\r' || line.back() == '\n')) line.pop_back();
        double t;
        if (ParseImageListLine(line, t)) out.push_back(std::make_pair(t, line));
    }
    return true;
}

bool ParseImuLine(const std::string &line, IMU::Point &imu)
{
    std::istringstream sin(line);  // Examples/Demo/RealSenseD435i.cpp:118-127
    double a[3], w[3];
    std::string str_time;
    if (!(sin >> str_time >> a[0] >> a[1] >> a[2] >> w[0] >> w[1] >> w[2])) return false;
    char *end = nullptr;
    const long ns = std::strtol(str_time.c_str(), &end, 10);
    if (end == str_time.c_str()) return false;
    imu.a.x = a[0], imu.a.y = a[1], imu.a.z = a[2];
    imu.w.x = w[0], imu.w.y = w[1], imu.w.z = w[2];
    imu.t = ns * 1e-9;
    return true;
}

bool LoadImu(const std::string &path, std::vector<IMU::Point> &out)
{
    out.clear();
    std::ifstream fin(path.c_str());
    if (!fin.is_open()) return false;
    std::string line;
    while (getline(fin, line)) {
        IMU::Point p;
        if (ParseImuLine(line, p)) out.push_back(p);
    }
    return true;
}

ImuWindow::ImuWindow(std::vector<IMU::Point> all, double delay) : all_(std::move(all)), delay_(delay)
{
    getNext(last_);  // :196  (valid_imu starts true regardless, :197)
}

bool ImuWindow::getNext(IMU::Point &p)
{
    if (pos_ >= all_.size()) return false;  // getline fails: the sample is left as it was
    p = all_[pos_++];
    return true;
}

std::vector<IMU::Point> ImuWindow::Next(double t_prev, double t_cur)
{
    std::vector<IMU::Point> vImuMeas;
    if (t_prev != 0) {  // :208
        while (last_.t < t_prev - delay_ && getNext(last_)) continue;  // :209-210
        while (last_.t < t_cur - delay_ && valid_) {                   // :213-216
            vImuMeas.push_back(last_);
            valid_ = getNext(last_);
        }
    }
    return vImuMeas;
}

}  // namespace pagk_seq
