#include "FrameLoader.h"
#include <dirent.h>
#include <algorithm>
#include <cstdio>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include "ImageIO.h"

namespace {

bool hasSuffix(const std::string& s, const std::string& suffix)
{
    return s.size() >= suffix.size() && std::equal(suffix.rbegin(), suffix.rend(), s.rbegin());
}

bool isImageName(const std::string& n)
{
    return hasSuffix(n, ".png") || hasSuffix(n, ".jpg") || hasSuffix(n, ".pgm") || hasSuffix(n, ".ppm");
}

// "<timestamp>.<ext>" listing of a directory, ascending by timestamp (src/FrameLoader.cpp:25-56)
std::vector<Frame> listImages(std::string directory)
{
    if (directory.empty() || directory.back() != '/') directory += '/';
    std::vector<Frame> names;
    if (DIR* d = opendir(directory.c_str())) {
        while (struct dirent* e = readdir(d)) {
            const std::string n = e->d_name;
            if (!isImageName(n)) continue;
            Frame f;
            double ts = 0;
            std::sscanf(n.c_str(), "%lf", &ts);
            f.meta.timestamp = ts;
            f.meta.framename = directory + n;
            names.push_back(f);
        }
        closedir(d);
    }
    std::sort(names.begin(), names.end());
    return names;
}

cv::Mat readImage(const std::string& path, int flags)
{
    cv::Mat m;
    std::string err;
    if (!imageio::imread(path, m, flags, &err)) throw std::runtime_error("FrameLoader: " + err);
    return m;
}

void loadImgFileList(const std::string& directory, int begin_frame, int end_frame, DataManager& data, int step)
{
    const std::vector<Frame> names = listImages(directory);
    for (int i = begin_frame; i < std::min(end_frame, (int)names.size()); i += step) {
        Frame f;
        f.meta.timestamp = names[(size_t)i].meta.timestamp;
        f.meta.framename = names[(size_t)i].meta.framename;
        f.meta.frameID = i;                                             // the index in the sorted listing, not the position in data.frames
        f.frameBuffer = readImage(f.meta.framename, imageio::LOAD_UNCHANGED);
        data.frames.push_back(f);
    }
}

void loadDepthFileList(const std::string& directory, int begin_frame, int end_frame, DataManager& data, int step)
{
    const std::vector<Frame> names = listImages(directory);
    size_t count = 0;
    for (int i = begin_frame; i < std::min(end_frame, (int)names.size()) && count < data.frames.size(); i += step)
        data.frames[count++].depthBuffer = readImage(names[(size_t)i].meta.framename, imageio::LOAD_ANYDEPTH_F32);
}

// unit quaternion + translation -> 3x4 (src/FrameLoader.cpp:103-121)
void loadRT(cv::Mat& Rt, double tx, double ty, double tz, double qx, double qy, double qz, double qw)
{
    Rt = cv::Mat::zeros(3, 4, cv::CV_64F);
    Rt.at<double>(0, 0) = 1 - 2 * qy * qy - 2 * qz * qz; Rt.at<double>(0, 1) = 2 * qx * qy - 2 * qz * qw; Rt.at<double>(0, 2) = 2 * qx * qz + 2 * qy * qw;
    Rt.at<double>(1, 0) = 2 * qx * qy + 2 * qz * qw; Rt.at<double>(1, 1) = 1 - 2 * qx * qx - 2 * qz * qz; Rt.at<double>(1, 2) = 2 * qy * qz - 2 * qx * qw;
    Rt.at<double>(2, 0) = 2 * qx * qz - 2 * qy * qw; Rt.at<double>(2, 1) = 2 * qy * qz + 2 * qx * qw; Rt.at<double>(2, 2) = 1 - 2 * qx * qx - 2 * qy * qy;
    Rt.at<double>(0, 3) = tx; Rt.at<double>(1, 3) = ty; Rt.at<double>(2, 3) = tz;
}

// TUM groundtruth.txt: 3 comment lines, then "timestamp tx ty tz qx qy qz qw"; every frame takes the first record whose
// timestamp is within 20 ms before the frame's or later (src/FrameLoader.cpp:123-175)
void loadGroundTruth(std::string filename, DataManager& data)
{
    if (data.frames.empty()) return;
    if (!filename.empty() && filename.back() == '/') filename.pop_back();
    std::ifstream fin(filename);
    if (!fin.is_open()) { std::fprintf(stderr, "Can't find ground truth file %s\n", filename.c_str()); return; }
    std::string line;
    double ts_gt = 0, tx = 0, ty = 0, tz = 0, qx = 0, qy = 0, qz = 0, qw = 1;
    auto next = [&]() {
        if (!std::getline(fin, line)) return false;
        std::sscanf(line.c_str(), "%lf %lf %lf %lf %lf %lf %lf %lf", &ts_gt, &tx, &ty, &tz, &qx, &qy, &qz, &qw);
        return true;
    };
    for (int k = 0; k < 3; k++) std::getline(fin, line);
    if (!next()) return;
    for (size_t i = 0; i < data.frames.size(); i++) {
        const double ts = data.frames[i].meta.timestamp;
        while (ts >= ts_gt + 0.02) if (!next()) break;
        cv::Mat Rt, res;
        loadRT(Rt, tx, ty, tz, qx, qy, qz, qw);
        WorldRtToRT(Rt, res);
        data.frames[i].RtGt = res;
    }
}

void setIntrinsics(DataManager& data, double fx, double fy, double cx, double cy)
{
    cv::Mat K = cv::Mat::zeros(3, 3, cv::CV_64F);
    K.at<double>(0, 0) = fx; K.at<double>(0, 2) = cx; K.at<double>(1, 1) = fy; K.at<double>(1, 2) = cy; K.at<double>(2, 2) = 1.0;
    for (Frame& f : data.frames) f.K = K.clone();
    data.camera_intrinsics = K.clone();
}

// Middlebury "<name>_par.txt": count, then per image "name k11..k33 r11..r33 t1 t2 t3" (src/FrameLoader.cpp:239-280)
void loadMiddlebury(DataManager& data, const std::string& filename, int begin_frame, int end_frame, int step)
{
    std::ifstream file(filename);
    if (!file) throw std::runtime_error("FrameLoader: cannot open " + filename);
    int numImages = 0;
    file >> numImages;
    std::string tmp;
    std::getline(file, tmp);
    for (int i = 0; i < begin_frame; i++) std::getline(file, tmp);
    for (int j = begin_frame; j < std::min(end_frame, numImages); j += step) {
        const size_t i = (size_t)((j - begin_frame) / step);
        if (i >= data.frames.size()) break;
        file >> tmp;
        Frame& frame = data.frames[i];
        frame.K = cv::Mat::zeros(3, 3, cv::CV_64F);
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) file >> frame.K.at<double>(r, c);
        cv::Mat gt = cv::Mat::zeros(3, 4, cv::CV_64F), res;
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) file >> gt.at<double>(r, c);
        for (int r = 0; r < 3; r++) file >> gt.at<double>(r, 3);
        WorldRtToRT(gt, res);
        frame.RtGt = res;
        std::getline(file, tmp);
        for (int s = 0; s < step - 1; s++) std::getline(file, tmp);
    }
}

} // namespace

// as written in the reference: R is kept, t' = -R t (world pose -> stored pose) and t' = -R^-1 t the other way round
void WorldRtToRT(const cv::Mat& Rt, cv::Mat& Rt_res)
{
    cv::Mat out = cv::Mat::zeros(3, 4, cv::CV_64F);
    for (int r = 0; r < 3; r++) {
        double acc = 0;
        for (int c = 0; c < 3; c++) { out.at<double>(r, c) = Rt.at<double>(r, c); acc += Rt.at<double>(r, c) * Rt.at<double>(c, 3); }
        out.at<double>(r, 3) = -acc;
    }
    Rt_res = out;
}

void RtToWorldRT(const cv::Mat& Rt, cv::Mat& Rt_res)
{
    // t' = -R^-1 t by the adjugate (R need not be exactly orthonormal after optimisation)
    double R[9];
    for (int i = 0; i < 9; i++) R[i] = Rt.at<double>(i / 3, i % 3);
    const double det = R[0] * (R[4] * R[8] - R[5] * R[7]) - R[1] * (R[3] * R[8] - R[5] * R[6]) + R[2] * (R[3] * R[7] - R[4] * R[6]);
    const double inv[9] = { (R[4] * R[8] - R[5] * R[7]) / det, (R[2] * R[7] - R[1] * R[8]) / det, (R[1] * R[5] - R[2] * R[4]) / det,
                            (R[5] * R[6] - R[3] * R[8]) / det, (R[0] * R[8] - R[2] * R[6]) / det, (R[2] * R[3] - R[0] * R[5]) / det,
                            (R[3] * R[7] - R[4] * R[6]) / det, (R[1] * R[6] - R[0] * R[7]) / det, (R[0] * R[4] - R[1] * R[3]) / det };
    cv::Mat out = cv::Mat::zeros(3, 4, cv::CV_64F);
    for (int r = 0; r < 3; r++) {
        double acc = 0;
        for (int c = 0; c < 3; c++) { out.at<double>(r, c) = R[r * 3 + c]; acc += inv[r * 3 + c] * Rt.at<double>(c, 3); }
        out.at<double>(r, 3) = -acc;
    }
    Rt_res = out;
}

void FrameLoader::load(DataManager& data)
{
    auto has = [&](const char* s) { return directory.find(s) != std::string::npos; };
    if (has("rgbd_dataset_freiburg1_desk2_secret")) {
        loadImgFileList(directory, begin_frame, end_frame, data, step);
        setIntrinsics(data, 517.306408, 516.469215, 318.643040, 255.313989);       // loadCameraIntrinsics_TUM1
    } else if (has("f1") || has("f3")) {
        loadImgFileList(directory + "/rgb/", begin_frame, end_frame, data, step);
        loadGroundTruth(directory + "/groundtruth.txt/", data);
        loadDepthFileList(directory + "/depth/", begin_frame, end_frame, data, step);
        setIntrinsics(data, 517.3, 516.5, 318.6, 255.3);                            // loadCameraIntrinsics_kinect
    } else if (has("temple")) {
        loadImgFileList(directory + "/", begin_frame, end_frame, data, step);
        loadMiddlebury(data, directory + "/temple_par.txt", begin_frame, end_frame, step);
    } else if (has("MinitaurHoppingVideo")) {
        loadImgFileList(directory, begin_frame, end_frame, data, step);
        setIntrinsics(data, 712.9293, 683.2151, 913.5476, 550.1913);               // loadCameraIntrinsics_Minitaur
    } else {
        loadImgFileList(directory, begin_frame, end_frame, data, step);
        setIntrinsics(data, default_fx, default_fy, default_cx, default_cy);
    }
}
