// FrameLoader -- the reference's sequence loader (src/FrameLoader.h:14-33, src/FrameLoader.cpp:36-301): every image of a
// directory (file name = "<timestamp>.png"), sorted by timestamp, frames [begin_frame, end_frame) with a stride, plus the
// data set specific extras chosen by the directory NAME exactly as the reference does (:282-301):
//   "...rgbd_dataset_freiburg1_desk2_secret..."  images in the directory itself, TUM-1 intrinsics
//   "...f1..." / "...f3..."                      rgb/ + groundtruth.txt + depth/, Kinect intrinsics
//   "...temple..."                               images + temple_par.txt (per-frame K and ground-truth R|t)
//   "...MinitaurHoppingVideo..."                 images, Minitaur intrinsics
// Differences: images are decoded by host/ImageIO (PNG and binary PGM/PPM; a .jpg raises an error instead of being read),
// and a directory matching none of the names loads its images with the intrinsics given to the constructor
// (the reference silently loads nothing).
#pragma once
#include <string>
#include "DataManager.h"

class FrameLoader {
public:
    FrameLoader(std::string directory, int begin_frame, int end_frame, int step)
        : directory(directory), begin_frame(begin_frame), end_frame(end_frame), step(step) {}
    void load(DataManager& data);
    double default_fx = 517.3, default_fy = 516.5, default_cx = 318.6, default_cy = 255.3; // used for unnamed data sets only
private:
    std::string directory;
    int begin_frame, end_frame, step;
};

// camera-to-world ground truth -> the convention the reference stores in Frame::RtGt (src/PointCloudVisualizer.cpp:221-231)
void WorldRtToRT(const cv::Mat& Rt, cv::Mat& Rt_res);
void RtToWorldRT(const cv::Mat& Rt, cv::Mat& Rt_res);
