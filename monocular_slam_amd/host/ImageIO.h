// ImageIO -- cv::imread for the file types the sequence loader meets, without OpenCV: PNG (all colour types, 8/16 bit,
// non-interlaced; inflate by zlib) and binary PGM/PPM.  Output follows imread: 8-bit, BGR order for colour
// (CV_LOAD_IMAGE_UNCHANGED keeps 1/3/4 channels, alpha last), or CV_32F single channel for depth maps
// (imread(ANYDEPTH) + convertTo(CV_32F), src/FrameLoader.cpp:97-98).  JPEG is not supported (no decoder in the image).
#pragma once
#include <string>
#include "Mat.h"

namespace imageio {
enum { LOAD_UNCHANGED = -1, LOAD_ANYDEPTH_F32 = 2 };
// false + message in `err` on failure
bool imread(const std::string& path, cv::Mat& out, int flags, std::string* err = nullptr);
// test helper / tools: 8-bit gray or BGR -> PNG (stored deflate blocks are enough for fixtures; uses zlib's compress)
bool imwritePng(const std::string& path, const cv::Mat& img, std::string* err = nullptr);
}
