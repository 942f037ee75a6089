// util_check -- CPU-only hook for the Util helpers that never touch the GPU:
//   util_check computef Frames.csv fx fy cx cy id1 id2   -> "F f00 .. f22"   (Util::ComputeF on two loaded frames)
//   util_check info out_info.txt                          -> "id name" per entry (Util::loadFrameIdToImageFileName)
#include <cstdio>
#include <cstring>
#include "Util.h"

int main(int argc, char** argv)
{
    if (argc >= 9 && !std::strcmp(argv[1], "computef")) {
        std::vector<Frame> frames = Util::loadFramesOnlyIdRt(argv[2]);
        cv::Mat K = cv::Mat::eye(3, 3, cv::CV_64F);
        K.at<double>(0, 0) = std::atof(argv[3]); K.at<double>(1, 1) = std::atof(argv[4]);
        K.at<double>(0, 2) = std::atof(argv[5]); K.at<double>(1, 2) = std::atof(argv[6]);
        for (Frame& f : frames) f.K = K;
        Frame* a = Util::findFrameById(frames, std::atoi(argv[7]));
        Frame* b = Util::findFrameById(frames, std::atoi(argv[8]));
        if (!a || !b) return 1;
        cv::Mat F = Util::ComputeF(*a, *b);
        if (F.empty()) return 1;
        std::printf("F");
        for (int i = 0; i < 9; i++) std::printf(" %.17g", F.at<double>(i / 3, i % 3));
        std::printf("\n");
        return 0;
    }
    if (argc >= 3 && !std::strcmp(argv[1], "info")) {
        for (const auto& kv : Util::loadFrameIdToImageFileName(argv[2])) std::printf("%d %s\n", kv.first, kv.second.c_str());
        return 0;
    }
    return 2;
}
