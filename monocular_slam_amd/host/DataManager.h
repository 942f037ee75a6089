// The blackboard data model of the reference (src/MapPoint.h:16-28, src/Frame.h:22-72, src/DataManager.h:23-36),
// re-stated with identical member names so that stage code written against the reference compiles against it.
#pragma once
#include <cassert>
#include "Common.h"

using cv::Mat;
using cv::Point2d;
using cv::Point2f;
using cv::Point3d;

class MapPoint {
    friend class Util;
public:
    MapPoint(Point3d worldPosition, int id) : id((long unsigned int)id), worldPosition(worldPosition) {}
    long unsigned int id;   // identifier (== index in DataManager::mapPoints)
    Point3d worldPosition;  // 3D world coordinate
    // observer bookkeeping (reference src/MapPoint.cpp:8-28): observing frame id -> feature index, -1 if not an observer
    void addObservingFrame(int frameIdx, int featureIdx) { assert(featureIdx >= 0); observerToIndex[frameIdx] = featureIdx; }
    int getFeatureIdxFromObservingFrame(int frameIdx) const { auto it = observerToIndex.find(frameIdx); return it == observerToIndex.end() ? -1 : it->second; }
    void deleteObservingFrame(int frameIdx) { observerToIndex.erase(frameIdx); }
    const std::map<int, int>& observers() const { return observerToIndex; }
private:
    std::map<int, int> observerToIndex; // observing frame id -> feature index in that frame
};

struct Features {
    std::vector<Point2d> positions;     // N feature positions (level-0 pixels)
    Mat descriptors;                    // N x 32 CV_8U
    std::vector<int> mapPointsIndices;  // map point of each feature or -1
    std::vector<double> scales;         // keypoint size
};

struct FrameMeta { double timestamp = 0; int frameID = 0; std::string framename; };

typedef Mat RawBuffer;

struct Frame {
    FrameMeta meta;
    RawBuffer frameBuffer;  // pixels (CV_8U, 1/3/4 channels)
    RawBuffer depthBuffer;
    Mat Rt;                 // 3x4 CV_64F world -> camera
    Mat RtGt;
    Features features;
    Mat K;                  // 3x3 CV_64F
    bool operator<(const Frame& o) const { return o.meta.timestamp > meta.timestamp; }
};

class DataManager {
public:
    Mat camera_intrinsics;               // deprecated in the reference too
    std::vector<Frame> frames;
    std::vector<Mat> relative_frame_poses;
    std::vector<MapPoint> mapPoints;
};
