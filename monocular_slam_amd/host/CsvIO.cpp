// CSV fixtures in the reference's formats (src/Util.cpp:384-588) -- the only on-disk format the reference defines
// (read by UnitTest/g2o/UnitTestg2o.cpp:31-82) -- plus the writers the reference lacks.
//   Frames.csv            per frame:  "id,r00,r01,r02,r10,r11,r12,r20,r21,r22,t0,t1,t2"
//                                     one "x,y,sigma" line per feature, blank line,
//                                     one map-point index line per feature, blank line
//   MapPoints.csv         per point:  "id,x,y,z", one "observerFrameId,featureIdx" line per observation, blank line
//   FramesOptimised.csv   per frame:  "id,R(9 row-major),t(3)"
//   MapPointsOptimised.csv per point: "id,x,y,z"
#include <fstream>
#include <sstream>
#include "Util.h"

namespace {

std::vector<double> numbers(const std::string& line)
{
    std::vector<double> v;
    std::string tok;
    std::istringstream is(line);
    while (std::getline(is, tok, ',')) {
        std::istringstream ts(tok);
        double d;
        if (ts >> d) v.push_back(d);
    }
    return v;
}

bool blank(const std::string& s) { return s.find_first_not_of(" \t\r") == std::string::npos; }

cv::Mat rtFrom(const std::vector<double>& v) // id, R (9), t (3)
{
    cv::Mat Rt = cv::Mat::zeros(3, 4, cv::CV_64F);
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) Rt.at<double>(i, j) = v[1 + (size_t)i * 3 + j]; Rt.at<double>(i, 3) = v[10 + (size_t)i]; }
    return Rt;
}

void writeRtLine(std::ofstream& out, const Frame& f)
{
    out << f.meta.frameID;
    out.precision(17);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) out << "," << f.Rt.at<double>(i, j);
    for (int i = 0; i < 3; i++) out << "," << f.Rt.at<double>(i, 3);
    out << "\n";
}

} // namespace

std::vector<Frame> Util::loadFrames(std::string frame_filename)
{
    std::vector<Frame> frames;
    std::ifstream in(frame_filename);
    std::string line;
    while (std::getline(in, line)) {
        if (blank(line)) continue;
        const std::vector<double> hdr = numbers(line);
        if (hdr.size() < 13) continue;
        Frame f;
        f.meta.frameID = (int)hdr[0];
        f.Rt = rtFrom(hdr);
        while (std::getline(in, line) && !blank(line)) {
            const std::vector<double> v = numbers(line);
            if (v.size() >= 3) { f.features.positions.push_back(Point2d(v[0], v[1])); f.features.scales.push_back(v[2]); }
        }
        while (std::getline(in, line) && !blank(line)) f.features.mapPointsIndices.push_back((int)numbers(line).at(0));
        frames.push_back(f);
    }
    return frames;
}

std::vector<MapPoint> Util::loadMapPoints(std::string map_points_filename)
{
    std::vector<MapPoint> pts;
    std::ifstream in(map_points_filename);
    std::string line;
    while (std::getline(in, line)) {
        if (blank(line)) continue;
        const std::vector<double> hdr = numbers(line);
        if (hdr.size() < 4) continue;
        MapPoint mp(Point3d(hdr[1], hdr[2], hdr[3]), (int)hdr[0]);
        while (std::getline(in, line) && !blank(line)) {
            const std::vector<double> v = numbers(line);
            if (v.size() >= 2) mp.addObservingFrame((int)v[0], (int)v[1]);
        }
        pts.push_back(mp);
    }
    return pts;
}

std::vector<Frame> Util::loadFramesOnlyIdRt(std::string frame_filename)
{
    std::vector<Frame> frames;
    std::ifstream in(frame_filename);
    std::string line;
    while (std::getline(in, line)) {
        if (blank(line)) continue;
        const std::vector<double> hdr = numbers(line);
        if (hdr.size() < 13) continue;
        Frame f;
        f.meta.frameID = (int)hdr[0];
        f.Rt = rtFrom(hdr);
        frames.push_back(f);
    }
    return frames;
}

std::vector<MapPoint> Util::loadMapPointsOnlyIdXYZ(std::string map_points_filename)
{
    std::vector<MapPoint> pts;
    std::ifstream in(map_points_filename);
    std::string line;
    while (std::getline(in, line)) {
        if (blank(line)) continue;
        const std::vector<double> v = numbers(line);
        if (v.size() >= 4) pts.push_back(MapPoint(Point3d(v[1], v[2], v[3]), (int)v[0]));
    }
    return pts;
}

void Util::saveFrames(const std::vector<Frame>& frames, std::string frame_filename)
{
    std::ofstream out(frame_filename);
    out.precision(17);
    for (const Frame& f : frames) {
        writeRtLine(out, f);
        for (size_t i = 0; i < f.features.positions.size(); i++)
            out << f.features.positions[i].x << "," << f.features.positions[i].y << "," << f.features.scales[i] << "\n";
        out << "\n";
        for (int m : f.features.mapPointsIndices) out << m << "\n";
        out << "\n";
    }
}

void Util::saveMapPoints(const std::vector<MapPoint>& map_points, std::string map_points_filename)
{
    std::ofstream out(map_points_filename);
    out.precision(17);
    for (const MapPoint& m : map_points) {
        out << m.id << "," << m.worldPosition.x << "," << m.worldPosition.y << "," << m.worldPosition.z << "\n";
        for (const auto& ob : m.observers()) out << ob.first << "," << ob.second << "\n";
        out << "\n";
    }
}

void Util::saveFramesOnlyIdRt(const std::vector<Frame>& frames, std::string frame_filename)
{
    std::ofstream out(frame_filename);
    for (const Frame& f : frames) if (!f.Rt.empty()) writeRtLine(out, f);
}

void Util::saveMapPointsOnlyIdXYZ(const std::vector<MapPoint>& map_points, std::string map_points_filename)
{
    std::ofstream out(map_points_filename);
    out.precision(17);
    for (const MapPoint& m : map_points) out << m.id << "," << m.worldPosition.x << "," << m.worldPosition.y << "," << m.worldPosition.z << "\n";
}

std::map<int, std::string> Util::loadFrameIdToImageFileName(std::string info_filename)
{
    std::map<int, std::string> frame_id_to_file;
    std::ifstream in(info_filename);
    std::string line;
    while (std::getline(in, line)) {
        std::istringstream is(line);
        int id;
        std::string name;
        if (is >> id >> name) frame_id_to_file[id] = name;
    }
    return frame_id_to_file;
}
