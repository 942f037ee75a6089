#include "Optimiser.h"
#include "Util.h"
#include <algorithm>
#include <vector>

// window <= 0: what the reference's stage does today -- full BA over every frame and map point with frame 0 fixed
// (src/Optimiser.cpp:6-18 -> Util::GlobalBundleAdjustemnt, src/Util.cpp:28-32).
// window = k > 0: the sliding-window BA the reference only declares (Util::LocalBundleAdjustment, src/Util.h:27;
// LocalMapper::localBA is a stub, src/LocalMapper.h:30-42): the last k frames that carry a pose are free, every earlier
// frame is held fixed, and the points the free frames observe are optimised.
void Optimiser::localBA(DataManager& data, int n_iterations, int window_)
{
    if (window_ <= 0) { Util::GlobalBundleAdjustemnt(data, n_iterations); return; }
    // frame ids of the posed frames, ascending; the window's first id is the k-th largest (but never frame id 0's slot:
    // with fewer than k + 1 posed frames the oldest one stays the gauge, as in the global problem)
    std::vector<int> ids;
    for (const Frame& f : data.frames)
        if (!f.Rt.empty()) ids.push_back(f.meta.frameID);
    if (ids.size() < 2) return;
    std::sort(ids.begin(), ids.end());
    const size_t k = (size_t)window_ < ids.size() - 1 ? (size_t)window_ : ids.size() - 1;
    Util::LocalBundleAdjustment(data, ids[ids.size() - k], n_iterations);
}

void Optimiser::process(DataManager& data, int frameIdx)
{
    if (frameIdx != 0) localBA(data, n_iterations, window);
}

bool Optimiser::validationCheck(DataManager&, int) { return true; }
