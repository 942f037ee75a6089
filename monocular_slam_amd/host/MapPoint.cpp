// MapPoint bookkeeping (reference src/MapPoint.cpp:8-28): observer frame id -> feature index.
#include <cassert>
#include "DataManager.h"

void MapPoint::addObservingFrame(int frameIdx, int featureIdx)
{
    assert(featureIdx >= 0);
    observerToIndex[frameIdx] = featureIdx;
}

int MapPoint::getFeatureIdxFromObservingFrame(int frameIdx)
{
    auto it = observerToIndex.find(frameIdx);
    return it == observerToIndex.end() ? -1 : it->second;
}

void MapPoint::deleteObservingFrame(int frameIdx) { observerToIndex.erase(frameIdx); }
