// Optimiser -- the reference's stage (src/Optimiser.h:5-18, src/Optimiser.cpp:6-18): for frameIdx != 0 it runs the
// full bundle adjustment over everything in the DataManager.  `localBA` is the new entry point north_star names
// (the reference only declares Util::LocalBundleAdjustment, src/Util.h:27, and never defines it); process() calls it.
#pragma once
#include "ProcessingNode.h"

class Optimiser : public ProcessingNode {
public:
    Optimiser() : ProcessingNode("Optimiser") {}
    void localBA(DataManager& data, int n_iterations = FULL_BA_ITER);
    virtual void process(DataManager& data, int frameIdx) override;
    virtual bool validationCheck(DataManager& data, int frameIdx) override;
};
