#include "Optimiser.h"
#include "Util.h"

void Optimiser::localBA(DataManager& data, int n_iterations) { Util::GlobalBundleAdjustemnt(data, n_iterations); }

void Optimiser::process(DataManager& data, int frameIdx)
{
    if (frameIdx != 0) localBA(data, FULL_BA_ITER);
}

bool Optimiser::validationCheck(DataManager&, int) { return true; }
