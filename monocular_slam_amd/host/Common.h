// Common.h / ParamConfig.h of the reference (src/Common.h:26, src/ParamConfig.h:5-25) re-stated: same constant names and values.
#pragma once
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <map>
#include <string>
#include <vector>
#include "Mat.h"

#define EPSILON (1e-03)

// feature matching
const static float FEATURE_MATCH_RATIO_TEST = 0.85f;
// Huber thresholds (float, as in the reference: the truncation to float is part of the semantics)
const float THRESH_HUBER_FULL_BA = (float)std::sqrt(5.99);
const float THRESH_HUBER = (float)std::sqrt(5.991);
const float CHI2_THRESH = 5.991f;
const int POSE_BA_ITER = 10;
const int FULL_BA_ITER = 15;
// Fundamental matrix parameters
const double MAX_DISTANCE = 3.;
const double CONFIDENCE = 0.85;
