// Synth.h -- seeded synthetic inputs for the C++ host tools (no dataset exists on the GPU box): corner-rich gray frames,
// the next frame of a stream (same scene shifted, fresh noise) and 3D-2D problems of BASELINE configs[2] shape.
// Generators only; they follow the recipe of monocular_slam_amd/synth.py with the xorshift64* PRNG of
// Geometry.h, so the values differ from the Python ones while the statistics (corner density, inlier ratio) match.
#pragma once
#include <cstdint>
#include <vector>

namespace synth {

void texturedFrame(int w, int h, uint64_t seed, uint8_t* out);                                   // w*h bytes
void shiftedFrame(const uint8_t* img, int w, int h, int dx, int dy, uint64_t seed, uint8_t* out);
// P points in the frustum of K (z in [2, 10] m), small random pose, sigma 0.5 px noise, 30 % uniform outliers
void pnpProblem(int P, uint64_t seed, const double K[9], int w, int h, double* X /*P*3*/, double* x /*P*2*/);
extern const double K_MINITAUR[9];   // src/FrameLoader.cpp:196-199

} // namespace synth
