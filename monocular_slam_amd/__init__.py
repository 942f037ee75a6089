"""monocular_slam_amd -- MI355X (gfx950) implementation of the tracking/mapping hot path of a
monocular ORB SLAM pipeline (ORB extract, Hamming match, RANSAC F / PnP, BA linearisation)
behind the stage interface of eastgeneral2007/Monocular_SLAM.

The compute path is libmslam_hip.so (hand-written HIP, C ABI in include/mslam.h).  This
package is the thin host-side mirror used by the tests and the benchmark; it never falls
back to a CPU implementation: importing `monocular_slam_amd.api` without the built
library raises.
"""
__version__ = "0.1.0"
