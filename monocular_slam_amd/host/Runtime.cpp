#include "Runtime.h"
#include <cstdlib>

MslamRuntime& MslamRuntime::instance() { static MslamRuntime r; return r; }

void MslamRuntime::ensure(int w, int h, int max_kpts)
{
    if (ctx_ && w <= w_ && h <= h_ && max_kpts <= kp_) return;
    if (ctx_) { mslam_destroy(ctx_); ctx_ = nullptr; }
    w_ = w > w_ ? w : w_; h_ = h > h_ ? h : h_; kp_ = max_kpts > kp_ ? max_kpts : kp_;
    const char* dev = std::getenv("MSLAM_DEVICE");
    // no fallback: without a gfx950 device this throws and the pipeline stops
    check(mslam_create(&ctx_, dev ? std::atoi(dev) : 0, w_, h_, kp_, 1), "mslam_create");
}

MslamRuntime::~MslamRuntime() { if (ctx_) mslam_destroy(ctx_); }
