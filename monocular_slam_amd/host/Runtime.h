// Runtime.h -- process-wide handle on the C ABI context (one per GPU) shared by the stages.
#pragma once
#include <stdexcept>
#include <string>
#include "../../include/mslam.h"

class MslamRuntime {
public:
    // device from $MSLAM_DEVICE (default 0); sized for the largest frame the pipeline will see
    static MslamRuntime& instance();
    mslam_ctx* ctx() { return ctx_; }
    void ensure(int w, int h, int max_kpts);
    static void check(int status, const char* what) {
        if (status != MSLAM_OK) throw std::runtime_error(std::string(what) + ": " + mslam_strerror(status));
    }
    ~MslamRuntime();
private:
    MslamRuntime() = default;
    mslam_ctx* ctx_ = nullptr;
    int w_ = 0, h_ = 0, kp_ = 0;
};
