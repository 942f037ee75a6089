#include "Runtime.h"
#include <cstdlib>

bool MslamRuntime::alive_ = false;

MslamRuntime& MslamRuntime::instance() { static MslamRuntime r; alive_ = true; return r; }

void MslamRuntime::release()
{
    if (!ctx_) return;
    for (void* p : owned_) (void)mslam_dev_free(ctx_, p);
    owned_.clear(); pools_.clear();
    for (int i = 0; i < 4; i++) { if (scratch_[i]) (void)mslam_dev_free(ctx_, scratch_[i]); scratch_[i] = nullptr; scratch_bytes_[i] = 0; }
    mslam_destroy(ctx_);
    ctx_ = nullptr;
    gen_++;   // twins handed out before this point are dead
}

void MslamRuntime::ensure(int w, int h, int max_kpts)
{
    if (ctx_ && w <= w_ && h <= h_ && max_kpts <= kp_) return;
    release();
    w_ = w > w_ ? w : w_; h_ = h > h_ ? h : h_; kp_ = max_kpts > kp_ ? max_kpts : kp_;
    const char* dev = std::getenv("MSLAM_DEVICE");
    // no fallback: without a gfx950 device this throws and the pipeline stops
    check(mslam_create(&ctx_, dev ? std::atoi(dev) : 0, w_, h_, kp_, 1), "mslam_create");
}

void* MslamRuntime::scratch(int slot, size_t bytes)
{
    if (bytes > scratch_bytes_[slot]) {
        if (scratch_[slot]) check(mslam_dev_free(ctx_, scratch_[slot]), "mslam_dev_free");
        scratch_[slot] = nullptr; scratch_bytes_[slot] = 0;
        const size_t want = bytes + bytes / 4 + 4096;
        check(mslam_dev_alloc(ctx_, &scratch_[slot], want), "mslam_dev_alloc");
        scratch_bytes_[slot] = want;
    }
    return scratch_[slot];
}

std::shared_ptr<DeviceFeatures> MslamRuntime::newDeviceFeatures(int cap)
{
    Pool* pool = nullptr;
    for (Pool& p : pools_) if (p.cap == cap) pool = &p;
    if (!pool) { pools_.push_back(Pool{ cap, {} }); pool = &pools_.back(); }
    void* buf = nullptr;
    if (!pool->free.empty()) { buf = pool->free.back(); pool->free.pop_back(); }
    else {
        check(mslam_dev_alloc(ctx_, &buf, (size_t)cap * 32 + 256), "mslam_dev_alloc");
        owned_.push_back(buf);
    }
    DeviceFeatures* d = new DeviceFeatures();
    d->gen = gen_; d->cap = cap; d->desc = (uint8_t*)buf; d->n_dev = (int32_t*)((uint8_t*)buf + (size_t)cap * 32);
    return std::shared_ptr<DeviceFeatures>(d, [](DeviceFeatures* x) {
        // back to the pool of the context it came from, if that context still exists (the runtime frees the rest itself)
        if (MslamRuntime::alive_) {
            MslamRuntime& r = MslamRuntime::instance();
            if (r.ctx_ && x->gen == r.gen_)
                for (Pool& p : r.pools_) if (p.cap == x->cap) { p.free.push_back(x->desc); break; }
        }
        delete x;
    });
}

DeviceFeatures* MslamRuntime::deviceFeatures(const std::shared_ptr<void>& handle) const
{
    DeviceFeatures* d = static_cast<DeviceFeatures*>(handle.get());
    return (d && ctx_ && d->gen == gen_) ? d : nullptr;
}

MslamRuntime::~MslamRuntime() { release(); alive_ = false; }
