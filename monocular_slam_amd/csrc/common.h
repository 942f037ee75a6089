// common.h -- internal definitions of libmslam_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <unordered_map>
#include <vector>
#include "../../include/mslam.h"

#define MSLAM_ABI_VERSION 1000

struct OrbPlan;   // orb.hip

// kernel ids for the optional per-kernel HIP-event timing (mslam_profile_*)
enum KernelId {
    K_GRAY = 0, K_RESIZE, K_FAST_NMS, K_SELECT, K_DESCRIBE, K_MATCH_PARTIAL, K_MATCH_FINALIZE,
    K_FMAT_PREPARE, K_FMAT_SOLVE, K_FMAT_SCORE, K_FMAT_FINALIZE, K_RANSAC_SELECT,
    K_PNP_PREPARE, K_PNP_SOLVE, K_PNP_SCORE, K_PNP_FINALIZE, K_TRIANGULATE,
    K_BA_POINTS, K_BA_POSES, K_BA_FINAL, K_BA_SCHUR, K_BA_SCHUR_TILES, K_BA_SCHUR_FINAL, K_BA_BACKSUB, K_PYRAMID2, K_MATCH_EXPAND, K_MATCH_MFMA, K_PNP_SCORE32, K_FMAT7_SOLVE, K_FMAT7_LMEDS, K_COUNT
};
extern const char* const kKernelNames[K_COUNT];

struct Prof {
    bool on = false;
    std::vector<hipEvent_t> ev;   // pairs: 2*i start, 2*i+1 stop
    std::vector<int> kid;         // kernel id of pair i
    size_t used = 0;
};

struct mslam_ctx {
    int device = -1;
    hipStream_t stream = nullptr;
    int max_w = 0, max_h = 0, max_kpts = 0, batch = 1;
    std::string last_hip_error;
    OrbPlan* orb = nullptr;            // geometry-dependent ORB scratch (lazily built)
    // generic scratch arena, grown on demand (never inside a timed loop once warm)
    void* scratch = nullptr;
    size_t scratch_bytes = 0;
    // second arena for the host-pointer convenience wrappers (staging of inputs/outputs)
    void* stage = nullptr;
    size_t stage_bytes = 0;
    Prof prof;
    // optional replay of captured launch chains (mslam_set_graphs): key = hash of the call's arguments
    bool graphs = false;
    std::unordered_map<uint64_t, hipGraphExec_t> graph_cache;
};

// 1: the chain was replayed from the cache (nothing left to enqueue); 0: capture has begun -- enqueue, then ctx_graph_end;
// -1: graphs are off (enqueue directly).  ctx_graph_end instantiates, caches and launches the captured chain.
int ctx_graph_begin(mslam_ctx* ctx, uint64_t key);
int ctx_graph_end(mslam_ctx* ctx, uint64_t key);
void ctx_graph_clear(mslam_ctx* ctx);
uint64_t ctx_hash(const void* data, size_t n, uint64_t seed);

// returns the pair index or -1 (profiling off / pool exhausted)
int prof_begin(mslam_ctx* ctx, int kid);
void prof_end(mslam_ctx* ctx, int pair);

// launch on the context's stream, bracketed by HIP events when profiling is enabled
#define LAUNCH(ctx, KID, kernel, grid, block, shmem, ...)                                    \
    do {                                                                                     \
        int _pp = prof_begin((ctx), (KID));                                                  \
        hipLaunchKernelGGL(kernel, grid, block, shmem, (ctx)->stream, __VA_ARGS__);          \
        prof_end((ctx), _pp);                                                                \
    } while (0)

// record + return on HIP failure
#define HIP_TRY(ctx, expr)                                                         \
    do {                                                                           \
        hipError_t _e = (expr);                                                    \
        if (_e != hipSuccess) {                                                    \
            (ctx)->last_hip_error = std::string(#expr) + ": " + hipGetErrorString(_e); \
            return MSLAM_ERR_HIP;                                                  \
        }                                                                          \
    } while (0)

#define MSLAM_TRY(expr)            \
    do {                           \
        int _s = (expr);           \
        if (_s != MSLAM_OK) return _s; \
    } while (0)

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// grows (never shrinks) the scratch arena; contents are not preserved
int ctx_scratch(mslam_ctx* ctx, size_t bytes, void** out);
int ctx_stage(mslam_ctx* ctx, size_t bytes, void** out);
int ctx_set_device(mslam_ctx* ctx);

// bump allocator over an arena
struct Bump {
    char* base; size_t off; size_t cap;
    template <typename T> T* take(size_t n) {
        off = align_up(off, 256);
        T* p = reinterpret_cast<T*>(base + off);
        off += n * sizeof(T);
        return p;
    }
};

void orb_plan_destroy(mslam_ctx* ctx);

#ifdef __HIPCC__
// XCD-aware work-item order: the dispatcher deals consecutive workgroups round-robin over the 8 XCDs (each with its own
// L2), so workgroups orig, orig + 8, ... share an L2.  This bijection gives every XCD one CONTIGUOUS range of the
// logical tile order, so tiles that share image rows / cache lines (tile halos, neighbouring keypoint patches) are
// fetched into one L2 instead of up to eight.  Speed only: any placement computes the same result.
__device__ __forceinline__ unsigned xcd_contiguous(unsigned orig, unsigned nwg)
{
    const unsigned q = nwg >> 3, r = nwg & 7u, xcd = orig & 7u;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
}
#endif
