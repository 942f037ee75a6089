// ctx.hip -- context, memory and timing helpers of the C ABI (include/mslam.h).
#include "common.h"
#include <cstdlib>
#include <cstring>

const char* const kKernelNames[K_COUNT] = {
    "bgr2gray_k", "resize_k", "fast_nms_k", "select_k", "describe_k", "hamming_knn2_partial", "knn2_finalize",
    "fmat_prepare_k", "fmat_solve_k", "fmat_score_k", "fmat_finalize_k", "ransac_select_k",
    "pnp_prepare_k", "pnp_solve_k", "pnp_score_k", "pnp_finalize_k", "triangulate_k",
    "ba_points_k", "ba_edges_k", "ba_final_k", "ba_schur_points_k", "ba_schur_tiles_k", "ba_schur_final_k", "ba_backsub_k", "pyramid2_k", "expand_pm1_k", "hamming_mfma_k", "pnp_score32_k", "fmat7_solve_k", "fmat7_lmeds_k" };

int prof_begin(mslam_ctx* ctx, int kid)
{
    Prof& p = ctx->prof;
    if (!p.on || p.used >= p.kid.size()) return -1;
    const int i = (int)p.used++;
    p.kid[i] = kid;
    (void)hipEventRecord(p.ev[2 * i], ctx->stream);
    return i;
}

void prof_end(mslam_ctx* ctx, int pair)
{
    if (pair >= 0) (void)hipEventRecord(ctx->prof.ev[2 * pair + 1], ctx->stream);
}

uint64_t ctx_hash(const void* data, size_t n, uint64_t h)
{
    const unsigned char* p = static_cast<const unsigned char*>(data); // FNV-1a
    for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 1099511628211ull; }
    return h;
}

void ctx_graph_clear(mslam_ctx* ctx)
{
    for (auto& kv : ctx->graph_cache) (void)hipGraphExecDestroy(kv.second);
    ctx->graph_cache.clear();
}

int ctx_graph_begin(mslam_ctx* ctx, uint64_t key)
{
    if (!ctx->graphs || ctx->prof.on) return -1;
    auto it = ctx->graph_cache.find(key);
    if (it != ctx->graph_cache.end()) {
        if (hipGraphLaunch(it->second, ctx->stream) == hipSuccess) return 1;
        (void)hipGetLastError();
        (void)hipGraphExecDestroy(it->second);
        ctx->graph_cache.erase(it);
        return -1;
    }
    if (ctx->graph_cache.size() >= 64) ctx_graph_clear(ctx); // callers cycle through a handful of argument sets
    if (hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) { (void)hipGetLastError(); return -1; }
    return 0;
}

int ctx_graph_end(mslam_ctx* ctx, uint64_t key)
{
    hipGraph_t graph = nullptr;
    HIP_TRY(ctx, hipStreamEndCapture(ctx->stream, &graph));
    hipGraphExec_t exec = nullptr;
    hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    HIP_TRY(ctx, e);
    ctx->graph_cache[key] = exec;
    HIP_TRY(ctx, hipGraphLaunch(exec, ctx->stream));
    return MSLAM_OK;
}

extern "C" {

int mslam_abi_version(void) { return MSLAM_ABI_VERSION; }

const char* mslam_strerror(int s)
{
    switch (s) {
        case MSLAM_OK: return "ok";
        case MSLAM_ERR_INVALID: return "invalid argument";
        case MSLAM_ERR_HIP: return "HIP runtime error";
        case MSLAM_ERR_CAPACITY: return "result exceeds the given capacity";
        case MSLAM_ERR_TOO_FEW: return "too few rows/points";
        case MSLAM_ERR_NO_DEVICE: return "no usable gfx950 device";
        case MSLAM_ERR_DEGENERATE: return "no valid model";
        default: return "unknown status";
    }
}

int mslam_create(mslam_ctx** out, int device, int max_w, int max_h, int max_kpts, int batch)
{
    if (!out || max_w < 16 || max_h < 16 || max_w > 4095 || max_h > 4095 || max_kpts < 8 || max_kpts > 65535 || batch < 1)
        return MSLAM_ERR_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev)
        return MSLAM_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return MSLAM_ERR_NO_DEVICE;
    // this library carries gfx950 code objects only
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return MSLAM_ERR_NO_DEVICE;
    mslam_ctx* ctx = new mslam_ctx();
    ctx->device = device;
    ctx->max_w = max_w; ctx->max_h = max_h; ctx->max_kpts = max_kpts; ctx->batch = batch;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return MSLAM_ERR_HIP;
    }
    if (const char* g = std::getenv("MSLAM_GRAPHS")) ctx->graphs = std::atoi(g) != 0;
    *out = ctx;
    return MSLAM_OK;
}

int mslam_set_graphs(mslam_ctx* ctx, int enable)
{
    if (!ctx) return MSLAM_ERR_INVALID;
    MSLAM_TRY(ctx_set_device(ctx));
    if (!enable) { HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); ctx_graph_clear(ctx); }
    ctx->graphs = enable != 0;
    return MSLAM_OK;
}

void mslam_destroy(mslam_ctx* ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    ctx_graph_clear(ctx);
    orb_plan_destroy(ctx);
    for (hipEvent_t e : ctx->prof.ev) (void)hipEventDestroy(e);
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    if (ctx->stage) (void)hipFree(ctx->stage);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int mslam_sync(mslam_ctx* ctx)
{
    if (!ctx) return MSLAM_ERR_INVALID;
    MSLAM_TRY(ctx_set_device(ctx));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MSLAM_OK;
}

void* mslam_stream(mslam_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

const char* mslam_last_hip_error(mslam_ctx* ctx) { return ctx ? ctx->last_hip_error.c_str() : ""; }

int mslam_dev_alloc(mslam_ctx* ctx, void** dptr, uint64_t bytes)
{
    if (!ctx || !dptr) return MSLAM_ERR_INVALID;
    MSLAM_TRY(ctx_set_device(ctx));
    HIP_TRY(ctx, hipMalloc(dptr, bytes ? bytes : 1));
    return MSLAM_OK;
}

int mslam_dev_free(mslam_ctx* ctx, void* dptr)
{
    if (!ctx) return MSLAM_ERR_INVALID;
    MSLAM_TRY(ctx_set_device(ctx));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipFree(dptr));
    return MSLAM_OK;
}

int mslam_memcpy_h2d(mslam_ctx* ctx, void* dst, const void* src, uint64_t bytes)
{
    if (!ctx || (bytes && (!dst || !src))) return MSLAM_ERR_INVALID;
    MSLAM_TRY(ctx_set_device(ctx));
    HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MSLAM_OK;
}

int mslam_memcpy_h2d_async(mslam_ctx* ctx, void* dst, const void* src, uint64_t bytes)
{
    if (!ctx || (bytes && (!dst || !src))) return MSLAM_ERR_INVALID;
    MSLAM_TRY(ctx_set_device(ctx));
    HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return MSLAM_OK;
}

int mslam_memcpy_d2h_async(mslam_ctx* ctx, void* dst, const void* src, uint64_t bytes)
{
    if (!ctx || (bytes && (!dst || !src))) return MSLAM_ERR_INVALID;
    MSLAM_TRY(ctx_set_device(ctx));
    HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    return MSLAM_OK;
}

int mslam_memcpy_d2h_2d_async(mslam_ctx* ctx, void* dst, uint64_t dst_pitch, const void* src, uint64_t src_pitch, uint64_t width, uint64_t rows)
{
    if (!ctx || width > dst_pitch || width > src_pitch || (width && rows && (!dst || !src))) return MSLAM_ERR_INVALID;
    if (!width || !rows) return MSLAM_OK;
    MSLAM_TRY(ctx_set_device(ctx));
    HIP_TRY(ctx, hipMemcpy2DAsync(dst, dst_pitch, src, src_pitch, width, rows, hipMemcpyDeviceToHost, ctx->stream));
    return MSLAM_OK;
}

int mslam_memcpy_d2d_async(mslam_ctx* ctx, void* dst, const void* src, uint64_t bytes)
{
    if (!ctx || (bytes && (!dst || !src))) return MSLAM_ERR_INVALID;
    MSLAM_TRY(ctx_set_device(ctx));
    HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return MSLAM_OK;
}

int mslam_host_alloc(mslam_ctx* ctx, void** hptr, uint64_t bytes)
{
    if (!ctx || !hptr) return MSLAM_ERR_INVALID;
    MSLAM_TRY(ctx_set_device(ctx));
    HIP_TRY(ctx, hipHostMalloc(hptr, bytes ? bytes : 1, hipHostMallocDefault));
    return MSLAM_OK;
}

int mslam_host_free(mslam_ctx* ctx, void* hptr)
{
    if (!ctx) return MSLAM_ERR_INVALID;
    MSLAM_TRY(ctx_set_device(ctx));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipHostFree(hptr));
    return MSLAM_OK;
}

int mslam_stream_wait_event(mslam_ctx* ctx, void* ev)
{
    if (!ctx || !ev) return MSLAM_ERR_INVALID;
    MSLAM_TRY(ctx_set_device(ctx));
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, (hipEvent_t)ev, 0));
    return MSLAM_OK;
}

int mslam_memcpy_d2h(mslam_ctx* ctx, void* dst, const void* src, uint64_t bytes)
{
    if (!ctx || (bytes && (!dst || !src))) return MSLAM_ERR_INVALID;
    MSLAM_TRY(ctx_set_device(ctx));
    HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MSLAM_OK;
}

int mslam_event_create(mslam_ctx* ctx, void** ev)
{
    if (!ctx || !ev) return MSLAM_ERR_INVALID;
    MSLAM_TRY(ctx_set_device(ctx));
    hipEvent_t e;
    HIP_TRY(ctx, hipEventCreate(&e));
    *ev = (void*)e;
    return MSLAM_OK;
}

int mslam_event_destroy(mslam_ctx* ctx, void* ev)
{
    if (!ctx || !ev) return MSLAM_ERR_INVALID;
    HIP_TRY(ctx, hipEventDestroy((hipEvent_t)ev));
    return MSLAM_OK;
}

int mslam_event_record(mslam_ctx* ctx, void* ev)
{
    if (!ctx || !ev) return MSLAM_ERR_INVALID;
    MSLAM_TRY(ctx_set_device(ctx));
    HIP_TRY(ctx, hipEventRecord((hipEvent_t)ev, ctx->stream));
    return MSLAM_OK;
}

int mslam_event_elapsed_ms(mslam_ctx* ctx, void* a, void* b, float* ms)
{
    if (!ctx || !a || !b || !ms) return MSLAM_ERR_INVALID;
    HIP_TRY(ctx, hipEventSynchronize((hipEvent_t)b));
    HIP_TRY(ctx, hipEventElapsedTime(ms, (hipEvent_t)a, (hipEvent_t)b));
    return MSLAM_OK;
}

// ---- per-kernel timing with HIP events on the context's stream
int mslam_profile_enable(mslam_ctx* ctx, int max_launches)
{
    if (!ctx || max_launches < 0) return MSLAM_ERR_INVALID;
    MSLAM_TRY(ctx_set_device(ctx));
    Prof& p = ctx->prof;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    while (p.ev.size() < 2 * (size_t)max_launches) {
        hipEvent_t e;
        HIP_TRY(ctx, hipEventCreate(&e));
        p.ev.push_back(e);
    }
    p.kid.assign(p.ev.size() / 2, -1);
    p.used = 0;
    p.on = max_launches > 0;
    return MSLAM_OK;
}

int mslam_profile_kernel_count(void) { return K_COUNT; }
const char* mslam_profile_kernel_name(int kid) { return (kid >= 0 && kid < K_COUNT) ? kKernelNames[kid] : ""; }

// total milliseconds and launch count per kernel id since the last enable (syncs the stream)
int mslam_profile_read(mslam_ctx* ctx, double* total_ms /*K_COUNT*/, int32_t* launches /*K_COUNT*/)
{
    if (!ctx || !total_ms || !launches) return MSLAM_ERR_INVALID;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < K_COUNT; k++) { total_ms[k] = 0.0; launches[k] = 0; }
    Prof& p = ctx->prof;
    for (size_t i = 0; i < p.used; i++) {
        float ms = 0.f;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, p.ev[2 * i], p.ev[2 * i + 1]));
        total_ms[p.kid[i]] += ms;
        launches[p.kid[i]]++;
    }
    return MSLAM_OK;
}

int mslam_argmax_counts(const int32_t* counts, int n, int* best, int* best_count)
{
    if (!counts || n <= 0 || !best) return MSLAM_ERR_INVALID;
    int b = 0;
    for (int i = 1; i < n; i++)
        if (counts[i] > counts[b]) b = i; // strict: lowest index wins ties
    *best = b;
    if (best_count) *best_count = counts[b];
    return MSLAM_OK;
}

} // extern "C"

int ctx_set_device(mslam_ctx* ctx)
{
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return MSLAM_OK;
}

static int grow(mslam_ctx* ctx, void** buf, size_t* have, size_t bytes)
{
    if (bytes <= *have) return MSLAM_OK;
    MSLAM_TRY(ctx_set_device(ctx));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (*buf) HIP_TRY(ctx, hipFree(*buf));
    *buf = nullptr; *have = 0;
    size_t want = align_up(bytes + bytes / 4, 1 << 20);
    HIP_TRY(ctx, hipMalloc(buf, want));
    *have = want;
    return MSLAM_OK;
}

int ctx_scratch(mslam_ctx* ctx, size_t bytes, void** out)
{
    MSLAM_TRY(grow(ctx, &ctx->scratch, &ctx->scratch_bytes, bytes));
    *out = ctx->scratch;
    return MSLAM_OK;
}

int ctx_stage(mslam_ctx* ctx, size_t bytes, void** out)
{
    MSLAM_TRY(grow(ctx, &ctx->stage, &ctx->stage_bytes, bytes));
    *out = ctx->stage;
    return MSLAM_OK;
}
