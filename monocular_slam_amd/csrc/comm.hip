// comm.hip -- the one collective of this path, at the C ABI: an int32 all-gather over RCCL (xGMI inside a node).
//
// The tracking path shards by independent streams (no data-path collective).  The only exchange is SURVEY.md 8e's
// "secondary" split: the H hypotheses of ONE RANSAC problem (the solvePnPRansac call site, reference
// src/CameraPoseEstimator.cpp:472) are scored H / nranks per GPU and the int32 inlier counts are all-gathered, after
// which every rank takes the same argmax (mslam_argmax_counts).  Messages are <= 16 KiB: latency-bound, one RCCL
// call per RANSAC, enqueued on the context's own stream so that it orders behind the scoring kernels without a host sync.
//
// RCCL is bound lazily (dlopen of librccl.so.1 on the first communicator): the single-GPU path never needs it and
// must not fail to load where RCCL is absent.  Two forms:
//   * one process per GPU (the deployment form): rank 0 calls mslam_comm_unique_id, the host side hands the 128 bytes
//     to the other ranks by whatever channel it has (file, socket, MPI, torch.distributed store), every rank calls
//     mslam_comm_create(ctx, id, rank, nranks).
//   * one process, several GPUs: mslam_allgather_i32(ctxs, ngpu, ...) -- SURVEY.md 8b's literal signature; communicators
//     are created once per distinct context list (ncclCommInitAll) and cached.
#include "common.h"
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

namespace {

struct Rccl {
    void* h = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string err, path;
};

Rccl g_rccl;
std::mutex g_rccl_mu;

bool rccl_load()
{
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.h) return true;
    // The RCCL that belongs to THIS library's HIP runtime: the librccl.so.1 that sits beside the libamdhip64 this library
    // is bound to, by absolute path.  A process may hold a second ROCm stack (a PyTorch wheel bundles its own libamdhip64 and
    // librccl under the same SONAMEs); binding by name would pick whichever was loaded first, and an RCCL running on the other
    // HIP runtime cannot use this runtime's streams or device pointers.  RTLD_NOW | RTLD_DEEPBIND: its own symbols first, bound now.
    // MSLAM_RCCL_PATH replaces the candidate list by one explicit path (deployments with RCCL elsewhere; the loader's failure test).
    std::vector<std::string> names;
    const char* forced = std::getenv("MSLAM_RCCL_PATH");
    if (forced && *forced) {
        names.push_back(forced);
    } else {
        Dl_info info;
        if (dladdr(reinterpret_cast<void*>(&hipGetDeviceCount), &info) && info.dli_fname) {
            std::string dir(info.dli_fname);
            const size_t slash = dir.rfind('/');
            if (slash != std::string::npos) names.push_back(dir.substr(0, slash) + "/librccl.so.1");
        }
        names.push_back("/opt/rocm/lib/librccl.so.1");
        names.push_back("librccl.so.1");
    }
    void* h = nullptr;
    std::string why;
    for (const std::string& n : names) {
        h = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL | RTLD_DEEPBIND);
        if (h) { g_rccl.path = n; break; }
        const char* e = dlerror();   // one call: dlerror() clears the state it returns
        why += std::string(why.empty() ? "" : "; ") + (e ? e : (n + ": not found"));
    }
    if (!h) { g_rccl.err = "dlopen librccl: " + why; return false; }
#define SYM(field, name) \
    *reinterpret_cast<void**>(&g_rccl.field) = dlsym(h, name); \
    if (!g_rccl.field) { g_rccl.err = std::string("dlsym ") + name; dlclose(h); return false; }
    SYM(GetVersion, "ncclGetVersion") SYM(GetUniqueId, "ncclGetUniqueId") SYM(CommInitRank, "ncclCommInitRank")
    SYM(CommInitAll, "ncclCommInitAll") SYM(CommDestroy, "ncclCommDestroy") SYM(AllGather, "ncclAllGather")
    SYM(GroupStart, "ncclGroupStart") SYM(GroupEnd, "ncclGroupEnd") SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
    g_rccl.h = h;
    return true;
}

} // namespace

struct mslam_comm {
    mslam_ctx* ctx = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, nranks = 1;
    int32_t* stage = nullptr;   // device staging for the host-pointer form: [n] send + [n * nranks] recv
    size_t stage_elems = 0;
};

#define RCCL_TRY(ctx, expr)                                                                                   \
    do {                                                                                                      \
        ncclResult_t _r = (expr);                                                                             \
        if (_r != ncclSuccess) {                                                                              \
            (ctx)->last_hip_error = std::string(#expr) + ": " + g_rccl.GetErrorString(_r);                    \
            return MSLAM_ERR_HIP;                                                                             \
        }                                                                                                     \
    } while (0)

// single-process multi-GPU communicators, keyed by the device list
// (process-lifetime: the communicators are destroyed by mslam_comm_shutdown or left to process exit; g_all_mu serialises
// the whole single-process call, so two host threads cannot race on the cache or interleave their group calls)
struct AllComm { std::vector<int> devs; std::vector<ncclComm_t> comms; };
static std::vector<AllComm> g_all;
static std::mutex g_all_mu;

extern "C" {

int mslam_comm_preload(char* path_out, int cap)
{
    if (!rccl_load()) return MSLAM_ERR_NO_DEVICE;
    if (path_out && cap > 0) { std::strncpy(path_out, g_rccl.path.c_str(), (size_t)cap - 1); path_out[cap - 1] = 0; }
    return MSLAM_OK;
}

int mslam_comm_unique_id(uint8_t id[MSLAM_COMM_ID_BYTES])
{
    if (!id) return MSLAM_ERR_INVALID;
    static_assert(MSLAM_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    if (!rccl_load()) return MSLAM_ERR_NO_DEVICE;
    ncclUniqueId u;
    if (g_rccl.GetUniqueId(&u) != ncclSuccess) return MSLAM_ERR_HIP;
    std::memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
    return MSLAM_OK;
}

int mslam_comm_create(mslam_ctx* ctx, const uint8_t id[MSLAM_COMM_ID_BYTES], int rank, int nranks, mslam_comm** out)
{
    if (!ctx || !id || !out || nranks < 1 || rank < 0 || rank >= nranks) return MSLAM_ERR_INVALID;
    if (!rccl_load()) { ctx->last_hip_error = g_rccl.err; return MSLAM_ERR_NO_DEVICE; }
    MSLAM_TRY(ctx_set_device(ctx));
    ncclUniqueId u;
    std::memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
    mslam_comm* c = new mslam_comm();
    c->ctx = ctx; c->rank = rank; c->nranks = nranks;
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, nranks, u, rank);
    if (r != ncclSuccess) {
        ctx->last_hip_error = std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r);
        delete c;
        return MSLAM_ERR_HIP;
    }
    *out = c;
    return MSLAM_OK;
}

int mslam_comm_info(mslam_comm* c, int* rank, int* nranks, int* rccl_version)
{
    if (!c) return MSLAM_ERR_INVALID;
    if (rank) *rank = c->rank;
    if (nranks) *nranks = c->nranks;
    if (rccl_version) { int v = 0; (void)g_rccl.GetVersion(&v); *rccl_version = v; }
    return MSLAM_OK;
}

int mslam_comm_allgather_i32_dev(mslam_comm* c, const int32_t* send_dev, int n, int32_t* recv_dev)
{
    if (!c || !send_dev || !recv_dev || n < 1) return MSLAM_ERR_INVALID;
    MSLAM_TRY(ctx_set_device(c->ctx));
    RCCL_TRY(c->ctx, g_rccl.AllGather(send_dev, recv_dev, (size_t)n, ncclInt32, c->comm, c->ctx->stream));
    return MSLAM_OK;
}

int mslam_comm_allgather_i32(mslam_comm* c, const int32_t* send_host, int n, int32_t* recv_host)
{
    if (!c || !send_host || !recv_host || n < 1) return MSLAM_ERR_INVALID;
    mslam_ctx* ctx = c->ctx;
    MSLAM_TRY(ctx_set_device(ctx));
    const size_t need = (size_t)n * (size_t)(c->nranks + 1);
    if (need > c->stage_elems) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (c->stage) HIP_TRY(ctx, hipFree(c->stage));
        c->stage = nullptr; c->stage_elems = 0;
        HIP_TRY(ctx, hipMalloc(&c->stage, need * sizeof(int32_t)));
        c->stage_elems = need;
    }
    HIP_TRY(ctx, hipMemcpyAsync(c->stage, send_host, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    MSLAM_TRY(mslam_comm_allgather_i32_dev(c, c->stage, n, c->stage + n));
    HIP_TRY(ctx, hipMemcpyAsync(recv_host, c->stage + n, (size_t)n * 4 * c->nranks, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MSLAM_OK;
}

int mslam_comm_destroy(mslam_comm* c)
{
    if (!c) return MSLAM_ERR_INVALID;
    (void)hipSetDevice(c->ctx->device);
    (void)hipStreamSynchronize(c->ctx->stream);
    if (c->comm) (void)g_rccl.CommDestroy(c->comm);
    if (c->stage) (void)hipFree(c->stage);
    delete c;
    return MSLAM_OK;
}

// One process driving ngpu contexts (one per device): rank g contributes send + g*n (host), every rank's gathered vector
// is identical, so one copy of it (ngpu * n) is returned in recv.
int mslam_allgather_i32(mslam_ctx* const* ctxs, int ngpu, const int32_t* send, int n, int32_t* recv)
{
    if (!ctxs || ngpu < 1 || !send || !recv || n < 1) return MSLAM_ERR_INVALID;
    for (int g = 0; g < ngpu; g++) if (!ctxs[g]) return MSLAM_ERR_INVALID;
    mslam_ctx* c0 = ctxs[0];
    if (!rccl_load()) { c0->last_hip_error = g_rccl.err; return MSLAM_ERR_NO_DEVICE; }
    std::vector<int> devs(ngpu);
    for (int g = 0; g < ngpu; g++) devs[g] = ctxs[g]->device;
    for (int a = 0; a < ngpu; a++) for (int b = a + 1; b < ngpu; b++) if (devs[a] == devs[b]) return MSLAM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(g_all_mu);
    AllComm* ac = nullptr;
    for (auto& e : g_all) if (e.devs == devs) ac = &e;
    if (!ac) {
        AllComm e; e.devs = devs; e.comms.resize(ngpu);
        RCCL_TRY(c0, g_rccl.CommInitAll(e.comms.data(), ngpu, devs.data()));
        g_all.push_back(e);
        ac = &g_all.back();
    }
    // after a failure every stream that may already hold work of this call is drained before the caller sees the error
    auto drain = [&](int upto) { for (int g = 0; g < upto; g++) { (void)hipSetDevice(devs[g]); (void)hipStreamSynchronize(ctxs[g]->stream); } };
    std::vector<int32_t*> bufs(ngpu, nullptr);
    int status = MSLAM_OK;
    for (int g = 0; g < ngpu && status == MSLAM_OK; g++) {
        void* p = nullptr;
        status = ctx_stage(ctxs[g], (size_t)n * 4 * (ngpu + 1), &p);   // sets the device
        bufs[g] = (int32_t*)p;
        if (status == MSLAM_OK && hipMemcpyAsync(bufs[g], send + (size_t)g * n, (size_t)n * 4, hipMemcpyHostToDevice, ctxs[g]->stream) != hipSuccess)
            status = MSLAM_ERR_HIP;
    }
    if (status != MSLAM_OK) { drain(ngpu); return status; }
    ncclResult_t r = g_rccl.GroupStart();
    if (r != ncclSuccess) { drain(ngpu); c0->last_hip_error = std::string("ncclGroupStart: ") + g_rccl.GetErrorString(r); return MSLAM_ERR_HIP; }
    for (int g = 0; g < ngpu && r == ncclSuccess; g++) {
        (void)hipSetDevice(devs[g]);
        r = g_rccl.AllGather(bufs[g], bufs[g] + n, (size_t)n, ncclInt32, ac->comms[g], ctxs[g]->stream);
    }
    const ncclResult_t re = g_rccl.GroupEnd();   // always closes the group, also after a failed enqueue
    if (r == ncclSuccess) r = re;
    if (r != ncclSuccess) { drain(ngpu); c0->last_hip_error = std::string("ncclAllGather: ") + g_rccl.GetErrorString(r); return MSLAM_ERR_HIP; }
    (void)hipSetDevice(devs[0]);
    hipError_t he = hipMemcpyAsync(recv, bufs[0] + n, (size_t)n * 4 * ngpu, hipMemcpyDeviceToHost, ctxs[0]->stream);
    for (int g = 0; g < ngpu; g++) {
        (void)hipSetDevice(devs[g]);
        const hipError_t hs = hipStreamSynchronize(ctxs[g]->stream);
        if (he == hipSuccess) he = hs;
    }
    if (he != hipSuccess) { c0->last_hip_error = std::string("mslam_allgather_i32: ") + hipGetErrorString(he); return MSLAM_ERR_HIP; }
    return MSLAM_OK;
}

// Destroys the cached single-process communicators (call after the contexts' streams are idle and before the contexts are
// destroyed; optional: without it they live until process exit).
int mslam_comm_shutdown(void)
{
    std::lock_guard<std::mutex> lk(g_all_mu);
    for (auto& e : g_all)
        for (size_t g = 0; g < e.comms.size(); g++) { (void)hipSetDevice(e.devs[g]); if (e.comms[g]) (void)g_rccl.CommDestroy(e.comms[g]); }
    g_all.clear();
    return MSLAM_OK;
}

} // extern "C"
