// match.hip -- K7: brute-force 256-bit Hamming kNN-2 + Lowe ratio test (gfx950).
//
// Stands in for matchFeatures (reference src/CameraPoseEstimator.cpp:200-213):
// BFMatcher(NORM_HAMMING,false).knnMatch(q, t, raw, 2) then `d0 < d1 * ratio`.
//
// Data layout: descriptors are 32-byte rows (8 dwords), row-major, 16-byte aligned.
// Kernel 1 (hamming_knn2_partial): a lane owns QPL query rows in registers; the train
//   set is cut into chunks of TC rows; a workgroup stages one chunk in LDS (coalesced
//   16-B loads) and every lane scans it with broadcast LDS reads, v_xor + v_bcnt (popcount
//   with fused accumulate), keeping (best, second) as packed keys (dist << 16 | train)
//   so that "strict < in scan order" == unsigned min of the key (ties -> lower index).
//   grid = (query tiles, train chunks, frames): >= 2048 workgroups at batch 64.
// Kernel 2 (knn2_finalize): one workgroup per frame merges the per-chunk top-2 in chunk
//   order, writes idx/dist, applies the float ratio test and compacts the survivors in
//   query order with a ballot/prefix scan (deterministic, no atomics).
// No MFMA: there is no dense contraction here, only xor/popcount/min.
#include "common.h"

namespace {

constexpr int TC = 256;        // train rows per chunk (8 KiB of LDS)
constexpr int QPL = 2;         // query rows per lane
constexpr int MATCH_THREADS = 256;
constexpr uint32_t KEY_NONE = 0xFFFFFFFFu;

__device__ __forceinline__ void top2_update(uint32_t& best, uint32_t& second, uint32_t k)
{
    // invariant best <= second: the new second is the median of (best, second, k), the new best the minimum.
    // Written in the min/max form the compiler folds into one v_med3_u32.
    const uint32_t mn = best < second ? best : second, mx = best < second ? second : best;
    const uint32_t t = mn > k ? mn : k;
    second = mx < t ? mx : t;
    best = best < k ? best : k;
}

__global__ __launch_bounds__(MATCH_THREADS)
void hamming_knn2_partial(const uint8_t* __restrict__ q, const int32_t* __restrict__ nq_arr,
                          const uint8_t* __restrict__ t, const int32_t* __restrict__ nt_arr,
                          int cap, int nchunks, uint2* __restrict__ part)
{
    const int b = blockIdx.z;
    const int nq = nq_arr[b], nt = nt_arr[b];
    const int c0 = blockIdx.y * TC;
    const int q0 = blockIdx.x * (MATCH_THREADS * QPL);
    if (c0 >= nt || q0 >= nq) return; // wave-uniform exit
    __shared__ uint4 tile[TC * 2];
    const uint4* tsrc = reinterpret_cast<const uint4*>(t + ((size_t)b * cap + c0) * 32);
    const int cnt = min(TC, nt - c0);
    for (int i = threadIdx.x; i < TC * 2; i += MATCH_THREADS)
        tile[i] = (i < cnt * 2) ? tsrc[i] : make_uint4(0, 0, 0, 0);
    uint32_t qr[QPL][8];
    const uint4* qsrc = reinterpret_cast<const uint4*>(q + (size_t)b * cap * 32);
#pragma unroll
    for (int k = 0; k < QPL; k++) {
        int qi = q0 + k * MATCH_THREADS + threadIdx.x;
        int qc = qi < nq ? qi : nq - 1; // clamp: lanes past the end compute a discarded duplicate
        uint4 a = qsrc[(size_t)qc * 2], c = qsrc[(size_t)qc * 2 + 1];
        qr[k][0] = a.x; qr[k][1] = a.y; qr[k][2] = a.z; qr[k][3] = a.w;
        qr[k][4] = c.x; qr[k][5] = c.y; qr[k][6] = c.z; qr[k][7] = c.w;
    }
    __syncthreads();
    uint32_t best[QPL], second[QPL];
#pragma unroll
    for (int k = 0; k < QPL; k++) { best[k] = KEY_NONE; second[k] = KEY_NONE; }
#pragma unroll 2
    for (int j = 0; j < cnt; j++) {
        uint4 a = tile[2 * j], c = tile[2 * j + 1]; // same address in every lane: LDS broadcast
        const uint32_t tr[8] = { a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w };
        const uint32_t lo = (uint32_t)(c0 + j);
#pragma unroll
        for (int k = 0; k < QPL; k++) {
            uint32_t d = 0;
#pragma unroll
            for (int w = 0; w < 8; w++) d += __popc(qr[k][w] ^ tr[w]);
            top2_update(best[k], second[k], (d << 16) | lo);
        }
    }
    uint2* dst = part + ((size_t)b * nchunks + blockIdx.y) * cap;
#pragma unroll
    for (int k = 0; k < QPL; k++) {
        int qi = q0 + k * MATCH_THREADS + threadIdx.x;
        if (qi < nq) dst[qi] = make_uint2(best[k], second[k]);
    }
}

__global__ __launch_bounds__(256)
void knn2_finalize(const uint2* __restrict__ part, const int32_t* __restrict__ nq_arr,
                   const int32_t* __restrict__ nt_arr, int cap, int nchunks,
                   int32_t* __restrict__ idx, int32_t* __restrict__ dist, float ratio,
                   int32_t* __restrict__ m_query, int32_t* __restrict__ m_train, float* __restrict__ m_dist,
                   int32_t* __restrict__ m_count)
{
    const int b = blockIdx.x;
    const int nq = nq_arr[b], nt = nt_arr[b];
    const int used = (nt + TC - 1) / TC;
    __shared__ int wave_cnt[4];
    __shared__ int base_s;
    if (threadIdx.x == 0) base_s = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int q0 = 0; q0 < nq; q0 += 256) {
        const int qi = q0 + threadIdx.x;
        bool keep = false;
        uint32_t best = KEY_NONE, second = KEY_NONE;
        if (qi < nq && nt >= 2) {
            for (int c = 0; c < used; c++) {
                uint2 p = part[((size_t)b * nchunks + c) * cap + qi];
                top2_update(best, second, p.x);
                if (p.y != KEY_NONE) top2_update(best, second, p.y);
            }
        }
        if (qi < nq) {
            const size_t o = ((size_t)b * cap + qi) * 2;
            if (nt >= 2) {
                idx[o] = (int32_t)(best & 0xFFFFu); idx[o + 1] = (int32_t)(second & 0xFFFFu);
                dist[o] = (int32_t)(best >> 16); dist[o + 1] = (int32_t)(second >> 16);
                // DMatch.distance is float; `d0 < d1 * ratio` evaluated in float
                keep = ratio > 0.f && (float)(best >> 16) < (float)(second >> 16) * ratio;
            } else {
                idx[o] = -1; idx[o + 1] = -1; dist[o] = 0; dist[o + 1] = 0;
            }
        }
        if (m_count) { // block-wide ordered compaction
            unsigned long long bal = __ballot(keep);
            if (lane == 0) wave_cnt[wave] = __popcll(bal);
            __syncthreads();
            int pre = base_s;
            for (int w = 0; w < wave; w++) pre += wave_cnt[w];
            int pos = pre + __popcll(bal & ((1ull << lane) - 1ull));
            if (keep) {
                const size_t o = (size_t)b * cap + pos;
                m_query[o] = qi; m_train[o] = (int32_t)(best & 0xFFFFu); m_dist[o] = (float)(best >> 16);
            }
            __syncthreads();
            if (threadIdx.x == 0) base_s += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
            __syncthreads();
        }
    }
    if (m_count && threadIdx.x == 0) m_count[b] = base_s;
}

} // namespace

extern "C" {

int mslam_match_knn2_batch_dev(mslam_ctx* ctx, const uint8_t* q, const int32_t* nq, const uint8_t* t,
                               const int32_t* nt, int cap, int B, int32_t* idx, int32_t* dist, float ratio,
                               int32_t* m_query, int32_t* m_train, float* m_dist, int32_t* m_count)
{
    if (!ctx || !q || !nq || !t || !nt || !idx || !dist || cap < 1 || cap > 65535 || B < 1) return MSLAM_ERR_INVALID;
    if (ratio > 0.f && (!m_query || !m_train || !m_dist || !m_count)) return MSLAM_ERR_INVALID;
    if (((uintptr_t)q | (uintptr_t)t) & 15) return MSLAM_ERR_INVALID; // 16-B aligned rows
    MSLAM_TRY(ctx_set_device(ctx));
    const int nchunks = (cap + TC - 1) / TC;
    void* scr = nullptr;
    MSLAM_TRY(ctx_scratch(ctx, sizeof(uint2) * (size_t)B * nchunks * cap, &scr));
    uint2* part = (uint2*)scr;
    dim3 grid((cap + MATCH_THREADS * QPL - 1) / (MATCH_THREADS * QPL), nchunks, B);
    LAUNCH(ctx, K_MATCH_PARTIAL, hamming_knn2_partial, grid, dim3(MATCH_THREADS), 0, q, nq, t, nt, cap, nchunks, part);
    LAUNCH(ctx, K_MATCH_FINALIZE, knn2_finalize, dim3(B), dim3(256), 0, part, nq, nt, cap, nchunks, idx, dist,
                       ratio, m_query, m_train, m_dist, ratio > 0.f ? m_count : (int32_t*)nullptr);
    HIP_TRY(ctx, hipGetLastError());
    return MSLAM_OK;
}

int mslam_match_knn2(mslam_ctx* ctx, const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx, int32_t* dist)
{
    if (!ctx || !q || !t || !idx || !dist || nq < 0 || nt < 0 || nq > 65535 || nt > 65535) return MSLAM_ERR_INVALID;
    if (nt < 2) return MSLAM_ERR_TOO_FEW;
    if (nq == 0) return MSLAM_OK;
    MSLAM_TRY(ctx_set_device(ctx));
    const int cap = nq > nt ? nq : nt;
    // staging: q, t, counts, idx, dist
    size_t bytes = align_up((size_t)cap * 32, 256) * 2 + 512 + align_up(sizeof(int32_t) * cap * 2, 256) * 2;
    void* st = nullptr;
    MSLAM_TRY(ctx_stage(ctx, bytes, &st));
    Bump bp{ (char*)st, 0, bytes };
    uint8_t* dq = bp.take<uint8_t>((size_t)cap * 32);
    uint8_t* dt = bp.take<uint8_t>((size_t)cap * 32);
    int32_t* dn = bp.take<int32_t>(2);
    int32_t* didx = bp.take<int32_t>((size_t)cap * 2);
    int32_t* ddist = bp.take<int32_t>((size_t)cap * 2);
    int32_t hn[2] = { nq, nt };
    HIP_TRY(ctx, hipMemcpyAsync(dq, q, (size_t)nq * 32, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(dt, t, (size_t)nt * 32, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(dn, hn, sizeof(hn), hipMemcpyHostToDevice, ctx->stream));
    // the copies above read pageable host memory: make sure they are done before hn leaves scope
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    MSLAM_TRY(mslam_match_knn2_batch_dev(ctx, dq, dn, dt, dn + 1, cap, 1, didx, ddist, 0.f, nullptr, nullptr, nullptr, nullptr));
    HIP_TRY(ctx, hipMemcpyAsync(idx, didx, sizeof(int32_t) * (size_t)nq * 2, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(dist, ddist, sizeof(int32_t) * (size_t)nq * 2, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MSLAM_OK;
}

int mslam_ratio_filter(const int32_t* idx, const int32_t* dist, int nq, float ratio, int32_t* query_idx,
                       int32_t* train_idx, float* d, int* n_out)
{
    if (!idx || !dist || !query_idx || !train_idx || !d || !n_out || nq < 0) return MSLAM_ERR_INVALID;
    int n = 0;
    for (int i = 0; i < nq; i++) {
        float d0 = (float)dist[2 * i], d1 = (float)dist[2 * i + 1];
        if (d0 < d1 * ratio) { query_idx[n] = i; train_idx[n] = idx[2 * i]; d[n] = d0; n++; }
    }
    *n_out = n;
    return MSLAM_OK;
}

} // extern "C"
