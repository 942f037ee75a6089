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
// MFMA form (default; MSLAM_MATCH_MFMA=0 selects the kernel above): with every descriptor bit b stored as an int8 value
//   +-v, the dot product of two descriptors is v^2 (#equal bits - #different bits) = v^2 (256 - 2 * Hamming), so the 2000 x 2000
//   distance table of a frame pair is an int8 Gram matrix and v_mfma_i32_32x32x32_i8 produces it exactly.  The tracking
//   path as a whole is bound by VALU issue: this moves the xor/popcount work (16 of the 19 VALU
//   instructions per pair and lane) onto the otherwise idle matrix pipe and leaves the top-2 bookkeeping on the VALU.
//   expand_pm1_k writes the +-1 bytes in MFMA operand order (tile of 32 descriptors x k-step of 32 bits = 64 lanes x 16 B),
//   so every operand load is one lane-linear 1 KiB read; hamming_mfma_k keeps 64 query columns per wavefront in registers
//   (B operands), stages the train tiles through LDS (double-buffered, shared by the workgroup's two wavefronts) and folds
//   each 32 x 32 tile into per-lane (best, second) keys.  The bytes are +-127 with the query side negated and the accumulator
//   starts at 127^2 * 256 + row, so a tile element IS the key 127^2 * 2 * Hamming + row (distance major, train index minor):
//   per element only v_min + v_med3 remain.  Train indices must stay below 127^2 = 16129; larger sets take the VALU form.
#include "common.h"
#include <cstdio>
#include <vector>

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

// ---- MFMA form ------------------------------------------------------------------------------------------------------
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
constexpr int MF_THREADS = 128;          // two wavefronts
constexpr int MF_QPW = 64;               // query columns per wavefront (two 32-wide B blocks)

// E[b][tile][s][lane] (16 bytes): lane = half * 32 + r holds bits [32 s + 16 half, +16) of descriptor 32 tile + r as +-1
// bytes (zero bytes for rows past n: their dot products are never used).
__global__ __launch_bounds__(256)
void expand_pm1_k(const uint8_t* __restrict__ q, const int32_t* __restrict__ nq_arr, const uint8_t* __restrict__ t, const int32_t* __restrict__ nt_arr,
                  int cap, int tiles_cap, int B, uint4* __restrict__ Eq, uint4* __restrict__ Et)
{
    // blockIdx.z < B: query side of frame blockIdx.z, stored negated; otherwise the train side of frame blockIdx.z - B
    const bool train = (int)blockIdx.z >= B;
    const int b = train ? blockIdx.z - B : blockIdx.z, s = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    const int n = train ? nt_arr[b] : nq_arr[b];
    if (i >= ((n + 31) & ~31)) return;
    const uint32_t flip = train ? 0x81818181u : 0x7F7F7F7Fu;   // bit -> 0x7F / 0x81 (+127 / -127), or the negated pair
    uint32_t w = 0;
    const bool live = i < n;
    if (live) w = reinterpret_cast<const uint32_t*>((train ? t : q) + ((size_t)b * cap + i) * 32)[s];
    uint4* dst = (train ? Et : Eq) + (((size_t)b * tiles_cap + (i >> 5)) * 8 + s) * 64 + (i & 31);
#pragma unroll
    for (int half = 0; half < 2; half++) {
        uint32_t o[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t x = (w >> (16 * half + 4 * j)) & 0xFu;
            const uint32_t m = (x * 0x00204081u) & 0x01010101u;          // bit j of the nibble -> bit 0 of byte j
            o[j] = live ? (((m << 8) - (m << 1)) ^ flip) : 0u;            // m * 0xFE per byte, then the flip
        }
        dst[half * 32] = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

constexpr int MF_SCALE = 127 * 127;      // a tile element is MF_SCALE * 2 * Hamming + row: train indices must stay below it
template <bool MASK>
__device__ __forceinline__ void fold_tile(const v16i& acc, const v16i& c, int limit, uint32_t& best, uint32_t& second)
{
#pragma unroll
    for (int i = 0; i < 16; i++) {
        uint32_t key = (uint32_t)acc[i];   // the MFMA's own output is the key: no vector instruction to form it
        if (MASK) key = c[i] < limit ? key : KEY_NONE;
        top2_update(best, second, key);
    }
}

__global__ __launch_bounds__(MF_THREADS, 2)
void hamming_mfma_k(const uint4* __restrict__ Eq, const int32_t* __restrict__ nq_arr, const uint4* __restrict__ Et,
                    const int32_t* __restrict__ nt_arr, int cap, int tiles_cap, int nsplit, uint2* __restrict__ part, unsigned long long* __restrict__ stamps)
{
    // development hook (MSLAM_MATCH_DEBUG=<file>): entry / exit time stamps per wavefront (100 MHz) and where it ran
    const unsigned long long stamp0 = stamps ? __builtin_amdgcn_s_memrealtime() : 0ull;
    // blockIdx.y = which share of the train tiles this workgroup scans: more wavefronts per SIMD, so that one wavefront's
    // top-2 bookkeeping (VALU) runs under another's MFMAs; knn2_finalize merges the nsplit partial keys per query
    // XCD-contiguous work order with the frame slowest: the workgroups that share a frame's train tiles run on one XCD, so
    // the tiles come out of one L2 instead of being fetched from HBM by up to eight
    const unsigned nwg = gridDim.x * gridDim.y * gridDim.z;
    const unsigned work = xcd_contiguous(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), nwg);
    const int bx = (int)(work % gridDim.x), by = (int)((work / gridDim.x) % gridDim.y), b = (int)(work / (gridDim.x * gridDim.y));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nq = nq_arr[b], nt = nt_arr[b];
    const int q0 = (bx * 2 + wave) * MF_QPW;                        // first query column of this wavefront
    if (bx * 2 * MF_QPW >= nq) return;                                // workgroup-uniform (its stamp slots stay zero)
    __shared__ uint4 s_tile[2][8 * 64];                              // two train tiles of 8 KiB
    const int nTall = (nt + 31) >> 5, per = (nTall + nsplit - 1) / nsplit;
    const int t_lo = min(by * per, nTall), nT = min(t_lo + per, nTall) - t_lo;   // tiles [t_lo, t_lo + nT)
    const uint4* et = Et + ((size_t)b * tiles_cap + t_lo) * 512;
    // B operands: this wavefront's two query blocks, 8 k-steps each (columns past nq hold zero bytes or stale tiles; never written out)
    v4i bq[2][8];
    {
        const uint4* eq = Eq + (size_t)b * tiles_cap * 512;
        const int qt = q0 >> 5;
#pragma unroll
        for (int blk = 0; blk < 2; blk++) {
            const int t = min(qt + blk, tiles_cap - 1);
#pragma unroll
            for (int s = 0; s < 8; s++) {
                const uint4 v = eq[((size_t)t * 8 + s) * 64 + lane];
                bq[blk][s] = v4i{ (int)v.x, (int)v.y, (int)v.z, (int)v.w };
            }
        }
    }
    // row (train index within the tile) of accumulator element i of this lane: 8 (i / 4) + 4 (lane / 32) + i % 4
    // The accumulator starts at MF_SCALE * 256 + row and the operand bytes are +-127 (query side negated), so an element
    // comes out as MF_SCALE * (256 - dot) + row = MF_SCALE * 2 * Hamming + row: already the key (distance major, row minor).
    v16i c;
#pragma unroll
    for (int i = 0; i < 16; i++) c[i] = MF_SCALE * 256 + 32 * t_lo + 8 * (i >> 2) + 4 * (lane >> 5) + (i & 3);
    const int limit = MF_SCALE * 256 + nt;
    uint32_t best[2] = { KEY_NONE, KEY_NONE }, second[2] = { KEY_NONE, KEY_NONE };
    if (nT > 0) {
#pragma unroll
        for (int j = 0; j < 4; j++) s_tile[0][(wave * 4 + j) * 64 + lane] = et[(size_t)(wave * 4 + j) * 64 + lane];
    }
    __syncthreads();
    for (int t = 0; t < nT; t++) {
        const bool more = t + 1 < nT;
        // next tile: global -> registers now, registers -> the other LDS buffer after this tile's arithmetic (no branch around
        // either: the last iteration re-reads its own tile into the buffer nobody looks at again)
        const uint4* nx = et + ((size_t)(more ? t + 1 : t) * 8 + wave * 4) * 64 + lane;
        const uint4 p0 = nx[0], p1 = nx[64], p2 = nx[128], p3 = nx[192];
        const uint4* tile = s_tile[t & 1];
        v16i acc0 = c, acc1 = c;
#pragma unroll
        for (int s = 0; s < 8; s++) {
            const uint4 v = tile[s * 64 + lane];
            const v4i a = v4i{ (int)v.x, (int)v.y, (int)v.z, (int)v.w };
            acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bq[0][s], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bq[1][s], acc1, 0, 0, 0);
        }
        if (more) {
            fold_tile<false>(acc0, c, limit, best[0], second[0]);
            fold_tile<false>(acc1, c, limit, best[1], second[1]);
        } else {   // the last tile may hold rows past nt
            fold_tile<true>(acc0, c, limit, best[0], second[0]);
            fold_tile<true>(acc1, c, limit, best[1], second[1]);
        }
        uint4* nb = s_tile[(t + 1) & 1] + wave * 4 * 64 + lane;
        nb[0] = p0; nb[64] = p1; nb[128] = p2; nb[192] = p3;
#pragma unroll
        for (int i = 0; i < 16; i++) c[i] += 32;   // next tile's rows
        __syncthreads();
    }
    // lanes l and l ^ 32 hold the same query column (different rows): merge, then the lower half writes
#pragma unroll
    for (int blk = 0; blk < 2; blk++) {
        const uint32_t ob = (uint32_t)__shfl_xor((int)best[blk], 32), os = (uint32_t)__shfl_xor((int)second[blk], 32);
        top2_update(best[blk], second[blk], ob);
        top2_update(best[blk], second[blk], os);
        const int qi = q0 + blk * 32 + (lane & 31);
        // keys are MF_SCALE * 2 * Hamming + row: back to (Hamming << 16 | row), the form knn2_finalize reads
        const uint32_t hb = best[blk] / (uint32_t)MF_SCALE, hs = second[blk] / (uint32_t)MF_SCALE;
        const uint32_t kb = best[blk] == KEY_NONE ? KEY_NONE : ((hb >> 1) << 16) | (best[blk] - hb * (uint32_t)MF_SCALE);
        const uint32_t ks = second[blk] == KEY_NONE ? KEY_NONE : ((hs >> 1) << 16) | (second[blk] - hs * (uint32_t)MF_SCALE);
        if (lane < 32 && qi < nq) part[((size_t)b * nsplit + by) * cap + qi] = make_uint2(kb, ks);
    }
    if (stamps && lane == 0) {
        unsigned long long* o = stamps + 4 * ((size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * (MF_THREADS / 64) + wave);
        o[0] = stamp0; o[1] = __builtin_amdgcn_s_memrealtime();
        o[2] = 1u | (__builtin_amdgcn_s_getreg(20 | (31 << 11)) << 8);   // XCC_ID
        o[3] = __builtin_amdgcn_s_getreg(4 | (31 << 11));                  // HW_ID (wave, SIMD, CU, SE)
    }
}

__global__ __launch_bounds__(256)
void knn2_finalize(const uint2* __restrict__ part, const int32_t* __restrict__ nq_arr,
                   const int32_t* __restrict__ nt_arr, int cap, int nchunks,
                   int32_t* __restrict__ idx, int32_t* __restrict__ dist, float ratio,
                   int32_t* __restrict__ m_query, int32_t* __restrict__ m_train, float* __restrict__ m_dist,
                   int32_t* __restrict__ m_count, int all_chunks)
{
    const int b = blockIdx.x;
    const int nq = nq_arr[b], nt = nt_arr[b];
    const int used = all_chunks ? nchunks : (nt + TC - 1) / TC;   // all_chunks: every chunk holds keys (KEY_NONE if it saw no rows): hamming_mfma_k
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
            for (int c0 = 0; c0 < used; c0 += 4) {   // four partial results in flight together (one memory latency, not four)
                uint2 p[4];
#pragma unroll
                for (int u = 0; u < 4; u++) p[u] = c0 + u < used ? part[((size_t)b * nchunks + c0 + u) * cap + qi] : make_uint2(KEY_NONE, KEY_NONE);
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    if (c0 + u < used) {
                        top2_update(best, second, p[u].x);
                        if (p[u].y != KEY_NONE) top2_update(best, second, p[u].y);
                    }
                }
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
    static const int use_mfma = [] { const char* e = getenv("MSLAM_MATCH_MFMA"); return e ? atoi(e) : 1; }();
    void* scr = nullptr;
    if (use_mfma && cap < MF_SCALE) {   // (train indices live below MF_SCALE in the keys; larger capacities take the VALU form)
        const int tiles_cap = (cap + 31) / 32;
        static int stamp_calls = 0;
        // (development hook: the buffer lives on the device that was current at the first call and is used by contexts of that device only)
        static int stamps_device = -1;
        static unsigned long long* stamps_buf = [] { unsigned long long* p = nullptr; if (getenv("MSLAM_MATCH_DEBUG") && hipGetDevice(&stamps_device) == hipSuccess && hipMalloc(&p, 8 * 4 * 65536) == hipSuccess) (void)hipMemset(p, 0, 8 * 4 * 65536); return p; }();
        unsigned long long* stamps = stamps_device == ctx->device ? stamps_buf : nullptr;
        static const int nsplit = [] { const char* e = getenv("MSLAM_MATCH_SPLIT"); const int v = e ? atoi(e) : 4; return v < 1 ? 1 : (v > 16 ? 16 : v); }();
        const size_t e_bytes = align_up((size_t)B * tiles_cap * 8192, 256), p_bytes = align_up(sizeof(uint2) * (size_t)B * nsplit * cap, 256);
        MSLAM_TRY(ctx_scratch(ctx, 2 * e_bytes + p_bytes, &scr));
        uint4* Eq = (uint4*)scr; uint4* Et = (uint4*)((char*)scr + e_bytes);
        uint2* part = (uint2*)((char*)scr + 2 * e_bytes);
        LAUNCH(ctx, K_MATCH_EXPAND, expand_pm1_k, dim3((tiles_cap * 32 + 255) / 256, 8, 2 * B), dim3(256), 0, q, nq, t, nt, cap, tiles_cap, B, Eq, Et);
        const size_t nw = (size_t)((cap + 2 * MF_QPW - 1) / (2 * MF_QPW)) * nsplit * B * (MF_THREADS / 64);   // wavefronts of the launch
        unsigned long long* st = nw <= 65536 ? stamps : nullptr;
        LAUNCH(ctx, K_MATCH_MFMA, hamming_mfma_k, dim3((cap + 2 * MF_QPW - 1) / (2 * MF_QPW), nsplit, B), dim3(MF_THREADS), 0, (const uint4*)Eq, nq, (const uint4*)Et, nt,
               cap, tiles_cap, nsplit, part, st);
        if (st && ++stamp_calls == 8) {   // the eighth call of the process: warm caches, steady clocks
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            std::vector<unsigned long long> h(4 * nw);
            HIP_TRY(ctx, hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
            if (FILE* f = std::fopen(getenv("MSLAM_MATCH_DEBUG"), "w")) {
                for (size_t i = 0; i < nw; i++) std::fprintf(f, "%zu %llu %llu %llu %llu\n", i, h[4 * i], h[4 * i + 1], h[4 * i + 2], h[4 * i + 3]);
                std::fclose(f);
            }
        }
        LAUNCH(ctx, K_MATCH_FINALIZE, knn2_finalize, dim3(B), dim3(256), 0, (const uint2*)part, nq, nt, cap, nsplit, idx, dist,
               ratio, m_query, m_train, m_dist, ratio > 0.f ? m_count : (int32_t*)nullptr, 1);
        HIP_TRY(ctx, hipGetLastError());
        return MSLAM_OK;
    }
    const int nchunks = (cap + TC - 1) / TC;
    MSLAM_TRY(ctx_scratch(ctx, sizeof(uint2) * (size_t)B * nchunks * cap, &scr));
    uint2* part = (uint2*)scr;
    dim3 grid((cap + MATCH_THREADS * QPL - 1) / (MATCH_THREADS * QPL), nchunks, B);
    LAUNCH(ctx, K_MATCH_PARTIAL, hamming_knn2_partial, grid, dim3(MATCH_THREADS), 0, q, nq, t, nt, cap, nchunks, part);
    LAUNCH(ctx, K_MATCH_FINALIZE, knn2_finalize, dim3(B), dim3(256), 0, (const uint2*)part, nq, nt, cap, nchunks, idx, dist,
                       ratio, m_query, m_train, m_dist, ratio > 0.f ? m_count : (int32_t*)nullptr, 0);
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
