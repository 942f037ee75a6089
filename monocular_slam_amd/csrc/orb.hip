// orb.hip -- K1-K6: ORB keypoint detection + 256-bit rBRIEF description (gfx950).
//
// Stands in for cv::ORB as called by FeatureExtractor::process (reference
// src/FeatureExtractor.cpp:17,19): detect() then compute() on the same frame.  The
// reference builds the pyramid twice (once per call); here it is built once and both
// passes read it.
//
// HBM layout (per context, batch B frames, level l of size lw x lh):
//   level[l]   u8  [B][lh][pitch_l]   pitch_l = align64(lw); level 0 aliases the caller's
//                                      gray frame when it is 1-channel and 4-byte aligned
//   cand[l]    u32 [B][cap_l]         FAST corners after NMS: score<<24 | y<<12 | x
//   candA[l]   u32 [B][cap_l]         survivors of retainBest(2n) + respA f32 (Harris)
//   cnt        u32 [B][L]             candidate counts (the FAST score histogram is rebuilt in LDS by select_k)
//   sel        u32x2 [B][L][max_kpts] final per-level keypoints (key, response), row-major
//   cap_l = ceil(roi_w/2)*ceil(roi_h/2): NMS survivors are never 8-adjacent, so the
//   candidate list cannot overflow.
// Kernels (one launch each per batch unless noted):
//   bgr2gray_k      BGR[A] -> gray, fixed point (only for 3/4-channel input)
//   resize_k        level l-1 -> l, 11-bit fixed-point bilinear (serial dependency between levels)
//   pyramid2_k      levels l and l+1 from level l-1 in one launch (level l stays in LDS for the second resize); used for
//                   the pairs (2,3), (4,5), (6,7): 4 pyramid launches instead of 7, levels 2/4/6 are never read back
//   fast_nms_k      all levels x frames: 128x64 score tiles (124x62 outputs), pixels staged in LDS by LDS-DMA with a halo;
//                   3-phase (high-speed reject -> dense survivor list -> exact score) so that
//                   lanes stay busy; 3x3 NMS on an LDS score tile; wave-aggregated append
//   select_k        one workgroup per (frame, level): histogram threshold (retainBest by
//                   FAST score), Harris response, radix-select of the n-th best response,
//                   rank-sort of the survivors by (y, x)
//   describe_k      one wavefront per keypoint: 43x43 patch in LDS by LDS-DMA, intensity-centroid
//                   angle (v_dot4 over the disc's rows + wave reduce), 7x7 Gaussian as a row pass over the
//                   patch and a column pass evaluated at the 512 sample points only (the blurred pyramid
//                   is never materialised), 256 rotated tests, 4 ballots = 32 descriptor bytes
// All integer/float arithmetic is the plain-IEEE evaluation documented in DESIGN.md (built
// with -ffp-contract=off); cvRound = round-half-even (__float2int_rn / rint).
#include "common.h"
#include "sincos_pio4.h"
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace {

__constant__ __attribute__((aligned(16))) int8_t c_pattern[256 * 4] = {   // (read as one dword per test by describe_k)
#include "orb_pattern_31.inc"
};
// u_max of IC_Angle for half patch 15 (OpenCV computeKeyPoints)
// half widths of the circular patch rows (ORB's umax for patch size 31): { 15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3 },
// baked into the weight bytes below (tools/gen_pattern.py style: generated, not edited by hand)
// IC_Angle weights per (|v|, half of the row): 4 dwords of (u + 15) bytes for the columns inside the circular patch (0 outside),
// then 4 dwords of 0/1 bytes (inside / outside); half 0 = columns -15..0, half 1 = columns 1..16
__constant__ uint32_t c_icw[32][8] = {
    { 0x03020100u, 0x07060504u, 0x0b0a0908u, 0x0f0e0d0cu, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u },
    { 0x13121110u, 0x17161514u, 0x1b1a1918u, 0x001e1d1cu, 0x01010101u, 0x01010101u, 0x01010101u, 0x00010101u },
    { 0x03020100u, 0x07060504u, 0x0b0a0908u, 0x0f0e0d0cu, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u },
    { 0x13121110u, 0x17161514u, 0x1b1a1918u, 0x001e1d1cu, 0x01010101u, 0x01010101u, 0x01010101u, 0x00010101u },
    { 0x03020100u, 0x07060504u, 0x0b0a0908u, 0x0f0e0d0cu, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u },
    { 0x13121110u, 0x17161514u, 0x1b1a1918u, 0x001e1d1cu, 0x01010101u, 0x01010101u, 0x01010101u, 0x00010101u },
    { 0x03020100u, 0x07060504u, 0x0b0a0908u, 0x0f0e0d0cu, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u },
    { 0x13121110u, 0x17161514u, 0x1b1a1918u, 0x001e1d1cu, 0x01010101u, 0x01010101u, 0x01010101u, 0x00010101u },
    { 0x03020100u, 0x07060504u, 0x0b0a0908u, 0x0f0e0d0cu, 0x01010100u, 0x01010101u, 0x01010101u, 0x01010101u },
    { 0x13121110u, 0x17161514u, 0x1b1a1918u, 0x00001d1cu, 0x01010101u, 0x01010101u, 0x01010101u, 0x00000101u },
    { 0x03020100u, 0x07060504u, 0x0b0a0908u, 0x0f0e0d0cu, 0x01010100u, 0x01010101u, 0x01010101u, 0x01010101u },
    { 0x13121110u, 0x17161514u, 0x1b1a1918u, 0x00001d1cu, 0x01010101u, 0x01010101u, 0x01010101u, 0x00000101u },
    { 0x03020100u, 0x07060504u, 0x0b0a0908u, 0x0f0e0d0cu, 0x01010100u, 0x01010101u, 0x01010101u, 0x01010101u },
    { 0x13121110u, 0x17161514u, 0x1b1a1918u, 0x00001d1cu, 0x01010101u, 0x01010101u, 0x01010101u, 0x00000101u },
    { 0x03020000u, 0x07060504u, 0x0b0a0908u, 0x0f0e0d0cu, 0x01010000u, 0x01010101u, 0x01010101u, 0x01010101u },
    { 0x13121110u, 0x17161514u, 0x1b1a1918u, 0x0000001cu, 0x01010101u, 0x01010101u, 0x01010101u, 0x00000001u },
    { 0x03020000u, 0x07060504u, 0x0b0a0908u, 0x0f0e0d0cu, 0x01010000u, 0x01010101u, 0x01010101u, 0x01010101u },
    { 0x13121110u, 0x17161514u, 0x1b1a1918u, 0x0000001cu, 0x01010101u, 0x01010101u, 0x01010101u, 0x00000001u },
    { 0x03000000u, 0x07060504u, 0x0b0a0908u, 0x0f0e0d0cu, 0x01000000u, 0x01010101u, 0x01010101u, 0x01010101u },
    { 0x13121110u, 0x17161514u, 0x1b1a1918u, 0x00000000u, 0x01010101u, 0x01010101u, 0x01010101u, 0x00000000u },
    { 0x00000000u, 0x07060504u, 0x0b0a0908u, 0x0f0e0d0cu, 0x00000000u, 0x01010101u, 0x01010101u, 0x01010101u },
    { 0x13121110u, 0x17161514u, 0x001a1918u, 0x00000000u, 0x01010101u, 0x01010101u, 0x00010101u, 0x00000000u },
    { 0x00000000u, 0x07060500u, 0x0b0a0908u, 0x0f0e0d0cu, 0x00000000u, 0x01010100u, 0x01010101u, 0x01010101u },
    { 0x13121110u, 0x17161514u, 0x00001918u, 0x00000000u, 0x01010101u, 0x01010101u, 0x00000101u, 0x00000000u },
    { 0x00000000u, 0x07060000u, 0x0b0a0908u, 0x0f0e0d0cu, 0x00000000u, 0x01010000u, 0x01010101u, 0x01010101u },
    { 0x13121110u, 0x17161514u, 0x00000018u, 0x00000000u, 0x01010101u, 0x01010101u, 0x00000001u, 0x00000000u },
    { 0x00000000u, 0x07000000u, 0x0b0a0908u, 0x0f0e0d0cu, 0x00000000u, 0x01000000u, 0x01010101u, 0x01010101u },
    { 0x13121110u, 0x17161514u, 0x00000000u, 0x00000000u, 0x01010101u, 0x01010101u, 0x00000000u, 0x00000000u },
    { 0x00000000u, 0x00000000u, 0x0b0a0900u, 0x0f0e0d0cu, 0x00000000u, 0x00000000u, 0x01010100u, 0x01010101u },
    { 0x13121110u, 0x00001514u, 0x00000000u, 0x00000000u, 0x01010101u, 0x00000101u, 0x00000000u, 0x00000000u },
    { 0x00000000u, 0x00000000u, 0x00000000u, 0x0f0e0d0cu, 0x00000000u, 0x00000000u, 0x00000000u, 0x01010101u },
    { 0x00121110u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00010101u, 0x00000000u, 0x00000000u, 0x00000000u },
};
// 7x7 sigma=2 Gaussian, 8-bit fixed point (getGaussianKernel(7,2,CV_32F) * 256, cvRound)
__constant__ int c_gk[7] = { 18, 34, 49, 55, 49, 34, 18 };

// FAST tile geometry: the SCORE region (outputs + 1 px ring for NMS) is 128 x 64 so that a row of it is 32 groups of
// 4 pixels = 32 LDS dwords; outputs are the inner 126 x 62.  The pixel tile (score region + 3 px ring) is staged so
// that score column 0 sits on a dword boundary: LDS byte (c + 1) of a row holds tile column c.
constexpr int FS_W = 128, FS_H = 64;              // score region
constexpr int FT_W = FS_W - 4, FT_H = FS_H - 2;   // outputs per tile (124 x 62): the tile step in x is a multiple of 4 so that
                                                  // every tile's pixel window starts on an image dword (aligned LDS-DMA staging)
constexpr int FP_H = FS_H + 6;                    // pixel tile rows (70)
constexpr int FP_PITCH = 144;                     // LDS row pitch in bytes: 9 chunks of 16 B (134 columns used)
constexpr int FAST_THREADS = 512;                 // 128 x 64 tiles with 8 wavefronts: 0.273 -> 0.260 ms vs 128 x 32 / 4 (128 x 128 / 16: 0.275)
constexpr int CNT_STRIDE = 32;                    // candidate counters one per 128-byte line: tiles of different
                                                  // (frame, level) pairs must not serialise on one L2 atomic unit

struct LevelDesc {
    uint8_t* img; int w, h, pitch; long long frame_stride; // bytes
    int roi_x0, roi_y0, roi_w, roi_h;  // [edge, w-edge) x [edge, h-edge)
    int tiles_x, tiles_y, tile_base;   // tile_base: first tile index of this level in the all-level grid
    int tile_x0;                       // image x of the first tile's first output column: <= roi_x0, == 1 (mod 4)
    int cand_cap; uint32_t* cand; uint32_t* candA; float* respA;
    int nfeat; float sf, inv_sf;
    int2* xtab; int4* ytab;            // resize tables for producing THIS level from l-1
    int4* p2col; int4* p2row;          // pyramid2_k with THIS level as B: two int4 per tile column / tile row (regions and footprints)
    int fused_with_next;               // 1: this level and the next one are produced together from level l-1 (pyramid2_k)
};

struct OrbDev {
    LevelDesc lv[MSLAM_MAX_LEVELS];
    int L, B, max_kpts, edge, fast_thr, score_type, tiles_per_frame, ablate;
    uint32_t* cnt;    // [B][L] candidate counts
    uint2* sel;       // [B][L][max_kpts]
    int32_t* sel_cnt; // [B][L]
    int32_t* frame_status; // [B]
    const int4* fast_tiles; // [tiles_per_frame] {level, tile x, tile y, 0}: fast_nms_k's tile -> level map, built on the host
    uint32_t tpf_magic;     // floor(2^32 / tiles_per_frame) + 1 (tiles_per_frame >= 2): workgroup / tiles_per_frame == __umulhi(workgroup, tpf_magic)
};

// LDS-DMA: 16 bytes per lane from a per-lane global address straight into LDS at (wave-uniform) lds_wave_base + lane * 16.
// No VGPR round trip; completion is tracked by vmcnt (a following __syncthreads() waits for it).
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base)
{
    typedef __attribute__((address_space(1))) const void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    __builtin_amdgcn_global_load_lds((gptr_t)gsrc, (lptr_t)lds_wave_base, 16, 0, 0);
}

// ---------------------------------------------------------------- K1: gray + pyramid
__global__ void bgr2gray_k(const uint8_t* __restrict__ src, int w, int h, int stride, long long fstride, int ch,
                           uint8_t* __restrict__ dst, int dpitch, long long dfstride)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, b = blockIdx.z;
    if (x >= w) return;
    const uint8_t* s = src + b * fstride + (size_t)y * stride + (size_t)x * ch;
    int v = ch == 1 ? s[0] : (s[0] * 1868 + s[1] * 9617 + s[2] * 4899 + (1 << 13)) >> 14;
    dst[b * dfstride + (size_t)y * dpitch + x] = (uint8_t)v;
}

// A workgroup produces a 256 x 8 destination tile: the source footprint (<= ~310 x 11 px at scale 1.2) is staged in
// LDS with coalesced aligned dword loads (byte gathers from global memory are VMEM-issue bound: ~22 CU-cycles per
// wave-instruction), then each lane computes 4 px x 2 rows from LDS bytes and writes packed 32-bit words.
// Column table entry {sx0 | sx1 << 16, a0 | a1 << 16}; row table entry {sy0, sy1, b0, b1} (wave-uniform).
// Separable evaluation inside the tile (same integer operations as OpenCV's two passes): the horizontal sums
// S = src[sx0] a0 + src[sx1] a1 are formed ONCE per source row of the tile's footprint (about 1.2 + 1/8 rows per output
// row at the pyramid's scale instead of 2) and kept as S >> 4 (u16) in LDS; the vertical pass then reads two of them.
constexpr int RS_TW = 256, RS_TH = 8, RS_LDS_DW = 1536, RS_HROWS = 16; // 6 + 8 KiB of LDS: 8 workgroups per CU
constexpr int RS_CPR = RS_LDS_DW / 4 / RS_HROWS;                        // 24 chunks of 16 bytes: the fixed LDS row pitch of the fast path
__global__ __launch_bounds__(256)
void resize_k(const uint8_t* __restrict__ src, int spitch, long long sfstride,
              uint8_t* __restrict__ dst, int dw, int dh, int dpitch, long long dfstride,
              const int2* __restrict__ xtab, const int4* __restrict__ ytab)
{
    __shared__ uint32_t s_src[RS_LDS_DW];
    __shared__ uint32_t s_h[RS_HROWS * RS_TW / 2];
    const int tid = threadIdx.x, b = blockIdx.z; // (an XCD-contiguous order was measured slower here: +11 %)
    const int x0 = blockIdx.x * RS_TW, y0 = blockIdx.y * RS_TH;
    const int xl = min(x0 + RS_TW, dw) - 1, yl = min(y0 + RS_TH, dh) - 1;
    // source footprint of the tile (tables are monotone)
    const int sx_lo = xtab[x0].x & 0xFFFF, sx_hi = (int)((unsigned)xtab[xl].x >> 16);
    const int sy_lo = ytab[y0].x, sy_hi = ytab[yl].y;
    const int sxa = sx_lo & ~3;
    const int nrows = sy_hi - sy_lo + 1;
    // LDS-DMA staging (16-byte chunks, lane-linear in LDS = row-major with a pitch of cpr chunks) whenever the last chunk of a
    // row stays inside the row's pitch; the tile column at the right border falls back to dword loads through VGPRs
    const int cpr = ((sx_hi - sxa) >> 4) + 1;
    const bool dma = sxa + cpr * 16 <= spitch;
    const int ndw = dma ? cpr * 4 : ((sx_hi - sxa) >> 2) + 1;
    const bool use_lds = ndw * nrows < RS_LDS_DW && nrows <= RS_HROWS && ndw <= 128; // true for pyramid scales up to ~1.6
    const uint8_t* sb = src + b * sfstride;
    const int tx = tid & 63, ty = __builtin_amdgcn_readfirstlane(tid >> 6);   // (the wavefront number as a scalar: row-table entries by scalar loads, row offsets on the scalar unit)
    const int x4 = x0 + tx * 4;
    uint8_t* db = dst + b * dfstride;
    if (dma && cpr <= RS_CPR && nrows <= RS_HROWS) {
        // The pyramid's own case (scale 1.2: 20-21 chunks per source row): a fixed LDS row pitch of RS_CPR chunks.  Row offsets are
        // then immediates of the LDS instructions in the unrolled row loops below (no pointer arithmetic per row), and the two
        // row-table entries of this thread are fetched before the staging instead of after the second barrier.
        int4 yt2[2];
#pragma unroll
        for (int r = 0; r < 2; r++) yt2[r] = ytab[min(y0 + ty * 2 + r, dh - 1)];
        {
            const uint8_t* s0 = sb + (size_t)sy_lo * spitch + sxa;
            const int nch = nrows * RS_CPR;
#pragma unroll
            for (int c0 = 0; c0 < RS_HROWS * RS_CPR; c0 += 256) {
                const int cb = c0 + (tid & ~63), c = cb + (tid & 63);
                const int r = c / RS_CPR, k = c - r * RS_CPR;
                if (c < nch && k < cpr) glds16(s0 + __umul24((unsigned)r, (unsigned)spitch) + 16 * k, reinterpret_cast<uint8_t*>(s_src) + cb * 16);
            }
        }
        int o0[4], a0[4], a1[4];
        {
            const int4* t4 = reinterpret_cast<const int4*>(xtab + min(x4, (dw + 3) & ~3));
            const int4 ta = t4[0], tb = t4[1];
            const int tx_[4] = { ta.x, ta.z, tb.x, tb.z }, ty_[4] = { ta.y, ta.w, tb.y, tb.w };
#pragma unroll
            for (int i = 0; i < 4; i++) { o0[i] = (tx_[i] & 0xFFFF) - sxa + ty * (RS_CPR * 16); a0[i] = ty_[i] & 0xFFFF; a1[i] = (int)((unsigned)ty_[i] >> 16); }
        }
        __syncthreads();
        const uint8_t* lds = reinterpret_cast<const uint8_t*>(s_src);
        uint32_t* hrow = s_h + (ty * RS_TW + tx * 4) / 2;
#pragma unroll
        for (int k = 0; k < RS_HROWS / 4; k++) {
            if (ty + 4 * k < nrows) {   // wave-uniform
                const uint8_t* row = lds + 4 * k * (RS_CPR * 16);
                uint32_t h[4];
#pragma unroll
                for (int i = 0; i < 4; i++) h[i] = (uint32_t)(__umul24(row[o0[i]], a0[i]) + __umul24(row[o0[i] + 1], a1[i])) >> 4; // <= 32640
                *reinterpret_cast<uint2*>(hrow + 4 * k * (RS_TW / 2)) = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
            }
        }
        __syncthreads();
        if (x4 >= dw) return;
        const uint8_t* hb8 = reinterpret_cast<const uint8_t*>(s_h) + 8 * tx;
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const int y = y0 + ty * 2 + r;
            if (y >= dh) break;
            const int4 yt = yt2[r]; // {sy0, sy1, b0, b1}
            const uint16_t* ha = reinterpret_cast<const uint16_t*>(hb8 + (yt.x - sy_lo) * (RS_TW * 2));
            const uint16_t* hb = reinterpret_cast<const uint16_t*>(hb8 + (yt.y - sy_lo) * (RS_TW * 2));
            uint32_t out = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t A = __umul24((uint32_t)yt.z, ha[i]), Bv = __umul24((uint32_t)yt.w, hb[i]);
                typedef unsigned short us2 __attribute__((ext_vector_type(2)));
                const uint32_t hh = __builtin_amdgcn_perm(Bv, A, 0x07060302u);
                const uint32_t v = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, hh), (us2){1, 1}, 2u, false) >> 2;
                out |= v << (8 * i);
            }
            *reinterpret_cast<uint32_t*>(db + (__umul24((unsigned)y, (unsigned)dpitch) + (unsigned)x4)) = out;
        }
        return;
    }
    if (use_lds) {
        if (dma) {   // nrows * cpr chunks: one global_load_lds_dwordx4 moves 64 of them, no VGPR round trip
            const int nch = nrows * cpr, inv = (65536 + cpr - 1) / cpr;   // c / cpr == (c * inv) >> 16 for c < 640, cpr < 40
            const uint8_t* s0 = sb + (size_t)sy_lo * spitch + sxa;
            for (int c0 = 0; c0 < nch; c0 += 256) {
                const int cb = c0 + (tid & ~63);     // wave-uniform first chunk of this instruction
                const int c = cb + (tid & 63);
                if (c < nch) {
                    const int r = (int)(__umul24((unsigned)c, (unsigned)inv) >> 16), k = c - __mul24(r, cpr);
                    glds16(s0 + __umul24((unsigned)r, (unsigned)spitch) + 16 * k, reinterpret_cast<uint8_t*>(s_src) + cb * 16);
                }
            }
        } else {   // stage the footprint: 128 lanes per source row, two rows per pass, RS_HROWS / 2 passes; all loads of a lane are
            // issued before its first LDS store (one memory latency per tile, not one per pass)
            const int max_dw = spitch / 4 - 1, spitch_dw = spitch / 4;
            const uint32_t* sb32 = reinterpret_cast<const uint32_t*>(sb);
            const int c = tid & 127, rh = tid >> 7;
            const unsigned col = (unsigned)min(sxa / 4 + c, max_dw);
            const bool c_ok = c < ndw;
            uint32_t v[RS_HROWS / 2];
#pragma unroll
            for (int k = 0; k < RS_HROWS / 2; k++) {
                const int r = 2 * k + rh;
                v[k] = (c_ok && r < nrows) ? sb32[__umul24((unsigned)(sy_lo + r), (unsigned)spitch_dw) + col] : 0u; // 24-bit multiplies: full rate (v_mul_lo_u32 is quarter rate)
            }
#pragma unroll
            for (int k = 0; k < RS_HROWS / 2; k++) {
                const int r = 2 * k + rh;
                if (c_ok && r < nrows) s_src[__umul24((unsigned)r, (unsigned)ndw) + c] = v[k];
            }
        }
        __syncthreads();
        int o0[4], a0[4], a1[4];
        {   // the table is padded with copies of its last column: one group of 4 = two 16-byte loads, no clamps
            const int4* t4 = reinterpret_cast<const int4*>(xtab + min(x4, (dw + 3) & ~3));
            const int4 ta = t4[0], tb = t4[1];
            const int tx_[4] = { ta.x, ta.z, tb.x, tb.z }, ty_[4] = { ta.y, ta.w, tb.y, tb.w };
#pragma unroll
            for (int i = 0; i < 4; i++) {
                // sx1 == sx0 + 1 except at the clamped right border, where a1 == 0 (fx = 0): reading sx0 + 1 is then harmless
                o0[i] = (tx_[i] & 0xFFFF) - sxa; a0[i] = ty_[i] & 0xFFFF; a1[i] = (int)((unsigned)ty_[i] >> 16);
            }
        }
        const uint8_t* lds = reinterpret_cast<const uint8_t*>(s_src);
        const int lpitch = ndw * 4;
        for (int r = ty; r < nrows; r += 4) { // wave-uniform row
            const uint8_t* row = lds + r * lpitch;
            uint32_t h[4];
#pragma unroll
            for (int i = 0; i < 4; i++) h[i] = (uint32_t)(row[o0[i]] * a0[i] + row[o0[i] + 1] * a1[i]) >> 4; // <= 32640
            *reinterpret_cast<uint2*>(&s_h[(r * RS_TW + tx * 4) / 2]) = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
        }
        __syncthreads();
        if (x4 >= dw) return;
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const int y = y0 + ty * 2 + r;
            if (y >= dh) break;
            const int4 yt = ytab[y]; // {sy0, sy1, b0, b1}
            const uint16_t* ha = reinterpret_cast<const uint16_t*>(s_h) + (yt.x - sy_lo) * RS_TW + tx * 4; // u16 reads: no unpacking
            const uint16_t* hb = reinterpret_cast<const uint16_t*>(s_h) + (yt.y - sy_lo) * RS_TW + tx * 4;
            uint32_t out = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) { // v <= 255: no masking needed
                const uint32_t A = __umul24((uint32_t)yt.z, ha[i]), Bv = __umul24((uint32_t)yt.w, hb[i]); // <= 2048 x 32640
                // (A >> 16) + (Bv >> 16) + 2: high halves gathered by one v_perm, summed by v_dot2_u32_u16 with weights (1, 1)
                typedef unsigned short us2 __attribute__((ext_vector_type(2)));
                const uint32_t hh = __builtin_amdgcn_perm(Bv, A, 0x07060302u);
                const uint32_t v = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, hh), (us2){1, 1}, 2u, false) >> 2;
                out |= v << (8 * i);
            }
            *reinterpret_cast<uint32_t*>(db + (__umul24((unsigned)y, (unsigned)dpitch) + (unsigned)x4)) = out;
        }
        return;
    }
    // generic path (footprint larger than the LDS tiles): straight from global memory
    if (x4 >= dw) return;
    int sx0[4], sx1[4], a0[4], a1[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int2 t = xtab[min(x4 + i, dw - 1)];
        sx0[i] = t.x & 0xFFFF; sx1[i] = (int)((unsigned)t.x >> 16); a0[i] = t.y & 0xFFFF; a1[i] = (int)((unsigned)t.y >> 16);
    }
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const int y = y0 + ty * 2 + r;
        if (y >= dh) break;
        const int4 yt = ytab[y];
        uint32_t out = 0;
        const uint8_t* r0 = sb + (size_t)yt.x * spitch;
        const uint8_t* r1 = sb + (size_t)yt.y * spitch;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int S0 = r0[sx0[i]] * a0[i] + r0[sx1[i]] * a1[i];
            const int S1 = r1[sx0[i]] * a0[i] + r1[sx1[i]] * a1[i];
            const int v = (((yt.z * (S0 >> 4)) >> 16) + ((yt.w * (S1 >> 4)) >> 16) + 2) >> 2;
            out |= (uint32_t)(v & 255) << (8 * i);
        }
        *reinterpret_cast<uint32_t*>(db + (size_t)y * dpitch + x4) = out;
    }
}

// Two pyramid levels per launch: level A = l+1 and level B = l+2 from level S = l.  A workgroup owns a 128 x 16 tile of B.
// It stages the source footprint of the A pixels that tile needs, produces those A pixels in LDS (same two integer passes as
// resize_k), stores the part of them it OWNS to HBM (A is partitioned over the workgroups by the first A column / row each B
// tile needs, rounded down to a multiple of 4 columns so that every store is an aligned dword), and produces the B tile
// from the A pixels in LDS.  Against two resize_k launches: level A is not read back from HBM (1.35 x its area with the tile
// halos) and one launch is saved; the price is the recomputation of the A pixels two neighbouring tiles share.
constexpr int P2_BW = 128, P2_BH = 16;           // B tile
constexpr int P2_AW = 176, P2_AH = 24;           // A region bound (columns multiple of 4, rows)
constexpr int P2_SDW = 56, P2_SH = 32;           // source footprint bound: dwords per row, rows
struct Pyr2Args {
    const uint8_t* S; int spitch; long long sfstride;
    uint8_t* A; int wA, hA, apitch; long long afstride; const int2* xtA; const int4* ytA;
    uint8_t* Bm; int wB, hB, bpitch; long long bfstride; const int2* xtB; const int4* ytB;
    const int4* colD; const int4* rowD;   // per tile column / row: {ox0, ox1, cx1, sx_lo}{sx_hi} / {oy0, oy1, cy1, sy_lo}{sy_hi}
};
__global__ __launch_bounds__(256)
void pyramid2_k(Pyr2Args a)
{
    // (+ 8 dwords each: the 4-column groups at the right edge of a region read a few bytes past it; those columns are discarded)
    __shared__ uint32_t s_src[P2_SH * P2_SDW + 8];        // source footprint (bytes)
    __shared__ uint16_t s_h[P2_SH * P2_AW];               // horizontal sums >> 4: first for A (source rows), then for B (A rows)
    // the A region (bytes, pitch P2_AW) takes the place of the source footprint, which is dead once A's horizontal sums are formed
    // (the barrier after that pass separates the last read of s_src from the first write of s_A): 19 KiB of LDS instead of 23,
    // eight workgroups per CU instead of six
    uint32_t* const s_A = s_src;
    static_assert(P2_AH * P2_AW / 4 + 8 <= P2_SH * P2_SDW + 8, "the A region must fit in the source footprint's storage");
    __shared__ int4 s_yt[P2_AH + P2_BH];                  // row-table entries of the tile's A rows, then of its B rows
    const int tid = threadIdx.x, b = blockIdx.z;
    const int xB0 = blockIdx.x * P2_BW, yB0 = blockIdx.y * P2_BH;
    const int xB1 = min(xB0 + P2_BW, a.wB) - 1, yB1 = min(yB0 + P2_BH, a.hB) - 1;
    // A region: owned = [ox0, ox1) x [oy0, oy1), computed = [ox0, cx1] x [oy0, cy1], and its source footprint: from the
    // per-tile descriptors the host derived from the tables (two independent 32-byte scalar loads)
    const int4 c0 = a.colD[2 * blockIdx.x], c1 = a.colD[2 * blockIdx.x + 1];
    const int4 r0 = a.rowD[2 * blockIdx.y], r1 = a.rowD[2 * blockIdx.y + 1];
    const int ox0 = c0.x, ox1 = c0.y, cx1 = c0.z, sx_lo = c0.w, sx_hi = c1.x;
    const int oy0 = r0.x, oy1 = r0.y, cy1 = r0.z, sy_lo = r0.w, sy_hi = r1.x;
    const int caw = cx1 - ox0 + 1, cah = cy1 - oy0 + 1;   // <= P2_AW, P2_AH (checked on the host against the tables)
    const int sxa = sx_lo & ~3;
    const int ndw = ((sx_hi - sxa) >> 2) + 1, nrows = sy_hi - sy_lo + 1;   // <= P2_SDW, P2_SH
    const int cprn = (ndw + 3) >> 2;                       // 16-byte chunks of a source row that hold needed columns
    if (sxa + 16 * cprn <= a.spitch) {
        // LDS-DMA: the LDS row pitch is P2_SDW dwords = 14 chunks, so chunk c of the lane-linear destination is (row c / 14,
        // chunk c % 14); chunks past the needed columns are masked off.  No VGPR round trip, 2 instructions per wavefront.
        static_assert(P2_SDW % 4 == 0, "whole 16-byte chunks per LDS row");
        constexpr int CPR = P2_SDW / 4;
        const uint8_t* s0 = a.S + b * a.sfstride + (size_t)sy_lo * a.spitch + sxa;
        const int nch = nrows * CPR;
        for (int c0 = 0; c0 < nch; c0 += 256) {
            const int cb = c0 + (tid & ~63), c = cb + (tid & 63);
            const int r = c / CPR, k = c - r * CPR;
            if (c < nch && k < cprn) glds16(s0 + __umul24((unsigned)r, (unsigned)a.spitch) + 16 * k, reinterpret_cast<uint8_t*>(s_src) + cb * 16);
        }
    } else {   // the tile column at the right border: a chunk would cross the end of the row
        const uint32_t* sb32 = reinterpret_cast<const uint32_t*>(a.S + b * a.sfstride);
        const int max_dw = a.spitch / 4 - 1, spitch_dw = a.spitch / 4;
        for (int i = tid; i < nrows * P2_SDW; i += 256) {
            const int r = i / P2_SDW, c = i - r * P2_SDW;
            if (c < ndw) s_src[r * P2_SDW + c] = sb32[__umul24((unsigned)(sy_lo + r), (unsigned)spitch_dw) + (unsigned)min(sxa / 4 + c, max_dw)];
        }
    }
    // the column-table entries of this thread's A and B column groups: fetched here, before the barrier, so that their latency
    // passes under the staging (they were loaded after the barriers, in front of the loops that need them)
    int4 tA0 = make_int4(0, 0, 0, 0), tA1 = tA0, tB0 = tA0, tB1 = tA0;
    if (tid / 44 < 5 && 4 * (tid % 44) < caw) { const int4* t4 = reinterpret_cast<const int4*>(a.xtA + ox0 + 4 * (tid % 44)); tA0 = t4[0]; tA1 = t4[1]; }
    if (xB0 + 4 * (tid & 31) <= xB1) { const int4* t4 = reinterpret_cast<const int4*>(a.xtB + xB0 + 4 * (tid & 31)); tB0 = t4[0]; tB1 = t4[1]; }
    // the row-table entries of the tile's A rows and B rows, as LDS byte offsets into s_h: {first row, second row, b0, b1} -- read
    // in the vertical passes from LDS instead of from global memory inside the row loops (an exposed load per row)
    if (tid < cah) {
        const int4 yt = a.ytA[oy0 + tid];
        s_yt[tid] = make_int4((yt.x - sy_lo) * (P2_AW * 2), (yt.y - sy_lo) * (P2_AW * 2), yt.z, yt.w);
    } else if (tid >= 64 && tid < 64 + P2_BH && yB0 + tid - 64 <= yB1) {
        const int4 yt = a.ytB[yB0 + tid - 64];
        s_yt[P2_AH + tid - 64] = make_int4((yt.x - oy0) * (P2_AW * 2), (yt.y - oy0) * (P2_AW * 2), yt.z, yt.w);
    }
    __syncthreads();
    // The row loops below are written with compile-time trip counts and a guard per step: row offsets become immediates of the LDS
    // instructions instead of pointer increments (5 of the 21 vector instructions of a horizontal step were pointer arithmetic).
    // ---- A, horizontal: thread = (column group of 4, row lane); 44 groups x 5 row lanes
    {
        const int g = tid % 44, rl = tid / 44;
        if (rl < 5 && 4 * g < caw) {
            int o0[4], a0[4], a1[4];
            {
                const int4 ta = tA0, tb = tA1;   // (padded table: no clamps)
                const int tx_[4] = { ta.x, ta.z, tb.x, tb.z }, ty_[4] = { ta.y, ta.w, tb.y, tb.w };
#pragma unroll
                for (int i = 0; i < 4; i++) { o0[i] = (tx_[i] & 0xFFFF) - sxa + rl * (P2_SDW * 4); a0[i] = ty_[i] & 0xFFFF; a1[i] = (int)((unsigned)ty_[i] >> 16); }
            }
            const uint8_t* lds = reinterpret_cast<const uint8_t*>(s_src);
            uint16_t* hrow = s_h + rl * P2_AW + 4 * g;
#pragma unroll
            for (int k = 0; k < (P2_SH + 4) / 5; k++) {
                if (rl + 5 * k < nrows) {
                    const uint8_t* row = lds + 5 * k * (P2_SDW * 4);
                    uint32_t h[4];
#pragma unroll
                    for (int i = 0; i < 4; i++) h[i] = (uint32_t)(__umul24(row[o0[i]], a0[i]) + __umul24(row[o0[i] + 1], a1[i])) >> 4;
                    *reinterpret_cast<uint2*>(hrow + 5 * k * P2_AW) = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
                }
            }
        }
    }
    __syncthreads();
    // ---- A, vertical: into LDS, and the owned part to HBM
    {
        const int g = tid % 44, rl = tid / 44;
        if (rl < 5 && 4 * g < caw) {
            uint8_t* ab = a.A + b * a.afstride + (__umul24((unsigned)(oy0 + rl), (unsigned)a.apitch) + (unsigned)(ox0 + 4 * g));
            const uint8_t* hb8 = reinterpret_cast<const uint8_t*>(s_h) + 8 * g;
            const bool own_x = ox0 + 4 * g < ox1;
#pragma unroll
            for (int k = 0; k < (P2_AH + 4) / 5; k++) {
                const int r = rl + 5 * k;
                if (r < cah) {
                    const int4 yt = s_yt[r];
                    const uint16_t* ha = reinterpret_cast<const uint16_t*>(hb8 + yt.x);
                    const uint16_t* hb = reinterpret_cast<const uint16_t*>(hb8 + yt.y);
                    uint32_t out = 0;
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const uint32_t Av = __umul24((uint32_t)yt.z, ha[i]), Bv = __umul24((uint32_t)yt.w, hb[i]);
                        typedef unsigned short us2 __attribute__((ext_vector_type(2)));
                        const uint32_t hh = __builtin_amdgcn_perm(Bv, Av, 0x07060302u);
                        const uint32_t v = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, hh), (us2){1, 1}, 2u, false) >> 2;
                        out |= v << (8 * i);
                    }
                    s_A[r * (P2_AW / 4) + g] = out;
                    if (own_x && oy0 + r < oy1) *reinterpret_cast<uint32_t*>(ab + __umul24((unsigned)(5 * k), (unsigned)a.apitch)) = out;
                }
            }
        }
    }
    __syncthreads();
    // ---- B, horizontal from the A region in LDS: thread = (column group of 4, row lane); 32 groups x 8 row lanes
    {
        const int g = tid & 31, rl = tid >> 5;
        if (xB0 + 4 * g <= xB1) {
            int o0[4], a0[4], a1[4];
            {
                const int4 ta = tB0, tb = tB1;   // (padded table: no clamps)
                const int tx_[4] = { ta.x, ta.z, tb.x, tb.z }, ty_[4] = { ta.y, ta.w, tb.y, tb.w };
#pragma unroll
                for (int i = 0; i < 4; i++) { o0[i] = (tx_[i] & 0xFFFF) - ox0 + rl * P2_AW; a0[i] = ty_[i] & 0xFFFF; a1[i] = (int)((unsigned)ty_[i] >> 16); }
            }
            const uint8_t* lds = reinterpret_cast<const uint8_t*>(s_A);
            uint16_t* hrow = s_h + rl * P2_AW + 4 * g;
#pragma unroll
            for (int k = 0; k < (P2_AH + 7) / 8; k++) {
                if (rl + 8 * k < cah) {
                    const uint8_t* row = lds + 8 * k * P2_AW;
                    uint32_t h[4];
#pragma unroll
                    for (int i = 0; i < 4; i++) h[i] = (uint32_t)(__umul24(row[o0[i]], a0[i]) + __umul24(row[o0[i] + 1], a1[i])) >> 4;
                    *reinterpret_cast<uint2*>(hrow + 8 * k * P2_AW) = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
                }
            }
        }
    }
    __syncthreads();
    // ---- B, vertical: 16 rows x 32 groups, two rows per thread
    {
        const int g = tid & 31, rl = tid >> 5;
        uint8_t* bb = a.Bm + b * a.bfstride + (__umul24((unsigned)(yB0 + rl), (unsigned)a.bpitch) + (unsigned)(xB0 + 4 * g));
        const uint8_t* hb8 = reinterpret_cast<const uint8_t*>(s_h) + 8 * g;
        if (xB0 + 4 * g <= xB1) {
#pragma unroll
            for (int k = 0; k < P2_BH / 8; k++) {
                const int r = rl + 8 * k;
                if (yB0 + r <= yB1) {
                    const int4 yt = s_yt[P2_AH + r];
                    const uint16_t* ha = reinterpret_cast<const uint16_t*>(hb8 + yt.x);
                    const uint16_t* hb = reinterpret_cast<const uint16_t*>(hb8 + yt.y);
                    uint32_t out = 0;
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const uint32_t Av = __umul24((uint32_t)yt.z, ha[i]), Bv = __umul24((uint32_t)yt.w, hb[i]);
                        typedef unsigned short us2 __attribute__((ext_vector_type(2)));
                        const uint32_t hh = __builtin_amdgcn_perm(Bv, Av, 0x07060302u);
                        const uint32_t v = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, hh), (us2){1, 1}, 2u, false) >> 2;
                        out |= v << (8 * i);
                    }
                    *reinterpret_cast<uint32_t*>(bb + __umul24((unsigned)(8 * k), (unsigned)a.bpitch)) = out;
                }
            }
        }
    }
}

// ---------------------------------------------------------------- K2: FAST-9/16 + NMS
__device__ __forceinline__ int ring_ofs(int k)
{
    // Bresenham circle r=3 in FAST's order, as byte offsets in the LDS pixel tile
    constexpr int dx[16] = { 0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1 };
    constexpr int dy[16] = { 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3 };
    return dy[k] * FP_PITCH + dx[k];
}

// exact cornerScore<16>: max over the 16 arcs of 9 of min(d) / min(-d), minus 1, with d[k] = centre - ring pixel k.
// Evaluated on packed 16-bit pairs D[k] = (d[k], d[k + 8]), k = 0..7 (|d| <= 255): v_pk_min/max_i16 work on two ring
// positions per instruction and the "+8" partner of a pair is a half swap, which the VOP3P op_sel modifiers provide for free.
// Two arcs that start at k and k + 1 share the window d[k+1 .. k+8], so only the 8 windows of 8 that start at an odd position
// are formed (by doubling: 2, 4, 8 consecutive, 4 packed instructions per level), and each is extended by the ring pixel on
// either side (46 packed instructions for both polarities; doubling over all 16 positions took 78).
typedef short s2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s2v swp(s2v a) { return __builtin_shufflevector(a, a, 1, 0); }
__device__ __forceinline__ int fast_exact_score_pk(const s2v* D)
{
#define PMIN(a, b) __builtin_elementwise_min(a, b)
#define PMAX(a, b) __builtin_elementwise_max(a, b)
    // windows of 2 and 4 starting at ring positions 1, 3, 5, 7 (low halves) and 9, 11, 13, 15 (high halves)
    const s2v D8 = swp(D[0]);   // (d[8], d[0]) = position 8 | 16
    const s2v m2_1 = PMIN(D[1], D[2]), m2_3 = PMIN(D[3], D[4]), m2_5 = PMIN(D[5], D[6]), m2_7 = PMIN(D[7], D8);
    const s2v n2_1 = PMAX(D[1], D[2]), n2_3 = PMAX(D[3], D[4]), n2_5 = PMAX(D[5], D[6]), n2_7 = PMAX(D[7], D8);
    const s2v m4_1 = PMIN(m2_1, m2_3), m4_3 = PMIN(m2_3, m2_5), m4_5 = PMIN(m2_5, m2_7), m4_7 = PMIN(m2_7, swp(m2_1));
    const s2v n4_1 = PMAX(n2_1, n2_3), n4_3 = PMAX(n2_3, n2_5), n4_5 = PMAX(n2_5, n2_7), n4_7 = PMAX(n2_7, swp(n2_1));
    // windows of 8: j .. j + 7
    const s2v m8_1 = PMIN(m4_1, m4_5), m8_3 = PMIN(m4_3, m4_7), m8_5 = PMIN(m4_5, swp(m4_1)), m8_7 = PMIN(m4_7, swp(m4_3));
    const s2v n8_1 = PMAX(n4_1, n4_5), n8_3 = PMAX(n4_3, n4_7), n8_5 = PMAX(n4_5, swp(n4_1)), n8_7 = PMAX(n4_7, swp(n4_3));
    // arcs of 9: the window extended by position j - 1, or by position j + 8 (the partner half of D[j]);
    // max(min(w, x), min(w, y)) = min(w, max(x, y)), so each window costs one instruction per polarity and one to pair x, y
    s2v A = PMIN(m8_1, PMAX(D[0], swp(D[1])));
    s2v Bv = PMAX(n8_1, PMIN(D[0], swp(D[1])));
    A = PMAX(A, PMIN(m8_3, PMAX(D[2], swp(D[3]))));
    Bv = PMIN(Bv, PMAX(n8_3, PMIN(D[2], swp(D[3]))));
    A = PMAX(A, PMIN(m8_5, PMAX(D[4], swp(D[5]))));
    Bv = PMIN(Bv, PMAX(n8_5, PMIN(D[4], swp(D[5]))));
    A = PMAX(A, PMIN(m8_7, PMAX(D[6], swp(D[7]))));
    Bv = PMIN(Bv, PMAX(n8_7, PMIN(D[6], swp(D[7]))));
#undef PMIN
#undef PMAX
    const int amax = max((int)A.x, (int)A.y), bmin = min((int)Bv.x, (int)Bv.y);
    return max(amax, -bmin) - 1;
}

__device__ __forceinline__ uint32_t byte_of(uint32_t w, int k) { return (w >> (8 * k)) & 255u; }

__global__ __launch_bounds__(FAST_THREADS)
void fast_nms_k(OrbDev od)
{
    __shared__ uint32_t s_pix[FP_H * FP_PITCH / 4];
    __shared__ uint8_t s_score[FS_H * FS_W];
    __shared__ uint16_t s_list[FS_H * FS_W];
    __shared__ int s_n1, s_ne, s_base, s_nc;
    // NMS maxima (<= 63 x 31 per tile) are collected in the pixel tile's storage, which is dead after phase 2
    uint32_t* s_emit = s_pix;
    static_assert(((FT_W + 1) / 2) * ((FT_H + 1) / 2) * 4 <= FP_H * FP_PITCH, "emit buffer must fit in the pixel tile");

    // tile -> (frame, level, tx, ty): blocks of one frame are consecutive, so an XCD works through whole frames (a 2-D grid
    // with the frame in y, each frame's tiles dealt over the XCDs, measured 40 % slower).  One multiply-high instead of the
    // division by tiles_per_frame, one descriptor load instead of a search over the levels and a division by tiles_x.
    const unsigned wg = xcd_contiguous(blockIdx.x, gridDim.x);
    const int b = od.tiles_per_frame > 1 ? (int)__umulhi(wg, od.tpf_magic) : (int)wg;
    const int4 td = od.fast_tiles[wg - (unsigned)b * (unsigned)od.tiles_per_frame];
    const int l = td.x, tx = td.y, ty = td.z;
    const LevelDesc& lv = od.lv[l];
    const int ox = lv.tile_x0 + tx * FT_W, oy = lv.roi_y0 + ty * FT_H; // first output pixel (ox % 4 == 1; ox <= roi_x0 for tx == 0)
    const uint8_t* img = lv.img + b * lv.frame_stride;
    const int tid = threadIdx.x;

    if (tid == 0) { s_n1 = 0; s_ne = 0; s_nc = 0; }
    for (int i = tid; i < FS_H * FS_W / 4; i += FAST_THREADS) reinterpret_cast<uint32_t*>(s_score)[i] = 0;
    // ---- stage the pixel tile by LDS-DMA: LDS byte i of row r = image byte x = (ox - 5 + i) of row (oy - 4 + r).  The tile
    // grid is laid out so that ox - 5 is a multiple of 4 (level rows are dword aligned), hence score column 0 sits on an LDS
    // dword boundary with no realignment.  One global_load_lds_dwordx4 moves 64 chunks of 16 B (lane-linear in LDS); 630
    // chunks per tile = 10 wave-instructions for the whole workgroup, no VGPR round trip.  Chunks that would start beyond
    // (row end - 16) are clamped: they only hold columns past the ROI + 4, which nothing reads.
    {
        const int gx0 = ox - 5;                 // >= 16, multiple of 4
        constexpr int NCHUNK = FP_H * (FP_PITCH / 16);
        static_assert(FP_PITCH % 16 == 0, "16-byte LDS-DMA chunks");
        const int wv = tid >> 6, ln = tid & 63;
#pragma unroll
        for (int c0 = 0; c0 < NCHUNK; c0 += FAST_THREADS) {
            const int cb = c0 + wv * 64;        // wave-uniform first chunk of this instruction
            const int c = cb + ln;
            if (c < NCHUNK) {
                const int r = c / (FP_PITCH / 16), k = c - r * (FP_PITCH / 16);
                const uint8_t* g = img + (size_t)min(oy - 4 + r, lv.h - 1) * lv.pitch + min(gx0 + 16 * k, lv.pitch - 16);
                glds16(g, reinterpret_cast<uint8_t*>(s_pix) + cb * 16);
            }
        }
    }
    __syncthreads();
    if (od.ablate == 1) return;
    const int thr = od.fast_thr;
    // score (sx, sy) <-> image pixel (ox - 1 + sx, oy - 1 + sy) <-> LDS byte (sy + 3) * FP_PITCH + sx + 4
    const int lim_x = min(FS_W, lv.roi_x0 + lv.roi_w + 1 - (ox - 1)); // scores only needed up to ROI + 1
    const int lim_y = min(FS_H, lv.roi_y0 + lv.roi_h + 1 - (oy - 1));
    // ---- phase 1: high-speed rejection on 4 pixels per lane from 5 dword reads: an arc of 9 equal-polarity ring
    // pixels contains pixel 0 or 8 AND pixel 4 or 12 of that polarity.  All predicates are wavefront lane masks
    // (v_cmp -> SGPR pair, logic on the scalar unit); the survivors are appended to s_list by ballot compaction.
    const int lane = tid & 63;
    {
        const int g = tid & 31, ry = tid >> 5;
        unsigned long long colok[4]; // lanes whose pixel 4g + k lies inside the needed score columns
#pragma unroll
        for (int k = 0; k < 4; k++) colok[k] = __builtin_amdgcn_ballot_w64(4 * g + k < lim_x);
        unsigned long long bal[FS_H / (FAST_THREADS / 32)][4];
        int tot = 0;
#pragma unroll
        for (int m = 0; m < FS_H / (FAST_THREADS / 32); m++) {
            const int sy = ry + (FAST_THREADS / 32) * m;
            const unsigned long long rowok = __builtin_amdgcn_ballot_w64(sy < lim_y);
            const uint32_t* rowp = s_pix + (sy + 3) * (FP_PITCH / 4) + g; // rows/columns exist in LDS for every lane
            const uint32_t Lw = rowp[0], Cw = rowp[1], Rw = rowp[2];
            const uint32_t Tw = rowp[1 - 3 * (FP_PITCH / 4)], Bw = rowp[1 + 3 * (FP_PITCH / 4)];
            // two pixels per packed 16-bit instruction: E = pixels (0, 2), O = pixels (1, 3) of the lane's four, each ring pixel
            // pair picked out of the dwords by one v_perm_b32 / v_and_b32 (selector 0x0c = constant zero byte)
            typedef unsigned short u2v __attribute__((ext_vector_type(2)));
#define U2(x) __builtin_bit_cast(u2v, (uint32_t)(x))
#define UMIN(a, b) __builtin_elementwise_min(a, b)
#define UMAX(a, b) __builtin_elementwise_max(a, b)
            const u2v cE = U2(Cw & 0x00FF00FFu), cO = U2(__builtin_amdgcn_perm(0u, Cw, 0x0c030c01u));
            const u2v tE = U2(Tw & 0x00FF00FFu), tO = U2(__builtin_amdgcn_perm(0u, Tw, 0x0c030c01u));
            const u2v bE = U2(Bw & 0x00FF00FFu), bO = U2(__builtin_amdgcn_perm(0u, Bw, 0x0c030c01u));
            // x + 3: pixel k is byte k + 3 of (Cw, Rw); x - 3: pixel k is byte k + 1 of (Lw, Cw)
            const u2v rE = U2(__builtin_amdgcn_perm(Rw, Cw, 0x0c050c03u)), rO = U2(Rw & 0x00FF00FFu);
            const u2v lE = U2(__builtin_amdgcn_perm(0u, Lw, 0x0c030c01u)), lO = U2(__builtin_amdgcn_perm(Cw, Lw, 0x0c040c02u));
            // brighter: min(max(p0, p8), max(p4, p12)) - v > thr; darker: v - max(min(p0, p8), min(p4, p12)) > thr
            const s2v mE = __builtin_elementwise_max(__builtin_bit_cast(s2v, UMIN(UMAX(tE, bE), UMAX(rE, lE)) - cE),
                                                      __builtin_bit_cast(s2v, cE - UMAX(UMIN(tE, bE), UMIN(rE, lE))));
            const s2v mO = __builtin_elementwise_max(__builtin_bit_cast(s2v, UMIN(UMAX(tO, bO), UMAX(rO, lO)) - cO),
                                                      __builtin_bit_cast(s2v, cO - UMAX(UMIN(tO, bO), UMIN(rO, lO))));
#undef U2
#undef UMIN
#undef UMAX
            const short m4[4] = { mE.x, mO.x, mE.y, mO.y };
#pragma unroll
            for (int k = 0; k < 4; k++) {
                bal[m][k] = __builtin_amdgcn_ballot_w64((int)m4[k] > thr) & rowok & colok[k];
                tot += __popcll(bal[m][k]);
            }
        }
        // one reservation in the survivor list per wavefront (the 32 lane masks stay in SGPRs), then the entries by rank
        if (tot) { // wave-uniform
            int base = 0;
            if (lane == 0) base = atomicAdd(&s_n1, tot);
            // byte offset of the wavefront's next entry, held on the scalar unit (the running sums of the slot counts are scalar
            // additions, and a slot's address is one v_lshl_add_u32 of its rank)
            uint32_t boff = 2u * (uint32_t)__builtin_amdgcn_readfirstlane(base);
            asm volatile("" : "+s"(boff));   // (opaque: otherwise the factor 2 is pulled out of the sum again and every slot shifts twice)
            uint8_t* const lb = reinterpret_cast<uint8_t*>(s_list);
#pragma unroll
            for (int m = 0; m < FS_H / (FAST_THREADS / 32); m++) {
                const int sy = ry + (FAST_THREADS / 32) * m;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    // rank of this lane among the survivors of slot k: v_mbcnt_lo/hi take the lane mask straight from SGPRs
                    const uint32_t rank_k = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal[m][k] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal[m][k], 0u));
                    if (__builtin_amdgcn_inverse_ballot_w64(bal[m][k])) *reinterpret_cast<uint16_t*>(lb + (boff + 2u * rank_k)) = (uint16_t)(sy * FS_W + 4 * g + k);
                    boff += 2u * (uint32_t)__popcll(bal[m][k]);
                }
            }
        }
    }
    __syncthreads();
    if (od.ablate == 2) return;
    const uint8_t* pix = reinterpret_cast<const uint8_t*>(s_pix);
    // ---- phase 2: segment test + exact score for the survivors (dense lanes).  The corners (about 40 % of the survivors)
    // are appended to a list of their own in the upper half of s_list, so that phase 3 runs on dense lanes as well; a tile with
    // more survivors than half the list holds (never seen outside synthetic noise) walks the survivor list again instead.
    const int n1 = s_n1;
    constexpr int CL0 = FS_H * FS_W / 2;
    const bool clist = n1 <= CL0;   // workgroup-uniform
    for (int j0 = 0; j0 < n1; j0 += FAST_THREADS) {
        const int j = j0 + tid;
        int i = 0, sc = -1;
        if (j < n1) {
            i = s_list[j];
            const int sy = i / FS_W, sx = i % FS_W;
            const uint8_t* p = pix + (sy + 3) * FP_PITCH + (sx + 4);
            const uint32_t v = p[0];
            // packed differences D[k] = (v - p[k], v - p[k + 8]).  The segment test needs no masks of its own: an arc of 9 ring
            // pixels all darker than v - thr exists iff max over arcs of min(d) > thr, all brighter iff min over arcs of
            // max(d) < -thr -- the two quantities the exact score is made of.  So: score + 1 > thr <=> corner.
            // (ds_read_u8_d16_hi would drop the packing instructions, but with SRAM ECC a d16 load clears the other half.)
            const s2v V2 = __builtin_bit_cast(s2v, v | (v << 16));
            s2v D[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const uint32_t pk = (uint32_t)p[ring_ofs(k)] | ((uint32_t)p[ring_ofs(k + 8)] << 16);
                D[k] = V2 - __builtin_bit_cast(s2v, pk);
            }
            sc = fast_exact_score_pk(D);
        }
        const bool corner = sc >= thr;
        if (corner) s_score[i] = (uint8_t)sc;
        if (clist) {
            const unsigned long long cm = __builtin_amdgcn_ballot_w64(corner);
            if (cm) { // wave-uniform
                int base = 0;
                if (lane == 0) base = atomicAdd(&s_nc, __popcll(cm));
                uint32_t boff = 2u * (uint32_t)(CL0 + __builtin_amdgcn_readfirstlane(base));   // (scalar, as in phase 1)
                asm volatile("" : "+s"(boff));
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(cm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)cm, 0u));
                if (corner) *reinterpret_cast<uint16_t*>(reinterpret_cast<uint8_t*>(s_list) + (boff + 2u * rank)) = (uint16_t)i;
            }
        }
    }
    __syncthreads();
    if (od.ablate == 3) return;
    // ---- phase 3: 3x3 non-max suppression on the score tile, outputs inside the ROI only
    const int out_w = min(FT_W, lv.roi_x0 + lv.roi_w - ox), out_h = min(FT_H, lv.roi_y0 + lv.roi_h - oy);
    const int out_x0 = 1 + max(0, lv.roi_x0 - ox);   // the first tile column starts up to 3 px left of the ROI
    const int n3 = clist ? s_nc : n1;
    const uint16_t* l3 = clist ? s_list + CL0 : s_list;
    for (int j = tid; j < n3; j += FAST_THREADS) {
        const int i = l3[j];
        const int sy = i / FS_W, sx = i % FS_W;
        const int sc = s_score[i];
        // branch-free: eight neighbour bytes, their maximum by v_max3, one compare (a chain of && here compiles to a chain of
        // exec-mask branches that every wavefront walks to the end); positions that are no candidates read a safe interior cell
        const bool cand = (sc != 0) & (sx >= out_x0) & (sy >= 1) & (sx <= out_w) & (sy <= out_h);
        const uint8_t* c = s_score + (cand ? i : FS_W + 1);
        const uint32_t m0 = max(max((uint32_t)c[-FS_W - 1], (uint32_t)c[-FS_W]), (uint32_t)c[-FS_W + 1]);
        const uint32_t m1 = max(max((uint32_t)c[-1], (uint32_t)c[1]), (uint32_t)c[FS_W - 1]);
        const uint32_t m2 = max(max((uint32_t)c[FS_W], (uint32_t)c[FS_W + 1]), m0);
        if (cand & ((uint32_t)sc > max(m1, m2))) {
            const int e = atomicAdd(&s_ne, 1);
            const int gx = ox - 1 + sx, gy = oy - 1 + sy;
            s_emit[e] = ((uint32_t)sc << 24) | ((uint32_t)gy << 12) | (uint32_t)gx;
        }
    }
    __syncthreads();
    const int ne = s_ne;
    if (ne == 0 || od.ablate == 4) return;
    if (tid == 0) s_base = (int)atomicAdd(&od.cnt[(b * od.L + l) * CNT_STRIDE], (uint32_t)ne);
    __syncthreads();
    uint32_t* cand = lv.cand + (size_t)b * lv.cand_cap;
    for (int j = tid; j < ne; j += FAST_THREADS)
        if (s_base + j < lv.cand_cap) cand[s_base + j] = s_emit[j];
}

// ---------------------------------------------------------------- K3: select
constexpr int SEL_THREADS = 1024;
constexpr int SEL_LEVEL_CAP = 4096; // survivors of one level are rank-sorted in LDS (32 KiB)
constexpr int SEL_RC = 16;           // candidates per thread that select_k keeps in registers between its two passes over the list

__device__ __forceinline__ float harris_response(const uint8_t* __restrict__ img, int pitch, int x, int y)
{
    // HarrisResponses(img, pts, 7, 0.04f): 7x7 block of Sobel-like gradients from a 9x9 window
    int a = 0, bb = 0, c = 0;
    int win[9][9];
    // nine bytes per row from three aligned dwords (level rows are dword aligned and a keypoint is >= 31 px from the border):
    // 27 loads per lane instead of 81 byte loads, which are bound by the issue rate of the vector memory pipeline
    const uint8_t* p = img + (size_t)(y - 4) * pitch + (x - 4);
    const unsigned o = (unsigned)(uintptr_t)p & 3u;
    const uint32_t* p4 = reinterpret_cast<const uint32_t*>(p - o);
    const int pitch4 = pitch >> 2;
#pragma unroll
    for (int r = 0; r < 9; r++) {
        const uint32_t d0 = p4[r * pitch4], d1 = p4[r * pitch4 + 1], d2 = p4[r * pitch4 + 2];
        const uint32_t w[3] = { __builtin_amdgcn_alignbyte(d1, d0, o), __builtin_amdgcn_alignbyte(d2, d1, o), __builtin_amdgcn_alignbyte(0u, d2, o) };
#pragma unroll
        for (int j = 0; j < 9; j++) win[r][j] = (int)((w[j >> 2] >> (8 * (j & 3))) & 255u);
    }
#pragma unroll
    for (int r = 1; r <= 7; r++)
#pragma unroll
        for (int j = 1; j <= 7; j++) {
            int Ix = (win[r][j + 1] - win[r][j - 1]) * 2 + (win[r - 1][j + 1] - win[r - 1][j - 1]) + (win[r + 1][j + 1] - win[r + 1][j - 1]);
            int Iy = (win[r + 1][j] - win[r - 1][j]) * 2 + (win[r + 1][j - 1] - win[r - 1][j - 1]) + (win[r + 1][j + 1] - win[r - 1][j + 1]);
            a += Ix * Ix; bb += Iy * Iy; c += Ix * Iy;
        }
    float scale = (1 << 2) * 7 * 255.0f;
    scale = 1.0f / scale;
    float scale_sq_sq = scale * scale * scale * scale;
    return ((float)a * bb - (float)c * c - 0.04f * ((float)a + bb) * ((float)a + bb)) * scale_sq_sq;
}

__device__ __forceinline__ uint32_t float_key(float f)
{
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u); // order-preserving
}

// Wavefront reductions and scans by DPP operands -- no LDS round trips (a butterfly of __shfl_xor is six dependent ds_bpermute
// latencies, with four vector instructions of address arithmetic each).  An inclusive scan inside each row of 16 lanes by row shifts
// (1, 2, 4, 8), then the row totals passed on by row_bcast:15 (into rows 1, 3) and row_bcast:31 (into rows 2, 3): lane l holds
// the result over lanes 0..l, lane 63 the result over the wavefront.  Every lane of the wavefront must be active at the call (an
// inactive lane feeds its neighbours stale register contents): the callers are whole wavefronts (select_k under `tid < 64`,
// describe_k outside any lane-dependent branch).
__device__ __forceinline__ int wave_scan_add_dpp(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);   // row_shr:1 (zeros shifted in)
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);   // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31
    return v;
}
__device__ __forceinline__ int wave_sum_dpp(int v) { return __builtin_amdgcn_readlane(wave_scan_add_dpp(v), 63); }   // -> scalar register
// (max / min: a lane without a source keeps its own value)
#define MSLAM_DPP_FOLD(OP, v)                                                              \
    v = OP(v, __builtin_amdgcn_update_dpp(v, v, 0x111, 0xf, 0xf, false));                  \
    v = OP(v, __builtin_amdgcn_update_dpp(v, v, 0x112, 0xf, 0xf, false));                  \
    v = OP(v, __builtin_amdgcn_update_dpp(v, v, 0x114, 0xf, 0xf, false));                  \
    v = OP(v, __builtin_amdgcn_update_dpp(v, v, 0x118, 0xf, 0xf, false));                  \
    v = OP(v, __builtin_amdgcn_update_dpp(v, v, 0x142, 0xa, 0xf, false));                  \
    v = OP(v, __builtin_amdgcn_update_dpp(v, v, 0x143, 0xc, 0xf, false));
__device__ __forceinline__ int wave_max_dpp(int v) { MSLAM_DPP_FOLD(max, v) return __builtin_amdgcn_readlane(v, 63); }
__device__ __forceinline__ int wave_min_dpp(int v) { MSLAM_DPP_FOLD(min, v) return __builtin_amdgcn_readlane(v, 63); }

// Suffix sums of the 256-bin histogram by the first wavefront (lane l owns bins 4 l .. 4 l + 3): inc[q] = number of entries in
// bins >= 4 l + q.  Replaces a scan by one thread, whose 256 dependent LDS reads were the longest step of select_k.
__device__ __forceinline__ void hist_suffix(const uint32_t* s_hist, int lane, uint32_t (&h)[4], uint32_t (&inc)[4])
{
    const uint4 hv = *reinterpret_cast<const uint4*>(s_hist + 4 * lane);
    h[0] = hv.x; h[1] = hv.y; h[2] = hv.z; h[3] = hv.w;
    const uint32_t own = h[0] + h[1] + h[2] + h[3];
    // suffix over the lanes from the prefix scan: sum(l..63) = total - sum(0..l) + own
    const uint32_t pre = (uint32_t)wave_scan_add_dpp((int)own);
    const uint32_t sfx = (uint32_t)__builtin_amdgcn_readlane((int)pre, 63) - pre + own;
    inc[3] = sfx - own + h[3]; inc[2] = inc[3] + h[2]; inc[1] = inc[2] + h[1]; inc[0] = inc[1] + h[0];
}

// One digit of the radix select, by the first wavefront: the bin that holds the k-th largest key is the highest bin >= 1 whose
// inclusive suffix count reaches k, else bin 0; k then drops by the entries in the bins above it.
__device__ __forceinline__ void radix_step(const uint32_t* s_hist, int lane, int shift, uint32_t prefix, uint32_t mask, int* s_k, uint32_t* s_prefix, uint32_t* s_mask)
{
    uint32_t h[4], inc[4];
    hist_suffix(s_hist, lane, h, inc);
    const int k = *s_k;
    int bl = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) if ((int)inc[q] >= k && 4 * lane + q >= 1) bl = 4 * lane + q;
    const int bin = wave_max_dpp(bl);
    if (lane == (bin >> 2)) {
        *s_k = k - (int)(inc[bin & 3] - h[bin & 3]);
        *s_prefix = prefix | ((uint32_t)bin << shift); *s_mask = mask | (255u << shift);
    }
}

__global__ __launch_bounds__(SEL_THREADS)
void select_k(OrbDev od)
{
    extern __shared__ uint8_t dyn[];  // survivors: uint2[max_kpts]
    __shared__ __attribute__((aligned(16))) uint32_t s_hist[256];   // (read as uint4 by hist_suffix)
    __shared__ int s_T, s_nA, s_nS, s_nKeep;
    __shared__ uint32_t s_prefix, s_mask;
    __shared__ int s_k;
    const int l = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const LevelDesc& lv = od.lv[l];
    const uint8_t* img = lv.img + b * lv.frame_stride;
    const int ncand = min((int)od.cnt[(b * od.L + l) * CNT_STRIDE], lv.cand_cap);
    const uint32_t* cand = lv.cand + (size_t)b * lv.cand_cap;
    uint32_t* candA = lv.candA + (size_t)b * lv.cand_cap;
    float* respA = lv.respA + (size_t)b * lv.cand_cap;
    uint2* surv = reinterpret_cast<uint2*>(dyn);
    const int nf = lv.nfeat;
    const bool harris = od.score_type == 0;
    const int n1 = harris ? 2 * nf : nf; // first retainBest budget (by FAST score)

    // FAST score histogram of this (frame, level) from its candidate list
    if (tid < 256) s_hist[tid] = 0;
    if (tid == 0) { s_nA = 0; s_nS = 0; s_T = 256; }
    __syncthreads();
    // (the first SEL_RC x 1024 candidates are read ONCE, all loads of a thread in flight together, and kept in registers for the
    // compaction below: a loop of load -> LDS atomic exposes the memory latency once per iteration)
    uint32_t creg[SEL_RC];
#pragma unroll
    for (int j = 0; j < SEL_RC; j++) { const int i = tid + j * SEL_THREADS; creg[j] = i < ncand ? cand[i] : 0u; }
#pragma unroll
    for (int j = 0; j < SEL_RC; j++) if (tid + j * SEL_THREADS < ncand) atomicAdd(&s_hist[creg[j] >> 24], 1u);
    for (int i = tid + SEL_RC * SEL_THREADS; i < ncand; i += SEL_THREADS) atomicAdd(&s_hist[cand[i] >> 24], 1u);
    __syncthreads();
    if (tid < 64) {
        // retainBest(n1) as a set: keep score s iff #(score > s) < n1, i.e. T = the lowest such s; n1 == 0 clears (T = 256)
        uint32_t h[4], inc[4];
        hist_suffix(s_hist, tid, h, inc);
        int Tl = 256;
#pragma unroll
        for (int q = 3; q >= 0; q--) if ((int)(inc[q] - h[q]) < n1) Tl = 4 * tid + q;   // (true for a range of bins that ends at 255)
        const int Tw = wave_min_dpp(Tl);
        if (tid == 0) { s_T = Tw; if (Tw >= 256) s_nKeep = 0; }
        if (Tw < 256 && tid == (Tw >> 2)) s_nKeep = (int)inc[Tw & 3];   // candidates with score >= T
    }
    __syncthreads();
    const int T = s_T;
    const bool clear_all = (nf == 0) || (harris && n1 == 0);
    if (s_nKeep <= od.max_kpts) {
        // ---- the kept candidates fit the survivor storage in LDS (the normal case: 2 nf + ties): candidate and response live there
        // as pairs from here on; no global round trip between the steps
        auto keep_one = [&](int i, uint32_t c) {
            bool keep = i < ncand && (int)(c >> 24) >= T;
            unsigned long long bal = __ballot(keep);
            int base = 0;
            if ((tid & 63) == 0 && bal) base = atomicAdd(&s_nA, __popcll(bal));
            base = __builtin_amdgcn_readfirstlane(base);   // (every lane is active here: lane 0 holds the reservation)
            if (keep) surv[base + __popcll(bal & ((1ull << (tid & 63)) - 1ull))].x = c;
        };
#pragma unroll
        for (int j = 0; j < SEL_RC; j++) if (j * SEL_THREADS < ncand) keep_one(tid + j * SEL_THREADS, creg[j]);
        for (int i0 = SEL_RC * SEL_THREADS; i0 < ncand; i0 += SEL_THREADS) { const int i = i0 + tid; keep_one(i, i < ncand ? cand[i] : 0u); }
        __syncthreads();
        const int nA = s_nA;
        for (int i = tid; i < nA; i += SEL_THREADS) {
            const uint32_t c = surv[i].x;
            surv[i].y = __float_as_uint(harris ? harris_response(img, lv.pitch, (int)(c & 0xFFF), (int)((c >> 12) & 0xFFF)) : (float)(c >> 24));
        }
        __syncthreads();
        uint32_t Tkey = 0; // keep everything
        if (harris && nf > 0 && nA > nf) {
            if (tid == 0) { s_prefix = 0; s_mask = 0; s_k = nf; }
            __syncthreads();
            for (int shift = 24; shift >= 0; shift -= 8) {
                if (tid < 256) s_hist[tid] = 0;
                __syncthreads();
                const uint32_t prefix = s_prefix, mask = s_mask;
                for (int i = tid; i < nA; i += SEL_THREADS) {
                    uint32_t k = float_key(__uint_as_float(surv[i].y));
                    if ((k & mask) == prefix) atomicAdd(&s_hist[(k >> shift) & 255], 1u);
                }
                __syncthreads();
                if (tid < 64) radix_step(s_hist, tid, shift, prefix, mask, &s_k, &s_prefix, &s_mask);
                __syncthreads();
            }
            Tkey = s_prefix;
        }
        // survivors, compacted in place: entry i moves to pos <= i, and every entry of a chunk is read before any is written
        for (int i0 = 0; i0 < nA; i0 += SEL_THREADS) {
            int i = i0 + tid;
            const uint2 e = i < nA ? surv[i] : make_uint2(0u, 0u);
            bool keep = i < nA && !clear_all && float_key(__uint_as_float(e.y)) >= Tkey;
            unsigned long long bal = __ballot(keep);
            int base = 0;
            if ((tid & 63) == 0 && bal) base = atomicAdd(&s_nS, __popcll(bal));
            base = __builtin_amdgcn_readfirstlane(base);   // (every lane is active here: lane 0 holds the reservation)
            int pos = base + __popcll(bal & ((1ull << (tid & 63)) - 1ull));
            __syncthreads();
            if (keep) surv[pos] = make_uint2(e.x & 0xFFFFFFu, e.y);
            __syncthreads();
        }
    } else {
        // ---- more kept candidates than the LDS storage holds (a long run of equal FAST scores): through the global lists
        // compact the kept candidates (unordered; order is restored by the final rank sort)
        for (int i0 = 0; i0 < ncand; i0 += SEL_THREADS) {
            int i = i0 + tid;
            uint32_t c = i < ncand ? cand[i] : 0;
            bool keep = i < ncand && (int)(c >> 24) >= T;
            unsigned long long bal = __ballot(keep);
            int base = 0;
            if ((tid & 63) == 0 && bal) base = atomicAdd(&s_nA, __popcll(bal));
            base = __builtin_amdgcn_readfirstlane(base);   // (every lane is active here: lane 0 holds the reservation)
            if (keep) candA[base + __popcll(bal & ((1ull << (tid & 63)) - 1ull))] = c;
        }
        __syncthreads();
        const int nA = s_nA;
        for (int i = tid; i < nA; i += SEL_THREADS) {
            uint32_t c = candA[i];
            respA[i] = harris ? harris_response(img, lv.pitch, (int)(c & 0xFFF), (int)((c >> 12) & 0xFFF)) : (float)(c >> 24);
        }
        __syncthreads();
        // second retainBest(nf) by response: radix-select the nf-th largest key, keep key >= it
        uint32_t Tkey = 0; // keep everything
        if (harris && nf > 0 && nA > nf) {
            if (tid == 0) { s_prefix = 0; s_mask = 0; s_k = nf; }
            __syncthreads();
            for (int shift = 24; shift >= 0; shift -= 8) {
                if (tid < 256) s_hist[tid] = 0;
                __syncthreads();
                const uint32_t prefix = s_prefix, mask = s_mask;
                for (int i = tid; i < nA; i += SEL_THREADS) {
                    uint32_t k = float_key(respA[i]);
                    if ((k & mask) == prefix) atomicAdd(&s_hist[(k >> shift) & 255], 1u);
                }
                __syncthreads();
                if (tid < 64) radix_step(s_hist, tid, shift, prefix, mask, &s_k, &s_prefix, &s_mask);
                __syncthreads();
            }
            Tkey = s_prefix;
        }
        // survivors -> LDS
        for (int i0 = 0; i0 < nA; i0 += SEL_THREADS) {
            int i = i0 + tid;
            float r = i < nA ? respA[i] : 0.f;
            bool keep = i < nA && !clear_all && float_key(r) >= Tkey;
            unsigned long long bal = __ballot(keep);
            int base = 0;
            if ((tid & 63) == 0 && bal) base = atomicAdd(&s_nS, __popcll(bal));
            base = __builtin_amdgcn_readfirstlane(base);   // (every lane is active here: lane 0 holds the reservation)
            int pos = base + __popcll(bal & ((1ull << (tid & 63)) - 1ull));
            if (keep && pos < od.max_kpts) surv[pos] = make_uint2(candA[i] & 0xFFFFFFu, __float_as_uint(r));
        }
    }
    __syncthreads();
    const int nS = s_nS;
    if (nS > od.max_kpts) { // response ties overflow the per-frame capacity
        if (tid == 0) { od.sel_cnt[b * od.L + l] = 0; atomicMin(&od.frame_status[b], (int)MSLAM_ERR_CAPACITY); }
        return;
    }
    // rank sort by key = y<<12 | x (unique) -> row-major order inside the level
    uint2* out = od.sel + ((size_t)b * od.L + l) * od.max_kpts;
    for (int i = tid; i < nS; i += SEL_THREADS) {
        uint2 me = surv[i];
        int rank = 0;
        for (int j = 0; j < nS; j++) rank += surv[j].x < me.x;
        out[rank] = me;
    }
    if (tid == 0) od.sel_cnt[b * od.L + l] = nS;
}

// ---------------------------------------------------------------- K4-K6: describe
constexpr int DP_R = 21;               // raw patch radius: 18 (pattern reach) + 3 (blur)
constexpr int DP_N = 2 * DP_R + 1;     // 43
constexpr int DP_PITCH = 48;           // LDS bytes per raw row: three 16-byte LDS-DMA chunks (45 bytes + misalignment)
constexpr int DH_RP = 22, DH_CG = 10;  // row pass: 22 row pairs x 10 groups of 4 columns (44 x 40 >= 43 x 37)
constexpr int HT_DW = 22;              // dwords per column of the transposed row-pass result (44 u16 rows)
constexpr int DESC_WAVES = 4;

__device__ __forceinline__ float fast_atan2_deg(float y, float x)
{
    const float p1 = 0.9997878412794807f * (float)(180 / 3.1415926535897932384626433832795);
    const float p3 = -0.3258083974640975f * (float)(180 / 3.1415926535897932384626433832795);
    const float p5 = 0.1555786518463281f * (float)(180 / 3.1415926535897932384626433832795);
    const float p7 = -0.04432655554792128f * (float)(180 / 3.1415926535897932384626433832795);
    float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)2.2204460492503131e-16);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)2.2204460492503131e-16);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// Row pass of the 7x7 Gaussian for one alignment XO of the patch inside its dword-aligned LDS rows.  The four output columns
// of an item read source bytes XO + k .. XO + k + 6 of the 16 the item's four dwords hold; instead of shifting the bytes
// into place (nine v_alignbyte per row) the taps are shifted: weight dword m of output k holds g[4 m + b - (XO + k)] in byte
// b (zero outside 0..6), which leaves two or three v_dot4 per output -- ten per row for every XO, and no byte shuffles.
__host__ __device__ constexpr uint32_t blur_w(int xo, int k, int m)
{
    constexpr int g[7] = { 18, 34, 49, 55, 49, 34, 18 };
    uint32_t w = 0;
    for (int b = 0; b < 4; b++) {
        const int j = 4 * m + b - (xo + k);
        if (j >= 0 && j <= 6) w |= (uint32_t)g[j] << (8 * b);
    }
    return w;
}

template <int XO>
__device__ __forceinline__ void blur_rows(const uint32_t* __restrict__ raw32, uint32_t* __restrict__ hT, int lane)
{
    int rp = lane / DH_CG, cg = lane - rp * DH_CG; // item = rp * DH_CG + cg advances by 64 = 6 * DH_CG + 4
    for (int i = lane; i < DH_RP * DH_CG; i += 64, rp += 6, cg += 4) {
        if (cg >= DH_CG) { cg -= DH_CG; rp++; }
        uint32_t h[2][4];
#pragma unroll
        for (int dr = 0; dr < 2; dr++) {
            const uint32_t* q = raw32 + (2 * rp + dr) * (DP_PITCH / 4) + cg;
            const uint32_t d[4] = { q[0], q[1], q[2], q[3] };
#pragma unroll
            for (int k = 0; k < 4; k++) {
                uint32_t acc = 0;
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    const uint32_t w = blur_w(XO, k, m);
                    if (w) acc = __builtin_amdgcn_udot4(d[m], w, acc, false);   // (resolved at compile time: w is a constant)
                }
                h[dr][k] = acc; // <= 65535
            }
        }
#pragma unroll
        for (int k = 0; k < 4; k++) hT[(4 * cg + k) * HT_DW + rp] = h[0][k] | (h[1][k] << 16);
    }
}

struct DescArgs {
    mslam_keypoint* kps; uint8_t* desc; int32_t* n_out; int32_t* status; int cap;
};

__global__ __launch_bounds__(64 * DESC_WAVES)
void describe_k(OrbDev od, DescArgs out)
{
    __shared__ uint32_t s_raw[DESC_WAVES][2 * DH_RP * DP_PITCH / 4 + 4];   // (+4: the last row's column groups read a dword past it)
    __shared__ uint32_t s_h[DESC_WAVES][4 * DH_CG * HT_DW];
    // (the wavefront number as a scalar: the keypoint index and the level search below then run on the scalar unit -- as vector
    // code the 16-level search was 85 of the kernel's ~730 vector instructions)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const unsigned wg = xcd_contiguous(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);
    const int b = (int)(wg / gridDim.x), bx = (int)(wg % gridDim.x);
    const int g = bx * DESC_WAVES + wave; // keypoint index inside the frame
    // level offsets (<= 16 counts)
    int total = 0, l = -1, off = 0;
    const int st = od.frame_status[b];
    {   // (all counts fetched before the first is used: one memory latency, not one per level)
        int cnt[MSLAM_MAX_LEVELS];
#pragma unroll
        for (int i = 0; i < MSLAM_MAX_LEVELS; i++) cnt[i] = i < od.L ? od.sel_cnt[b * od.L + i] : 0;
#pragma unroll
        for (int i = 0; i < MSLAM_MAX_LEVELS; i++) {
            if (l < 0 && i < od.L && g < total + cnt[i]) { l = i; off = total; }
            total += cnt[i];
        }
    }
    const bool bad = st != MSLAM_OK || total > out.cap;
    if (g == 0 && lane == 0) {
        out.n_out[b] = bad ? 0 : total;
        if (out.status) out.status[b] = bad ? (int)MSLAM_ERR_CAPACITY : (int)MSLAM_OK;
    }
    // blocks entirely beyond the keypoint count have nothing to do (the grid is sized for the capacity); the condition is
    // uniform over the block, so no barrier is skipped by part of it
    if (bx > 0 && (bad || bx * DESC_WAVES >= total)) return;
    const bool active = !bad && l >= 0; // wave-uniform; inactive waves of a partial block still reach the barriers
    if (l < 0) l = 0;
    const LevelDesc& lv = od.lv[l];
    const uint2 sel = active ? od.sel[((size_t)b * od.L + l) * od.max_kpts + (g - off)] : make_uint2((DP_R << 12) | DP_R, 0);
    const int x = (int)(sel.x & 0xFFF), y = (int)(sel.x >> 12);
    const uint8_t* img = lv.img + b * lv.frame_stride;
    // ---- raw 43x43 patch by LDS-DMA (always inside the image: edge >= 22): 43 rows x 3 chunks of 16 bytes from the dword-aligned
    // column xa, lane-linear in LDS = row-major with a pitch of 48 bytes; 3 instructions per keypoint, no VGPR round trip.
    // A chunk may run up to 4 bytes past the end of an image row into the next one (rows <= h - 2 here): columns nothing reads.
    const int px0 = x - DP_R, py0 = y - DP_R;
    const int xa = px0 & ~3, xo = px0 & 3;
    uint32_t* raw32 = s_raw[wave];
    // The lane's constant-table entries -- IC-angle weights of its half row, pattern points of its four tests -- are fetched here, so
    // that their latency passes under the patch staging: read where they are used they sit behind the barriers (a load is not moved
    // across one), one exposed memory latency each.
    const int icv = (min(lane, 61) >> 1) - 15;
    const uint4* wt = reinterpret_cast<const uint4*>(c_icw[abs(icv) * 2 + (lane & 1)]);
    const uint4 wx = wt[0], wm = wt[1];
    uint32_t pat[4];
#pragma unroll
    for (int j = 0; j < 4; j++) pat[j] = reinterpret_cast<const uint32_t*>(c_pattern)[j * 64 + lane];
    {
        constexpr int NCH = DP_N * (DP_PITCH / 16);
        const uint8_t* g0 = img + (size_t)py0 * lv.pitch + xa;
#pragma unroll
        for (int c0 = 0; c0 < NCH; c0 += 64) {
            const int c = c0 + lane, r = c / (DP_PITCH / 16), k = c - r * (DP_PITCH / 16);
            if (active && c < NCH) glds16(g0 + __umul24((unsigned)r, (unsigned)lv.pitch) + 16 * k, reinterpret_cast<uint8_t*>(raw32) + c0 * 16);
        }
    }
    __syncthreads();
    // ---- IC_Angle: lane (v + 15) * 2 + half sums one half of row v of the circular patch (integer sums: any order) with packed
    // dot products: sum (u + 15) I(u) and sum I(u) over the half row's 16 columns against the weight bytes of c_icw (zero
    // outside the circle); m10 = the first minus 15 times the second
    int m10 = 0, m01 = 0;
    if (lane < 62) {
        const int v = (lane >> 1) - 15, half = lane & 1;
        const int start = (DP_R + v) * DP_PITCH + DP_R + (half ? 1 : -15) + xo;   // LDS byte of the half row's first column
        const uint32_t* q = raw32 + (start >> 2);
        const int sh = start & 3;
        const uint32_t d0 = q[0], d1 = q[1], d2 = q[2], d3 = q[3], d4 = q[4];
        const uint32_t e0 = __builtin_amdgcn_alignbyte(d1, d0, sh), e1 = __builtin_amdgcn_alignbyte(d2, d1, sh),
                       e2 = __builtin_amdgcn_alignbyte(d3, d2, sh), e3 = __builtin_amdgcn_alignbyte(d4, d3, sh);
        const uint32_t sx = __builtin_amdgcn_udot4(e3, wx.w, __builtin_amdgcn_udot4(e2, wx.z, __builtin_amdgcn_udot4(e1, wx.y, __builtin_amdgcn_udot4(e0, wx.x, 0u, false), false), false), false);
        const uint32_t rs = __builtin_amdgcn_udot4(e3, wm.w, __builtin_amdgcn_udot4(e2, wm.z, __builtin_amdgcn_udot4(e1, wm.y, __builtin_amdgcn_udot4(e0, wm.x, 0u, false), false), false), false);
        m10 = (int)sx - 15 * (int)rs;
        m01 = __mul24(v, (int)rs);
    }
    m10 = wave_sum_dpp(m10); m01 = wave_sum_dpp(m01);
    const float angle = fast_atan2_deg((float)m01, (float)m10);
    // ---- 7x7 Gaussian of the patch with packed dot products (integer arithmetic: any association is exact).
    // Rows: one item = 2 patch rows x 4 output columns by v_dot4_u32_u8 (blur_rows, one instantiation per byte alignment of the
    // patch); the result is stored transposed (hT[col][row], u16) so that the column taps of a pixel are vertical neighbours
    // packed in consecutive dwords for v_dot2_u32_u16.  Rows/columns beyond 37 are scratch (computed from whatever LDS holds,
    // never read back).
    uint32_t* hT = s_h[wave];
    switch (xo) {   // wave-uniform
    case 0: blur_rows<0>(raw32, hT, lane); break;
    case 1: blur_rows<1>(raw32, hT, lane); break;
    case 2: blur_rows<2>(raw32, hT, lane); break;
    default: blur_rows<3>(raw32, hT, lane); break;
    }
    __syncthreads();
    // ---- 256 rotated binary tests; test t = j*64 + lane, ballot j = descriptor bytes 8j..8j+7
    float ang = angle;
    ang *= (float)(3.1415926535897932384626433832795 / 180.f);
    // (float)cos((double)ang), (float)sin((double)ang): quadrant-reduced kernels instead of the library calls (5x fewer
    // f64 instructions); equal to libm after the float cast for every float in [0, 2 pi] (tests/trig_check.c, exhaustive)
    double sd, cd;
    mslam_sincos_2pi((double)ang, &sd, &cd);
    const float ca = (float)cd, sb = (float)sd;
    // The column pass of the blur is evaluated only where the tests look: the blurred pixel (c, r) (column / row offsets
    // 0..36 from the patch corner) is min((sum_k g[k] hT[c][r + k] + 2^15) >> 16, 255), seven vertical neighbours that sit in
    // four consecutive dwords of the transposed row-pass result; an odd r starts in the high half of the first dword, which the
    // weight pairs absorb ({0,18},{34,49},{55,49},{34,18} instead of {18,34},{49,55},{49,34},{18,0}).  512 samples per
    // keypoint instead of the 37 x 37 blurred patch (1369 pixels): the same integers, fewer of them.
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    auto blurred = [&](int cx, int ry) -> int {   // cx, ry in [-18, 18] (pattern reach), relative to the keypoint
        const int c = cx + 18, r = ry + 18;
        const uint32_t* q = hT + __mul24(c, HT_DW) + (r >> 1);
        const bool odd = r & 1;
        const uint32_t w0 = odd ? (18u << 16) : (18u | (34u << 16)), w1 = odd ? (34u | (49u << 16)) : (49u | (55u << 16)),
                       w2 = odd ? (55u | (49u << 16)) : (49u | (34u << 16)), w3 = odd ? (34u | (18u << 16)) : 18u;
        uint32_t o = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, q[0]), __builtin_bit_cast(us2, w0), 1u << 15, false);
        o = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, q[1]), __builtin_bit_cast(us2, w1), o, false);
        o = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, q[2]), __builtin_bit_cast(us2, w2), o, false);
        o = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, q[3]), __builtin_bit_cast(us2, w3), o, false);
        return (int)min(o >> 16, 255u);
    };
    unsigned long long bits[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int8_t pt[4] = { (int8_t)(pat[j] & 255u), (int8_t)((pat[j] >> 8) & 255u), (int8_t)((pat[j] >> 16) & 255u), (int8_t)(pat[j] >> 24) };
        float x0 = pt[0] * ca - pt[1] * sb, y0 = pt[0] * sb + pt[1] * ca;
        float x1 = pt[2] * ca - pt[3] * sb, y1 = pt[2] * sb + pt[3] * ca;
        const int t0 = blurred(__float2int_rn(x0), __float2int_rn(y0));
        const int t1 = blurred(__float2int_rn(x1), __float2int_rn(y1));
        bits[j] = __ballot(t0 < t1);
    }
    if (!active) return;
    if (lane < 4) {
        unsigned long long v = lane == 0 ? bits[0] : lane == 1 ? bits[1] : lane == 2 ? bits[2] : bits[3];
        reinterpret_cast<unsigned long long*>(out.desc + ((size_t)b * out.cap + g) * 32)[lane] = v;
    }
    if (lane == 0) {
        // coordinates: detect() scales level coordinates by sf, compute() scales by 1/sf, describes at
        // cvRound() of that (== x, y) and scales by sf again -- three float products (ORB::operator())
        float fx = (float)x, fy = (float)y;
        if (l != 0) {
            fx *= lv.sf; fy *= lv.sf;
            fx *= lv.inv_sf; fy *= lv.inv_sf;
            fx *= lv.sf; fy *= lv.sf;
        }
        mslam_keypoint k;
        k.x = fx; k.y = fy; k.size = 31 * lv.sf; k.angle = angle;
        k.response = __uint_as_float(sel.y); k.octave = l;
        out.kps[(size_t)b * out.cap + g] = k;
    }
}

// standalone full-image blur (test hook / K5): same taps, BORDER_REFLECT_101
__device__ __forceinline__ int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while ((unsigned)p >= (unsigned)len) p = p < 0 ? -p : 2 * (len - 1) - p;
    return p;
}
__global__ void blur7_k(const uint8_t* __restrict__ src, int w, int h, int pitch, uint8_t* __restrict__ dst)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    int acc = 0;
    for (int j = 0; j < 7; j++) {
        const uint8_t* r = src + (size_t)reflect101(y + j - 3, h) * pitch;
        int hacc = 0;
        for (int i = 0; i < 7; i++) hacc += c_gk[i] * r[reflect101(x + i - 3, w)];
        acc += c_gk[j] * hacc;
    }
    int v = (acc + (1 << 15)) >> 16;
    dst[(size_t)y * w + x] = (uint8_t)(v > 255 ? 255 : v);
}

inline int cv_round_host(double v) { return (int)std::lrint(v); }
inline int cv_floor_host(double v) { int i = (int)v; return i - (i > v); }
inline int sat_short_host(int v) { return v < -32768 ? -32768 : v > 32767 ? 32767 : v; }

} // namespace

// geometry-dependent plan: level sizes, resize tables, scratch buffers
struct OrbPlan {
    int w = 0, h = 0, B = 0, channels = 0;
    mslam_orb_params prm{};
    OrbDev dev{};
    void* arena = nullptr;     // one allocation for everything below
    size_t arena_bytes = 0;
    uint8_t* level0 = nullptr; // internal level-0 buffer (BGR input or unaligned gray)
    int level0_pitch = 0;
    long long level0_fstride = 0;
    uint8_t* last_l0 = nullptr; int last_l0_pitch = 0;      // level 0 of the last call (may alias caller memory)
    uint32_t* counters = nullptr; size_t counters_bytes = 0; // cnt + sel_cnt + frame_status (zeroed per batch)
};

void orb_plan_destroy(mslam_ctx* ctx)
{
    if (!ctx->orb) return;
    ctx_graph_clear(ctx); // captured chains point into the plan's arena
    if (ctx->orb->arena) (void)hipFree(ctx->orb->arena);
    delete ctx->orb;
    ctx->orb = nullptr;
}

static bool same_params(const mslam_orb_params& a, const mslam_orb_params& b)
{
    return a.nfeatures == b.nfeatures && a.scale_factor == b.scale_factor && a.nlevels == b.nlevels &&
           a.edge_threshold == b.edge_threshold && a.first_level == b.first_level && a.wta_k == b.wta_k &&
           a.score_type == b.score_type && a.patch_size == b.patch_size && a.fast_threshold == b.fast_threshold;
}

static int orb_plan_get(mslam_ctx* ctx, int w, int h, int B, int channels, const mslam_orb_params* p, OrbPlan** out)
{
    if (ctx->orb && ctx->orb->w == w && ctx->orb->h == h && ctx->orb->B >= B && ctx->orb->channels == channels &&
        same_params(ctx->orb->prm, *p)) {
        *out = ctx->orb;
        return MSLAM_OK;
    }
    if (p->nlevels < 1 || p->nlevels > MSLAM_MAX_LEVELS || p->first_level != 0 || p->wta_k != 2 || p->patch_size != 31 ||
        !(p->scale_factor > 1.0f) || p->edge_threshold < 22 || p->edge_threshold > 255 || p->nfeatures < 0 ||
        p->fast_threshold < 1 || p->fast_threshold > 254 || (p->score_type != 0 && p->score_type != 1))
        return MSLAM_ERR_INVALID;
    if (w > ctx->max_w || h > ctx->max_h || B > ctx->batch) return MSLAM_ERR_CAPACITY;
    MSLAM_TRY(ctx_set_device(ctx));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    orb_plan_destroy(ctx);
    OrbPlan* pl = new OrbPlan();
    ctx->orb = pl;
    pl->w = w; pl->h = h; pl->B = B; pl->channels = channels; pl->prm = *p;
    OrbDev& od = pl->dev;
    od.L = p->nlevels; od.B = B; od.edge = p->edge_threshold;
    od.max_kpts = ctx->max_kpts < SEL_LEVEL_CAP ? ctx->max_kpts : SEL_LEVEL_CAP; // per-level list capacity
    od.fast_thr = p->fast_threshold; od.score_type = p->score_type;
    od.ablate = getenv("MSLAM_FAST_ABLATE") ? atoi(getenv("MSLAM_FAST_ABLATE")) : 0; // development only
    const int L = od.L, edge = od.edge;
    // level geometry and feature budget -- ORB::operator() / computeKeyPoints (OpenCV 2.4.13)
    const double scaleFactor = (double)p->scale_factor;
    {
        float factor = (float)(1.0 / scaleFactor);
        float nd = p->nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)L));
        int sum = 0;
        for (int l = 0; l < L - 1; l++) {
            od.lv[l].nfeat = cv_round_host(nd);
            sum += od.lv[l].nfeat;
            nd *= factor;
        }
        od.lv[L - 1].nfeat = p->nfeatures - sum > 0 ? p->nfeatures - sum : 0;
    }
    size_t bytes = 0;
    auto reserve = [&](size_t n) { bytes = align_up(bytes, 256); size_t o = bytes; bytes += n; return o; };
    size_t off_img[MSLAM_MAX_LEVELS], off_cand[MSLAM_MAX_LEVELS], off_candA[MSLAM_MAX_LEVELS], off_resp[MSLAM_MAX_LEVELS],
        off_xt[MSLAM_MAX_LEVELS], off_yt[MSLAM_MAX_LEVELS], off_p2c[MSLAM_MAX_LEVELS], off_p2r[MSLAM_MAX_LEVELS];
    int tile_base = 0;
    for (int l = 0; l < L; l++) {
        LevelDesc& lv = od.lv[l];
        float s = (float)std::pow(scaleFactor, (double)l);
        float inv = 1 / s;
        lv.sf = s; lv.inv_sf = inv;
        lv.w = cv_round_host(w * inv); lv.h = cv_round_host(h * inv);
        if (lv.w < 1 || lv.h < 1) { orb_plan_destroy(ctx); return MSLAM_ERR_INVALID; }   // (no half-built plan left for the cache check above)
        lv.pitch = (int)align_up((size_t)lv.w, 64);
        lv.frame_stride = (long long)lv.pitch * lv.h;
        off_img[l] = reserve((size_t)lv.frame_stride * B);
        lv.roi_x0 = edge; lv.roi_y0 = edge;
        lv.roi_w = lv.w - 2 * edge; lv.roi_h = lv.h - 2 * edge;
        if (lv.roi_w <= 0 || lv.roi_h <= 0) { lv.roi_w = 0; lv.roi_h = 0; }
        lv.tile_x0 = lv.roi_x0 - ((lv.roi_x0 - 1) & 3);
        lv.tiles_x = lv.roi_w > 0 ? (lv.roi_x0 + lv.roi_w - lv.tile_x0 + FT_W - 1) / FT_W : 0; lv.tiles_y = (lv.roi_h + FT_H - 1) / FT_H;
        lv.tile_base = tile_base;
        tile_base += lv.tiles_x * lv.tiles_y;
        lv.cand_cap = ((lv.roi_w + 1) / 2) * ((lv.roi_h + 1) / 2) + 1;
        off_cand[l] = reserve(sizeof(uint32_t) * (size_t)lv.cand_cap * B);
        off_candA[l] = reserve(sizeof(uint32_t) * (size_t)lv.cand_cap * B);
        off_resp[l] = reserve(sizeof(float) * (size_t)lv.cand_cap * B);
        off_xt[l] = reserve(sizeof(int2) * (align_up((size_t)lv.w, 4) + 4));   // padded: column groups of 4 load two int4 without clamps
        off_yt[l] = reserve(sizeof(int4) * (size_t)lv.h);
        off_p2c[l] = reserve(sizeof(int4) * 2 * (size_t)((lv.w + 127) / 128));
        off_p2r[l] = reserve(sizeof(int4) * 2 * (size_t)((lv.h + 15) / 16));
    }
    od.tiles_per_frame = tile_base;
    if (tile_base > 1) {   // fast_nms_k divides its workgroup index by tiles_per_frame with one multiply: (n * magic) >> 32 == n / tpf
        // for every n of a launch while n * (magic * tpf - 2^32) < 2^32 (fails only beyond ~10^4 tiles x 64 frames: refused)
        const uint64_t tpf = (uint64_t)tile_base, magic = (1ull << 32) / tpf + 1;
        if (tpf * (uint64_t)B * (magic * tpf - (1ull << 32)) >= (1ull << 32)) { orb_plan_destroy(ctx); return MSLAM_ERR_INVALID; }
    }
    const size_t off_ft = reserve(sizeof(int4) * (size_t)std::max(tile_base, 1));
    const size_t off_sel = reserve(sizeof(uint2) * (size_t)B * L * od.max_kpts);
    const size_t n_cnt = (size_t)B * L;
    pl->counters_bytes = sizeof(uint32_t) * ((CNT_STRIDE + 1) * n_cnt + B);
    const size_t off_cnt = reserve(pl->counters_bytes);
    HIP_TRY(ctx, hipMalloc(&pl->arena, bytes));
    pl->arena_bytes = bytes;
    char* base = (char*)pl->arena;
    pl->counters = (uint32_t*)(base + off_cnt);
    od.cnt = pl->counters;
    od.sel_cnt = (int32_t*)(od.cnt + CNT_STRIDE * n_cnt);
    od.frame_status = od.sel_cnt + n_cnt;
    od.sel = (uint2*)(base + off_sel);
    {   // fast_nms_k's tile map
        std::vector<int4> ft;
        for (int l = 0; l < L; l++)
            for (int ty = 0; ty < od.lv[l].tiles_y; ty++)
                for (int tx = 0; tx < od.lv[l].tiles_x; tx++) ft.push_back(make_int4(l, tx, ty, 0));
        if (!ft.empty()) HIP_TRY(ctx, hipMemcpy(base + off_ft, ft.data(), sizeof(int4) * ft.size(), hipMemcpyHostToDevice));
        od.fast_tiles = (const int4*)(base + off_ft);
        od.tpf_magic = tile_base > 1 ? (uint32_t)((1ull << 32) / (uint64_t)tile_base + 1) : 0u;
    }
    std::vector<int4> tab;
    std::vector<std::vector<int2>> h_xt((size_t)L);
    std::vector<std::vector<int4>> h_yt((size_t)L);
    for (int l = 0; l < L; l++) {
        LevelDesc& lv = od.lv[l];
        lv.fused_with_next = 0;
        lv.img = (uint8_t*)(base + off_img[l]);
        lv.cand = (uint32_t*)(base + off_cand[l]);
        lv.candA = (uint32_t*)(base + off_candA[l]);
        lv.respA = (float*)(base + off_resp[l]);
        lv.xtab = (int2*)(base + off_xt[l]);
        lv.ytab = (int4*)(base + off_yt[l]);
        lv.p2col = (int4*)(base + off_p2c[l]);
        lv.p2row = (int4*)(base + off_p2r[l]);
        if (l == 0) continue;
        // resize(INTER_LINEAR) coefficient tables for level l-1 -> l (OpenCV 2.4.13 imgwarp.cpp)
        const int sw = od.lv[l - 1].w, sh = od.lv[l - 1].h, dw = lv.w, dh = lv.h;
        const double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
        const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
        std::vector<int2> xt((size_t)dw);
        for (int dx = 0; dx < dw; dx++) {
            float fx = (float)((dx + 0.5) * scale_x - 0.5);
            int sx = cv_floor_host(fx);
            fx -= sx;
            if (sx < 0) { fx = 0; sx = 0; }
            if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
            const int sx1 = sx + 1 < sw ? sx + 1 : sx;
            const int a0 = sat_short_host(cv_round_host((1.f - fx) * 2048)), a1 = sat_short_host(cv_round_host(fx * 2048));
            xt[(size_t)dx] = make_int2(sx | (sx1 << 16), (a0 & 0xFFFF) | (a1 << 16)); // 0 <= a0, a1 <= 2048
        }
        h_xt[(size_t)l] = xt;
        xt.resize(align_up((size_t)dw, 4) + 4, xt.back());   // the device copy repeats the last column: groups of 4 need no clamp
        HIP_TRY(ctx, hipMemcpy(lv.xtab, xt.data(), sizeof(int2) * xt.size(), hipMemcpyHostToDevice));
        tab.assign((size_t)dh, make_int4(0, 0, 0, 0));
        for (int dy = 0; dy < dh; dy++) {
            float fy = (float)((dy + 0.5) * scale_y - 0.5);
            int sy = cv_floor_host(fy);
            fy -= sy;
            int s0 = sy < 0 ? 0 : (sy < sh ? sy : sh - 1);
            int s1 = sy + 1 < 0 ? 0 : (sy + 1 < sh ? sy + 1 : sh - 1);
            tab[dy] = make_int4(s0, s1, sat_short_host(cv_round_host((1.f - fy) * 2048)),
                                sat_short_host(cv_round_host(fy * 2048)));
        }
        HIP_TRY(ctx, hipMemcpy(lv.ytab, tab.data(), sizeof(int4) * (size_t)dh, hipMemcpyHostToDevice));
        h_yt[(size_t)l] = tab;
    }
    // Which level pairs (A = l, B = l + 1) can be produced together from level l - 1: every 128 x 16 tile of B must keep its
    // A region and that region's source footprint inside pyramid2_k's LDS bounds (true for scale factors up to ~1.25)
    // Level 1 (69 % of the pyramid's pixels together with level 2) stays a resize_k launch: fusing the pair (1, 2) costs 7 % more
    // time than two launches (the shared A pixels are recomputed), while the pairs (2, 3), (4, 5), (6, 7) are faster fused
    // (measured per 32 frames, tools/sweep_pyramid_fusion.sh: none 0.186 ms, from level 1: 0.194, from level 2: 0.180).
    // MSLAM_PYRAMID_FUSE (development): first level of the first fused pair (0 = no fusion).
    const int fuse_first = getenv("MSLAM_PYRAMID_FUSE") ? atoi(getenv("MSLAM_PYRAMID_FUSE")) : 2;
    for (int l = fuse_first; fuse_first >= 1 && l + 1 < L; l += 2) {
        const LevelDesc& A = od.lv[l];
        const LevelDesc& Bv = od.lv[l + 1];
        const std::vector<int2>& xtA = h_xt[(size_t)l]; const std::vector<int4>& ytA = h_yt[(size_t)l];
        const std::vector<int2>& xtB = h_xt[(size_t)l + 1]; const std::vector<int4>& ytB = h_yt[(size_t)l + 1];
        bool ok = true;
        // per tile column / row of B: the A region it owns and computes and that region's source footprint, so that a
        // workgroup reads two descriptors instead of walking four tables with dependent loads
        std::vector<int4> colD, rowD;
        for (int xB0 = 0; xB0 < Bv.w && ok; xB0 += P2_BW) {
            const int xB1 = std::min(xB0 + P2_BW, Bv.w) - 1;
            const bool lastx = xB0 + P2_BW >= Bv.w;
            const int ox0 = xB0 == 0 ? 0 : ((xtB[(size_t)xB0].x & 0xFFFF) & ~3);
            const int ox1 = lastx ? ((A.w + 3) & ~3) : ((xtB[(size_t)xB0 + P2_BW].x & 0xFFFF) & ~3);
            const int cx1 = lastx ? A.w - 1 : (int)((unsigned)xtB[(size_t)xB1].x >> 16);
            const int needed_lo = xtB[(size_t)xB0].x & 0xFFFF;
            if (cx1 - ox0 + 1 > P2_AW || cx1 < ox0 || ox0 > needed_lo) { ok = false; break; }
            const int sx_lo = xtA[(size_t)ox0].x & 0xFFFF, sx_hi = (int)((unsigned)xtA[(size_t)cx1].x >> 16);
            if (((sx_hi - (sx_lo & ~3)) >> 2) + 1 > P2_SDW) ok = false;
            colD.push_back(make_int4(ox0, ox1, cx1, sx_lo)); colD.push_back(make_int4(sx_hi, 0, 0, 0));
        }
        for (int yB0 = 0; yB0 < Bv.h && ok; yB0 += P2_BH) {
            const int yB1 = std::min(yB0 + P2_BH, Bv.h) - 1;
            const bool lasty = yB0 + P2_BH >= Bv.h;
            const int oy0 = yB0 == 0 ? 0 : ytB[(size_t)yB0].x;
            const int oy1 = lasty ? A.h : ytB[(size_t)yB0 + P2_BH].x;
            const int cy1 = lasty ? A.h - 1 : ytB[(size_t)yB1].y;
            if (cy1 - oy0 + 1 > P2_AH || cy1 < oy0 || oy0 > ytB[(size_t)yB0].x) { ok = false; break; }
            if (ytA[(size_t)cy1].y - ytA[(size_t)oy0].x + 1 > P2_SH) ok = false;
            rowD.push_back(make_int4(oy0, oy1, cy1, ytA[(size_t)oy0].x)); rowD.push_back(make_int4(ytA[(size_t)cy1].y, 0, 0, 0));
        }
        if (ok) {
            HIP_TRY(ctx, hipMemcpy(od.lv[l + 1].p2col, colD.data(), sizeof(int4) * colD.size(), hipMemcpyHostToDevice));
            HIP_TRY(ctx, hipMemcpy(od.lv[l + 1].p2row, rowD.data(), sizeof(int4) * rowD.size(), hipMemcpyHostToDevice));
        }
        od.lv[l].fused_with_next = ok ? 1 : 0;
    }
    pl->level0 = od.lv[0].img;
    pl->level0_pitch = od.lv[0].pitch;
    pl->level0_fstride = od.lv[0].frame_stride;
    *out = pl;
    return MSLAM_OK;
}

extern "C" {

void mslam_orb_default_params(mslam_orb_params* p)
{
    if (!p) return;
    p->nfeatures = 500; p->scale_factor = 1.2f; p->nlevels = 8; p->edge_threshold = 31; p->first_level = 0;
    p->wta_k = 2; p->score_type = 0; p->patch_size = 31; p->fast_threshold = 20;
}

int mslam_orb_extract_batch_dev(mslam_ctx* ctx, const uint8_t* imgs, int B, int w, int h, int stride,
                                int64_t frame_stride, int channels, const mslam_orb_params* params,
                                mslam_keypoint* kps, uint8_t* desc, int cap, int32_t* n_out, int32_t* status_dev)
{
    if (!ctx || !imgs || !params || !kps || !desc || !n_out || B < 1 || w < 16 || h < 16 || cap < 1 ||
        (channels != 1 && channels != 3 && channels != 4) || stride < w * channels || frame_stride < (int64_t)stride * h)
        return MSLAM_ERR_INVALID;
    if (((uintptr_t)desc & 7) || ((uintptr_t)kps & 3)) return MSLAM_ERR_INVALID;
    MSLAM_TRY(ctx_set_device(ctx)); // cached-plan path included: launches must not depend on the caller's current device
    OrbPlan* pl = nullptr;
    MSLAM_TRY(orb_plan_get(ctx, w, h, B, channels, params, &pl));
    OrbDev od = pl->dev;
    od.B = B;
    hipStream_t st = ctx->stream;
    // level 0: alias the caller's gray frames when dword loads are legal on them, else convert/copy
    const bool alias = channels == 1 && ((uintptr_t)imgs & 3) == 0 && (stride & 3) == 0 && (frame_stride & 3) == 0;
    if (alias) {
        od.lv[0].img = const_cast<uint8_t*>(imgs);
        od.lv[0].pitch = stride;
        od.lv[0].frame_stride = frame_stride;
    } else {
        od.lv[0].img = pl->level0; od.lv[0].pitch = pl->level0_pitch; od.lv[0].frame_stride = pl->level0_fstride;
    }
    pl->last_l0 = od.lv[0].img; pl->last_l0_pitch = od.lv[0].pitch;
    // optional graph replay: the whole chain below depends only on these values
    struct { const void* imgs; const void* kps; const void* desc; const void* n_out; const void* status; const void* plan;
             int64_t fstride; int B, w, h, stride, channels, cap; } key_src = { imgs, kps, desc, n_out, status_dev, pl, frame_stride, B, w, h, stride, channels, cap };
    const uint64_t key = ctx_hash(&key_src, sizeof(key_src), 14695981039346656037ull);
    const int g = ctx_graph_begin(ctx, key);
    if (g == 1) return MSLAM_OK;
    if (!alias)
        LAUNCH(ctx, K_GRAY, bgr2gray_k, dim3((w + 255) / 256, h, B), dim3(256), 0, imgs, w, h, stride,
                           (long long)frame_stride, channels, od.lv[0].img, od.lv[0].pitch, od.lv[0].frame_stride);
    hipError_t me = hipMemsetAsync(pl->counters, 0, pl->counters_bytes, st);
    for (int l = 1; l < od.L; l++) {
        const LevelDesc& s = od.lv[l - 1];
        const LevelDesc& d = od.lv[l];
        if (d.fused_with_next) {   // levels l and l + 1 from level l - 1 in one launch
            const LevelDesc& e = od.lv[l + 1];
            Pyr2Args pa{ s.img, s.pitch, s.frame_stride, d.img, d.w, d.h, d.pitch, d.frame_stride, d.xtab, d.ytab,
                         e.img, e.w, e.h, e.pitch, e.frame_stride, e.xtab, e.ytab, e.p2col, e.p2row };
            LAUNCH(ctx, K_PYRAMID2, pyramid2_k, dim3((e.w + P2_BW - 1) / P2_BW, (e.h + P2_BH - 1) / P2_BH, B), dim3(256), 0, pa);
            l++;
            continue;
        }
        LAUNCH(ctx, K_RESIZE, resize_k, dim3((d.w + RS_TW - 1) / RS_TW, (d.h + RS_TH - 1) / RS_TH, B), dim3(256), 0, s.img, s.pitch, s.frame_stride,
                           d.img, d.w, d.h, d.pitch, d.frame_stride, d.xtab, d.ytab);
    }
    if (od.tiles_per_frame > 0)
        LAUNCH(ctx, K_FAST_NMS, fast_nms_k, dim3(od.tiles_per_frame * B), dim3(FAST_THREADS), 0, od);
    LAUNCH(ctx, K_SELECT, select_k, dim3(od.L, B), dim3(SEL_THREADS), sizeof(uint2) * (size_t)od.max_kpts, od);
    DescArgs da{ kps, desc, n_out, status_dev, cap };
    const int max_total = cap < od.max_kpts * od.L ? cap : od.max_kpts * od.L;
    LAUNCH(ctx, K_DESCRIBE, describe_k, dim3((max_total + DESC_WAVES - 1) / DESC_WAVES + 1, B), dim3(64 * DESC_WAVES), 0, od, da);
    if (g == 0) MSLAM_TRY(ctx_graph_end(ctx, key)); // ends the capture on every path; launch errors surface here
    HIP_TRY(ctx, me);
    HIP_TRY(ctx, hipGetLastError());
    return MSLAM_OK;
}

int mslam_orb_extract(mslam_ctx* ctx, const uint8_t* img, int w, int h, int stride, int channels,
                      const mslam_orb_params* params, mslam_keypoint* kps, uint8_t* desc, int cap, int* n_out)
{
    if (!ctx || !img || !params || !kps || !desc || !n_out || cap < 1 || w < 16 || h < 16 || stride < w * channels)
        return MSLAM_ERR_INVALID;
    MSLAM_TRY(ctx_set_device(ctx));
    const size_t img_bytes = align_up((size_t)stride * h, 256);
    size_t bytes = img_bytes + align_up(sizeof(mslam_keypoint) * (size_t)cap, 256) + align_up((size_t)cap * 32, 256) + 512;
    void* stg = nullptr;
    MSLAM_TRY(ctx_stage(ctx, bytes, &stg));
    Bump bp{ (char*)stg, 0, bytes };
    uint8_t* dimg = bp.take<uint8_t>(img_bytes);
    mslam_keypoint* dk = bp.take<mslam_keypoint>(cap);
    uint8_t* dd = bp.take<uint8_t>((size_t)cap * 32);
    int32_t* dn = bp.take<int32_t>(2);
    HIP_TRY(ctx, hipMemcpyAsync(dimg, img, (size_t)stride * h, hipMemcpyHostToDevice, ctx->stream));
    MSLAM_TRY(mslam_orb_extract_batch_dev(ctx, dimg, 1, w, h, stride, (int64_t)img_bytes, channels, params, dk, dd, cap, dn, dn + 1));
    int32_t hn[2] = { 0, 0 };
    HIP_TRY(ctx, hipMemcpyAsync(hn, dn, sizeof(hn), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (hn[1] != MSLAM_OK) { *n_out = 0; return hn[1]; }
    *n_out = hn[0];
    if (hn[0] > 0) {
        HIP_TRY(ctx, hipMemcpyAsync(kps, dk, sizeof(mslam_keypoint) * (size_t)hn[0], hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(desc, dd, (size_t)hn[0] * 32, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    return MSLAM_OK;
}

int mslam_orb_debug_level(mslam_ctx* ctx, int level, int blurred, uint8_t* out, int* level_w, int* level_h)
{
    if (!ctx || !ctx->orb || !out || !level_w || !level_h || level < 0 || level >= ctx->orb->dev.L) return MSLAM_ERR_INVALID;
    MSLAM_TRY(ctx_set_device(ctx));
    OrbPlan* pl = ctx->orb;
    LevelDesc lv = pl->dev.lv[level];
    if (level == 0 && pl->last_l0) { lv.img = pl->last_l0; lv.pitch = pl->last_l0_pitch; }
    *level_w = lv.w; *level_h = lv.h;
    if (!blurred) {
        HIP_TRY(ctx, hipMemcpy2DAsync(out, lv.w, lv.img, lv.pitch, lv.w, lv.h, hipMemcpyDeviceToHost, ctx->stream));
    } else {
        void* scr = nullptr;
        MSLAM_TRY(ctx_scratch(ctx, (size_t)lv.w * lv.h, &scr));
        hipLaunchKernelGGL(blur7_k, dim3((lv.w + 255) / 256, lv.h), dim3(256), 0, ctx->stream, lv.img, lv.w, lv.h, lv.pitch, (uint8_t*)scr);
        HIP_TRY(ctx, hipMemcpyAsync(out, scr, (size_t)lv.w * lv.h, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MSLAM_OK;
}

} // extern "C"
