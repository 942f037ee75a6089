// ba.hip -- K9: bundle-adjustment linearisation on gfx950: reprojection residuals, Huber
// weights, analytic 2x3 / 2x6 Jacobians and the J^T w Omega J blocks, written straight into the
// block layout the host-side Schur / LDL^T solve consumes (Hpp, Hll, block-CSR Hpl, b); plus the Schur
// complement and the landmark back-substitution of the same system.
//
// Stands in for one pass of BlockSolver<6,3>::buildSystem (reference ThirdParty/g2o/g2o/core/
// block_solver.hpp:502-560) over the graph of Util::BundleAdjustment (src/Util.cpp:34-170):
// EdgeSE3ProjectXYZ::computeError / linearizeOplus (types/types_six_dof_expmap.h:90-95,
// types_six_dof_expmap.cpp:103-139), RobustKernelHuber::robustify (core/robust_kernel_impl.cpp:78-91),
// BaseBinaryEdge::constructQuadraticForm (core/base_binary_edge.hpp:55-120); and for
// BlockSolver::solve (block_solver.hpp:354-486).
//
// HBM layout (device-resident plan, built once per graph like g2o's buildStructure).  Edges are re-ordered once, on the
// host, into ROW ORDER k: rows = poses with at least one edge, free poses first (in free-index order) then fixed ones;
// inside a row by point.  For a free row k IS the index of the edge's 6x3 Hpl block in the block-CSR (so the kernel
// that evaluates an edge writes its block to consecutive memory), and every per-edge input is a streaming read:
//   state    poses f64[n_poses][7] (t, q xyzw), points f64[n_points][3]
//   rows     erowptr i32[R+1], row_pose i32[R]; rec i32[E] (the edge's point; its original id stays on the host), obs_k f64x2[E]
//   by point pt_ptr i32[n_points+1], pt_k i32[E]   (row-order indices of a point's edges, ascending)
//   kpose    i32[E]       pose of the edge at row-order position k
// Kernels of one linearisation (no float atomics anywhere: every sum has a fixed shape, results are run-to-run identical):
//   ba_edges_k   grid (chunk, row), one lane per edge: residual, Huber weight, Jacobians; err / chi2 in row order;
//                for free rows the 6x3 block goes through an LDS transpose and
//                leaves as 16-byte stores of consecutive memory, and the 27 pose-side sums are reduced by a wavefront
//                butterfly + 4 wavefront partials into the chunk's partial
//   ba_points_k  one lane per point: evaluates its edges (row order) a second time -> Hll, b_l, robust chi2 of the point
//   ba_final_k   adds the chunk partials per pose in chunk order, reduces the robust chi2
// Schur complement (tiles of at most SCHUR_TP consecutive points, ending at the seams of the pose graph; a tile's pose list and its dense point x local-pose block table
// are plan data, O(points x poses-per-tile) instead of the dense poses x points table):
//   ba_schur_points_k  (Hll + lambda I)^-1 and its product with b_l, per point
//   ba_schur_tiles_k   per tile: each wavefront takes every fourth point, its blocks prefetched into registers and staged through
//                      the wavefront's own LDS with B Dinv formed on the way; per local pose pair the sum over the points of
//                      (B Dinv)_j1 B_j2^T in registers; per local pose the sum of B (Dinv b_l)
//   ba_schur_final_k   S block (j1, j2) = [Hpp + lambda I] - sum of the tile partials in tile order; b_s
//   ba_backsub_k / ba_window_sums_k   x_l, X += x_l, and the per-window sums of LM's computeScale and chi2
// lambda may differ per WINDOW (mslam_ba_plan_set_windows): block-diagonal batches of independent local-BA windows then
// run one LM trial of every window per launch.
// f64 throughout (g2o is double; 1e-4 relative parity is an f64 target).  FMA contraction is
// allowed in this file: no discrete decision depends on the last bit here.
#include "common.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#pragma clang fp contract(fast)

namespace {

constexpr int EDGE_THREADS = 256;  // lanes of a ba_edges_k workgroup
constexpr int EDGE_EPL = 2;        // edges per lane
constexpr int EDGE_CHUNK = EDGE_THREADS * EDGE_EPL;    // edges per ba_edges_k workgroup
constexpr int T_PITCH = 258;       // LDS plane pitch (doubles) of the block transpose: 258 = 2 (mod 32) spreads the planes over banks
constexpr int SCHUR_TP = 64;       // points per Schur tile
constexpr int SCHUR_LMAX = 64;     // local poses per tile beyond which the device Schur path is not offered
constexpr int SCHUR_DD = 12;       // doubles of the packed per-point record a Schur tile reads: Dinv (9), Dinv b_l (3) = 6 pieces of 16 bytes

struct BaDev {
    int n_poses, n_points, n_edges, n_free, nnzb, n_rows, n_chunks_max;
    double huber_delta; float dsqr;
    const double* poses; const double* points;
    const int32_t* edge_pose; const double* info /* [E] in ROW order */; const double* intr;
    const int32_t* fidx;      // free index per pose or -1
    const int32_t* erowptr; const int32_t* row_pose;
    const int32_t* rec;       // [E] point of the edge, in row order
    const double2* obs_k;     // [E] observation in row order
    const int32_t* pt_ptr; const int32_t* pt_k;
    const int32_t* blk_row;   // [nnzb] free row of a block
    double* err; double* chi2; double* rho1; double* Hpp; double* Hll; double* Hpl_val; double* b;
    const int32_t* kpose;     // [E] pose of the edge at row-order position k (ba_points_k evaluates a point's edges itself)
    double* pt_rho;           // [n_points] robust chi2 per point
    double* chi_part;         // [ceil(n_points / 4096)] slice sums of pt_rho (large problems)
    double* pose_part;        // [n_free][n_chunks_max][27]
    double* robust_chi2;      // [1]
    // ---- Schur complement (f1): reduced pose system S x_p = b_s, landmark back-substitution
    const int32_t* col;       // [nnzb] point of a block
    int n_tiles; long long n_tile_pairs;
    const int32_t* tile_pose_ptr; const int32_t* tile_pose;   // local pose list of a tile (free row ids, ascending)
    const int32_t* tile_pt0;                                  // [n_tiles + 1] first point of a tile (at most SCHUR_TP points each)
    const int32_t* tile_blk_ptr;  const int32_t* tile_blk;    // [tile][points of the tile][L] block of (point, local pose) or -1
    const long long* tile_out_ptr;                            // first partial (36 doubles each) of a tile
    int n_gpairs; const int32_t* gpair_j; const int32_t* gpair_ptr; const long long* gpair_src; // global pose pairs and their partials
    const int32_t* gdiag;     // [n_free] global pair index of (j, j) or -1
    double* tile_part;        // [n_tile_pairs][36]
    double* tile_coef;        // [n_tile_pairs][6] (diagonal local pairs only)
    double* Dinv;             // [n_points][9]  (Hll + lambda I)^-1
    double* DD;               // [n_points][12] {Dinv, Dinv b_l} packed: what a Schur tile reads per point
    double* S;                // [6 n_free][6 n_free]
    double* bs;               // [6 n_free]
    double* x;                // [6 n_free + 3 n_points]
    double* points_bk;        // [n_points][3] snapshot for rejected LM steps
    double* pt_scale;         // [n_points] x_l . (lambda x_l + b_l)
    double* scale_part;       // [ceil(n_points/4096)]
    double* scale_l;          // [1]
    // ---- windows (independent sub-problems concatenated block-diagonally): lambda per window
    int n_windows;
    const int32_t* pose_win; const int32_t* point_win;        // window of a free row / of a point (null: one window)
    const int32_t* win_pt_ptr; const int32_t* win_pt;         // points of a window
    const double* lam_w;      // [n_windows] (device)
    double* win_out;          // [n_windows][2]: robust chi2, landmark part of computeScale
};

__device__ __forceinline__ double lam_pose(const BaDev& d, int f) { return d.lam_w[d.pose_win ? d.pose_win[f] : 0]; }
__device__ __forceinline__ double lam_point(const BaDev& d, int i) { return d.lam_w[d.point_win ? d.point_win[i] : 0]; }

struct EdgeLin { double e0, e1, A[6], B[12]; };

// computeError + linearizeOplus (Eigen's quaternion * vector and toRotationMatrix restated)
__device__ __forceinline__ void edge_eval(const double* __restrict__ pose, double X0, double X1, double X2, double ox, double oy,
                                          double fx, double fy, double cx, double cy, EdgeLin& L)
{
    const double tx_ = pose[0], ty_ = pose[1], tz_ = pose[2], qx = pose[3], qy = pose[4], qz = pose[5], qw = pose[6];
    double ux = qy * X2 - qz * X1, uy = qz * X0 - qx * X2, uz = qx * X1 - qy * X0;
    ux += ux; uy += uy; uz += uz;
    const double x = X0 + qw * ux + (qy * uz - qz * uy) + tx_;
    const double y = X1 + qw * uy + (qz * ux - qx * uz) + ty_;
    const double z = X2 + qw * uz + (qx * uy - qy * ux) + tz_;
    const double iz = 1.0 / z;
    L.e0 = ox - (x * iz * fx + cx);
    L.e1 = oy - (y * iz * fy + cy);
    const double z_2 = z * z;
    const double tx = 2 * qx, ty = 2 * qy, tz = 2 * qz;
    const double twx = tx * qw, twy = ty * qw, twz = tz * qw, txx = tx * qx, txy = ty * qx, txz = tz * qx, tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
    const double R[9] = { 1 - (tyy + tzz), txy - twz, txz + twy, txy + twz, 1 - (txx + tzz), tyz - twx, txz - twy, tyz + twx, 1 - (txx + tyy) };
    const double s = -iz;
    const double t00 = s * fx, t02 = s * (-x * iz * fx), t11 = s * fy, t12 = s * (-y * iz * fy);
#pragma unroll
    for (int j = 0; j < 3; j++) {
        L.A[j] = t00 * R[j] + t02 * R[6 + j];
        L.A[3 + j] = t11 * R[3 + j] + t12 * R[6 + j];
    }
    L.B[0] = x * y / z_2 * fx;       L.B[1] = -(1 + (x * x / z_2)) * fx; L.B[2] = y * iz * fx;
    L.B[3] = -iz * fx;               L.B[4] = 0;                         L.B[5] = x / z_2 * fx;
    L.B[6] = (1 + y * y / z_2) * fy; L.B[7] = -x * y / z_2 * fy;         L.B[8] = -x * iz * fy;
    L.B[9] = 0;                      L.B[10] = -iz * fy;                 L.B[11] = y / z_2 * fy;
}

__device__ __forceinline__ void huber(double chi2, double delta, float dsqr, double& rho0, double& rho1)
{
    rho0 = chi2; rho1 = 1.0;
    if (delta > 0 && !(chi2 <= (double)dsqr)) {
        const double sq = sqrt(chi2);
        rho0 = 2 * sq * delta - (double)dsqr;
        rho1 = delta / sq;
    }
}

// fixed-shape workgroup sum of NV values per lane: a DPP scan inside each wavefront (registers), then the 4 wavefront
// sums in index order by thread q < NV; the result of value q is returned to thread q (others get garbage)
template <int NV>
__device__ __forceinline__ double chunk_reduce(double (&v)[NV], double* s_red /* NV * 4 */, int tid)
{
    // wavefront sums by DPP moves (the two halves of a double travel as two v_mov_b32_dpp): an inclusive scan inside each row of
    // 16 lanes (row_shr 1, 2, 4, 8; +0.0 shifted in), row totals handed on by row_bcast:15 / :31, the total in lane 63 -- no LDS
    // round trips (the xor butterfly of __shfl_xor was 2 ds_bpermute per value and step: the kernel's limiter)
#define BA_DPP_ADD(ctrl, rmask, bc)                                                                                     \
    _Pragma("unroll") for (int q = 0; q < NV; q++) {                                                                    \
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v[q]), ctrl, rmask, 0xf, bc);                       \
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v[q]), ctrl, rmask, 0xf, bc);                       \
        v[q] += __hiloint2double(hi, lo);                                                                               \
    }
    BA_DPP_ADD(0x111, 0xf, true) BA_DPP_ADD(0x112, 0xf, true) BA_DPP_ADD(0x114, 0xf, true) BA_DPP_ADD(0x118, 0xf, true)
    BA_DPP_ADD(0x142, 0xa, false) BA_DPP_ADD(0x143, 0xc, false)
#undef BA_DPP_ADD
    if ((tid & 63) == 63)
#pragma unroll
        for (int q = 0; q < NV; q++) s_red[q * 4 + (tid >> 6)] = v[q];
    __syncthreads();
    double t = 0.0;
    if (tid < NV) t = ((s_red[tid * 4] + s_red[tid * 4 + 1]) + s_red[tid * 4 + 2]) + s_red[tid * 4 + 3];
    return t;
}

__global__ __launch_bounds__(EDGE_THREADS)
void ba_edges_k(BaDev d)
{
    const int r = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
    const int k0 = d.erowptr[r], k1 = d.erowptr[r + 1];
    const int kb = k0 + chunk * EDGE_CHUNK;
    if (kb >= k1) return; // whole workgroup out of range
    __shared__ double s_t[18 * T_PITCH];
    __shared__ double s_red[27 * 4];
    const int jp = d.row_pose[r];          // the whole row shares one pose: uniform (scalar) loads
    const int f = d.fidx[jp];
    double v[27];
#pragma unroll
    for (int q = 0; q < 27; q++) v[q] = 0.0;
    // EDGE_EPL edges per lane, one after the other: the 27 pose-side sums stay in the lane's registers across them, so the
    // cross-lane reduction at the end (27 values x 6 butterfly steps) is paid once per EDGE_EPL edges
    for (int sub = 0; sub < EDGE_EPL; sub++) {
    const int kbs = kb + sub * EDGE_THREADS;
    if (kbs >= k1) break;                  // workgroup-uniform
    const int k = kbs + tid;
    const bool live = k < k1;
    double Hb[18];
#pragma unroll
    for (int q = 0; q < 18; q++) Hb[q] = 0.0;
    if (live) {
        const double2 ob = d.obs_k[k];
        const int ip = d.rec[k];   // (the caller's edge id is only needed by the host: mslam_ba_plan_download)
        const double* in = d.intr + 4 * jp;
        EdgeLin L;
        edge_eval(d.poses + 7 * jp, d.points[3 * ip], d.points[3 * ip + 1], d.points[3 * ip + 2], ob.x, ob.y, in[0], in[1], in[2], in[3], L);
        const double om = d.info[k];   // (row order, like every other per-edge stream of this kernel: no scattered 8-byte accesses)
        const double chi2 = L.e0 * (om * L.e0) + L.e1 * (om * L.e1);
        double rho0, rho1;
        huber(chi2, d.huber_delta, d.dsqr, rho0, rho1);
        reinterpret_cast<double2*>(d.err)[k] = make_double2(L.e0, L.e1); d.chi2[k] = chi2;   // row order; mslam_ba_plan_download permutes (and forms rho1 from chi2)
        const double wom = rho1 * om;
        const double r0 = -(om * L.e0) * rho1, r1 = -(om * L.e1) * rho1;
        const double* A = L.A;
        (void)rho0;   // (the point side -- Hll, b_l, the robust chi2 -- is formed by ba_points_k from the same inputs: no per-edge scratch)
        if (f >= 0) {
#pragma unroll
            for (int a = 0; a < 6; a++) { // B^T (w Omega) A, row-major 6x3
                const double ba = wom * L.B[a], bb = wom * L.B[6 + a];
                Hb[a * 3 + 0] = ba * A[0] + bb * A[3];
                Hb[a * 3 + 1] = ba * A[1] + bb * A[4];
                Hb[a * 3 + 2] = ba * A[2] + bb * A[5];
            }
            int q = 0;
#pragma unroll
            for (int a = 0; a < 6; a++)
#pragma unroll
                for (int c = a; c < 6; c++) v[q++] += wom * (L.B[a] * L.B[c] + L.B[6 + a] * L.B[6 + c]);
#pragma unroll
            for (int a = 0; a < 6; a++) v[21 + a] += L.B[a] * r0 + L.B[6 + a] * r1;
        }
    }
    if (f < 0) continue; // fixed pose (uniform): no Hpl row, no pose block
    // the sub-chunk's blocks are consecutive in Hpl_val: transpose through LDS ([value][lane] planes) and write 16-byte
    // pieces of consecutive memory instead of 18 stride-144 doubles per lane
    if (sub) __syncthreads();   // the previous sub-chunk's planes have been read
#pragma unroll
    for (int q = 0; q < 18; q++) s_t[q * T_PITCH + tid] = Hb[q];
    __syncthreads();
    {
        const int n_live = min(EDGE_THREADS, k1 - kbs);
        double2* out = reinterpret_cast<double2*>(d.Hpl_val + (size_t)kbs * 18);   // (kbs * 144) bytes: 16-byte aligned
        for (int c = tid; c < n_live * 9; c += EDGE_THREADS) {
            const int kl = c / 9, q = 2 * (c - 9 * kl);
            out[c] = make_double2(s_t[q * T_PITCH + kl], s_t[(q + 1) * T_PITCH + kl]);
        }
    }
    }   // sub
    if (f < 0) return;
    const double t = chunk_reduce<27>(v, s_red, tid);
    if (tid < 27) d.pose_part[((size_t)f * d.n_chunks_max + chunk) * 27 + tid] = t;
}

__global__ __launch_bounds__(256)
void ba_points_k(BaDev d)
{
    // One lane per point: its edges (ascending row order: a fixed summation order) are evaluated HERE a second time -- residual,
    // Huber weight, the 2x3 point Jacobian -- instead of being read back from a per-edge scratch written by ba_edges_k.  The
    // linearisation is HBM-bound (ba_edges_k streams 150+ bytes of blocks per edge): 72 bytes written and 72 read per edge cost
    // more than the ~250 f64 instructions of a second evaluation, whose inputs (pose, intrinsics: cache-resident; observation,
    // information: 24 streamed bytes) are small.
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= d.n_points) return;
    double acc[10];
#pragma unroll
    for (int q = 0; q < 10; q++) acc[q] = 0.0;
    const double X0 = d.points[3 * (size_t)i], X1 = d.points[3 * (size_t)i + 1], X2 = d.points[3 * (size_t)i + 2];
    for (int kk = d.pt_ptr[i]; kk < d.pt_ptr[i + 1]; kk++) {
        const int k = d.pt_k[kk];
        const int jp = d.kpose[k];
        const double2 ob = d.obs_k[k];
        const double* in = d.intr + 4 * jp;
        EdgeLin L;
        edge_eval(d.poses + 7 * jp, X0, X1, X2, ob.x, ob.y, in[0], in[1], in[2], in[3], L);
        const double om = d.info[k];
        const double chi2 = L.e0 * (om * L.e0) + L.e1 * (om * L.e1);
        double rho0, rho1;
        huber(chi2, d.huber_delta, d.dsqr, rho0, rho1);
        const double wom = rho1 * om;
        const double r0 = -(om * L.e0) * rho1, r1 = -(om * L.e1) * rho1;
        const double* A = L.A;
        acc[0] += wom * (A[0] * A[0] + A[3] * A[3]); acc[1] += wom * (A[0] * A[1] + A[3] * A[4]);
        acc[2] += wom * (A[0] * A[2] + A[3] * A[5]); acc[3] += wom * (A[1] * A[1] + A[4] * A[4]);
        acc[4] += wom * (A[1] * A[2] + A[4] * A[5]); acc[5] += wom * (A[2] * A[2] + A[5] * A[5]);
        acc[6] += A[0] * r0 + A[3] * r1; acc[7] += A[1] * r0 + A[4] * r1; acc[8] += A[2] * r0 + A[5] * r1;
        acc[9] += rho0;
    }
    double* H = d.Hll + (size_t)i * 9;
    H[0] = acc[0]; H[1] = acc[1]; H[2] = acc[2];
    H[3] = acc[1]; H[4] = acc[3]; H[5] = acc[4];
    H[6] = acc[2]; H[7] = acc[4]; H[8] = acc[5];
    double* bl = d.b + 6 * d.n_free + 3 * (size_t)i;
    bl[0] = acc[6]; bl[1] = acc[7]; bl[2] = acc[8];
    d.pt_rho[i] = acc[9];
}

// Final reductions.  Small problems (one local-BA window): ONE workgroup does both the per-pose chunk sums and the
// robust chi2.  Large ones (many windows concatenated block-diagonally): ba_final_k runs as a grid -- workgroups
// [0, n_pose_wg) take 256 (pose, value) items each, the rest reduce fixed 4096-point slices of pt_rho into chi_part --
// and ba_final2_k adds the slice sums in index order.  Every sum has a fixed shape: bit-identical run to run.
constexpr int CHI_SLICE = 4096;
__global__ __launch_bounds__(256)
void ba_final_k(BaDev d, int n_pose_wg)
{
    const int tid = threadIdx.x;
    const bool single = gridDim.x == 1;
    if (single || (int)blockIdx.x < n_pose_wg) {
        // poses: one thread per (pose, value)
        const int i0 = single ? tid : (int)blockIdx.x * 256 + tid, step = single ? 256 : d.n_free * 27;
        for (int i = i0; i < d.n_free * 27; i += step) {
            const int f = i / 27, q = i % 27;
            const int nch = (d.erowptr[f + 1] - d.erowptr[f] + EDGE_CHUNK - 1) / EDGE_CHUNK;
            double s = 0.0;
            for (int c = 0; c < nch; c++) s += d.pose_part[((size_t)f * d.n_chunks_max + c) * 27 + q];
            if (q < 21) {
                int a = 0, rem = q; // unpack upper-triangle index
                while (rem >= 6 - a) { rem -= 6 - a; a++; }
                const int c2 = a + rem;
                d.Hpp[(size_t)f * 36 + a * 6 + c2] = s;
                d.Hpp[(size_t)f * 36 + c2 * 6 + a] = s;
            } else d.b[6 * f + (q - 21)] = s;
        }
        if (!single) return;
    }
    // robust chi2: fixed-order tree over points (all of them, or this workgroup's slice)
    __shared__ double s_r[256];
    const int slice = single ? 0 : (int)blockIdx.x - n_pose_wg;
    const int p0 = single ? 0 : slice * CHI_SLICE, p1 = single ? d.n_points : min(d.n_points, p0 + CHI_SLICE);
    double r = 0.0;
    for (int i = p0 + tid; i < p1; i += 256) r += d.pt_rho[i];
    s_r[tid] = r;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) s_r[tid] += s_r[tid + o]; __syncthreads(); }
    if (tid == 0) (single ? d.robust_chi2 : d.chi_part + slice)[0] = s_r[0];
}

// sum of a slice-partial array in index order (robust chi2 / landmark part of computeScale)
__global__ __launch_bounds__(256)
void ba_final2_k(const double* __restrict__ part, int n, double* __restrict__ out)
{
    __shared__ double s_r[256];
    const int tid = threadIdx.x;
    double r = 0.0;
    for (int i = tid; i < n; i += 256) r += part[i];
    s_r[tid] = r;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) s_r[tid] += s_r[tid + o]; __syncthreads(); }
    if (tid == 0) out[0] = s_r[0];
}

// ---- Schur complement on the device (BlockSolver::solve, block_solver.hpp:372-481)
__global__ __launch_bounds__(256)
void ba_schur_points_k(BaDev d)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= d.n_points) return;
    double M[9];
#pragma unroll
    for (int q = 0; q < 9; q++) M[q] = d.Hll[(size_t)i * 9 + q];
    const double lam = lam_point(d, i);
    M[0] += lam; M[4] += lam; M[8] += lam;
    const double det = M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
    const double id = 1.0 / det;
    double I[9];
    I[0] = (M[4] * M[8] - M[5] * M[7]) * id; I[1] = (M[2] * M[7] - M[1] * M[8]) * id; I[2] = (M[1] * M[5] - M[2] * M[4]) * id;
    I[3] = (M[5] * M[6] - M[3] * M[8]) * id; I[4] = (M[0] * M[8] - M[2] * M[6]) * id; I[5] = (M[2] * M[3] - M[0] * M[5]) * id;
    I[6] = (M[3] * M[7] - M[4] * M[6]) * id; I[7] = (M[1] * M[6] - M[0] * M[7]) * id; I[8] = (M[0] * M[4] - M[1] * M[3]) * id;
    const double* bl = d.b + 6 * d.n_free + 3 * (size_t)i;
#pragma unroll
    for (int q = 0; q < 9; q++) d.Dinv[(size_t)i * 9 + q] = I[q];
    double* dd = d.DD + (size_t)i * SCHUR_DD;
#pragma unroll
    for (int q = 0; q < 9; q++) dd[q] = I[q];
#pragma unroll
    for (int a = 0; a < 3; a++) dd[9 + a] = I[a * 3] * bl[0] + I[a * 3 + 1] * bl[1] + I[a * 3 + 2] * bl[2];
}

// piece 64 r + ln of point pi of a Schur tile: a 16-byte part of one of its blocks (missing block: block 0, never used) or of its
// packed record
template <int R> struct SchurPieces { typedef double type __attribute__((ext_vector_type(2 * R))); };   // (a value, not an array: stays in registers)
template <int R>
__device__ __forceinline__ typename SchurPieces<R>::type schur_fetch(const BaDev& d, const int32_t* s_tb, int L, int i0, int pi, int ln)
{
    typename SchurPieces<R>::type v;
    const int npieces = 9 * L + 6;
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int j = min(r * 64 + ln, npieces - 1);   // (lanes past the last piece repeat it)
        const int lq = j / 9;
        // branch-free: every lane issues exactly R loads per point, so the loads in flight are counted, not waited out
        const int k = s_tb[pi * L + min(lq, L - 1)];
        const double* sb = d.Hpl_val + (size_t)max(k, 0) * 18 + 2 * (j - 9 * lq);
        const double* sd = d.DD + (size_t)(i0 + pi) * SCHUR_DD + 2 * (j - 9 * L);
        const double2 t = *reinterpret_cast<const double2*>(lq < L ? sb : sd);
        v[2 * r] = t.x; v[2 * r + 1] = t.y;
    }
    return v;
}

// One workgroup per tile of at most SCHUR_TP consecutive points, four wavefronts that never meet inside the point loop: wavefront w
// takes the points w, w + 4, ... of the tile, one at a time, with lane = local pose pair (la <= lb).  A point's data -- its
// <= L blocks B (6x3, 144 bytes each) and its packed {Dinv, Dinv b_l} (96 bytes) -- are "pieces" of 16 bytes: lane j of load
// instruction r fetches piece 64 r + j into registers THREE points ahead of its use (the block indices come from the tile's table,
// copied to LDS once), writes it to the wavefront's own LDS stage when its turn comes, and the wavefront then forms B Dinv
// (one lane per block row) and, per pair, adds (B Dinv)_{la} B_{lb}^T into 36 registers (g2o's own loop shape,
// block_solver.hpp:381-432, with the point loop innermost).  The LDS traffic of one wavefront is ordered by program order, so no
// workgroup barrier sits between a point's loads and its use: the memory latency is covered by the three points in flight and by
// the other wavefronts.  Lanes j < 6 L also accumulate row j of B (Dinv b_l): the tile's share of the reduced right-hand
// side's coefficient, stored beside the diagonal pair (la, la).  The four wavefronts' sums are added as (0 + 2) + (1 + 3).
// R = load instructions per point: 9 L + 6 <= 64 R.
template <int R>
__global__ __launch_bounds__(256)
void ba_schur_tiles_k(BaDev d)
{
    constexpr int SLOT = R * 128;                       // doubles of one stage: R x 64 pieces
    constexpr int MAIN = 8 * SLOT > 2 * 64 * 36 ? 8 * SLOT : 2 * 64 * 36;   // 4 wavefronts x {stage, B Dinv}; later the cross-wavefront sum
    constexpr int LCAP = (64 * R - 6) / 9;              // local poses this instantiation can stage
    constexpr int CR = (6 * LCAP + 63) / 64;            // block rows per lane
    extern __shared__ __attribute__((aligned(16))) double s_dyn[];   // (16-byte pieces)
    int32_t* const s_tb = reinterpret_cast<int32_t*>(s_dyn + MAIN);   // [SCHUR_TP][L] block index table of the tile
    const int tile = blockIdx.x, tid = threadIdx.x, w = tid >> 6, ln = tid & 63;
    const int L = d.tile_pose_ptr[tile + 1] - d.tile_pose_ptr[tile];
    if (L == 0) return;
    const int npairs = L * (L + 1) / 2;
    const int32_t* tb = d.tile_blk + d.tile_blk_ptr[tile];
    const int i0 = d.tile_pt0[tile], np = d.tile_pt0[tile + 1] - i0;
    double* out = d.tile_part + (size_t)d.tile_out_ptr[tile] * 36;
    double* cout = d.tile_coef + (size_t)d.tile_out_ptr[tile] * 6;
    double* const sB = s_dyn + w * 2 * SLOT;            // this wavefront's stage: blocks at 18 lq, the packed record at 18 L
    double* const sBD = sB + SLOT;
    for (int i = tid; i < np * L; i += 256) s_tb[i] = tb[i];
    __syncthreads();
    const int n_my = np > w ? (np - w + 3) >> 2 : 0;    // points of this wavefront
    double coef[CR];
#pragma unroll
    for (int c = 0; c < CR; c++) coef[c] = 0.0;
    const int npass = (npairs + 63) / 64;       // pair passes (1 for L <= 10); wider tiles run the point loop once per pass
    for (int pass = 0; pass < npass; pass++) {
        const int p = ln + pass * 64;
        int la = 0, lb = 0;
        if (p < npairs) { int rem = p; while (rem >= L - la) { rem -= L - la; la++; } lb = la + rem; }
        double acc[36];
#pragma unroll
        for (int q = 0; q < 36; q++) acc[q] = 0.0;
        // pieces of the current point and of the next three.  The point loop is unrolled by four so that each of the four register
        // sets keeps its role (a register copy would have to wait for the load it copies) and has ONE exit (an exit between two
        // steps would merge "just loaded" into the loop header's state and bring the full wait back): a wavefront whose point count
        // is no multiple of four runs the missing steps on its last point with nothing accumulated, and fetches past the last point
        // repeat it (cached, unused), so that every step issues the same R loads
        const int n_u = __builtin_amdgcn_readfirstlane(n_my);
        const int pi_last = w + 4 * (n_u - 1);
        typename SchurPieces<R>::type v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0;
        auto step = [&](int it, const typename SchurPieces<R>::type& cur, typename SchurPieces<R>::type& fut) {
            const int pi = min(w + 4 * it, pi_last);
            fut = schur_fetch<R>(d, s_tb, L, i0, min(w + 4 * it + 12, pi_last), ln);
            // which of the point's blocks exist (wave-uniform mask over the local poses; none in a padding step)
            const unsigned long long vm = __builtin_amdgcn_ballot_w64(it < n_u && ln < L && s_tb[pi * L + ln] >= 0);
#pragma unroll
            for (int r = 0; r < R; r++)   // (every lane: the slots past the last piece exist and receive its copy -- no branch around the wait for the load)
                *reinterpret_cast<double2*>(sB + 2 * (r * 64 + ln)) = make_double2(cur[2 * r], cur[2 * r + 1]);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            {   // B Dinv and B (Dinv b_l), one block row per lane
                const double* DD = sB + 18 * L;
                const double D0 = DD[0], D1 = DD[1], D2 = DD[2], D3 = DD[3], D4 = DD[4], D5 = DD[5], D6 = DD[6], D7 = DD[7], D8 = DD[8];
                const double e0 = DD[9], e1 = DD[10], e2 = DD[11];
#pragma unroll
                for (int c = 0; c < CR; c++) {
                    const int j = c * 64 + ln;
                    if (j < 6 * L) {
                        const double b0 = sB[3 * j], b1 = sB[3 * j + 1], b2 = sB[3 * j + 2];
                        sBD[3 * j] = b0 * D0 + b1 * D3 + b2 * D6; sBD[3 * j + 1] = b0 * D1 + b1 * D4 + b2 * D7; sBD[3 * j + 2] = b0 * D2 + b1 * D5 + b2 * D8;
                        if (pass == 0 && ((vm >> (j / 6)) & 1ull)) coef[c] += b0 * e0 + b1 * e1 + b2 * e2;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (p < npairs && ((vm >> la) & (vm >> lb) & 1ull)) {
                const double2* a2 = reinterpret_cast<const double2*>(sBD + 18 * la);
                const double2* b2 = reinterpret_cast<const double2*>(sB + 18 * lb);
                double bb[18];
#pragma unroll
                for (int q = 0; q < 9; q++) { const double2 t = b2[q]; bb[2 * q] = t.x; bb[2 * q + 1] = t.y; }
#pragma unroll
                for (int ap = 0; ap < 3; ap++) {   // two rows of B Dinv (three 16-byte reads) at a time
                    const double2 t0 = a2[3 * ap], t1 = a2[3 * ap + 1], t2 = a2[3 * ap + 2];
                    const double aa[6] = { t0.x, t0.y, t1.x, t1.y, t2.x, t2.y };
#pragma unroll
                    for (int h = 0; h < 2; h++)
#pragma unroll
                        for (int c = 0; c < 6; c++)
                            acc[(2 * ap + h) * 6 + c] += aa[3 * h] * bb[c * 3] + aa[3 * h + 1] * bb[c * 3 + 1] + aa[3 * h + 2] * bb[c * 3 + 2];
                }
            }
            __builtin_amdgcn_wave_barrier();   // (the next point's stage writes stay behind these reads)
        };
        if (n_u > 0) {
            v0 = schur_fetch<R>(d, s_tb, L, i0, w, ln);
            v1 = schur_fetch<R>(d, s_tb, L, i0, min(w + 4, pi_last), ln);
            v2 = schur_fetch<R>(d, s_tb, L, i0, min(w + 8, pi_last), ln);
            for (int it = 0; it < n_u; it += 4) {
                step(it, v0, v3);
                step(it + 1, v1, v0);
                step(it + 2, v2, v1);
                step(it + 3, v3, v2);
            }
        }
        // wavefronts summed as (0 + 2) + (1 + 3), through the stage storage (element-major: consecutive lanes, consecutive doubles)
        __syncthreads();   // every wavefront is done with its stage
        if (w >= 2)
#pragma unroll
            for (int q = 0; q < 36; q++) s_dyn[((w - 2) * 36 + q) * 64 + ln] = acc[q];
        __syncthreads();
        if (w < 2)
#pragma unroll
            for (int q = 0; q < 36; q++) acc[q] += s_dyn[(w * 36 + q) * 64 + ln];
        __syncthreads();
        if (w == 1)
#pragma unroll
            for (int q = 0; q < 36; q++) s_dyn[q * 64 + ln] = acc[q];
        __syncthreads();
        if (w == 0 && p < npairs)
#pragma unroll
            for (int q = 0; q < 36; q++) out[(size_t)p * 36 + q] = acc[q] + s_dyn[q * 64 + ln];
        __syncthreads();   // (the next pass writes the stages again)
    }
    // the coefficient rows, the same way
    if (w >= 2)
#pragma unroll
        for (int c = 0; c < CR; c++) s_dyn[((w - 2) * CR + c) * 64 + ln] = coef[c];
    __syncthreads();
    if (w < 2)
#pragma unroll
        for (int c = 0; c < CR; c++) coef[c] += s_dyn[(w * CR + c) * 64 + ln];
    __syncthreads();
    if (w == 1)
#pragma unroll
        for (int c = 0; c < CR; c++) s_dyn[c * 64 + ln] = coef[c];
    __syncthreads();
    if (w == 0)
#pragma unroll
        for (int c = 0; c < CR; c++) {
            const int j = c * 64 + ln;
            if (j < 6 * L) {
                const int lq = j / 6;
                const int pd = lq * L - lq * (lq - 1) / 2;   // index of the diagonal pair (lq, lq) in the tile's upper-triangle order
                cout[(size_t)pd * 6 + (j - 6 * lq)] = coef[c] + s_dyn[c * 64 + ln];
            }
        }
}

// S (zeroed by the caller) <- one 6x6 block per global pose pair that shares a point; b_s
constexpr int FIN_SPLIT = 4;   // lanes per output element: partials q = r, r + 4, ..., then a fixed 2-step butterfly
__global__ __launch_bounds__(256)
void ba_schur_final_k(BaDev d)
{
    const int np6 = 6 * d.n_free;
    const long long n_el = (long long)d.n_gpairs * 36;
    const long long n_thr = (n_el * FIN_SPLIT + 255) / 256 * 256;
    for (long long t0 = (long long)blockIdx.x * 256 + threadIdx.x; t0 < n_thr; t0 += (long long)gridDim.x * 256) {
        const long long t = t0 / FIN_SPLIT;
        const int r = (int)(t0 % FIN_SPLIT);
        double sum = 0.0;
        int g = 0, e = 0;
        if (t < n_el) {
            g = (int)(t / 36); e = (int)(t % 36);
            // (four partials per step: their index and value loads are independent, so they are in flight together)
            const int q1 = d.gpair_ptr[g + 1];
            int q = d.gpair_ptr[g] + r;
            for (; q + 3 * FIN_SPLIT < q1; q += 4 * FIN_SPLIT) {
                const int s0 = d.gpair_src[q], s1 = d.gpair_src[q + FIN_SPLIT], s2 = d.gpair_src[q + 2 * FIN_SPLIT], s3 = d.gpair_src[q + 3 * FIN_SPLIT];
                const double v0 = d.tile_part[(size_t)s0 * 36 + e], v1 = d.tile_part[(size_t)s1 * 36 + e], v2 = d.tile_part[(size_t)s2 * 36 + e],
                             v3 = d.tile_part[(size_t)s3 * 36 + e];
                sum += v0; sum += v1; sum += v2; sum += v3;   // same order as one at a time
            }
            for (; q < q1; q += FIN_SPLIT) sum += d.tile_part[(size_t)d.gpair_src[q] * 36 + e];   // tile order
        }
#pragma unroll
        for (int o = FIN_SPLIT / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        if (t < n_el && r == 0) {
            const int a = e / 6, c = e % 6;
            const int j1 = d.gpair_j[2 * g], j2 = d.gpair_j[2 * g + 1];
            double val = -sum;
            if (j1 == j2) val += d.Hpp[(size_t)j1 * 36 + e] + (a == c ? lam_pose(d, j1) : 0.0);
            d.S[(size_t)(6 * j1 + a) * np6 + 6 * j2 + c] = val;
            if (j1 != j2) d.S[(size_t)(6 * j2 + c) * np6 + 6 * j1 + a] = val;
        }
    }
    // b_s = b_p - sum_i B_ji (Dinv b_l)_i: the tile shares stored beside the diagonal pairs, in tile order
    // (FIN_SPLIT lanes per element like the blocks above, four partials in flight per lane: one lane walking all the tiles of a
    // pose with two dependent loads per tile was the longest chain of the kernel)
    const long long n_bs = ((long long)np6 * FIN_SPLIT + 255) / 256 * 256;
    for (long long t0 = (long long)blockIdx.x * 256 + threadIdx.x; t0 < n_bs; t0 += (long long)gridDim.x * 256) {
        const int t = (int)(t0 / FIN_SPLIT), r = (int)(t0 % FIN_SPLIT);
        double sum = 0.0;
        if (t < np6) {
            const int j = t / 6, a = t % 6;
            const int g = d.gdiag[j];
            if (g >= 0) {
                const int q1 = d.gpair_ptr[g + 1];
                int q = d.gpair_ptr[g] + r;
                for (; q + 3 * FIN_SPLIT < q1; q += 4 * FIN_SPLIT) {
                    const int s0 = d.gpair_src[q], s1 = d.gpair_src[q + FIN_SPLIT], s2 = d.gpair_src[q + 2 * FIN_SPLIT], s3 = d.gpair_src[q + 3 * FIN_SPLIT];
                    const double v0 = d.tile_coef[(size_t)s0 * 6 + a], v1 = d.tile_coef[(size_t)s1 * 6 + a], v2 = d.tile_coef[(size_t)s2 * 6 + a],
                                 v3 = d.tile_coef[(size_t)s3 * 6 + a];
                    sum += v0; sum += v1; sum += v2; sum += v3;
                }
                for (; q < q1; q += FIN_SPLIT) sum += d.tile_coef[(size_t)d.gpair_src[q] * 6 + a];
            }
        }
#pragma unroll
        for (int o = FIN_SPLIT / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        if (t < np6 && r == 0) d.bs[t] = d.b[t] - sum;
    }
}

// x_l = Dinv (b_l - Hpl^T x_p); optional X += x_l; per point x_l . (lambda x_l + b_l) for computeScale
__global__ __launch_bounds__(256)
void ba_backsub_k(BaDev d, int apply, double* points_rw)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= d.n_points) return;
    const double* bl = d.b + 6 * d.n_free + 3 * (size_t)i;
    double cl[3] = { bl[0], bl[1], bl[2] };
    for (int q = d.pt_ptr[i]; q < d.pt_ptr[i + 1]; q++) {
        const int k = d.pt_k[q];
        if (k >= d.nnzb) break;   // ascending: the rest belongs to fixed rows
        const double* B = d.Hpl_val + (size_t)k * 18;
        const double* xp = d.x + 6 * d.blk_row[k];
#pragma unroll
        for (int c = 0; c < 3; c++)
#pragma unroll
            for (int a = 0; a < 6; a++) cl[c] -= B[a * 3 + c] * xp[a];
    }
    const double* Di = d.Dinv + (size_t)i * 9;
    const double lam = lam_point(d, i);
    double sc = 0.0;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const double xl = Di[a * 3] * cl[0] + Di[a * 3 + 1] * cl[1] + Di[a * 3 + 2] * cl[2];
        d.x[6 * d.n_free + 3 * (size_t)i + a] = xl;
        sc += xl * (lam * xl + bl[a]);
        if (apply) points_rw[3 * (size_t)i + a] += xl;
    }
    d.pt_scale[i] = sc;
}

// fixed-shape slice sums of a per-point array (4096 points per workgroup)
__global__ __launch_bounds__(256)
void ba_slice_sum_k(const double* __restrict__ v, int n, double* __restrict__ part)
{
    __shared__ double s_r[256];
    const int tid = threadIdx.x, p0 = blockIdx.x * CHI_SLICE, p1 = min(n, p0 + CHI_SLICE);
    double r = 0.0;
    for (int i = p0 + tid; i < p1; i += 256) r += v[i];
    s_r[tid] = r;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) s_r[tid] += s_r[tid + o]; __syncthreads(); }
    if (tid == 0) part[blockIdx.x] = s_r[0];
}

// one workgroup per window: robust chi2 (which = 0, from pt_rho) or the landmark part of computeScale (which = 1, from pt_scale)
__global__ __launch_bounds__(256)
void ba_window_sums_k(BaDev d, int which)
{
    __shared__ double s_r[256];
    const int w = blockIdx.x, tid = threadIdx.x;
    const double* v = which ? d.pt_scale : d.pt_rho;
    double r = 0.0;
    for (int q = d.win_pt_ptr[w] + tid; q < d.win_pt_ptr[w + 1]; q += 256) r += v[d.win_pt[q]];
    s_r[tid] = r;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) s_r[tid] += s_r[tid + o]; __syncthreads(); }
    if (tid == 0) d.win_out[2 * w + which] = s_r[0];
}

} // namespace

struct mslam_ba_plan {
    mslam_ctx* ctx = nullptr;
    void* arena = nullptr;
    void* win_arena = nullptr;
    BaDev dev{};
    double* d_poses = nullptr; double* d_points = nullptr;
    int32_t* d_rowptr = nullptr; int32_t* d_col = nullptr;
    double* d_lam = nullptr;                 // [max(1, n_windows)]
    std::vector<int32_t> h_rowptr, h_col;
    std::vector<int32_t> h_row_pose;         // pose of a free row
    double built_delta = 0.0; float built_dsqr = 0.f;   // robust kernel of the last build (rho1 is formed from chi2 at download)
    std::vector<int32_t> h_edge_of_k;        // original edge id of row-order position k (err / chi2 / rho1 / info live in row order on the device)
    bool has_schur = false;
    int schur_lmax = 0;       // widest Schur tile (local poses): selects ba_schur_tiles_k's instantiation
};

static int ba_validate(const mslam_ba_problem* p)
{
    if (!p || p->n_poses < 1 || p->n_points < 1 || p->n_edges < 1 || !p->poses || !p->pose_fixed || !p->points || !p->edge_pose ||
        !p->edge_point || !p->obs || !p->info || !p->intr)
        return MSLAM_ERR_INVALID;
    for (int e = 0; e < p->n_edges; e++)
        if (p->edge_pose[e] < 0 || p->edge_pose[e] >= p->n_poses || p->edge_point[e] < 0 || p->edge_point[e] >= p->n_points)
            return MSLAM_ERR_INVALID;
    return MSLAM_OK;
}

extern "C" {

int mslam_ba_sizes(const mslam_ba_problem* p, int* n_free_poses, int* nnzb)
{
    MSLAM_TRY(ba_validate(p));
    int nf = 0, nz = 0;
    for (int j = 0; j < p->n_poses; j++) nf += !p->pose_fixed[j];
    for (int e = 0; e < p->n_edges; e++) nz += !p->pose_fixed[p->edge_pose[e]];
    if (n_free_poses) *n_free_poses = nf;
    if (nnzb) *nnzb = nz;
    return MSLAM_OK;
}

// structure analysis on the host (g2o: SparseOptimizer::buildIndexMapping + BlockSolver::buildStructure)
int mslam_ba_plan_create(mslam_ctx* ctx, const mslam_ba_problem* p, mslam_ba_plan** out)
{
    if (!ctx || !out) return MSLAM_ERR_INVALID;
    MSLAM_TRY(ba_validate(p));
    MSLAM_TRY(ctx_set_device(ctx));
    const int NP = p->n_poses, NL = p->n_points, E = p->n_edges;
    std::vector<int32_t> fidx(NP);
    int nf = 0;
    for (int j = 0; j < NP; j++) fidx[j] = p->pose_fixed[j] ? -1 : nf++;
    // rows: free poses 0..nf-1 (free-index order, also those without edges, so that row == free index), then the fixed
    // poses that have edges
    std::vector<int32_t> row_of_pose(NP, -1), row_pose;
    for (int j = 0; j < NP; j++) if (fidx[j] >= 0) { row_of_pose[j] = (int)row_pose.size(); row_pose.push_back(j); }
    {
        std::vector<char> has(NP, 0);
        for (int e = 0; e < E; e++) has[p->edge_pose[e]] = 1;
        for (int j = 0; j < NP; j++) if (fidx[j] < 0 && has[j]) { row_of_pose[j] = (int)row_pose.size(); row_pose.push_back(j); }
    }
    const int R = (int)row_pose.size();
    std::vector<int32_t> erowptr(R + 1, 0);
    for (int e = 0; e < E; e++) erowptr[row_of_pose[p->edge_pose[e]] + 1]++;
    for (int r = 0; r < R; r++) erowptr[r + 1] += erowptr[r];
    const int nnzb = erowptr[nf];
    std::vector<std::pair<int32_t, int32_t>> pe(E);   // (point, edge) per row, sorted
    {
        std::vector<int32_t> fill(R, 0);
        for (int e = 0; e < E; e++) { const int r = row_of_pose[p->edge_pose[e]]; pe[erowptr[r] + fill[r]++] = { p->edge_point[e], e }; }
    }
    int max_row = 0;
    for (int r = 0; r < R; r++) {
        std::sort(pe.begin() + erowptr[r], pe.begin() + erowptr[r + 1]);
        max_row = std::max(max_row, erowptr[r + 1] - erowptr[r]);
        // two edges between the same pose and point would need one accumulated block (g2o adds them into one); this layout
        // gives every edge its own block, so such graphs are refused instead of being solved wrongly
        for (int k = erowptr[r] + 1; k < erowptr[r + 1]; k++)
            if (pe[k].first == pe[k - 1].first) return MSLAM_ERR_INVALID;
    }
    std::vector<int2> rec(E);
    std::vector<double2> obs_k(E);
    std::vector<int32_t> col(std::max(nnzb, 1)), blk_row(std::max(nnzb, 1)), pt_ptr(NL + 1, 0), pt_k(E);
    for (int r = 0; r < R; r++)
        for (int k = erowptr[r]; k < erowptr[r + 1]; k++) {
            rec[k] = make_int2(pe[k].first, pe[k].second);
            obs_k[k] = make_double2(p->obs[2 * pe[k].second], p->obs[2 * pe[k].second + 1]);
            if (k < nnzb) { col[k] = pe[k].first; blk_row[k] = r; }
            pt_ptr[pe[k].first + 1]++;
        }
    for (int i = 0; i < NL; i++) pt_ptr[i + 1] += pt_ptr[i];
    {
        std::vector<int32_t> fill(NL, 0);
        for (int k = 0; k < E; k++) { const int i = rec[k].x; pt_k[pt_ptr[i] + fill[i]++] = k; }   // ascending k
    }
    const int n_chunks_max = std::max(1, (max_row + EDGE_CHUNK - 1) / EDGE_CHUNK);

    // ---- Schur tiles: local pose lists, dense point x local-pose block tables, partial offsets, global pair lists
    // Schur tiles: runs of at most SCHUR_TP consecutive points; a tile also ends where the next point belongs to another
    // connected component of the pose graph (poses linked by a common point) -- the seam between two windows of a block-diagonal
    // batch: a tile across it would carry both windows' poses, i.e. four times the pose pairs, for nothing
    std::vector<int32_t> tile_pt0(1, 0);
    {
        std::vector<int32_t> parent(std::max(nf, 1));
        for (int f = 0; f < nf; f++) parent[f] = f;
        auto find = [&](int f) { while (parent[f] != f) { parent[f] = parent[parent[f]]; f = parent[f]; } return f; };
        auto first_row = [&](int i) { for (int q = pt_ptr[i]; q < pt_ptr[i + 1]; q++) if (pt_k[q] < nnzb) return (int)blk_row[pt_k[q]]; return -1; };
        for (int i = 0; i < NL; i++) {
            const int r0 = first_row(i);
            if (r0 < 0) continue;
            for (int q = pt_ptr[i]; q < pt_ptr[i + 1]; q++) { const int k = pt_k[q]; if (k < nnzb) { const int a = find(r0), c = find(blk_row[k]); if (a != c) parent[c] = a; } }
        }
        int cnt = 0, comp = -1;   // points and component of the tile being filled
        for (int i = 0; i < NL; i++) {
            const int r0 = first_row(i), ci = r0 >= 0 ? find(r0) : -1;
            if (cnt == SCHUR_TP || (cnt > 0 && ci >= 0 && comp >= 0 && ci != comp)) { tile_pt0.push_back(i); cnt = 0; comp = -1; }
            if (ci >= 0) comp = ci;
            cnt++;
        }
        tile_pt0.push_back(NL);
    }
    const int n_tiles = (int)tile_pt0.size() - 1;
    std::vector<int32_t> tile_pose_ptr(n_tiles + 1, 0), tile_pose, tile_blk_ptr(n_tiles + 1, 0), tile_blk;
    std::vector<long long> tile_out_ptr(n_tiles + 1, 0);
    bool want_schur = nf > 0 && nnzb > 0;
    int schur_lmax = 0;
    std::vector<int32_t> gpair_j, gpair_ptr;
    std::vector<long long> gpair_src;
    if (want_schur) {
        std::vector<int32_t> loc(nf, -1);
        for (int t = 0; t < n_tiles && want_schur; t++) {
            const int i0 = tile_pt0[t], i1 = tile_pt0[t + 1];
            std::vector<int32_t> lp;
            for (int i = i0; i < i1; i++)
                for (int q = pt_ptr[i]; q < pt_ptr[i + 1]; q++) { const int k = pt_k[q]; if (k < nnzb) lp.push_back(blk_row[k]); }
            std::sort(lp.begin(), lp.end());
            lp.erase(std::unique(lp.begin(), lp.end()), lp.end());
            const int L = (int)lp.size();
            if (L > SCHUR_LMAX) { want_schur = false; break; }   // a tile that sees too many poses: host Schur (MSLAM_ERR_CAPACITY)
            schur_lmax = std::max(schur_lmax, L);
            for (int a = 0; a < L; a++) loc[lp[a]] = a;
            const size_t base = tile_blk.size();
            tile_blk.resize(base + (size_t)(i1 - i0) * L, -1);
            for (int i = i0; i < i1; i++)
                for (int q = pt_ptr[i]; q < pt_ptr[i + 1]; q++) { const int k = pt_k[q]; if (k < nnzb) tile_blk[base + (size_t)(i - i0) * L + loc[blk_row[k]]] = k; }
            tile_pose.insert(tile_pose.end(), lp.begin(), lp.end());
            tile_pose_ptr[t + 1] = (int)tile_pose.size();
            if (tile_blk.size() > (size_t)0x7FFFFFFF) { want_schur = false; break; }
            tile_blk_ptr[t + 1] = (int)tile_blk.size();
            tile_out_ptr[t + 1] = tile_out_ptr[t] + (long long)L * (L + 1) / 2;
        }
    }
    if (want_schur) {
        // global pairs (j1 <= j2) <- (tile, local pair) partials, in tile order
        struct Src { int32_t j1, j2; long long idx; };
        std::vector<Src> src;
        src.reserve((size_t)tile_out_ptr[n_tiles]);
        for (int t = 0; t < n_tiles; t++) {
            const int p0 = tile_pose_ptr[t], L = tile_pose_ptr[t + 1] - p0;
            long long q = tile_out_ptr[t];
            for (int a = 0; a < L; a++)
                for (int b2 = a; b2 < L; b2++) src.push_back({ tile_pose[p0 + a], tile_pose[p0 + b2], q++ });
        }
        std::stable_sort(src.begin(), src.end(), [](const Src& a, const Src& b2) { return a.j1 != b2.j1 ? a.j1 < b2.j1 : a.j2 < b2.j2; });
        for (size_t q = 0; q < src.size(); q++) {
            if (q == 0 || src[q].j1 != src[q - 1].j1 || src[q].j2 != src[q - 1].j2) { gpair_j.push_back(src[q].j1); gpair_j.push_back(src[q].j2); gpair_ptr.push_back((int32_t)q); }
            gpair_src.push_back(src[q].idx);
        }
        gpair_ptr.push_back((int32_t)src.size());
        // free poses without any block still need their diagonal S block (Hpp = 0 there: a singular system, as in g2o)
        if (src.size() > (size_t)0x7FFFFFF0) want_schur = false;
    }
    const int n_gpairs = want_schur ? (int)gpair_j.size() / 2 : 0;
    std::vector<int32_t> gdiag(std::max(nf, 1), -1);
    for (int g = 0; g < n_gpairs; g++) if (gpair_j[2 * g] == gpair_j[2 * g + 1]) gdiag[gpair_j[2 * g]] = g;

    mslam_ba_plan* pl = new mslam_ba_plan();
    pl->ctx = ctx;
    size_t bytes = 0;
    auto reserve = [&](size_t n) { bytes = align_up(bytes, 256); size_t o = bytes; bytes += n; return o; };
    const size_t o_poses = reserve(sizeof(double) * 7 * NP), o_points = reserve(sizeof(double) * 3 * NL),
                 o_ep = reserve(sizeof(int32_t) * E), o_info = reserve(sizeof(double) * E), o_intr = reserve(sizeof(double) * 4 * NP),
                 o_fidx = reserve(sizeof(int32_t) * NP), o_erow = reserve(sizeof(int32_t) * (R + 1)), o_rowp = reserve(sizeof(int32_t) * R),
                 o_rec = reserve(sizeof(int32_t) * E), o_obs = reserve(sizeof(double2) * E),
                 o_ptp = reserve(sizeof(int32_t) * (NL + 1)), o_ptk = reserve(sizeof(int32_t) * E),
                 o_col = reserve(sizeof(int32_t) * std::max(nnzb, 1)), o_brow = reserve(sizeof(int32_t) * std::max(nnzb, 1)),
                 o_err = reserve(sizeof(double) * 2 * E), o_chi = reserve(sizeof(double) * E),
                 o_hpp = reserve(sizeof(double) * 36 * std::max(nf, 1)), o_hll = reserve(sizeof(double) * 9 * NL),
                 o_hpl = reserve(sizeof(double) * 18 * std::max(nnzb, 1)), o_b = reserve(sizeof(double) * (6 * nf + 3 * NL)),
                 o_kpose = reserve(sizeof(int32_t) * (size_t)E),
                 o_ptr = reserve(sizeof(double) * NL), o_chp = reserve(sizeof(double) * ((size_t)NL / CHI_SLICE + 1)),
                 o_pp = reserve(sizeof(double) * 27 * (size_t)std::max(nf, 1) * n_chunks_max), o_rc = reserve(sizeof(double)),
                 o_lam = reserve(sizeof(double));
    size_t o_tp0 = 0, o_tpp = 0, o_tp = 0, o_tbp = 0, o_tb = 0, o_top = 0, o_gj = 0, o_gp = 0, o_gs = 0, o_part = 0, o_dinv = 0, o_db = 0, o_gd = 0, o_tc = 0,
           o_S = 0, o_bs = 0, o_x = 0, o_bk = 0, o_psc = 0, o_sp = 0, o_sl = 0;
    if (want_schur) {
        o_tp0 = reserve(sizeof(int32_t) * (n_tiles + 1));
        o_tpp = reserve(sizeof(int32_t) * (n_tiles + 1)); o_tp = reserve(sizeof(int32_t) * std::max<size_t>(tile_pose.size(), 1));
        o_tbp = reserve(sizeof(int32_t) * (n_tiles + 1)); o_tb = reserve(sizeof(int32_t) * std::max<size_t>(tile_blk.size(), 1));
        o_top = reserve(sizeof(long long) * (n_tiles + 1));
        o_gj = reserve(sizeof(int32_t) * 2 * std::max(n_gpairs, 1)); o_gp = reserve(sizeof(int32_t) * (n_gpairs + 1));
        o_gs = reserve(sizeof(long long) * std::max<size_t>(gpair_src.size(), 1));
        o_part = reserve(sizeof(double) * 36 * (size_t)std::max<long long>(tile_out_ptr[n_tiles], 1));
        o_dinv = reserve(sizeof(double) * 9 * NL); o_db = reserve(sizeof(double) * 12 * NL);
        o_gd = reserve(sizeof(int32_t) * std::max(nf, 1)); o_tc = reserve(sizeof(double) * 6 * (size_t)std::max<long long>(tile_out_ptr[n_tiles], 1));
        o_S = reserve(sizeof(double) * 36 * (size_t)nf * nf); o_bs = reserve(sizeof(double) * 6 * nf);
        o_x = reserve(sizeof(double) * (6 * nf + 3 * NL)); o_bk = reserve(sizeof(double) * 3 * NL);
        o_psc = reserve(sizeof(double) * NL); o_sp = reserve(sizeof(double) * ((size_t)NL / CHI_SLICE + 1)); o_sl = reserve(sizeof(double));
    }
    hipError_t he = hipMalloc(&pl->arena, bytes);
    if (he != hipSuccess) { ctx->last_hip_error = std::string("hipMalloc(ba plan): ") + hipGetErrorString(he); delete pl; return MSLAM_ERR_HIP; }
    char* base = (char*)pl->arena;
    hipStream_t st = ctx->stream;
    auto up = [&](size_t off, const void* src, size_t n) { return n ? hipMemcpyAsync(base + off, src, n, hipMemcpyHostToDevice, st) : hipSuccess; };
    const double lam0 = 0.0;
    std::vector<double> info_k(E);
    pl->h_edge_of_k.resize(E);
    std::vector<int32_t> rec_pt(E), kpose(E);
    for (int r = 0; r < R; r++)
        for (int k = erowptr[r]; k < erowptr[r + 1]; k++) kpose[(size_t)k] = row_pose[(size_t)r];
    for (int k = 0; k < E; k++) { pl->h_edge_of_k[(size_t)k] = rec[(size_t)k].y; info_k[(size_t)k] = p->info[rec[(size_t)k].y]; rec_pt[(size_t)k] = rec[(size_t)k].x; }
    std::vector<hipError_t> errs = {
        up(o_poses, p->poses, sizeof(double) * 7 * NP), up(o_points, p->points, sizeof(double) * 3 * NL),
        up(o_ep, p->edge_pose, sizeof(int32_t) * E), up(o_info, info_k.data(), sizeof(double) * E), up(o_intr, p->intr, sizeof(double) * 4 * NP),
        up(o_fidx, fidx.data(), sizeof(int32_t) * NP), up(o_erow, erowptr.data(), sizeof(int32_t) * (R + 1)),
        up(o_rowp, row_pose.data(), sizeof(int32_t) * R), up(o_rec, rec_pt.data(), sizeof(int32_t) * E), up(o_kpose, kpose.data(), sizeof(int32_t) * E), up(o_obs, obs_k.data(), sizeof(double2) * E),
        up(o_ptp, pt_ptr.data(), sizeof(int32_t) * (NL + 1)), up(o_ptk, pt_k.data(), sizeof(int32_t) * E),
        up(o_col, col.data(), sizeof(int32_t) * nnzb), up(o_brow, blk_row.data(), sizeof(int32_t) * nnzb), up(o_lam, &lam0, sizeof(double)) };
    if (want_schur) {
        errs.push_back(up(o_tp0, tile_pt0.data(), sizeof(int32_t) * (n_tiles + 1)));
        errs.push_back(up(o_tpp, tile_pose_ptr.data(), sizeof(int32_t) * (n_tiles + 1)));
        errs.push_back(up(o_tp, tile_pose.data(), sizeof(int32_t) * tile_pose.size()));
        errs.push_back(up(o_tbp, tile_blk_ptr.data(), sizeof(int32_t) * (n_tiles + 1)));
        errs.push_back(up(o_tb, tile_blk.data(), sizeof(int32_t) * tile_blk.size()));
        errs.push_back(up(o_top, tile_out_ptr.data(), sizeof(long long) * (n_tiles + 1)));
        errs.push_back(up(o_gj, gpair_j.data(), sizeof(int32_t) * gpair_j.size()));
        errs.push_back(up(o_gp, gpair_ptr.data(), sizeof(int32_t) * gpair_ptr.size()));
        errs.push_back(up(o_gs, gpair_src.data(), sizeof(long long) * gpair_src.size()));
        errs.push_back(up(o_gd, gdiag.data(), sizeof(int32_t) * gdiag.size()));
    }
    errs.push_back(hipStreamSynchronize(st));
    for (hipError_t e : errs)
        if (e != hipSuccess) { ctx->last_hip_error = std::string("ba plan upload: ") + hipGetErrorString(e); (void)hipFree(pl->arena); delete pl; return MSLAM_ERR_HIP; }
    BaDev& d = pl->dev;
    d.n_poses = NP; d.n_points = NL; d.n_edges = E; d.n_free = nf; d.nnzb = nnzb; d.n_rows = R; d.n_chunks_max = n_chunks_max;
    d.huber_delta = p->huber_delta; d.dsqr = (float)(p->huber_delta * p->huber_delta); // RobustKernelHuber: float dsqr
    d.poses = (double*)(base + o_poses); d.points = (double*)(base + o_points);
    d.edge_pose = (int32_t*)(base + o_ep); d.info = (double*)(base + o_info); d.intr = (double*)(base + o_intr); d.fidx = (int32_t*)(base + o_fidx);
    d.erowptr = (int32_t*)(base + o_erow); d.row_pose = (int32_t*)(base + o_rowp); d.rec = (int32_t*)(base + o_rec); d.obs_k = (double2*)(base + o_obs);
    d.pt_ptr = (int32_t*)(base + o_ptp); d.pt_k = (int32_t*)(base + o_ptk); d.blk_row = (int32_t*)(base + o_brow);
    d.err = (double*)(base + o_err); d.chi2 = (double*)(base + o_chi); d.rho1 = nullptr;   // (the Huber weight is a function of chi2: not streamed out per edge)
    d.Hpp = (double*)(base + o_hpp); d.Hll = (double*)(base + o_hll); d.Hpl_val = (double*)(base + o_hpl); d.b = (double*)(base + o_b);
    d.kpose = (int32_t*)(base + o_kpose);
    d.pt_rho = (double*)(base + o_ptr); d.chi_part = (double*)(base + o_chp); d.pose_part = (double*)(base + o_pp); d.robust_chi2 = (double*)(base + o_rc);
    pl->d_poses = (double*)(base + o_poses); pl->d_points = (double*)(base + o_points);
    pl->d_rowptr = (int32_t*)(base + o_erow); pl->d_col = (int32_t*)(base + o_col);   // free rows come first: erowptr[0..nf] IS the block-CSR row pointer
    pl->h_rowptr.assign(erowptr.begin(), erowptr.begin() + nf + 1);
    pl->h_col.assign(col.begin(), col.begin() + nnzb);
    pl->h_row_pose.assign(row_pose.begin(), row_pose.begin() + nf);
    d.col = (int32_t*)(base + o_col);
    pl->d_lam = (double*)(base + o_lam);
    d.n_windows = 1; d.pose_win = nullptr; d.point_win = nullptr; d.win_pt_ptr = nullptr; d.win_pt = nullptr; d.lam_w = pl->d_lam; d.win_out = nullptr;
    if (want_schur) {
        d.n_tiles = n_tiles; d.n_tile_pairs = tile_out_ptr[n_tiles];
        d.tile_pt0 = (int32_t*)(base + o_tp0);
        d.tile_pose_ptr = (int32_t*)(base + o_tpp); d.tile_pose = (int32_t*)(base + o_tp);
        d.tile_blk_ptr = (int32_t*)(base + o_tbp); d.tile_blk = (int32_t*)(base + o_tb); d.tile_out_ptr = (long long*)(base + o_top);
        d.n_gpairs = n_gpairs; d.gpair_j = (int32_t*)(base + o_gj); d.gpair_ptr = (int32_t*)(base + o_gp); d.gpair_src = (long long*)(base + o_gs);
        d.tile_part = (double*)(base + o_part);
        d.Dinv = (double*)(base + o_dinv); d.DD = (double*)(base + o_db); d.gdiag = (int32_t*)(base + o_gd); d.tile_coef = (double*)(base + o_tc);
        d.S = (double*)(base + o_S); d.bs = (double*)(base + o_bs); d.x = (double*)(base + o_x); d.points_bk = (double*)(base + o_bk);
        d.pt_scale = (double*)(base + o_psc); d.scale_part = (double*)(base + o_sp); d.scale_l = (double*)(base + o_sl);
        pl->has_schur = true; pl->schur_lmax = schur_lmax;
    }
    *out = pl;
    return MSLAM_OK;
}

void mslam_ba_plan_destroy(mslam_ba_plan* pl)
{
    if (!pl) return;
    (void)hipSetDevice(pl->ctx->device);
    (void)hipStreamSynchronize(pl->ctx->stream);
    if (pl->win_arena) (void)hipFree(pl->win_arena);
    if (pl->arena) (void)hipFree(pl->arena);
    delete pl;
}

int mslam_ba_plan_view(mslam_ba_plan* pl, mslam_ba_dev_view* v)
{
    if (!pl || !v) return MSLAM_ERR_INVALID;
    const BaDev& d = pl->dev;
    v->poses = pl->d_poses; v->points = pl->d_points; v->err = d.err; v->chi2 = d.chi2; v->rho1 = d.rho1;
    v->Hpp = d.Hpp; v->Hll = d.Hll; v->Hpl_val = d.Hpl_val; v->b = d.b; v->Hpl_rowptr = pl->d_rowptr; v->Hpl_col = pl->d_col;
    v->robust_chi2 = d.robust_chi2; v->n_free_poses = d.n_free; v->nnzb = d.nnzb;
    return MSLAM_OK;
}

int mslam_ba_plan_set_state(mslam_ba_plan* pl, const double* poses, const double* points)
{
    if (!pl) return MSLAM_ERR_INVALID;
    mslam_ctx* ctx = pl->ctx;
    MSLAM_TRY(ctx_set_device(ctx));
    if (poses) HIP_TRY(ctx, hipMemcpyAsync(pl->d_poses, poses, sizeof(double) * 7 * pl->dev.n_poses, hipMemcpyHostToDevice, ctx->stream));
    if (points) HIP_TRY(ctx, hipMemcpyAsync(pl->d_points, points, sizeof(double) * 3 * pl->dev.n_points, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MSLAM_OK;
}

int mslam_ba_plan_set_robust(mslam_ba_plan* pl, const double* info, double huber_delta)
{
    if (!pl) return MSLAM_ERR_INVALID;
    mslam_ctx* ctx = pl->ctx;
    MSLAM_TRY(ctx_set_device(ctx));
    if (info) {   // the device copy is in row order
        std::vector<double> info_k((size_t)pl->dev.n_edges);
        for (int k = 0; k < pl->dev.n_edges; k++) info_k[(size_t)k] = info[pl->h_edge_of_k[(size_t)k]];
        HIP_TRY(ctx, hipMemcpyAsync(const_cast<double*>(pl->dev.info), info_k.data(), sizeof(double) * pl->dev.n_edges, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    pl->dev.huber_delta = huber_delta;
    pl->dev.dsqr = (float)(huber_delta * huber_delta);
    return MSLAM_OK;
}

int mslam_ba_plan_set_windows(mslam_ba_plan* pl, const int32_t* pose_window, const int32_t* point_window, int n_windows)
{
    if (!pl || !pose_window || !point_window || n_windows < 1) return MSLAM_ERR_INVALID;
    mslam_ctx* ctx = pl->ctx;
    MSLAM_TRY(ctx_set_device(ctx));
    BaDev& d = pl->dev;
    const int nf = d.n_free, NL = d.n_points;
    std::vector<int32_t> pw(std::max(nf, 1), 0), ptr(n_windows + 1, 0), lst(NL);
    for (int f = 0; f < nf; f++) {
        const int w = pose_window[pl->h_row_pose[f]];
        if (w < 0 || w >= n_windows) return MSLAM_ERR_INVALID;
        pw[f] = w;
    }
    for (int i = 0; i < NL; i++) { if (point_window[i] < 0 || point_window[i] >= n_windows) return MSLAM_ERR_INVALID; ptr[point_window[i] + 1]++; }
    for (int w = 0; w < n_windows; w++) ptr[w + 1] += ptr[w];
    {
        std::vector<int32_t> fill(n_windows, 0);
        for (int i = 0; i < NL; i++) lst[ptr[point_window[i]] + fill[point_window[i]]++] = i;
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (pl->win_arena) { HIP_TRY(ctx, hipFree(pl->win_arena)); pl->win_arena = nullptr; }
    size_t bytes = 0;
    auto reserve = [&](size_t n) { bytes = align_up(bytes, 256); size_t o = bytes; bytes += n; return o; };
    const size_t o_pw = reserve(sizeof(int32_t) * std::max(nf, 1)), o_qw = reserve(sizeof(int32_t) * NL), o_ptr = reserve(sizeof(int32_t) * (n_windows + 1)),
                 o_lst = reserve(sizeof(int32_t) * NL), o_lam = reserve(sizeof(double) * n_windows), o_out = reserve(sizeof(double) * 2 * n_windows);
    HIP_TRY(ctx, hipMalloc(&pl->win_arena, bytes));
    char* base = (char*)pl->win_arena;
    HIP_TRY(ctx, hipMemcpy(base + o_pw, pw.data(), sizeof(int32_t) * std::max(nf, 1), hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(base + o_qw, point_window, sizeof(int32_t) * NL, hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(base + o_ptr, ptr.data(), sizeof(int32_t) * (n_windows + 1), hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(base + o_lst, lst.data(), sizeof(int32_t) * NL, hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemset(base + o_lam, 0, sizeof(double) * n_windows));
    d.n_windows = n_windows; d.pose_win = (int32_t*)(base + o_pw); d.point_win = (int32_t*)(base + o_qw);
    d.win_pt_ptr = (int32_t*)(base + o_ptr); d.win_pt = (int32_t*)(base + o_lst); d.lam_w = (double*)(base + o_lam); d.win_out = (double*)(base + o_out);
    pl->d_lam = (double*)(base + o_lam);
    return MSLAM_OK;
}

int mslam_ba_plan_build_dev(mslam_ba_plan* pl)
{
    if (!pl) return MSLAM_ERR_INVALID;
    mslam_ctx* ctx = pl->ctx;
    MSLAM_TRY(ctx_set_device(ctx));
    const BaDev& d = pl->dev;
    pl->built_delta = d.huber_delta; pl->built_dsqr = d.dsqr;
    LAUNCH(ctx, K_BA_POSES, ba_edges_k, dim3(d.n_chunks_max, d.n_rows), dim3(EDGE_THREADS), 0, d);
    LAUNCH(ctx, K_BA_POINTS, ba_points_k, dim3((d.n_points + 255) / 256), dim3(256), 0, d);
    if (d.n_points <= 2 * CHI_SLICE && d.n_free * 27 <= 4096) {
        LAUNCH(ctx, K_BA_FINAL, ba_final_k, dim3(1), dim3(256), 0, d, 0);
    } else {
        const int n_pose_wg = (d.n_free * 27 + 255) / 256, n_slices = (d.n_points + CHI_SLICE - 1) / CHI_SLICE;
        LAUNCH(ctx, K_BA_FINAL, ba_final_k, dim3(n_pose_wg + n_slices), dim3(256), 0, d, n_pose_wg);
        LAUNCH(ctx, K_BA_FINAL, ba_final2_k, dim3(1), dim3(256), 0, (const double*)d.chi_part, n_slices, d.robust_chi2);
    }
    if (d.n_windows > 1) LAUNCH(ctx, K_BA_FINAL, ba_window_sums_k, dim3(d.n_windows), dim3(256), 0, d, 0);
    HIP_TRY(ctx, hipGetLastError());
    return MSLAM_OK;
}

int mslam_ba_plan_download(mslam_ba_plan* pl, mslam_ba_system* s)
{
    if (!pl || !s) return MSLAM_ERR_INVALID;
    mslam_ctx* ctx = pl->ctx;
    MSLAM_TRY(ctx_set_device(ctx));
    const BaDev& d = pl->dev;
    hipStream_t st = ctx->stream;
    auto dn = [&](void* dst, const void* src, size_t n) { return (dst && n) ? hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, st) : hipSuccess; };
    // err / chi2 / rho1 live in row order on the device (coalesced stores in ba_edges_k): they come back through a staging copy
    // and are put into the caller's edge order here
    const size_t E = (size_t)d.n_edges;
    std::vector<double> h_err(s->err ? 2 * E : 0), h_chi((s->chi2 || s->rho1) ? E : 0);
    HIP_TRY(ctx, dn(s->err ? h_err.data() : nullptr, d.err, sizeof(double) * 2 * E));
    HIP_TRY(ctx, dn(h_chi.empty() ? nullptr : h_chi.data(), d.chi2, sizeof(double) * E));
    HIP_TRY(ctx, dn(s->Hpp, d.Hpp, sizeof(double) * 36 * d.n_free));
    HIP_TRY(ctx, dn(s->Hll, d.Hll, sizeof(double) * 9 * d.n_points));
    HIP_TRY(ctx, dn(s->Hpl_val, d.Hpl_val, sizeof(double) * 18 * d.nnzb));
    HIP_TRY(ctx, dn(s->b, d.b, sizeof(double) * (6 * d.n_free + 3 * d.n_points)));
    HIP_TRY(ctx, dn(&s->robust_chi2, d.robust_chi2, sizeof(double)));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    for (size_t k = 0; k < E; k++) {
        const size_t e = (size_t)pl->h_edge_of_k[k];
        if (s->err) { s->err[2 * e] = h_err[2 * k]; s->err[2 * e + 1] = h_err[2 * k + 1]; }
        if (s->chi2) s->chi2[e] = h_chi[k];
        if (s->rho1) {   // RobustKernelHuber::robustify's rho' (robust_kernel_impl.cpp:78-91), the same operations as the kernel's huber()
            double r1 = 1.0;
            if (pl->built_delta > 0 && !(h_chi[k] <= (double)pl->built_dsqr)) r1 = pl->built_delta / std::sqrt(h_chi[k]);
            s->rho1[e] = r1;
        }
    }
    if (s->Hpl_rowptr) std::memcpy(s->Hpl_rowptr, pl->h_rowptr.data(), sizeof(int32_t) * (d.n_free + 1));
    if (s->Hpl_col && d.nnzb) std::memcpy(s->Hpl_col, pl->h_col.data(), sizeof(int32_t) * d.nnzb);
    s->n_free_poses = d.n_free; s->nnzb = d.nnzb;
    return MSLAM_OK;
}

static int schur_launch(mslam_ba_plan* pl)
{
    mslam_ctx* ctx = pl->ctx;
    const BaDev& d = pl->dev;
    const size_t n6 = 6 * (size_t)d.n_free;
    HIP_TRY(ctx, hipMemsetAsync(d.S, 0, sizeof(double) * n6 * n6, ctx->stream));   // pose pairs that share no point
    LAUNCH(ctx, K_BA_SCHUR, ba_schur_points_k, dim3((d.n_points + 255) / 256), dim3(256), 0, d);
    {   // the instantiation whose stage holds the widest tile: 9 L + 6 pieces of 16 bytes in R x 64
        const int Lm = pl->schur_lmax;
        auto lds = [&](int R) { return sizeof(double) * (size_t)std::max(8 * R * 128, 2 * 64 * 36) + sizeof(int32_t) * (size_t)SCHUR_TP * Lm; };
        if (Lm <= 6) { LAUNCH(ctx, K_BA_SCHUR_TILES, ba_schur_tiles_k<1>, dim3(d.n_tiles), dim3(256), lds(1), d); }
        else if (Lm <= 13) { LAUNCH(ctx, K_BA_SCHUR_TILES, ba_schur_tiles_k<2>, dim3(d.n_tiles), dim3(256), lds(2), d); }
        else if (Lm <= 27) { LAUNCH(ctx, K_BA_SCHUR_TILES, ba_schur_tiles_k<4>, dim3(d.n_tiles), dim3(256), lds(4), d); }
        else {
            static_assert(9 * SCHUR_LMAX + 6 <= 64 * 10, "widest tile fits the largest instantiation");
            // per launch, on the context's device (a cached result would be the first device's, failures included)
            HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&ba_schur_tiles_k<10>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)(sizeof(double) * 8 * 10 * 128 + sizeof(int32_t) * SCHUR_TP * SCHUR_LMAX)));
            LAUNCH(ctx, K_BA_SCHUR_TILES, ba_schur_tiles_k<10>, dim3(d.n_tiles), dim3(256), lds(10), d);
        }
    }
    const long long items = (long long)d.n_gpairs * 36 * FIN_SPLIT;
    LAUNCH(ctx, K_BA_SCHUR_FINAL, ba_schur_final_k, dim3((unsigned)std::min<long long>(4096, (items + 255) / 256 + 1)), dim3(256), 0, d);
    HIP_TRY(ctx, hipGetLastError());
    return MSLAM_OK;
}

int mslam_ba_plan_schur_dev(mslam_ba_plan* pl, double lambda)
{
    if (!pl) return MSLAM_ERR_INVALID;
    if (!pl->has_schur) return MSLAM_ERR_CAPACITY;
    mslam_ctx* ctx = pl->ctx;
    MSLAM_TRY(ctx_set_device(ctx));
    std::vector<double> lam((size_t)pl->dev.n_windows, lambda);   // one lambda for every window
    HIP_TRY(ctx, hipMemcpyAsync(pl->d_lam, lam.data(), sizeof(double) * lam.size(), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return schur_launch(pl);
}

int mslam_ba_plan_schur_windows_dev(mslam_ba_plan* pl, const double* lambda_per_window)
{
    if (!pl || !lambda_per_window) return MSLAM_ERR_INVALID;
    if (!pl->has_schur) return MSLAM_ERR_CAPACITY;
    mslam_ctx* ctx = pl->ctx;
    MSLAM_TRY(ctx_set_device(ctx));
    HIP_TRY(ctx, hipMemcpyAsync(pl->d_lam, lambda_per_window, sizeof(double) * pl->dev.n_windows, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // lambda_per_window may be a temporary of the caller
    return schur_launch(pl);
}

int mslam_ba_plan_download_schur(mslam_ba_plan* pl, double* S, double* bs)
{
    if (!pl || !pl->has_schur) return MSLAM_ERR_INVALID;
    mslam_ctx* ctx = pl->ctx;
    MSLAM_TRY(ctx_set_device(ctx));
    const BaDev& d = pl->dev;
    const size_t n6 = 6 * (size_t)d.n_free;
    if (S) HIP_TRY(ctx, hipMemcpyAsync(S, d.S, sizeof(double) * n6 * n6, hipMemcpyDeviceToHost, ctx->stream));
    if (bs) HIP_TRY(ctx, hipMemcpyAsync(bs, d.bs, sizeof(double) * n6, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MSLAM_OK;
}

int mslam_ba_plan_backsub_dev(mslam_ba_plan* pl, const double* x_pose_host, int apply_to_points)
{
    if (!pl || !pl->has_schur || !x_pose_host) return MSLAM_ERR_INVALID;
    mslam_ctx* ctx = pl->ctx;
    MSLAM_TRY(ctx_set_device(ctx));
    const BaDev& d = pl->dev;
    HIP_TRY(ctx, hipMemcpyAsync(d.x, x_pose_host, sizeof(double) * 6 * d.n_free, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); // x_pose_host may be a temporary of the caller
    LAUNCH(ctx, K_BA_BACKSUB, ba_backsub_k, dim3((d.n_points + 255) / 256), dim3(256), 0, d, apply_to_points, pl->d_points);
    const int n_slices = (d.n_points + CHI_SLICE - 1) / CHI_SLICE;
    LAUNCH(ctx, K_BA_BACKSUB, ba_slice_sum_k, dim3(n_slices), dim3(256), 0, (const double*)d.pt_scale, d.n_points, d.scale_part);
    LAUNCH(ctx, K_BA_BACKSUB, ba_final2_k, dim3(1), dim3(256), 0, (const double*)d.scale_part, n_slices, d.scale_l);
    if (d.n_windows > 1) LAUNCH(ctx, K_BA_BACKSUB, ba_window_sums_k, dim3(d.n_windows), dim3(256), 0, d, 1);
    HIP_TRY(ctx, hipGetLastError());
    return MSLAM_OK;
}

int mslam_ba_plan_download_x(mslam_ba_plan* pl, double* x, double* scale_landmarks)
{
    if (!pl || !pl->has_schur) return MSLAM_ERR_INVALID;
    mslam_ctx* ctx = pl->ctx;
    MSLAM_TRY(ctx_set_device(ctx));
    const BaDev& d = pl->dev;
    if (x) HIP_TRY(ctx, hipMemcpyAsync(x, d.x, sizeof(double) * (6 * d.n_free + 3 * (size_t)d.n_points), hipMemcpyDeviceToHost, ctx->stream));
    if (scale_landmarks) HIP_TRY(ctx, hipMemcpyAsync(scale_landmarks, d.scale_l, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MSLAM_OK;
}

int mslam_ba_plan_download_window_sums(mslam_ba_plan* pl, double* robust_chi2, double* scale_landmarks)
{
    if (!pl || pl->dev.n_windows < 2 || !pl->dev.win_out) return MSLAM_ERR_INVALID;
    mslam_ctx* ctx = pl->ctx;
    MSLAM_TRY(ctx_set_device(ctx));
    const int W = pl->dev.n_windows;
    std::vector<double> h((size_t)2 * W);
    HIP_TRY(ctx, hipMemcpyAsync(h.data(), pl->dev.win_out, sizeof(double) * 2 * W, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (int w = 0; w < W; w++) { if (robust_chi2) robust_chi2[w] = h[2 * w]; if (scale_landmarks) scale_landmarks[w] = h[2 * w + 1]; }
    return MSLAM_OK;
}

int mslam_ba_plan_points_snapshot_dev(mslam_ba_plan* pl, int restore)
{
    if (!pl || !pl->has_schur) return MSLAM_ERR_INVALID;
    mslam_ctx* ctx = pl->ctx;
    MSLAM_TRY(ctx_set_device(ctx));
    const BaDev& d = pl->dev;
    const size_t n = sizeof(double) * 3 * (size_t)d.n_points;
    if (restore) HIP_TRY(ctx, hipMemcpyAsync(pl->d_points, d.points_bk, n, hipMemcpyDeviceToDevice, ctx->stream));
    else HIP_TRY(ctx, hipMemcpyAsync(d.points_bk, pl->d_points, n, hipMemcpyDeviceToDevice, ctx->stream));
    return MSLAM_OK;
}

int mslam_ba_plan_get_state(mslam_ba_plan* pl, double* poses, double* points)
{
    if (!pl) return MSLAM_ERR_INVALID;
    mslam_ctx* ctx = pl->ctx;
    MSLAM_TRY(ctx_set_device(ctx));
    if (poses) HIP_TRY(ctx, hipMemcpyAsync(poses, pl->d_poses, sizeof(double) * 7 * pl->dev.n_poses, hipMemcpyDeviceToHost, ctx->stream));
    if (points) HIP_TRY(ctx, hipMemcpyAsync(points, pl->d_points, sizeof(double) * 3 * pl->dev.n_points, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MSLAM_OK;
}

int mslam_ba_build(mslam_ctx* ctx, const mslam_ba_problem* p, mslam_ba_system* s)
{
    if (!ctx || !s) return MSLAM_ERR_INVALID;
    mslam_ba_plan* pl = nullptr;
    MSLAM_TRY(mslam_ba_plan_create(ctx, p, &pl));
    int rc = mslam_ba_plan_build_dev(pl);
    if (rc == MSLAM_OK) rc = mslam_ba_plan_download(pl, s);
    mslam_ba_plan_destroy(pl);
    return rc;
}

} // extern "C"
