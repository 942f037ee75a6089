// ba.hip -- K9: bundle-adjustment linearisation on gfx950: reprojection residuals, Huber
// weights, analytic 2x3 / 2x6 Jacobians and the J^T w Omega J blocks, written straight into the
// block layout the host-side Schur / LDL^T solve consumes (Hpp, Hll, block-CSR Hpl, b).
//
// Stands in for one pass of BlockSolver<6,3>::buildSystem (reference ThirdParty/g2o/g2o/core/
// block_solver.hpp:502-560) over the graph of Util::BundleAdjustment (src/Util.cpp:34-170):
// EdgeSE3ProjectXYZ::computeError / linearizeOplus (types/types_six_dof_expmap.h:90-95,
// types_six_dof_expmap.cpp:103-139), RobustKernelHuber::robustify (core/robust_kernel_impl.cpp:78-91),
// BaseBinaryEdge::constructQuadraticForm (core/base_binary_edge.hpp:55-120).
//
// HBM layout (device-resident plan, built once per graph like g2o's buildStructure):
//   state    poses f64[n_poses][7] (t, q xyzw), points f64[n_points][3]
//   edges    edge_pose/edge_point i32[E], obs f64[E][2], info f64[E], intr f64[n_poses][4]
//   by point pt_ptr i32[n_points+1], pt_edge i32[E]   (edges of a point, ascending edge id)
//   Hpl      rowptr i32[fp+1], col i32[nnzb], val f64[nnzb][6][3] (row = free pose, col = point,
//            sorted by point), blk_edge i32[nnzb] (edge of a block), slot i32[E] (block of an edge)
// Kernels:
//   ba_points_k  16 lanes per point: lane l takes the point's edges l, l+16, ...: residual, Huber
//                weight, Jacobians, writes err/chi2/rho1 and the edge's 6x3 Hpl block, keeps the 3x3
//                Hll / b_l contribution in registers; a fixed-order butterfly over the 16 lanes sums
//                them (no float atomics: results are run-to-run identical)
//   ba_poses_k   grid (free pose, chunk): one lane per Hpl block of the pose recomputes B and the
//                weight (cheaper than re-reading 216 B per edge) and a fixed-tree block reduction
//                produces the chunk's 6x6 (21 unique) + 6 sums
//   ba_final_k   adds the chunk partials per pose in chunk order, reduces the robust chi2
// f64 throughout (g2o is double; 1e-4 relative parity is an f64 target).  FMA contraction is
// allowed in this file: no discrete decision depends on the last bit here.
#include "common.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#pragma clang fp contract(fast)

namespace {

constexpr int GROUP = 16;          // lanes per point
constexpr int BA_THREADS = 256;
constexpr int POSE_CHUNK = 256;    // Hpl blocks per pose-chunk workgroup

struct BaDev {
    int n_poses, n_points, n_edges, n_free, nnzb, n_chunks_max;
    double huber_delta; float dsqr;
    const double* poses; const double* points;
    const int32_t* edge_pose; const int32_t* edge_point; const double* obs; const double* info; const double* intr;
    const int32_t* fidx;      // free index per pose or -1
    const int32_t* pt_ptr; const int32_t* pt_edge;
    const int32_t* rowptr; const int32_t* blk_edge; const int32_t* slot;
    double* err; double* chi2; double* rho1; double* Hpp; double* Hll; double* Hpl_val; double* b;
    double* pt_rho;           // [n_points] robust chi2 per point
    double* chi_part;         // [ceil(n_points / 4096)] slice sums of pt_rho (large problems)
    double* pose_part;        // [n_free][n_chunks_max][27]
    double* robust_chi2;      // [1]
    // ---- Schur complement (f1): reduced pose system S x_p = b_s, landmark back-substitution
    const int32_t* col;       // [nnzb] point of a block
    const int32_t* blk_of;    // [n_free][n_points] block of (pose row, point) or -1
    double* Dinv;             // [n_points][9]  (Hll + lambda I)^-1
    double* db;               // [n_points][3]  Dinv b_l
    double* BD;               // [nnzb][18]     B Dinv
    double* cblk;             // [nnzb][6]      B (Dinv b_l)
    double* pair_part;        // [npairs][n_chunks_max][36]
    double* coef_part;        // [n_free][n_chunks_max][6]
    double* S;                // [6 n_free][6 n_free]
    double* bs;               // [6 n_free]
    double* x;                // [6 n_free + 3 n_points]
    double* points_bk;        // [n_points][3] snapshot for rejected LM steps
    double* scale_part;       // [ceil(n_points/256)] partial sums of x_l (lambda x_l + b_l)
    double* scale_l;          // [1]
    double lambda;
};

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

__global__ __launch_bounds__(BA_THREADS)
void ba_points_k(BaDev d)
{
    const int gid = (blockIdx.x * BA_THREADS + threadIdx.x) / GROUP; // point
    const int gl = threadIdx.x & (GROUP - 1);
    const bool live = gid < d.n_points;
    double acc[13]; // Hll upper (6): 00 01 02 11 12 22; b_l (3); rho (1); pad
#pragma unroll
    for (int k = 0; k < 13; k++) acc[k] = 0.0;
    if (live) {
        const double X0 = d.points[3 * gid], X1 = d.points[3 * gid + 1], X2 = d.points[3 * gid + 2];
        const int e0 = d.pt_ptr[gid], e1 = d.pt_ptr[gid + 1];
        for (int k = e0 + gl; k < e1; k += GROUP) {
            const int e = d.pt_edge[k];
            const int jp = d.edge_pose[e];
            const double* in = d.intr + 4 * jp;
            EdgeLin L;
            edge_eval(d.poses + 7 * jp, X0, X1, X2, d.obs[2 * e], d.obs[2 * e + 1], in[0], in[1], in[2], in[3], L);
            const double om = d.info[e];
            const double chi2 = L.e0 * (om * L.e0) + L.e1 * (om * L.e1);
            double rho0, rho1;
            huber(chi2, d.huber_delta, d.dsqr, rho0, rho1);
            d.err[2 * e] = L.e0; d.err[2 * e + 1] = L.e1; d.chi2[e] = chi2; d.rho1[e] = rho1;
            const double wom = rho1 * om;
            const double r0 = -(om * L.e0) * rho1, r1 = -(om * L.e1) * rho1;
            const double* A = L.A;
            acc[0] += wom * (A[0] * A[0] + A[3] * A[3]); acc[1] += wom * (A[0] * A[1] + A[3] * A[4]);
            acc[2] += wom * (A[0] * A[2] + A[3] * A[5]); acc[3] += wom * (A[1] * A[1] + A[4] * A[4]);
            acc[4] += wom * (A[1] * A[2] + A[4] * A[5]); acc[5] += wom * (A[2] * A[2] + A[5] * A[5]);
            acc[6] += A[0] * r0 + A[3] * r1; acc[7] += A[1] * r0 + A[4] * r1; acc[8] += A[2] * r0 + A[5] * r1;
            acc[9] += rho0;
            const int sl = d.slot[e];
            if (sl >= 0) { // B^T (w Omega) A, row-major 6x3
                double* Hb = d.Hpl_val + (size_t)sl * 18;
#pragma unroll
                for (int a = 0; a < 6; a++) {
                    const double ba = wom * L.B[a], bb = wom * L.B[6 + a];
                    Hb[a * 3 + 0] = ba * A[0] + bb * A[3];
                    Hb[a * 3 + 1] = ba * A[1] + bb * A[4];
                    Hb[a * 3 + 2] = ba * A[2] + bb * A[5];
                }
            }
        }
    }
    // fixed-order butterfly over the 16 lanes of the group
#pragma unroll
    for (int o = GROUP / 2; o > 0; o >>= 1)
#pragma unroll
        for (int k = 0; k < 10; k++) acc[k] += __shfl_xor(acc[k], o, GROUP);
    if (live && gl == 0) {
        double* H = d.Hll + (size_t)gid * 9;
        H[0] = acc[0]; H[1] = acc[1]; H[2] = acc[2];
        H[3] = acc[1]; H[4] = acc[3]; H[5] = acc[4];
        H[6] = acc[2]; H[7] = acc[4]; H[8] = acc[5];
        double* bl = d.b + 6 * d.n_free + 3 * (size_t)gid;
        bl[0] = acc[6]; bl[1] = acc[7]; bl[2] = acc[8];
        d.pt_rho[gid] = acc[9];
    }
}

__global__ __launch_bounds__(POSE_CHUNK)
void ba_poses_k(BaDev d)
{
    const int f = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
    const int k0 = d.rowptr[f], k1 = d.rowptr[f + 1];
    const int k = k0 + chunk * POSE_CHUNK + tid;
    if (k0 + chunk * POSE_CHUNK >= k1) return; // whole block out of range
    __shared__ double s_red[27 * (POSE_CHUNK / 64)];
    double v[27];
#pragma unroll
    for (int q = 0; q < 27; q++) v[q] = 0.0;
    if (k < k1) {
        const int e = d.blk_edge[k];
        const int jp = d.edge_pose[e], ip = d.edge_point[e];
        const double* in = d.intr + 4 * jp;
        EdgeLin L;
        edge_eval(d.poses + 7 * jp, d.points[3 * ip], d.points[3 * ip + 1], d.points[3 * ip + 2], d.obs[2 * e], d.obs[2 * e + 1],
                  in[0], in[1], in[2], in[3], L);
        const double om = d.info[e];
        const double chi2 = L.e0 * (om * L.e0) + L.e1 * (om * L.e1);
        double rho0, rho1;
        huber(chi2, d.huber_delta, d.dsqr, rho0, rho1);
        const double wom = rho1 * om;
        const double r0 = -(om * L.e0) * rho1, r1 = -(om * L.e1) * rho1;
        int q = 0;
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
            for (int c = a; c < 6; c++) v[q++] = wom * (L.B[a] * L.B[c] + L.B[6 + a] * L.B[6 + c]);
#pragma unroll
        for (int a = 0; a < 6; a++) v[21 + a] = L.B[a] * r0 + L.B[6 + a] * r1;
    }
    // fixed-shape reduction: xor butterfly inside each wavefront (registers), then the 4 wavefront sums in index order
    // (replaces a 55 KiB LDS tree that allowed only two workgroups per CU)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int q = 0; q < 27; q++) v[q] += __shfl_xor(v[q], o);
    if ((tid & 63) == 0)
#pragma unroll
        for (int q = 0; q < 27; q++) s_red[q * (POSE_CHUNK / 64) + (tid >> 6)] = v[q];
    __syncthreads();
    if (tid < 27) {
        double t = s_red[tid * (POSE_CHUNK / 64)];
#pragma unroll
        for (int w = 1; w < POSE_CHUNK / 64; w++) t += s_red[tid * (POSE_CHUNK / 64) + w];
        d.pose_part[((size_t)f * d.n_chunks_max + chunk) * 27 + tid] = t;
    }
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
            const int nch = (d.rowptr[f + 1] - d.rowptr[f] + POSE_CHUNK - 1) / POSE_CHUNK;
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

__global__ __launch_bounds__(256)
void ba_final2_k(BaDev d, int n_slices)
{
    __shared__ double s_r[256];
    const int tid = threadIdx.x;
    double r = 0.0;
    for (int i = tid; i < n_slices; i += 256) r += d.chi_part[i];
    s_r[tid] = r;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) s_r[tid] += s_r[tid + o]; __syncthreads(); }
    if (tid == 0) d.robust_chi2[0] = s_r[0];
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
    M[0] += d.lambda; M[4] += d.lambda; M[8] += d.lambda;
    const double det = M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
    const double id = 1.0 / det;
    double I[9];
    I[0] = (M[4] * M[8] - M[5] * M[7]) * id; I[1] = (M[2] * M[7] - M[1] * M[8]) * id; I[2] = (M[1] * M[5] - M[2] * M[4]) * id;
    I[3] = (M[5] * M[6] - M[3] * M[8]) * id; I[4] = (M[0] * M[8] - M[2] * M[6]) * id; I[5] = (M[2] * M[3] - M[0] * M[5]) * id;
    I[6] = (M[3] * M[7] - M[4] * M[6]) * id; I[7] = (M[1] * M[6] - M[0] * M[7]) * id; I[8] = (M[0] * M[4] - M[1] * M[3]) * id;
    const double* bl = d.b + 6 * d.n_free + 3 * (size_t)i;
#pragma unroll
    for (int q = 0; q < 9; q++) d.Dinv[(size_t)i * 9 + q] = I[q];
#pragma unroll
    for (int a = 0; a < 3; a++) d.db[(size_t)i * 3 + a] = I[a * 3] * bl[0] + I[a * 3 + 1] * bl[1] + I[a * 3 + 2] * bl[2];
}

__global__ __launch_bounds__(256)
void ba_schur_blocks_k(BaDev d)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= d.nnzb) return;
    const int i = d.col[k];
    const double* B = d.Hpl_val + (size_t)k * 18;
    const double* Di = d.Dinv + (size_t)i * 9;
    const double* dbi = d.db + (size_t)i * 3;
#pragma unroll
    for (int a = 0; a < 6; a++) {
#pragma unroll
        for (int c = 0; c < 3; c++) d.BD[(size_t)k * 18 + a * 3 + c] = B[a * 3] * Di[c] + B[a * 3 + 1] * Di[3 + c] + B[a * 3 + 2] * Di[6 + c];
        d.cblk[(size_t)k * 6 + a] = B[a * 3] * dbi[0] + B[a * 3 + 1] * dbi[1] + B[a * 3 + 2] * dbi[2];
    }
}

// grid (chunk, pair): pair p <-> (j1 <= j2); lanes walk the blocks of pose row j1, look up the block of (j2, same point)
__global__ __launch_bounds__(POSE_CHUNK)
void ba_schur_pairs_k(BaDev d)
{
    const int chunk = blockIdx.x, tid = threadIdx.x;
    int j1 = 0, rem = blockIdx.y; // unpack the upper-triangle pair index
    while (rem >= d.n_free - j1) { rem -= d.n_free - j1; j1++; }
    const int j2 = j1 + rem;
    const int k0 = d.rowptr[j1], k1 = d.rowptr[j1 + 1];
    if (k0 + chunk * POSE_CHUNK >= k1) return;
    const int k = k0 + chunk * POSE_CHUNK + tid;
    __shared__ double s_red[42 * POSE_CHUNK];
    double v[42];
#pragma unroll
    for (int q = 0; q < 42; q++) v[q] = 0.0;
    if (k < k1) {
        const int i = d.col[k];
        const int k2 = d.blk_of[(size_t)j2 * d.n_points + i];
        if (k2 >= 0) {
            const double* BDk = d.BD + (size_t)k * 18;
            const double* Bj = d.Hpl_val + (size_t)k2 * 18;
#pragma unroll
            for (int a = 0; a < 6; a++)
#pragma unroll
                for (int c = 0; c < 6; c++) v[a * 6 + c] = BDk[a * 3] * Bj[c * 3] + BDk[a * 3 + 1] * Bj[c * 3 + 1] + BDk[a * 3 + 2] * Bj[c * 3 + 2];
        }
        if (j1 == j2)
#pragma unroll
            for (int a = 0; a < 6; a++) v[36 + a] = d.cblk[(size_t)k * 6 + a];
    }
#pragma unroll
    for (int q = 0; q < 42; q++) s_red[q * POSE_CHUNK + tid] = v[q];
    __syncthreads();
    for (int o = POSE_CHUNK / 2; o > 0; o >>= 1) {
        if (tid < o)
#pragma unroll
            for (int q = 0; q < 42; q++) s_red[q * POSE_CHUNK + tid] += s_red[q * POSE_CHUNK + tid + o];
        __syncthreads();
    }
    if (tid < 36) d.pair_part[((size_t)blockIdx.y * d.n_chunks_max + chunk) * 36 + tid] = s_red[tid * POSE_CHUNK];
    else if (tid < 42 && j1 == j2) d.coef_part[((size_t)j1 * d.n_chunks_max + chunk) * 6 + (tid - 36)] = s_red[tid * POSE_CHUNK];
}

__global__ __launch_bounds__(256)
void ba_schur_final_k(BaDev d)
{
    const int np6 = 6 * d.n_free;
    const int npairs = d.n_free * (d.n_free + 1) / 2;
    for (int t = blockIdx.x * 256 + threadIdx.x; t < npairs * 36; t += gridDim.x * 256) {
        const int p = t / 36, e = t % 36, a = e / 6, c = e % 6;
        int j1 = 0, rem = p;
        while (rem >= d.n_free - j1) { rem -= d.n_free - j1; j1++; }
        const int j2 = j1 + rem;
        const int nch = (d.rowptr[j1 + 1] - d.rowptr[j1] + POSE_CHUNK - 1) / POSE_CHUNK;
        double sum = 0.0;
        for (int ch = 0; ch < nch; ch++) sum += d.pair_part[((size_t)p * d.n_chunks_max + ch) * 36 + e];
        double val = -sum;
        if (j1 == j2) val += d.Hpp[(size_t)j1 * 36 + e] + (a == c ? d.lambda : 0.0);
        d.S[(size_t)(6 * j1 + a) * np6 + 6 * j2 + c] = val;
        if (j1 != j2) d.S[(size_t)(6 * j2 + c) * np6 + 6 * j1 + a] = val;
    }
    for (int t = blockIdx.x * 256 + threadIdx.x; t < np6; t += gridDim.x * 256) {
        const int j = t / 6, a = t % 6;
        const int nch = (d.rowptr[j + 1] - d.rowptr[j] + POSE_CHUNK - 1) / POSE_CHUNK;
        double sum = 0.0;
        for (int ch = 0; ch < nch; ch++) sum += d.coef_part[((size_t)j * d.n_chunks_max + ch) * 6 + a];
        d.bs[t] = d.b[t] - sum;
    }
}

// x_l = Dinv (b_l - Hpl^T x_p); optional X += x_l; per-block partial of x_l . (lambda x_l + b_l) for computeScale
__global__ __launch_bounds__(256)
void ba_backsub_k(BaDev d, int apply, double* points_rw)
{
    const int i = blockIdx.x * 256 + threadIdx.x, tid = threadIdx.x;
    double sc = 0.0;
    if (i < d.n_points) {
        const double* bl = d.b + 6 * d.n_free + 3 * (size_t)i;
        double cl[3] = { bl[0], bl[1], bl[2] };
        for (int q = d.pt_ptr[i]; q < d.pt_ptr[i + 1]; q++) {
            const int e = d.pt_edge[q];
            const int sl = d.slot[e];
            if (sl < 0) continue;
            const double* B = d.Hpl_val + (size_t)sl * 18;
            const double* xp = d.x + 6 * d.fidx[d.edge_pose[e]];
#pragma unroll
            for (int c = 0; c < 3; c++)
#pragma unroll
                for (int a = 0; a < 6; a++) cl[c] -= B[a * 3 + c] * xp[a];
        }
        const double* Di = d.Dinv + (size_t)i * 9;
        double xl[3];
#pragma unroll
        for (int a = 0; a < 3; a++) xl[a] = Di[a * 3] * cl[0] + Di[a * 3 + 1] * cl[1] + Di[a * 3 + 2] * cl[2];
#pragma unroll
        for (int a = 0; a < 3; a++) {
            d.x[6 * d.n_free + 3 * (size_t)i + a] = xl[a];
            sc += xl[a] * (d.lambda * xl[a] + bl[a]);
            if (apply) points_rw[3 * (size_t)i + a] += xl[a];
        }
    }
    __shared__ double s_r[256];
    s_r[tid] = sc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) s_r[tid] += s_r[tid + o]; __syncthreads(); }
    if (tid == 0) d.scale_part[blockIdx.x] = s_r[0];
}

__global__ void ba_scale_final_k(BaDev d, int nblocks)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double s = 0.0;
        for (int q = 0; q < nblocks; q++) s += d.scale_part[q];
        d.scale_l[0] = s;
    }
}

} // namespace

struct mslam_ba_plan {
    mslam_ctx* ctx = nullptr;
    void* arena = nullptr;
    BaDev dev{};
    double* d_poses = nullptr; double* d_points = nullptr;
    int32_t* d_rowptr = nullptr; int32_t* d_col = nullptr;
    std::vector<int32_t> h_rowptr, h_col;
    bool has_schur = false;
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
    std::vector<int32_t> fidx(NP), pt_ptr(NL + 1, 0), pt_edge(E), slot(E, -1);
    int nf = 0;
    for (int j = 0; j < NP; j++) fidx[j] = p->pose_fixed[j] ? -1 : nf++;
    for (int e = 0; e < E; e++) pt_ptr[p->edge_point[e] + 1]++;
    for (int i = 0; i < NL; i++) pt_ptr[i + 1] += pt_ptr[i];
    {
        std::vector<int32_t> fill(NL, 0);
        for (int e = 0; e < E; e++) { int i = p->edge_point[e]; pt_edge[pt_ptr[i] + fill[i]++] = e; } // ascending edge id
    }
    std::vector<int32_t> rowptr(nf + 1, 0);
    for (int e = 0; e < E; e++) { int f = fidx[p->edge_pose[e]]; if (f >= 0) rowptr[f + 1]++; }
    for (int f = 0; f < nf; f++) rowptr[f + 1] += rowptr[f];
    const int nnzb = rowptr[nf];
    std::vector<std::pair<int32_t, int32_t>> pe(nnzb);
    {
        std::vector<int32_t> fill(nf, 0);
        for (int e = 0; e < E; e++) { int f = fidx[p->edge_pose[e]]; if (f >= 0) pe[rowptr[f] + fill[f]++] = { p->edge_point[e], e }; }
    }
    int max_row = 0;
    for (int f = 0; f < nf; f++) {
        std::sort(pe.begin() + rowptr[f], pe.begin() + rowptr[f + 1]);
        max_row = std::max(max_row, rowptr[f + 1] - rowptr[f]);
    }
    std::vector<int32_t> col(nnzb), blk_edge(nnzb);
    for (int k = 0; k < nnzb; k++) { col[k] = pe[k].first; blk_edge[k] = pe[k].second; slot[pe[k].second] = k; }
    const int n_chunks_max = std::max(1, (max_row + POSE_CHUNK - 1) / POSE_CHUNK);

    mslam_ba_plan* pl = new mslam_ba_plan();
    pl->ctx = ctx;
    size_t bytes = 0;
    auto reserve = [&](size_t n) { bytes = align_up(bytes, 256); size_t o = bytes; bytes += n; return o; };
    const size_t o_poses = reserve(sizeof(double) * 7 * NP), o_points = reserve(sizeof(double) * 3 * NL),
                 o_ep = reserve(sizeof(int32_t) * E), o_el = reserve(sizeof(int32_t) * E), o_obs = reserve(sizeof(double) * 2 * E),
                 o_info = reserve(sizeof(double) * E), o_intr = reserve(sizeof(double) * 4 * NP), o_fidx = reserve(sizeof(int32_t) * NP),
                 o_ptp = reserve(sizeof(int32_t) * (NL + 1)), o_pte = reserve(sizeof(int32_t) * E),
                 o_row = reserve(sizeof(int32_t) * (nf + 1)), o_col = reserve(sizeof(int32_t) * std::max(nnzb, 1)),
                 o_be = reserve(sizeof(int32_t) * std::max(nnzb, 1)), o_slot = reserve(sizeof(int32_t) * E),
                 o_err = reserve(sizeof(double) * 2 * E), o_chi = reserve(sizeof(double) * E), o_rho = reserve(sizeof(double) * E),
                 o_hpp = reserve(sizeof(double) * 36 * std::max(nf, 1)), o_hll = reserve(sizeof(double) * 9 * NL),
                 o_hpl = reserve(sizeof(double) * 18 * std::max(nnzb, 1)), o_b = reserve(sizeof(double) * (6 * nf + 3 * NL)),
                 o_ptr = reserve(sizeof(double) * NL), o_chp = reserve(sizeof(double) * ((size_t)NL / 4096 + 1)), o_pp = reserve(sizeof(double) * 27 * (size_t)std::max(nf, 1) * n_chunks_max),
                 o_rc = reserve(sizeof(double));
    // Schur scratch (dense pose-row x point lookup: fine for local-BA sizes; skipped when it would be huge)
    const bool want_schur = (size_t)nf * NL <= ((size_t)1 << 26) && nf > 0 && nnzb > 0;
    const int npairs = nf * (nf + 1) / 2;
    size_t o_blk = 0, o_dinv = 0, o_db = 0, o_bd = 0, o_cb = 0, o_pp2 = 0, o_cp = 0, o_S = 0, o_bs = 0, o_x = 0, o_bk = 0, o_sp = 0, o_sl = 0;
    if (want_schur) {
        o_blk = reserve(sizeof(int32_t) * (size_t)nf * NL); o_dinv = reserve(sizeof(double) * 9 * NL); o_db = reserve(sizeof(double) * 3 * NL);
        o_bd = reserve(sizeof(double) * 18 * nnzb); o_cb = reserve(sizeof(double) * 6 * nnzb);
        o_pp2 = reserve(sizeof(double) * 36 * (size_t)npairs * n_chunks_max); o_cp = reserve(sizeof(double) * 6 * (size_t)nf * n_chunks_max);
        o_S = reserve(sizeof(double) * 36 * (size_t)nf * nf); o_bs = reserve(sizeof(double) * 6 * nf);
        o_x = reserve(sizeof(double) * (6 * nf + 3 * NL)); o_bk = reserve(sizeof(double) * 3 * NL);
        o_sp = reserve(sizeof(double) * ((NL + 255) / 256)); o_sl = reserve(sizeof(double));
    }
    hipError_t he = hipMalloc(&pl->arena, bytes);
    if (he != hipSuccess) { ctx->last_hip_error = std::string("hipMalloc(ba plan): ") + hipGetErrorString(he); delete pl; return MSLAM_ERR_HIP; }
    char* base = (char*)pl->arena;
    hipStream_t st = ctx->stream;
    auto up = [&](size_t off, const void* src, size_t n) { return n ? hipMemcpyAsync(base + off, src, n, hipMemcpyHostToDevice, st) : hipSuccess; };
    hipError_t errs[] = {
        up(o_poses, p->poses, sizeof(double) * 7 * NP), up(o_points, p->points, sizeof(double) * 3 * NL),
        up(o_ep, p->edge_pose, sizeof(int32_t) * E), up(o_el, p->edge_point, sizeof(int32_t) * E),
        up(o_obs, p->obs, sizeof(double) * 2 * E), up(o_info, p->info, sizeof(double) * E), up(o_intr, p->intr, sizeof(double) * 4 * NP),
        up(o_fidx, fidx.data(), sizeof(int32_t) * NP), up(o_ptp, pt_ptr.data(), sizeof(int32_t) * (NL + 1)),
        up(o_pte, pt_edge.data(), sizeof(int32_t) * E), up(o_row, rowptr.data(), sizeof(int32_t) * (nf + 1)),
        up(o_col, col.data(), sizeof(int32_t) * nnzb), up(o_be, blk_edge.data(), sizeof(int32_t) * nnzb),
        up(o_slot, slot.data(), sizeof(int32_t) * E), hipStreamSynchronize(st) };
    for (hipError_t e : errs)
        if (e != hipSuccess) { ctx->last_hip_error = std::string("ba plan upload: ") + hipGetErrorString(e); (void)hipFree(pl->arena); delete pl; return MSLAM_ERR_HIP; }
    BaDev& d = pl->dev;
    d.n_poses = NP; d.n_points = NL; d.n_edges = E; d.n_free = nf; d.nnzb = nnzb; d.n_chunks_max = n_chunks_max;
    d.huber_delta = p->huber_delta; d.dsqr = (float)(p->huber_delta * p->huber_delta); // RobustKernelHuber: float dsqr
    d.poses = (double*)(base + o_poses); d.points = (double*)(base + o_points);
    d.edge_pose = (int32_t*)(base + o_ep); d.edge_point = (int32_t*)(base + o_el); d.obs = (double*)(base + o_obs);
    d.info = (double*)(base + o_info); d.intr = (double*)(base + o_intr); d.fidx = (int32_t*)(base + o_fidx);
    d.pt_ptr = (int32_t*)(base + o_ptp); d.pt_edge = (int32_t*)(base + o_pte);
    d.rowptr = (int32_t*)(base + o_row); d.blk_edge = (int32_t*)(base + o_be); d.slot = (int32_t*)(base + o_slot);
    d.err = (double*)(base + o_err); d.chi2 = (double*)(base + o_chi); d.rho1 = (double*)(base + o_rho);
    d.Hpp = (double*)(base + o_hpp); d.Hll = (double*)(base + o_hll); d.Hpl_val = (double*)(base + o_hpl); d.b = (double*)(base + o_b);
    d.pt_rho = (double*)(base + o_ptr); d.chi_part = (double*)(base + o_chp); d.pose_part = (double*)(base + o_pp); d.robust_chi2 = (double*)(base + o_rc);
    pl->d_poses = (double*)(base + o_poses); pl->d_points = (double*)(base + o_points);
    pl->d_rowptr = (int32_t*)(base + o_row); pl->d_col = (int32_t*)(base + o_col);
    pl->h_rowptr = rowptr; pl->h_col = col;
    d.col = (int32_t*)(base + o_col);
    if (want_schur) {
        std::vector<int32_t> blk_of((size_t)nf * NL, -1);
        for (int f = 0; f < nf; f++)
            for (int k = rowptr[f]; k < rowptr[f + 1]; k++) blk_of[(size_t)f * NL + col[k]] = k;
        hipError_t e2 = hipMemcpy(base + o_blk, blk_of.data(), sizeof(int32_t) * blk_of.size(), hipMemcpyHostToDevice);
        if (e2 != hipSuccess) { ctx->last_hip_error = std::string("ba plan upload: ") + hipGetErrorString(e2); (void)hipFree(pl->arena); delete pl; return MSLAM_ERR_HIP; }
        d.blk_of = (int32_t*)(base + o_blk); d.Dinv = (double*)(base + o_dinv); d.db = (double*)(base + o_db); d.BD = (double*)(base + o_bd);
        d.cblk = (double*)(base + o_cb); d.pair_part = (double*)(base + o_pp2); d.coef_part = (double*)(base + o_cp);
        d.S = (double*)(base + o_S); d.bs = (double*)(base + o_bs); d.x = (double*)(base + o_x); d.points_bk = (double*)(base + o_bk);
        d.scale_part = (double*)(base + o_sp); d.scale_l = (double*)(base + o_sl);
        pl->has_schur = true;
    }
    *out = pl;
    return MSLAM_OK;
}

void mslam_ba_plan_destroy(mslam_ba_plan* pl)
{
    if (!pl) return;
    (void)hipSetDevice(pl->ctx->device);
    (void)hipStreamSynchronize(pl->ctx->stream);
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
    if (info) {
        HIP_TRY(ctx, hipMemcpyAsync(const_cast<double*>(pl->dev.info), info, sizeof(double) * pl->dev.n_edges, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    pl->dev.huber_delta = huber_delta;
    pl->dev.dsqr = (float)(huber_delta * huber_delta);
    return MSLAM_OK;
}

int mslam_ba_plan_build_dev(mslam_ba_plan* pl)
{
    if (!pl) return MSLAM_ERR_INVALID;
    mslam_ctx* ctx = pl->ctx;
    MSLAM_TRY(ctx_set_device(ctx));
    const BaDev& d = pl->dev;
    hipStream_t st = ctx->stream;
    const int groups_per_block = BA_THREADS / GROUP;
    LAUNCH(ctx, K_BA_POINTS, ba_points_k, dim3((d.n_points + groups_per_block - 1) / groups_per_block), dim3(BA_THREADS), 0, d);
    if (d.n_free > 0 && d.nnzb > 0)
        LAUNCH(ctx, K_BA_POSES, ba_poses_k, dim3(d.n_chunks_max, d.n_free), dim3(POSE_CHUNK), 0, d);
    if (d.n_points <= 2 * CHI_SLICE && d.n_free * 27 <= 4096) {
        LAUNCH(ctx, K_BA_FINAL, ba_final_k, dim3(1), dim3(256), 0, d, 0);
    } else {
        const int n_pose_wg = (d.n_free * 27 + 255) / 256, n_slices = (d.n_points + CHI_SLICE - 1) / CHI_SLICE;
        LAUNCH(ctx, K_BA_FINAL, ba_final_k, dim3(n_pose_wg + n_slices), dim3(256), 0, d, n_pose_wg);
        LAUNCH(ctx, K_BA_FINAL, ba_final2_k, dim3(1), dim3(256), 0, d, n_slices);
    }
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
    HIP_TRY(ctx, dn(s->err, d.err, sizeof(double) * 2 * d.n_edges));
    HIP_TRY(ctx, dn(s->chi2, d.chi2, sizeof(double) * d.n_edges));
    HIP_TRY(ctx, dn(s->rho1, d.rho1, sizeof(double) * d.n_edges));
    HIP_TRY(ctx, dn(s->Hpp, d.Hpp, sizeof(double) * 36 * d.n_free));
    HIP_TRY(ctx, dn(s->Hll, d.Hll, sizeof(double) * 9 * d.n_points));
    HIP_TRY(ctx, dn(s->Hpl_val, d.Hpl_val, sizeof(double) * 18 * d.nnzb));
    HIP_TRY(ctx, dn(s->b, d.b, sizeof(double) * (6 * d.n_free + 3 * d.n_points)));
    HIP_TRY(ctx, dn(&s->robust_chi2, d.robust_chi2, sizeof(double)));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    if (s->Hpl_rowptr) std::memcpy(s->Hpl_rowptr, pl->h_rowptr.data(), sizeof(int32_t) * (d.n_free + 1));
    if (s->Hpl_col && d.nnzb) std::memcpy(s->Hpl_col, pl->h_col.data(), sizeof(int32_t) * d.nnzb);
    s->n_free_poses = d.n_free; s->nnzb = d.nnzb;
    return MSLAM_OK;
}

int mslam_ba_plan_schur_dev(mslam_ba_plan* pl, double lambda)
{
    if (!pl) return MSLAM_ERR_INVALID;
    if (!pl->has_schur) return MSLAM_ERR_CAPACITY;
    mslam_ctx* ctx = pl->ctx;
    MSLAM_TRY(ctx_set_device(ctx));
    BaDev d = pl->dev;
    d.lambda = lambda;
    pl->dev.lambda = lambda;
    const int npairs = d.n_free * (d.n_free + 1) / 2;
    LAUNCH(ctx, K_BA_SCHUR, ba_schur_points_k, dim3((d.n_points + 255) / 256), dim3(256), 0, d);
    LAUNCH(ctx, K_BA_SCHUR, ba_schur_blocks_k, dim3((d.nnzb + 255) / 256), dim3(256), 0, d);
    LAUNCH(ctx, K_BA_SCHUR, ba_schur_pairs_k, dim3(d.n_chunks_max, npairs), dim3(POSE_CHUNK), 0, d);
    LAUNCH(ctx, K_BA_SCHUR, ba_schur_final_k, dim3(std::min(64, (npairs * 36 + 255) / 256)), dim3(256), 0, d);
    HIP_TRY(ctx, hipGetLastError());
    return MSLAM_OK;
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
    const int nb = (d.n_points + 255) / 256;
    LAUNCH(ctx, K_BA_SCHUR, ba_backsub_k, dim3(nb), dim3(256), 0, d, apply_to_points, pl->d_points);
    LAUNCH(ctx, K_BA_SCHUR, ba_scale_final_k, dim3(1), dim3(64), 0, d, nb);
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
