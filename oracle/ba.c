/*
 * ba.c -- oracle (test infrastructure, see oracle.h): bundle-adjustment linearisation and the
 * Levenberg-Marquardt / Schur-complement loop around it.
 *
 * Follows, file by file (all under the reference root):
 *   src/Util.cpp:34-220            the graph Util::BundleAdjustment builds (pose vertices fixed iff id 0,
 *                                  marginalised point vertices, one EdgeSE3ProjectXYZ per observation,
 *                                  information = I * (1.0f/scale), Huber delta = (float)sqrt(5.99))
 *   src/Converter.cpp:8-31         Rt <-> SE3Quat
 *   ThirdParty/g2o/g2o/types/se3quat.h:58-60,104-117,217-257,280-285   SE3Quat ctor / product / map / exp / normalise
 *   ThirdParty/g2o/g2o/types/types_six_dof_expmap.h:73-76,90-95        oplus, computeError
 *   ThirdParty/g2o/g2o/types/types_six_dof_expmap.cpp:103-147          linearizeOplus, cam_project
 *   ThirdParty/g2o/g2o/types/types_sba.h:52-56                         point oplus
 *   ThirdParty/g2o/g2o/core/base_edge.h:58-61,96-102                   chi2, robustInformation (rho'' term dropped)
 *   ThirdParty/g2o/g2o/core/robust_kernel_impl.cpp:65-91               Huber (dsqr stored as float)
 *   ThirdParty/g2o/g2o/core/base_binary_edge.hpp:55-120                constructQuadraticForm
 *   ThirdParty/g2o/g2o/core/block_solver.hpp:354-486,502-604           Schur solve, buildSystem, setLambda/restoreDiagonal
 *   ThirdParty/g2o/g2o/core/optimization_algorithm_levenberg.cpp:61-189  LM control
 *   ThirdParty/g2o/g2o/core/sparse_optimizer.cpp:100-114,166-190,354-419 robust chi2, index mapping, optimize loop
 * g2o's arithmetic lives in Eigen 3.2.9 templates, which are NOT vendored: the Eigen pieces it relies on
 * (Quaterniond(Matrix3d), quaternion * vector, toRotationMatrix, 3x3 inverse, SimplicialLDLT) are restated
 * from Eigen's published algorithms.  The reduced pose system is solved by a dense LDL^T instead of Eigen's
 * sparse SimplicialLDLT with AMD ordering (same factorisation up to a symmetric permutation).
 * PARITY UNPINNED: the reference's only BA check (UnitTest/g2o/UnitTestg2o.cpp:31-82) needs CSV fixtures
 * from a git-ignored directory.
 */
#include "oracle.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* Eigen: Quaternion::toRotationMatrix */
static void quat_to_R(const double q[4] /* x y z w */, double R[9])
{
    double x = q[0], y = q[1], z = q[2], w = q[3];
    double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    double twx = tx * w, twy = ty * w, twz = tz * w;
    double txx = tx * x, txy = ty * x, txz = tz * x;
    double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}

/* Eigen: Quaternion::_transformVector  (v + w*uv + vec x uv with uv = 2 vec x v) */
static void quat_rotate(const double q[4], const double v[3], double out[3])
{
    double ux = q[1] * v[2] - q[2] * v[1], uy = q[2] * v[0] - q[0] * v[2], uz = q[0] * v[1] - q[1] * v[0];
    ux += ux; uy += uy; uz += uz;
    out[0] = v[0] + q[3] * ux + (q[1] * uz - q[2] * uy);
    out[1] = v[1] + q[3] * uy + (q[2] * ux - q[0] * uz);
    out[2] = v[2] + q[3] * uz + (q[0] * uy - q[1] * ux);
}

/* Eigen: Quaternion(Matrix3) */
static void R_to_quat(const double m[9], double q[4])
{
    double t = m[0] + m[4] + m[8];
    if (t > 0) {
        t = sqrt(t + 1.0);
        q[3] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (m[7] - m[5]) * t; q[1] = (m[2] - m[6]) * t; q[2] = (m[3] - m[1]) * t;
    } else {
        int i = 0;
        if (m[4] > m[0]) i = 1;
        if (m[8] > m[i * 3 + i]) i = 2;
        int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrt(m[i * 3 + i] - m[j * 3 + j] - m[k * 3 + k] + 1.0);
        q[i] = 0.5 * t;
        t = 0.5 / t;
        q[3] = (m[k * 3 + j] - m[j * 3 + k]) * t;
        q[j] = (m[j * 3 + i] + m[i * 3 + j]) * t;
        q[k] = (m[k * 3 + i] + m[i * 3 + k]) * t;
    }
}

/* SE3Quat::normalizeRotation (se3quat.h:280-285) */
static void quat_normalize(double q[4])
{
    if (q[3] < 0) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
    double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}

/* Converter::cvMatToSE3Quat -> SE3Quat(R, t): pose7 = tx,ty,tz,qx,qy,qz,qw */
void orc_Rt_to_se3quat(const double Rt[12], double p[7])
{
    double R[9] = { Rt[0], Rt[1], Rt[2], Rt[4], Rt[5], Rt[6], Rt[8], Rt[9], Rt[10] };
    R_to_quat(R, p + 3);
    quat_normalize(p + 3);
    p[0] = Rt[3]; p[1] = Rt[7]; p[2] = Rt[11];
}

/* Converter::SE3QuatToCvMat: to_homogeneous_matrix, divided by m(3,3) (== 1) */
void orc_se3quat_to_Rt(const double p[7], double Rt[12])
{
    double R[9];
    quat_to_R(p + 3, R);
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) Rt[i * 4 + j] = R[i * 3 + j] / 1.0; Rt[i * 4 + 3] = p[i] / 1.0; }
}

/* VertexSE3Expmap::oplusImpl: T <- SE3Quat::exp(delta) * T, delta = (omega, upsilon) */
void orc_se3_exp_left(const double d[6], double p[7])
{
    double om[3] = { d[0], d[1], d[2] }, up[3] = { d[3], d[4], d[5] };
    double theta = sqrt(om[0] * om[0] + om[1] * om[1] + om[2] * om[2]);
    double Om[9] = { 0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0 };
    double Om2[9], R[9], V[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) { double a = 0; for (int k = 0; k < 3; k++) a += Om[i * 3 + k] * Om[k * 3 + j]; Om2[i * 3 + j] = a; }
    if (theta < 0.00001) { /* "TODO: CHECK WHETHER THIS IS CORRECT!!!" -- restated as written (se3quat.h:237-243) */
        for (int i = 0; i < 9; i++) { double I = (i % 4 == 0) ? 1.0 : 0.0; R[i] = I + Om[i] + Om2[i]; V[i] = R[i]; }
    } else {
        double a = sin(theta) / theta, b = (1 - cos(theta)) / (theta * theta), c = (theta - sin(theta)) / pow(theta, 3);
        for (int i = 0; i < 9; i++) { double I = (i % 4 == 0) ? 1.0 : 0.0; R[i] = I + a * Om[i] + b * Om2[i]; V[i] = I + b * Om[i] + c * Om2[i]; }
    }
    double qe[4], te[3];
    R_to_quat(R, qe);
    quat_normalize(qe);
    for (int i = 0; i < 3; i++) te[i] = V[i * 3] * up[0] + V[i * 3 + 1] * up[1] + V[i * 3 + 2] * up[2];
    /* operator*: t = te + qe * t2, q = qe * q2, normalise */
    double rt[3];
    quat_rotate(qe, p, rt);
    double a[4] = { qe[0], qe[1], qe[2], qe[3] }, b[4] = { p[3], p[4], p[5], p[6] }, q[4];
    q[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
    q[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    q[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
    q[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
    quat_normalize(q);
    p[0] = te[0] + rt[0]; p[1] = te[1] + rt[1]; p[2] = te[2] + rt[2];
    p[3] = q[0]; p[4] = q[1]; p[5] = q[2]; p[6] = q[3];
}

/* computeError + linearizeOplus for one edge: e (2), A = d e/d point (2x3), Bm = d e/d pose (2x6) */
static void edge_eval(const double pose[7], const double X[3], const double obs[2], const double intr[4],
                      double e[2], double A[6], double Bm[12])
{
    double fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
    double pc[3];
    quat_rotate(pose + 3, X, pc);
    double x = pc[0] + pose[0], y = pc[1] + pose[1], z = pc[2] + pose[2];
    e[0] = obs[0] - (x / z * fx + cx);
    e[1] = obs[1] - (y / z * fy + cy);
    double z_2 = z * z;
    double tmp[6] = { fx, 0, -x / z * fx, 0, fy, -y / z * fy };
    double R[9];
    quat_to_R(pose + 3, R);
    double s = -1. / z;
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 3; j++)
            A[i * 3 + j] = (s * tmp[i * 3]) * R[j] + (s * tmp[i * 3 + 1]) * R[3 + j] + (s * tmp[i * 3 + 2]) * R[6 + j];
    Bm[0] = x * y / z_2 * fx;  Bm[1] = -(1 + (x * x / z_2)) * fx; Bm[2] = y / z * fx;
    Bm[3] = -1. / z * fx;      Bm[4] = 0;                         Bm[5] = x / z_2 * fx;
    Bm[6] = (1 + y * y / z_2) * fy; Bm[7] = -x * y / z_2 * fy;    Bm[8] = -x / z * fy;
    Bm[9] = 0;                 Bm[10] = -1. / z * fy;             Bm[11] = y / z_2 * fy;
}

int orc_ba_sizes(const orc_ba_problem* p, int* n_free, int* nnzb)
{
    int nf = 0, nz = 0;
    for (int j = 0; j < p->n_poses; j++) nf += !p->pose_fixed[j];
    for (int e = 0; e < p->n_edges; e++) nz += !p->pose_fixed[p->edge_pose[e]];
    if (n_free) *n_free = nf;
    if (nnzb) *nnzb = nz;
    return 0;
}

typedef struct { int point, edge; } pe_t;
static int cmp_pe(const void* a, const void* b)
{
    const pe_t* x = (const pe_t*)a; const pe_t* y = (const pe_t*)b;
    return x->point != y->point ? (x->point < y->point ? -1 : 1) : (x->edge < y->edge ? -1 : (x->edge > y->edge));
}

/* one pass of computeActiveErrors + buildSystem; edges are visited in index order (g2o: edge id order) */
int orc_ba_build(const orc_ba_problem* p, orc_ba_system* s)
{
    int nf = 0, nnzb = 0;
    orc_ba_sizes(p, &nf, &nnzb);
    s->n_free_poses = nf; s->nnzb = nnzb;
    int* fidx = (int*)malloc(sizeof(int) * (size_t)(p->n_poses + 1));
    for (int j = 0, k = 0; j < p->n_poses; j++) fidx[j] = p->pose_fixed[j] ? -1 : k++;
    /* block-CSR structure of Hpl: rows = free poses, columns = points ascending */
    int* slot = (int*)malloc(sizeof(int) * (size_t)(p->n_edges + 1));
    memset(s->Hpl_rowptr, 0, sizeof(int32_t) * (size_t)(nf + 1));
    for (int e = 0; e < p->n_edges; e++) { int f = fidx[p->edge_pose[e]]; if (f >= 0) s->Hpl_rowptr[f + 1]++; }
    for (int f = 0; f < nf; f++) s->Hpl_rowptr[f + 1] += s->Hpl_rowptr[f];
    pe_t* pe = (pe_t*)malloc(sizeof(pe_t) * (size_t)(nnzb + 1));
    int* fill = (int*)calloc((size_t)(nf + 1), sizeof(int));
    for (int e = 0; e < p->n_edges; e++) {
        int f = fidx[p->edge_pose[e]];
        if (f < 0) continue;
        int at = s->Hpl_rowptr[f] + fill[f]++;
        pe[at].point = p->edge_point[e]; pe[at].edge = e;
    }
    for (int f = 0; f < nf; f++) qsort(pe + s->Hpl_rowptr[f], (size_t)(s->Hpl_rowptr[f + 1] - s->Hpl_rowptr[f]), sizeof(pe_t), cmp_pe);
    for (int e = 0; e < p->n_edges; e++) slot[e] = -1;
    for (int k = 0; k < nnzb; k++) { s->Hpl_col[k] = pe[k].point; slot[pe[k].edge] = k; }
    free(pe); free(fill);

    memset(s->Hpp, 0, sizeof(double) * 36 * (size_t)nf);
    memset(s->Hll, 0, sizeof(double) * 9 * (size_t)p->n_points);
    memset(s->Hpl_val, 0, sizeof(double) * 18 * (size_t)nnzb);
    memset(s->b, 0, sizeof(double) * (size_t)(6 * nf + 3 * p->n_points));
    const double delta = p->huber_delta;
    const float dsqr = (float)(delta * delta); /* RobustKernelHuber::setDelta: float member */
    double rchi = 0;
    for (int e = 0; e < p->n_edges; e++) {
        int jp = p->edge_pose[e], ip = p->edge_point[e];
        double err[2], A[6], Bm[12];
        edge_eval(p->poses + 7 * jp, p->points + 3 * ip, p->obs + 2 * e, p->intr + 4 * jp, err, A, Bm);
        double om = p->info[e];
        double chi2 = err[0] * (om * err[0]) + err[1] * (om * err[1]); /* _error.dot(information()*_error) */
        double rho0 = chi2, rho1 = 1.0;
        if (delta > 0) {
            if (!(chi2 <= dsqr)) {
                double sq = sqrt(chi2);
                rho0 = 2 * sq * delta - dsqr;
                rho1 = delta / sq;
            }
        }
        s->err[2 * e] = err[0]; s->err[2 * e + 1] = err[1];
        s->chi2[e] = chi2; s->rho1[e] = rho1;
        rchi += rho0;
        double wom = rho1 * om;                                  /* robustInformation */
        double omr[2] = { -(om * err[0]) * rho1, -(om * err[1]) * rho1 }; /* omega_r = -omega*e, *= rho[1] */
        int f = fidx[jp];
        /* point (never fixed): b_i += A^T omega_r, Hll += A^T wOmega A */
        double* bl = s->b + 6 * nf + 3 * ip;
        double* Hl = s->Hll + 9 * ip;
        for (int a = 0; a < 3; a++) {
            bl[a] += A[a] * omr[0] + A[3 + a] * omr[1];
            for (int c = 0; c < 3; c++) Hl[a * 3 + c] += A[a] * wom * A[c] + A[3 + a] * wom * A[3 + c];
        }
        if (f >= 0) {
            double* bp = s->b + 6 * f;
            double* Hp = s->Hpp + 36 * f;
            double* Hb = s->Hpl_val + 18 * slot[e];
            for (int a = 0; a < 6; a++) {
                bp[a] += Bm[a] * omr[0] + Bm[6 + a] * omr[1];
                for (int c = 0; c < 6; c++) Hp[a * 6 + c] += Bm[a] * wom * Bm[c] + Bm[6 + a] * wom * Bm[6 + c];
                for (int c = 0; c < 3; c++) Hb[a * 3 + c] += Bm[a] * wom * A[c] + Bm[6 + a] * wom * A[3 + c]; /* B^T wOmega A */
            }
        }
    }
    s->robust_chi2 = rchi;
    free(fidx); free(slot);
    return 0;
}

static double robust_chi2_only(const orc_ba_problem* p)
{
    const double delta = p->huber_delta;
    const float dsqr = (float)(delta * delta);
    double rchi = 0;
    for (int e = 0; e < p->n_edges; e++) {
        int jp = p->edge_pose[e], ip = p->edge_point[e];
        double err[2], A[6], Bm[12];
        edge_eval(p->poses + 7 * jp, p->points + 3 * ip, p->obs + 2 * e, p->intr + 4 * jp, err, A, Bm);
        double om = p->info[e];
        double chi2 = err[0] * (om * err[0]) + err[1] * (om * err[1]);
        double rho0 = chi2;
        if (delta > 0 && !(chi2 <= dsqr)) rho0 = 2 * sqrt(chi2) * delta - dsqr;
        rchi += rho0;
    }
    return rchi;
}

static int inv3(const double* M, double* I)
{
    double det = M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
    double id = 1.0 / det;
    I[0] = (M[4] * M[8] - M[5] * M[7]) * id; I[1] = (M[2] * M[7] - M[1] * M[8]) * id; I[2] = (M[1] * M[5] - M[2] * M[4]) * id;
    I[3] = (M[5] * M[6] - M[3] * M[8]) * id; I[4] = (M[0] * M[8] - M[2] * M[6]) * id; I[5] = (M[2] * M[3] - M[0] * M[5]) * id;
    I[6] = (M[3] * M[7] - M[4] * M[6]) * id; I[7] = (M[1] * M[6] - M[0] * M[7]) * id; I[8] = (M[0] * M[4] - M[1] * M[3]) * id;
    return isfinite(id) ? 1 : 0;
}

/* dense LDL^T solve of the reduced pose system (upper triangle is authoritative) */
static int ldlt_solve(double* S, int n, const double* rhs, double* x)
{
    double* L = (double*)calloc((size_t)n * n, sizeof(double));
    double* D = (double*)calloc((size_t)n, sizeof(double));
    int ok = 1;
    for (int j = 0; j < n && ok; j++) {
        double d = S[j * n + j];
        for (int k = 0; k < j; k++) d -= L[j * n + k] * L[j * n + k] * D[k];
        if (d == 0.0 || !isfinite(d)) { ok = 0; break; }
        D[j] = d;
        L[j * n + j] = 1.0;
        for (int i = j + 1; i < n; i++) {
            double v = S[j * n + i]; /* upper triangle */
            for (int k = 0; k < j; k++) v -= L[i * n + k] * L[j * n + k] * D[k];
            L[i * n + j] = v / d;
        }
    }
    if (ok) {
        for (int i = 0; i < n; i++) { double v = rhs[i]; for (int k = 0; k < i; k++) v -= L[i * n + k] * x[k]; x[i] = v; }
        for (int i = 0; i < n; i++) x[i] /= D[i];
        for (int i = n - 1; i >= 0; i--) { double v = x[i]; for (int k = i + 1; k < n; k++) v -= L[k * n + i] * x[k]; x[i] = v; }
    }
    free(L); free(D);
    return ok;
}

/* BlockSolver::solve with Schur complement (block_solver.hpp:354-486) on a built system with lambda added */
static int schur_solve(const orc_ba_system* s, int nf, int npts, double lambda, double* x, double* S_out, double* bs_out)
{
    const int np6 = 6 * nf;
    double* S = (double*)calloc((size_t)np6 * np6 + 1, sizeof(double));
    double* coeff = (double*)calloc((size_t)np6 + 1, sizeof(double));
    double* Dinv = (double*)malloc(sizeof(double) * 9 * (size_t)(npts + 1));
    for (int f = 0; f < nf; f++)
        for (int a = 0; a < 6; a++)
            for (int c = 0; c < 6; c++) S[(6 * f + a) * np6 + 6 * f + c] = s->Hpp[36 * f + a * 6 + c] + (a == c ? lambda : 0.0);
    /* column view of Hpl: per point, the (pose row, block) pairs in ascending pose order */
    int* cptr = (int*)calloc((size_t)npts + 2, sizeof(int));
    for (int k = 0; k < s->nnzb; k++) cptr[s->Hpl_col[k] + 1]++;
    for (int i = 0; i < npts; i++) cptr[i + 1] += cptr[i];
    int* crow = (int*)malloc(sizeof(int) * (size_t)(s->nnzb + 1));
    int* cblk = (int*)malloc(sizeof(int) * (size_t)(s->nnzb + 1));
    int* cfill = (int*)calloc((size_t)npts + 1, sizeof(int));
    for (int f = 0; f < nf; f++)
        for (int k = s->Hpl_rowptr[f]; k < s->Hpl_rowptr[f + 1]; k++) {
            int i = s->Hpl_col[k];
            int at = cptr[i] + cfill[i]++;
            crow[at] = f; cblk[at] = k;
        }
    int ok = 1;
    for (int i = 0; i < npts; i++) {
        double D[9];
        memcpy(D, s->Hll + 9 * i, sizeof(D));
        D[0] += lambda; D[4] += lambda; D[8] += lambda;
        double* Di = Dinv + 9 * i;
        inv3(D, Di);
        const double* bl = s->b + np6 + 3 * i;
        double db[3];
        for (int a = 0; a < 3; a++) db[a] = Di[a * 3] * bl[0] + Di[a * 3 + 1] * bl[1] + Di[a * 3 + 2] * bl[2];
        for (int o = cptr[i]; o < cptr[i + 1]; o++) {
            int i1 = crow[o];
            const double* Bi = s->Hpl_val + 18 * cblk[o];
            double BD[18];
            for (int a = 0; a < 6; a++)
                for (int c = 0; c < 3; c++) BD[a * 3 + c] = Bi[a * 3] * Di[c] + Bi[a * 3 + 1] * Di[3 + c] + Bi[a * 3 + 2] * Di[6 + c];
            for (int a = 0; a < 6; a++) coeff[6 * i1 + a] += Bi[a * 3] * db[0] + Bi[a * 3 + 1] * db[1] + Bi[a * 3 + 2] * db[2];
            for (int o2 = o; o2 < cptr[i + 1]; o2++) { /* upper block triangle: i2 >= i1 */
                int i2 = crow[o2];
                const double* Bj = s->Hpl_val + 18 * cblk[o2];
                for (int a = 0; a < 6; a++)
                    for (int c = 0; c < 6; c++)
                        S[(6 * i1 + a) * np6 + 6 * i2 + c] -= BD[a * 3] * Bj[c * 3] + BD[a * 3 + 1] * Bj[c * 3 + 1] + BD[a * 3 + 2] * Bj[c * 3 + 2];
            }
        }
    }
    double* bs = (double*)malloc(sizeof(double) * (size_t)(np6 + 1));
    for (int a = 0; a < np6; a++) bs[a] = s->b[a] - coeff[a];
    if (S_out) /* full symmetric matrix (the upper block triangle was accumulated) */
        for (int a = 0; a < np6; a++)
            for (int c = 0; c < np6; c++) S_out[a * np6 + c] = (a / 6 <= c / 6) ? S[a * np6 + c] : S[c * np6 + a];
    if (bs_out) memcpy(bs_out, bs, sizeof(double) * (size_t)np6);
    ok = ldlt_solve(S, np6, bs, x);
    if (ok) { /* x_l = Dinv (b_l - Hpl^T x_p) */
        for (int i = 0; i < npts; i++) {
            double cl[3] = { s->b[np6 + 3 * i], s->b[np6 + 3 * i + 1], s->b[np6 + 3 * i + 2] };
            for (int o = cptr[i]; o < cptr[i + 1]; o++) {
                const double* Bi = s->Hpl_val + 18 * cblk[o];
                const double* xp = x + 6 * crow[o];
                for (int c = 0; c < 3; c++)
                    for (int a = 0; a < 6; a++) cl[c] -= Bi[a * 3 + c] * xp[a];
            }
            const double* Di = Dinv + 9 * i;
            for (int a = 0; a < 3; a++) x[np6 + 3 * i + a] = Di[a * 3] * cl[0] + Di[a * 3 + 1] * cl[1] + Di[a * 3 + 2] * cl[2];
        }
    }
    free(S); free(coeff); free(Dinv); free(cptr); free(crow); free(cblk); free(cfill); free(bs);
    return ok;
}

/* SparseOptimizer::optimize(iterations) with OptimizationAlgorithmLevenberg; poses/points updated in place.
 * chi2_history receives the robust chi2 after every outer iteration (optional). Returns iterations run. */
int orc_ba_optimize_ex(orc_ba_problem* p, double* poses, double* points, int iterations, int points_fixed, double* chi2_history, int* n_hist);

int orc_ba_optimize(orc_ba_problem* p, double* poses, double* points, int iterations, double* chi2_history, int* n_hist)
{
    return orc_ba_optimize_ex(p, poses, points, iterations, 0, chi2_history, n_hist);
}

/* points_fixed != 0: pose-only BA (Util::PoseBundleAdjustment, src/Util.cpp:222-358: EdgeSE3ProjectXYZOnlyPose unary edges,
 * BlockSolver without marginalised vertices): only the pose blocks enter the system, the points never move. */
int orc_ba_optimize_ex(orc_ba_problem* p, double* poses, double* points, int iterations, int points_fixed, double* chi2_history, int* n_hist)
{
    int nf = 0, nnzb = 0;
    p->poses = poses; p->points = points;
    orc_ba_sizes(p, &nf, &nnzb);
    const int np6 = 6 * nf, nx = np6 + 3 * p->n_points;
    orc_ba_system s;
    s.err = (double*)malloc(sizeof(double) * 2 * (size_t)(p->n_edges + 1));
    s.chi2 = (double*)malloc(sizeof(double) * (size_t)(p->n_edges + 1));
    s.rho1 = (double*)malloc(sizeof(double) * (size_t)(p->n_edges + 1));
    s.Hpp = (double*)malloc(sizeof(double) * 36 * (size_t)(nf + 1));
    s.Hll = (double*)malloc(sizeof(double) * 9 * (size_t)(p->n_points + 1));
    s.Hpl_rowptr = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nf + 2));
    s.Hpl_col = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nnzb + 1));
    s.Hpl_val = (double*)malloc(sizeof(double) * 18 * (size_t)(nnzb + 1));
    s.b = (double*)malloc(sizeof(double) * (size_t)(nx + 1));
    double* x = (double*)calloc((size_t)nx + 1, sizeof(double));
    double* bk_pose = (double*)malloc(sizeof(double) * 7 * (size_t)p->n_poses);
    double* bk_pts = (double*)malloc(sizeof(double) * 3 * (size_t)(p->n_points + 1));
    int* fidx = (int*)malloc(sizeof(int) * (size_t)(p->n_poses + 1));
    for (int j = 0, k = 0; j < p->n_poses; j++) fidx[j] = p->pose_fixed[j] ? -1 : k++;
    double lambda = -1, ni = 2;
    int nbad = 0, done = 0, hist = 0;
    const double good_upper = 2. / 3., good_lower = 1. / 3.;
    for (int it = 0; it < iterations; it++) {
        orc_ba_build(p, &s);
        double currentChi = s.robust_chi2, tempChi = currentChi, iniChi = currentChi;
        if (it == 0) { /* computeLambdaInit: tau * max |H_jj| */
            double maxd = 0;
            for (int f = 0; f < nf; f++) for (int a = 0; a < 6; a++) { double v = fabs(s.Hpp[36 * f + 7 * a]); if (v > maxd) maxd = v; }
            if (!points_fixed)
                for (int i = 0; i < p->n_points; i++) for (int a = 0; a < 3; a++) { double v = fabs(s.Hll[9 * i + 4 * a]); if (v > maxd) maxd = v; }
            lambda = 1e-5 * maxd; ni = 2; nbad = 0;
        }
        double rho = 0;
        int qmax = 0;
        do {
            memcpy(bk_pose, poses, sizeof(double) * 7 * (size_t)p->n_poses);   /* push */
            memcpy(bk_pts, points, sizeof(double) * 3 * (size_t)p->n_points);
            int ok2;
            if (points_fixed) {
                double* Hd = (double*)calloc((size_t)np6 * np6 + 1, sizeof(double));
                for (int f = 0; f < nf; f++)
                    for (int a = 0; a < 6; a++)
                        for (int c = 0; c < 6; c++) Hd[(6 * f + a) * np6 + 6 * f + c] = s.Hpp[36 * f + a * 6 + c] + (a == c ? lambda : 0.0);
                memset(x, 0, sizeof(double) * (size_t)nx);
                ok2 = ldlt_solve(Hd, np6, s.b, x);
                free(Hd);
            } else ok2 = schur_solve(&s, nf, p->n_points, lambda, x, NULL, NULL);
            /* update (oplus) */
            for (int j = 0; j < p->n_poses; j++) if (fidx[j] >= 0) orc_se3_exp_left(x + 6 * fidx[j], poses + 7 * j);
            for (int i = 0; i < 3 * p->n_points; i++) points[i] += x[np6 + i];
            tempChi = robust_chi2_only(p);
            if (!ok2) tempChi = DBL_MAX;
            rho = currentChi - tempChi;
            double scale = 0;
            for (int j = 0; j < (points_fixed ? np6 : nx); j++) scale += x[j] * (lambda * x[j] + s.b[j]);
            scale += 1e-3;
            rho /= scale;
            if (rho > 0 && isfinite(tempChi)) {
                double alpha = 1. - pow((2 * rho - 1), 3);
                alpha = alpha < good_upper ? alpha : good_upper;
                double sf = good_lower > alpha ? good_lower : alpha;
                lambda *= sf; ni = 2; currentChi = tempChi;
            } else {
                lambda *= ni; ni *= 2;
                memcpy(poses, bk_pose, sizeof(double) * 7 * (size_t)p->n_poses);   /* pop */
                memcpy(points, bk_pts, sizeof(double) * 3 * (size_t)p->n_points);
            }
            qmax++;
        } while (rho < 0 && qmax < 10);
        done++;
        if (chi2_history) chi2_history[hist] = currentChi;
        hist++;
        if (qmax == 10 || rho == 0) break;
        if ((iniChi - currentChi) * 1e3 < iniChi) nbad++; else nbad = 0;
        if (nbad >= 3) break;
    }
    if (n_hist) *n_hist = hist;
    free(s.err); free(s.chi2); free(s.rho1); free(s.Hpp); free(s.Hll); free(s.Hpl_rowptr); free(s.Hpl_col); free(s.Hpl_val);
    free(s.b); free(x); free(bk_pose); free(bk_pts); free(fidx);
    return done;
}

/* one linearisation + Schur complement with `lambda` on every diagonal: S (6nf x 6nf), b_s (6nf), x (6nf + 3 n_points) */
int orc_ba_schur(const orc_ba_problem* p, double lambda, double* S, double* bs, double* x)
{
    int nf = 0, nnzb = 0;
    orc_ba_sizes(p, &nf, &nnzb);
    orc_ba_system s;
    s.err = (double*)malloc(sizeof(double) * 2 * (size_t)(p->n_edges + 1));
    s.chi2 = (double*)malloc(sizeof(double) * (size_t)(p->n_edges + 1));
    s.rho1 = (double*)malloc(sizeof(double) * (size_t)(p->n_edges + 1));
    s.Hpp = (double*)malloc(sizeof(double) * 36 * (size_t)(nf + 1));
    s.Hll = (double*)malloc(sizeof(double) * 9 * (size_t)(p->n_points + 1));
    s.Hpl_rowptr = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nf + 2));
    s.Hpl_col = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nnzb + 1));
    s.Hpl_val = (double*)malloc(sizeof(double) * 18 * (size_t)(nnzb + 1));
    s.b = (double*)malloc(sizeof(double) * (size_t)(6 * nf + 3 * p->n_points + 1));
    orc_ba_build(p, &s);
    int ok = schur_solve(&s, nf, p->n_points, lambda, x, S, bs);
    free(s.err); free(s.chi2); free(s.rho1); free(s.Hpp); free(s.Hll); free(s.Hpl_rowptr); free(s.Hpl_col); free(s.Hpl_val); free(s.b);
    return ok ? 0 : -6;
}
