/* ba_oracle.c — CPU ORACLE (TEST INFRASTRUCTURE ONLY; see vido_oracle.h).
 *
 * Restates the windowed / global batch optimisation on the STATIC factor set:
 *   Optimizer.cc:43-1228   PartialBatchOptimization (STATIC_ONLY = true, :212): W camera-to-world VertexSE3,
 *                          one VertexPointXYZ per static tracklet (len >= 3), EdgeSE3 odometry (:244-258),
 *                          EdgeSE3PointXYZ (:297-350), EdgeSE3Prior on the first camera when N == W (:226-235)
 *   Optimizer.cc:1235-2178 FullBatchOptimization, static + odometry + prior factors (object-motion factors:
 *                          see DESIGN.md "next")
 * on g2o (vido_slam/3rdparty/g2o/g2o): LM policy core/optimization_algorithm_levenberg.cpp:61-189, gain stop
 * core/sparse_optimizer_terminate_action.cpp:49-85, outer loop core/sparse_optimizer.cpp:354-427, Huber
 * core/robust_kernel_impl.cpp:65-91, VertexSE3::oplusImpl types/vertex_se3.h:105-114 (X <- X * fromVectorMQT(d)),
 * EdgeSE3PointXYZ types/edge_se3_pointxyz.cpp:99-135, EdgeSE3 types/edge_se3.cpp:77-104 with
 * types/isometry3d_gradients.h:85-189 (the Jacobians are restated in closed form, see ba_edge_se3()),
 * EdgeSE3Prior types/edge_se3_prior.cpp:89-102, toVectorMQT/fromCompactQuaternion types/isometry3d_mappings.cpp:78-123.
 *
 * g2o solves the FULL pose+point system with sparse Cholesky under LM (no vertex is marginalised, SURVEY.md
 * fact 5).  Eliminating the points by Schur complement inside every LM trial is algebraically the same step,
 * so this restatement (and the HIP path) runs LM + point-Schur with a dense LDL^T on the reduced camera system.
 */
#include "vido_oracle.h"
#include <math.h>
#include <float.h>
#include <stdlib.h>
#include <string.h>

int vo_ldlt_solve(const double* A, const double* b, double* x, int n);

typedef struct {
    int32_t n_cam, n_pt, n_obs, n_odo, prior_cam, use_huber, max_iters, pad;
    double* cam_T;            /* [n_cam*12] camera-to-world, row-major 3x4 (in/out) */
    double* pt_xyz;           /* [n_pt*3] (in/out) */
    const int32_t* obs_cam; const int32_t* obs_pt; const double* obs_meas;   /* [n_obs], [n_obs], [n_obs*3] */
    const int32_t* odo_i; const int32_t* odo_j; const double* odo_T;         /* [n_odo], [n_odo], [n_odo*12] */
    double prior_T[12];
    double info_obs, info_odo, info_prior, huber_obs, huber_odo, gain_threshold;
} vo_ba_problem;

typedef struct { int32_t iterations, lm_trials; double chi2_initial, chi2_final, lambda_final; } vo_ba_result;

/* ---- isometry helpers (3x4 row-major: R | t) ---- */
static void iso_inv_mul(const double* A, const double* B, double* C)     /* C = A^-1 * B */
{
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) C[r * 4 + c] = A[0 * 4 + r] * B[0 * 4 + c] + A[1 * 4 + r] * B[1 * 4 + c] + A[2 * 4 + r] * B[2 * 4 + c];
        C[r * 4 + 3] = A[0 * 4 + r] * (B[3] - A[3]) + A[1 * 4 + r] * (B[7] - A[7]) + A[2 * 4 + r] * (B[11] - A[11]);
    }
}
/* Eigen Quaternion(R) + g2o normalize(): unit quaternion with w >= 0; q = (x,y,z,w) */
static void rot_to_quat(const double* M /*3x4*/, double* q)
{
    const double m00 = M[0], m01 = M[1], m02 = M[2], m10 = M[4], m11 = M[5], m12 = M[6], m20 = M[8], m21 = M[9], m22 = M[10];
    double t = m00 + m11 + m22;
    if (t > 0) {
        t = sqrt(t + 1.0); q[3] = 0.5 * t; t = 0.5 / t;
        q[0] = (m21 - m12) * t; q[1] = (m02 - m20) * t; q[2] = (m10 - m01) * t;
    } else {
        int i = 0; if (m11 > m00) i = 1; if (m22 > (i == 0 ? m00 : m11)) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        const double mm[3][3] = {{m00, m01, m02}, {m10, m11, m12}, {m20, m21, m22}};
        t = sqrt(mm[i][i] - mm[j][j] - mm[k][k] + 1.0);
        q[i] = 0.5 * t; t = 0.5 / t;
        q[3] = (mm[k][j] - mm[j][k]) * t; q[j] = (mm[j][i] + mm[i][j]) * t; q[k] = (mm[k][i] + mm[i][k]) * t;
    }
    const double nrm = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int a = 0; a < 4; a++) q[a] /= nrm;
    if (q[3] < 0) for (int a = 0; a < 4; a++) q[a] = -q[a];
}
/* VertexSE3::oplusImpl: X <- X * fromVectorMQT(d), d = (t, qx,qy,qz) */
static void iso_oplus(double* X, const double* d)
{
    double w = 1 - (d[3] * d[3] + d[4] * d[4] + d[5] * d[5]);
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (!(w < 0)) {
        w = sqrt(w);
        const double x = d[3], y = d[4], z = d[5];
        R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w); R[2] = 2 * (x * z + y * w);
        R[3] = 2 * (x * y + z * w); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
        R[6] = 2 * (x * z - y * w); R[7] = 2 * (y * z + x * w); R[8] = 1 - 2 * (x * x + y * y);
    }
    double N[12];
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) N[r * 4 + c] = X[r * 4] * R[c] + X[r * 4 + 1] * R[3 + c] + X[r * 4 + 2] * R[6 + c];
        N[r * 4 + 3] = X[r * 4] * d[0] + X[r * 4 + 1] * d[1] + X[r * 4 + 2] * d[2] + X[r * 4 + 3];
    }
    memcpy(X, N, sizeof N);
}

static void huber_w(double e2, double delta, int use, double* rho0, double* rho1)
{
    if (!use || e2 <= delta * delta) { *rho0 = e2; *rho1 = 1.0; return; }
    const double s = sqrt(e2); *rho0 = 2 * s * delta - delta * delta; *rho1 = delta / s;
}

/* EdgeSE3PointXYZ: e = R^T (p - t) - m;  de/ddt = -I, de/dv = 2[Zc]x, de/dp = R^T  (edge_se3_pointxyz.cpp:99-135) */
static void ba_edge_obs(const double* X, const double* p, const double* m, double* e, double* Jc /*3x6*/, double* Jp /*3x3*/)
{
    double Zc[3];
    for (int r = 0; r < 3; r++) Zc[r] = X[0 * 4 + r] * (p[0] - X[3]) + X[1 * 4 + r] * (p[1] - X[7]) + X[2 * 4 + r] * (p[2] - X[11]);
    for (int r = 0; r < 3; r++) e[r] = Zc[r] - m[r];
    if (Jc) {
        memset(Jc, 0, sizeof(double) * 18);
        Jc[0] = Jc[7] = Jc[14] = -1;
        Jc[0 * 6 + 4] = -2 * Zc[2]; Jc[0 * 6 + 5] = 2 * Zc[1];
        Jc[1 * 6 + 3] = 2 * Zc[2]; Jc[1 * 6 + 5] = -2 * Zc[0];
        Jc[2 * 6 + 3] = -2 * Zc[1]; Jc[2 * 6 + 4] = 2 * Zc[0];
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) Jp[r * 3 + c] = X[c * 4 + r];
    }
}

/* EdgeSE3 (and EdgeSE3Prior with Xi = identity): E = Z^-1 Xi^-1 Xj, e = (t_E, q_E.xyz).
 * Exact derivatives w.r.t. the right-multiplicative (dt, v) increments (v = quaternion vector part, R(v) ~ I+2[v]x):
 *   Jj = [ R_E  0 ; 0  Q ],  Ji = [ -R_A  2 R_A [t_B]x ; 0  -Q R_B^T ],  Q = w_E I + [q_E.xyz]x, A = Z^-1, B = Xi^-1 Xj.
 * g2o obtains the same matrices through dq/dR (isometry3d_gradients.h:85-189, dquat2mat.cpp). */
static void ba_edge_se3(const double* Z, const double* Xi, const double* Xj, double* e, double* Ji /*6x6*/, double* Jj /*6x6*/)
{
    double B[12], E[12], q[4];
    if (Xi) iso_inv_mul(Xi, Xj, B); else memcpy(B, Xj, sizeof B);
    iso_inv_mul(Z, B, E);
    rot_to_quat(E, q);
    e[0] = E[3]; e[1] = E[7]; e[2] = E[11]; e[3] = q[0]; e[4] = q[1]; e[5] = q[2];
    if (!Jj) return;
    const double Q[9] = {q[3], -q[2], q[1], q[2], q[3], -q[0], -q[1], q[0], q[3]};
    memset(Jj, 0, sizeof(double) * 36);
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { Jj[r * 6 + c] = E[r * 4 + c]; Jj[(3 + r) * 6 + 3 + c] = Q[r * 3 + c]; }
    if (Ji) {
        memset(Ji, 0, sizeof(double) * 36);
        const double tb[3] = {B[3], B[7], B[11]};
        const double S[9] = {0, -2 * tb[2], 2 * tb[1], 2 * tb[2], 0, -2 * tb[0], -2 * tb[1], 2 * tb[0], 0};     /* 2[t_B]x */
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) {
            const double ra = Z[c * 4 + r];                                                                   /* R_A = R_Z^T */
            Ji[r * 6 + c] = -ra;
            double s = 0, qq = 0;
            for (int k = 0; k < 3; k++) { s += Z[k * 4 + r] * S[k * 3 + c]; qq += Q[r * 3 + k] * B[c * 4 + k]; }   /* R_A S ; Q R_B^T */
            Ji[r * 6 + 3 + c] = s; Ji[(3 + r) * 6 + 3 + c] = -qq;
        }
    }
}

/* chi2 of the whole graph at (cam, pts) */
static double ba_chi2(const vo_ba_problem* p, const double* cam, const double* pts)
{
    double chi = 0, e[6], r0, r1;
    for (int k = 0; k < p->n_obs; k++) {
        ba_edge_obs(cam + 12 * p->obs_cam[k], pts + 3 * p->obs_pt[k], p->obs_meas + 3 * k, e, NULL, NULL);
        huber_w(p->info_obs * (e[0] * e[0] + e[1] * e[1] + e[2] * e[2]), p->huber_obs, p->use_huber, &r0, &r1); chi += r0;
    }
    for (int k = 0; k < p->n_odo; k++) {
        ba_edge_se3(p->odo_T + 12 * k, cam + 12 * p->odo_i[k], cam + 12 * p->odo_j[k], e, NULL, NULL);
        double s = 0; for (int a = 0; a < 6; a++) s += e[a] * e[a];
        huber_w(p->info_odo * s, p->huber_odo, p->use_huber, &r0, &r1); chi += r0;
    }
    if (p->prior_cam >= 0) {
        ba_edge_se3(p->prior_T, NULL, cam + 12 * p->prior_cam, e, NULL, NULL);
        double s = 0; for (int a = 0; a < 6; a++) s += e[a] * e[a];
        chi += p->info_prior * s;
    }
    return chi;
}

/* Linearisation at (cam, pts).  Camera-camera part Hcc (dense n6 x n6) + bc, per-point Hpp (3x3), bp (3), per-obs W (6x3).
 * pt_lo/pt_hi restrict the landmark factors to points in [pt_lo, pt_hi) (landmark shard); with_cam_factors adds the
 * odometry/prior factors (owned by exactly one shard). */
void vo_ba_linearize(const vo_ba_problem* p, int pt_lo, int pt_hi, int with_cam_factors,
                     double* Hcc, double* bc, double* Hpp, double* bp, double* W, double* chi2_out)
{
    const int n6 = 6 * p->n_cam;
    memset(Hcc, 0, sizeof(double) * (size_t)n6 * n6); memset(bc, 0, sizeof(double) * n6);
    memset(Hpp, 0, sizeof(double) * 9 * (size_t)p->n_pt); memset(bp, 0, sizeof(double) * 3 * (size_t)p->n_pt);
    double chi = 0;
    for (int k = 0; k < p->n_obs; k++) {
        const int c = p->obs_cam[k], l = p->obs_pt[k];
        double* Wk = W + 18 * (size_t)k;
        if (l < pt_lo || l >= pt_hi) { memset(Wk, 0, sizeof(double) * 18); continue; }
        double e[3], Jc[18], Jp[9], r0, w;
        ba_edge_obs(p->cam_T + 12 * c, p->pt_xyz + 3 * l, p->obs_meas + 3 * k, e, Jc, Jp);
        huber_w(p->info_obs * (e[0] * e[0] + e[1] * e[1] + e[2] * e[2]), p->huber_obs, p->use_huber, &r0, &w); chi += r0;
        const double wo = w * p->info_obs;
        for (int a = 0; a < 6; a++) {
            double s = 0; for (int r = 0; r < 3; r++) s += Jc[r * 6 + a] * e[r];
            bc[6 * c + a] -= wo * s;
            for (int b = 0; b < 6; b++) { double h = 0; for (int r = 0; r < 3; r++) h += Jc[r * 6 + a] * Jc[r * 6 + b]; Hcc[(size_t)(6 * c + a) * n6 + 6 * c + b] += wo * h; }
            for (int b = 0; b < 3; b++) { double h = 0; for (int r = 0; r < 3; r++) h += Jc[r * 6 + a] * Jp[r * 3 + b]; Wk[a * 3 + b] = wo * h; }
        }
        for (int a = 0; a < 3; a++) {
            double s = 0; for (int r = 0; r < 3; r++) s += Jp[r * 3 + a] * e[r];
            bp[3 * l + a] -= wo * s;
            for (int b = 0; b < 3; b++) { double h = 0; for (int r = 0; r < 3; r++) h += Jp[r * 3 + a] * Jp[r * 3 + b]; Hpp[9 * (size_t)l + a * 3 + b] += wo * h; }
        }
    }
    if (with_cam_factors) {
        for (int k = 0; k < p->n_odo + (p->prior_cam >= 0 ? 1 : 0); k++) {
            const int is_prior = (k == p->n_odo);
            const int i = is_prior ? -1 : p->odo_i[k], j = is_prior ? p->prior_cam : p->odo_j[k];
            double e[6], Ji[36], Jj[36], r0, w = 1;
            ba_edge_se3(is_prior ? p->prior_T : p->odo_T + 12 * k, is_prior ? NULL : p->cam_T + 12 * i, p->cam_T + 12 * j, e, is_prior ? NULL : Ji, Jj);
            double s2 = 0; for (int a = 0; a < 6; a++) s2 += e[a] * e[a];
            double info = is_prior ? p->info_prior : p->info_odo;
            if (is_prior) chi += info * s2; else { huber_w(info * s2, p->huber_odo, p->use_huber, &r0, &w); chi += r0; }
            const double wo = w * info;
            for (int a = 0; a < 6; a++) {
                double sj = 0, si = 0;
                for (int r = 0; r < 6; r++) { sj += Jj[r * 6 + a] * e[r]; if (!is_prior) si += Ji[r * 6 + a] * e[r]; }
                bc[6 * j + a] -= wo * sj; if (!is_prior) bc[6 * i + a] -= wo * si;
                for (int b = 0; b < 6; b++) {
                    double hjj = 0, hii = 0, hij = 0;
                    for (int r = 0; r < 6; r++) { hjj += Jj[r * 6 + a] * Jj[r * 6 + b]; if (!is_prior) { hii += Ji[r * 6 + a] * Ji[r * 6 + b]; hij += Ji[r * 6 + a] * Jj[r * 6 + b]; } }
                    Hcc[(size_t)(6 * j + a) * n6 + 6 * j + b] += wo * hjj;
                    if (!is_prior) {
                        Hcc[(size_t)(6 * i + a) * n6 + 6 * i + b] += wo * hii;
                        Hcc[(size_t)(6 * i + a) * n6 + 6 * j + b] += wo * hij; Hcc[(size_t)(6 * j + b) * n6 + 6 * i + a] += wo * hij;
                    }
                }
            }
        }
    }
    if (chi2_out) *chi2_out = chi;
}

static void inv3(const double* A, double* I)
{
    const double c0 = A[4] * A[8] - A[5] * A[7], c1 = A[5] * A[6] - A[3] * A[8], c2 = A[3] * A[7] - A[4] * A[6];
    const double d = 1.0 / (A[0] * c0 + A[1] * c1 + A[2] * c2);
    I[0] = c0 * d; I[1] = (A[2] * A[7] - A[1] * A[8]) * d; I[2] = (A[1] * A[5] - A[2] * A[4]) * d;
    I[3] = c1 * d; I[4] = (A[0] * A[8] - A[2] * A[6]) * d; I[5] = (A[2] * A[3] - A[0] * A[5]) * d;
    I[6] = c2 * d; I[7] = (A[1] * A[6] - A[0] * A[7]) * d; I[8] = (A[0] * A[4] - A[1] * A[3]) * d;
}

/* Reduced camera system of the landmark shard: S = [Hcc + lambda I] - sum_l W_l (Hpp_l + lambda I)^-1 W_l^T ; r likewise.
 * add_lambda: whether this shard adds lambda*I on the camera diagonal (exactly one shard does). */
void vo_ba_schur(const vo_ba_problem* p, int pt_lo, int pt_hi, double lambda, int add_lambda,
                 const double* Hcc, const double* bc, const double* Hpp, const double* bp, const double* W, double* S, double* r)
{
    const int n6 = 6 * p->n_cam;
    memcpy(S, Hcc, sizeof(double) * (size_t)n6 * n6); memcpy(r, bc, sizeof(double) * n6);
    if (add_lambda) for (int a = 0; a < n6; a++) S[(size_t)a * n6 + a] += lambda;
    /* group observations by point */
    int* start = (int*)calloc(p->n_pt + 2, sizeof(int)); int* order = (int*)malloc(sizeof(int) * (p->n_obs + 1));
    for (int k = 0; k < p->n_obs; k++) start[p->obs_pt[k] + 1]++;
    for (int l = 0; l < p->n_pt; l++) start[l + 1] += start[l];
    int* fill = (int*)malloc(sizeof(int) * (p->n_pt + 1)); memcpy(fill, start, sizeof(int) * (p->n_pt + 1));
    for (int k = 0; k < p->n_obs; k++) order[fill[p->obs_pt[k]]++] = k;
    for (int l = pt_lo; l < pt_hi; l++) {
        double D[9], Di[9]; memcpy(D, Hpp + 9 * (size_t)l, sizeof D); D[0] += lambda; D[4] += lambda; D[8] += lambda;
        inv3(D, Di);
        for (int a = start[l]; a < start[l + 1]; a++) {
            const int ka = order[a], ca = p->obs_cam[ka]; const double* Wa = W + 18 * (size_t)ka;
            double WD[18];
            for (int i = 0; i < 6; i++) for (int j = 0; j < 3; j++) WD[i * 3 + j] = Wa[i * 3] * Di[j] + Wa[i * 3 + 1] * Di[3 + j] + Wa[i * 3 + 2] * Di[6 + j];
            for (int i = 0; i < 6; i++) r[6 * ca + i] -= WD[i * 3] * bp[3 * l] + WD[i * 3 + 1] * bp[3 * l + 1] + WD[i * 3 + 2] * bp[3 * l + 2];
            for (int b = start[l]; b < start[l + 1]; b++) {
                const int kb = order[b], cb = p->obs_cam[kb]; const double* Wb = W + 18 * (size_t)kb;
                for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++)
                    S[(size_t)(6 * ca + i) * n6 + 6 * cb + j] -= WD[i * 3] * Wb[j * 3] + WD[i * 3 + 1] * Wb[j * 3 + 1] + WD[i * 3 + 2] * Wb[j * 3 + 2];
            }
        }
    }
    free(start); free(order); free(fill);
}

/* full LM run (single shard = everything) */
int vo_ba_optimize(vo_ba_problem* p, vo_ba_result* res)
{
    const int n6 = 6 * p->n_cam, np = p->n_pt;
    double* Hcc = (double*)malloc(sizeof(double) * (size_t)n6 * n6); double* S = (double*)malloc(sizeof(double) * (size_t)n6 * n6);
    double* bc = (double*)malloc(sizeof(double) * n6); double* r = (double*)malloc(sizeof(double) * n6); double* xc = (double*)calloc(n6, sizeof(double));
    double* Hpp = (double*)malloc(sizeof(double) * 9 * (size_t)np); double* bp = (double*)malloc(sizeof(double) * 3 * (size_t)np);
    double* xl = (double*)malloc(sizeof(double) * 3 * (size_t)np);
    double* W = (double*)malloc(sizeof(double) * 18 * (size_t)(p->n_obs + 1));
    double* cam_save = (double*)malloc(sizeof(double) * 12 * (size_t)p->n_cam); double* pt_save = (double*)malloc(sizeof(double) * 3 * (size_t)np);
    double lambda = -1, ni = 2, lastChi = 0, chi2_check = 0; int nBad = 0, trials = 0, it;
    res->chi2_initial = ba_chi2(p, p->cam_T, p->pt_xyz);
    for (it = 0; it < p->max_iters; it++) {
        double currentChi, iniChi;
        vo_ba_linearize(p, 0, np, 1, Hcc, bc, Hpp, bp, W, &currentChi);
        iniChi = currentChi;
        if (it == 0) {
            double md = 0;
            for (int a = 0; a < n6; a++) md = fmax(md, fabs(Hcc[(size_t)a * n6 + a]));
            for (int l = 0; l < np; l++) for (int a = 0; a < 3; a++) md = fmax(md, fabs(Hpp[9 * (size_t)l + 4 * a]));
            lambda = 1e-5 * md; ni = 2; nBad = 0;
        }
        double rho = 0; int qmax = 0;
        do {
            memcpy(cam_save, p->cam_T, sizeof(double) * 12 * p->n_cam); memcpy(pt_save, p->pt_xyz, sizeof(double) * 3 * np);
            vo_ba_schur(p, 0, np, lambda, 1, Hcc, bc, Hpp, bp, W, S, r);
            const int ok2 = vo_ldlt_solve(S, r, xc, n6);
            double scale = 0, tempChi;
            if (ok2) {
                /* back-substitution x_l = D^-1 (b_l - sum W^T x_c) */
                for (int l = 0; l < np; l++) { xl[3 * l] = bp[3 * l]; xl[3 * l + 1] = bp[3 * l + 1]; xl[3 * l + 2] = bp[3 * l + 2]; }
                for (int k = 0; k < p->n_obs; k++) {
                    const int c = p->obs_cam[k], l = p->obs_pt[k]; const double* Wk = W + 18 * (size_t)k;
                    for (int j = 0; j < 3; j++) { double s = 0; for (int i = 0; i < 6; i++) s += Wk[i * 3 + j] * xc[6 * c + i]; xl[3 * l + j] -= s; }
                }
                for (int l = 0; l < np; l++) {
                    double D[9], Di[9], t[3] = {xl[3 * l], xl[3 * l + 1], xl[3 * l + 2]};
                    memcpy(D, Hpp + 9 * (size_t)l, sizeof D); D[0] += lambda; D[4] += lambda; D[8] += lambda; inv3(D, Di);
                    for (int a = 0; a < 3; a++) xl[3 * l + a] = Di[a * 3] * t[0] + Di[a * 3 + 1] * t[1] + Di[a * 3 + 2] * t[2];
                }
                for (int c = 0; c < p->n_cam; c++) iso_oplus(p->cam_T + 12 * c, xc + 6 * c);
                for (int a = 0; a < 3 * np; a++) p->pt_xyz[a] += xl[a];
                for (int a = 0; a < n6; a++) scale += xc[a] * (lambda * xc[a] + bc[a]);
                for (int a = 0; a < 3 * np; a++) scale += xl[a] * (lambda * xl[a] + bp[a]);
                tempChi = ba_chi2(p, p->cam_T, p->pt_xyz);
            } else tempChi = DBL_MAX;
            rho = (currentChi - tempChi) / (scale + 1e-3);
            if (rho > 0 && isfinite(tempChi)) {
                double alpha = 1. - pow((2 * rho - 1), 3); alpha = fmin(alpha, 2. / 3.);
                lambda *= fmax(1. / 3., alpha); ni = 2; currentChi = tempChi;
            } else { lambda *= ni; ni *= 2; memcpy(p->cam_T, cam_save, sizeof(double) * 12 * p->n_cam); memcpy(p->pt_xyz, pt_save, sizeof(double) * 3 * np); }
            qmax++; trials++;
        } while (rho < 0 && qmax < 10);
        int terminate = (qmax == 10 || rho == 0);
        if (!terminate) { if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0; if (nBad >= 3) terminate = 1; }
        /* sparse_optimizer.cpp:393-396 (chi2 may not increase) + SparseOptimizerTerminateAction (relative gain) */
        const double chiNow = ba_chi2(p, p->cam_T, p->pt_xyz);
        if (chi2_check < chiNow && it > 0) terminate = 1;
        chi2_check = chiNow;
        if (it == 0) lastChi = chiNow;
        else { const double gain = (lastChi - chiNow) / chiNow; lastChi = chiNow; if (gain >= 0 && gain < p->gain_threshold) terminate = 1; }
        res->chi2_final = chiNow;
        if (terminate) { it++; break; }
    }
    res->iterations = it; res->lm_trials = trials; res->lambda_final = lambda;
    free(Hcc); free(S); free(bc); free(r); free(xc); free(Hpp); free(bp); free(xl); free(W); free(cam_save); free(pt_save);
    return 0;
}

/* test hooks */
void vo_ba_edge_se3(const double* Z, const double* Xi, const double* Xj, double* e, double* Ji, double* Jj) { ba_edge_se3(Z, Xi, Xj, e, Ji, Jj); }
void vo_ba_edge_obs(const double* X, const double* p, const double* m, double* e, double* Jc, double* Jp) { ba_edge_obs(X, p, m, e, Jc, Jp); }
void vo_iso_oplus(double* X, const double* d) { iso_oplus(X, d); }
double vo_ba_chi2(const vo_ba_problem* p) { return ba_chi2(p, p->cam_T, p->pt_xyz); }
