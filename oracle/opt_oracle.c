/* opt_oracle.c — CPU ORACLE (TEST INFRASTRUCTURE ONLY; see vido_oracle.h).
 *
 * Restates the four per-frame optimisers of the reference and the g2o machinery they run on:
 *   Optimizer.cc:2180-2334 PoseOptimizationNew      (EdgeSE3ProjectXYZOnlyPose, Huber, 1 round x100)
 *   Optimizer.cc:2622-2824 PoseOptimizationFlow2Cam (EdgeSE3ProjectFlow2 + EdgeFlowPrior, Schur on the
 *                                                    2-D flow vertices, 4 rounds x100)
 *   Optimizer.cc:2826-3035 PoseOptimizationObjMot   (EdgeSE3ProjectXYZOnlyObjMotion, no kernel, 1 round x200)
 *   Optimizer.cc:3037-3253 PoseOptimizationFlow2    (as Flow2Cam, prior 0.5, 1 round x200)
 * g2o (vido_slam/3rdparty/g2o/g2o): core/optimization_algorithm_levenberg.cpp:61-189 (LM policy),
 *   core/sparse_optimizer.cpp:354-427 (outer loop + added chi2 check), core/block_solver.hpp:354-486
 *   (Schur), solvers/linear_solver_dense.h:65-118 (LDLT), core/robust_kernel_impl.cpp:65-91 (Huber),
 *   core/base_unary_edge.hpp / base_binary_edge.hpp (quadratic forms), types/se3quat.h:221-262 (exp),
 *   types/types_six_dof_expmap.{h,cpp} (residuals/Jacobians; lines cited at each function).
 * Poses are kept as R(3x3)+t in double (g2o keeps a unit quaternion; difference ~1e-16).
 * The reference's addnoise=1 depth noise (time-seeded RNG) is NOT reproduced.
 */
#include "vido_oracle.h"
#include <math.h>
#include <float.h>
#include <stdlib.h>
#include <string.h>

typedef struct { double R[9], t[3]; } se3;

static void se3_from_mat(const double* M, se3* T) { for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) T->R[r * 3 + c] = M[r * 4 + c]; T->t[r] = M[r * 4 + 3]; } }
static void se3_to_mat(const se3* T, double* M) { for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) M[r * 4 + c] = T->R[r * 3 + c]; M[r * 4 + 3] = T->t[r]; } M[12] = M[13] = M[14] = 0; M[15] = 1; }
static void se3_map(const se3* T, const double* x, double* y) { for (int r = 0; r < 3; r++) y[r] = T->R[r * 3] * x[0] + T->R[r * 3 + 1] * x[1] + T->R[r * 3 + 2] * x[2] + T->t[r]; }
static void mat3_mul(const double* A, const double* B, double* C) { for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) C[r * 3 + c] = A[r * 3] * B[c] + A[r * 3 + 1] * B[3 + c] + A[r * 3 + 2] * B[6 + c]; }

/* SE3Quat::exp, se3quat.h:221-262: update = (omega, upsilon) */
static void se3_exp(const double* u, se3* T)
{
    const double w[3] = {u[0], u[1], u[2]};
    const double theta = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    const double O[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
    double O2[9]; mat3_mul(O, O, O2);
    double V[9];
    if (theta < 0.00001) {
        for (int i = 0; i < 9; i++) T->R[i] = (i % 4 == 0 ? 1.0 : 0.0) + O[i] + O2[i];
        memcpy(V, T->R, sizeof V);
    } else {
        const double a = sin(theta) / theta, b = (1 - cos(theta)) / (theta * theta), c = (theta - sin(theta)) / pow(theta, 3);
        for (int i = 0; i < 9; i++) { T->R[i] = (i % 4 == 0 ? 1.0 : 0.0) + a * O[i] + b * O2[i]; V[i] = (i % 4 == 0 ? 1.0 : 0.0) + b * O[i] + c * O2[i]; }
    }
    for (int r = 0; r < 3; r++) T->t[r] = V[r * 3] * u[3] + V[r * 3 + 1] * u[4] + V[r * 3 + 2] * u[5];
}
/* VertexSE3Expmap::oplusImpl, types_six_dof_expmap.h:80-83: T <- exp(update) * T */
static void se3_oplus_left(se3* T, const double* u)
{
    se3 E; se3_exp(u, &E);
    se3 N; mat3_mul(E.R, T->R, N.R);
    for (int r = 0; r < 3; r++) N.t[r] = E.R[r * 3] * T->t[0] + E.R[r * 3 + 1] * T->t[1] + E.R[r * 3 + 2] * T->t[2] + E.t[r];
    *T = N;
}

/* dense LDL^T (no pivoting) solve of an n x n SPD system; returns 0 if a pivot is not positive
 * (g2o: Eigen::LDLT + isPositive(), linear_solver_dense.h:107-112) */
int vo_ldlt_solve(const double* A, const double* b, double* x, int n)
{
    double* L = (double*)malloc(sizeof(double) * n * n); double* D = (double*)malloc(sizeof(double) * n);
    memcpy(L, A, sizeof(double) * n * n);
    int ok = 1;
    for (int j = 0; j < n && ok; j++) {
        double d = L[j * n + j];
        for (int k = 0; k < j; k++) d -= L[j * n + k] * L[j * n + k] * D[k];
        if (!(d > 0) || !isfinite(d)) { ok = 0; break; }
        D[j] = d;
        for (int i = j + 1; i < n; i++) {
            double s = L[i * n + j];
            for (int k = 0; k < j; k++) s -= L[i * n + k] * L[j * n + k] * D[k];
            L[i * n + j] = s / d;
        }
    }
    if (ok) {
        for (int i = 0; i < n; i++) { double s = b[i]; for (int k = 0; k < i; k++) s -= L[i * n + k] * x[k]; x[i] = s; }
        for (int i = 0; i < n; i++) x[i] /= D[i];
        for (int i = n - 1; i >= 0; i--) { double s = x[i]; for (int k = i + 1; k < n; k++) s -= L[k * n + i] * x[k]; x[i] = s; }
    }
    free(L); free(D);
    return ok;
}

/* ---- problem description (mirrors include/vido_c.h vido_pose_problem) ---- */
typedef struct {
    int32_t mode;              /* 0 REPROJ_CAM, 1 FLOW (Flow2Cam / Flow2), 2 REPROJ_OBJMOT */
    int32_t n;
    const double* Xw;          /* [n*3] modes 0,2 */
    const double* obs;         /* [n*2] */
    const double* flow0;       /* [n*2] mode 1 */
    const double* depth;       /* [n]   mode 1 */
    double Twl[16];            /* mode 1 */
    double P[12];              /* mode 2 */
    double fx, fy, cx, cy;
    double T_init[16];
    double info_edge, info_prior, huber_delta;
    int32_t use_huber, rounds, drop_kernel_after_round;
    int32_t iters[4];
    float chi2_th[4];
} vo_pose_problem;

typedef struct { double T[16]; int32_t n_inliers, lm_iterations; double chi2_final; } vo_pose_result;

/* residual (and optionally the 2x6 pose Jacobian) of edge i at pose T, flow f */
static void edge_eval(const vo_pose_problem* p, const se3* T, int i, const double* f, double* e, double* J)
{
    double X[3], q[3];
    if (p->mode == 1) {   /* EdgeSE3ProjectFlow2::computeError, types_six_dof_expmap.h:445-454 */
        const double d = p->depth[i];
        const double Xc[3] = {(p->obs[2 * i] - p->cx) * d / p->fx, (p->obs[2 * i + 1] - p->cy) * d / p->fy, d};
        for (int r = 0; r < 3; r++) X[r] = p->Twl[r * 4] * Xc[0] + p->Twl[r * 4 + 1] * Xc[1] + p->Twl[r * 4 + 2] * Xc[2] + p->Twl[r * 4 + 3];
    } else { X[0] = p->Xw[3 * i]; X[1] = p->Xw[3 * i + 1]; X[2] = p->Xw[3 * i + 2]; }
    se3_map(T, X, q);
    const double x = q[0], y = q[1], z = q[2];
    if (p->mode == 2) {   /* EdgeSE3ProjectXYZOnlyObjMotion, types_six_dof_expmap.cpp:394-443 */
        const double* P = p->P;
        const double m1 = P[0] * x + P[1] * y + P[2] * z + P[3], m2 = P[4] * x + P[5] * y + P[6] * z + P[7], m3 = P[8] * x + P[9] * y + P[10] * z + P[11];
        const double invm3 = 1.0 / m3;
        e[0] = p->obs[2 * i] - m1 * invm3; e[1] = p->obs[2 * i + 1] - m2 * invm3;
        if (J) {
            const double i2 = invm3 * invm3;
            const double t00 = i2 * (P[0] * m3 - P[8] * m1), t01 = i2 * (P[1] * m3 - P[9] * m1), t02 = i2 * (P[2] * m3 - P[10] * m1);
            const double t10 = i2 * (P[4] * m3 - P[8] * m2), t11 = i2 * (P[5] * m3 - P[9] * m2), t12 = i2 * (P[6] * m3 - P[10] * m2);
            J[0] = -1.0 * (y * t02 - z * t01); J[1] = -1.0 * (z * t00 - x * t02); J[2] = -1.0 * (x * t01 - y * t00); J[3] = -t00; J[4] = -t01; J[5] = -t02;
            J[6] = -1.0 * (y * t12 - z * t11); J[7] = -1.0 * (z * t10 - x * t12); J[8] = -1.0 * (x * t11 - y * t10); J[9] = -t10; J[10] = -t11; J[11] = -t12;
        }
        return;
    }
    const double u = x / z * p->fx + p->cx, v = y / z * p->fy + p->cy;           /* cam_project */
    if (p->mode == 1) { e[0] = (p->obs[2 * i] + f[0]) - u; e[1] = (p->obs[2 * i + 1] + f[1]) - v; }
    else { e[0] = p->obs[2 * i] - u; e[1] = p->obs[2 * i + 1] - v; }
    if (J) {
        if (p->mode == 0) {   /* EdgeSE3ProjectXYZOnlyPose::linearizeOplus, types_six_dof_expmap.cpp:266-288 */
            const double invz = 1.0 / z, invz_2 = invz * invz;
            J[0] = x * y * invz_2 * p->fx; J[1] = -(1 + (x * x * invz_2)) * p->fx; J[2] = y * invz * p->fx; J[3] = -invz * p->fx; J[4] = 0; J[5] = x * invz_2 * p->fx;
            J[6] = (1 + y * y * invz_2) * p->fy; J[7] = -x * y * invz_2 * p->fy; J[8] = -x * invz * p->fy; J[9] = 0; J[10] = -invz * p->fy; J[11] = y * invz_2 * p->fy;
        } else {              /* EdgeSE3ProjectFlow2::linearizeOplus, types_six_dof_expmap.cpp:813-845 */
            const double z_2 = z * z;
            J[0] = x * y / z_2 * p->fx; J[1] = -(1 + (x * x / z_2)) * p->fx; J[2] = y / z * p->fx; J[3] = -1. / z * p->fx; J[4] = 0; J[5] = x / z_2 * p->fx;
            J[6] = (1 + y * y / z_2) * p->fy; J[7] = -x * y / z_2 * p->fy; J[8] = -x / z * p->fy; J[9] = 0; J[10] = -1. / z * p->fy; J[11] = y / z_2 * p->fy;
        }
    }
}

/* RobustKernelHuber::robustify, robust_kernel_impl.cpp:78-91 */
static void huber(double e2, double delta, double* rho0, double* rho1)
{
    const double dsqr = delta * delta;
    if (e2 <= dsqr) { *rho0 = e2; *rho1 = 1.0; }
    else { const double s = sqrt(e2); *rho0 = 2 * s * delta - dsqr; *rho1 = delta / s; }
}

int vo_pose_optimize(const vo_pose_problem* p, vo_pose_result* res, uint8_t* outlier, double* flow_out)
{
    const int n = p->n, flowm = (p->mode == 1);
    se3 T, Tinit; se3_from_mat(p->T_init, &Tinit); T = Tinit;
    double* f = (double*)calloc(2 * (size_t)n + 2, sizeof(double));
    double* err = (double*)calloc(2 * (size_t)n + 2, sizeof(double));      /* edge _error as of the last computeActiveErrors */
    double* fsave = (double*)calloc(2 * (size_t)n + 2, sizeof(double));
    double* Hpl = (double*)calloc(12 * (size_t)n + 12, sizeof(double));
    double* Hll = (double*)calloc((size_t)n + 1, sizeof(double));            /* 2x2 blocks are scalar * I */
    double* bl = (double*)calloc(2 * (size_t)n + 2, sizeof(double));
    double* xl = (double*)calloc(2 * (size_t)n + 2, sizeof(double));
    uint8_t* has_kernel = (uint8_t*)malloc(n + 1);
    if (flowm) memcpy(f, p->flow0, sizeof(double) * 2 * n);
    memset(outlier, 0, n); memset(has_kernel, p->use_huber ? 1 : 0, n);
    int total_iters = 0;
    res->n_inliers = 0; res->chi2_final = 0;
    if (n < 3) { se3_to_mat(&T, res->T); res->lm_iterations = 0; goto done; }

    for (int round = 0; round < p->rounds; round++) {
        T = Tinit;                                                          /* vSE3->setEstimate(Init) each round */
        double lambda = -1, ni = 2; int nBad = 0; double chi2_check = 0;
        for (int it = 0; it < p->iters[round]; it++) {
            /* computeActiveErrors + activeRobustChi2 */
            double chi = 0;
            for (int i = 0; i < n; i++) {
                if (!outlier[i]) {
                    edge_eval(p, &T, i, f + 2 * i, err + 2 * i, NULL);
                    const double c2 = p->info_edge * (err[2 * i] * err[2 * i] + err[2 * i + 1] * err[2 * i + 1]);
                    double r0 = c2, r1 = 1; if (has_kernel[i]) huber(c2, p->huber_delta, &r0, &r1);
                    chi += r0;
                }
                if (flowm) { const double a = f[2 * i] - p->flow0[2 * i], b = f[2 * i + 1] - p->flow0[2 * i + 1]; chi += p->info_prior * (a * a + b * b); }
            }
            double currentChi = chi; const double iniChi = chi;
            /* buildSystem */
            double H[36], b6[6]; memset(H, 0, sizeof H); memset(b6, 0, sizeof b6);
            for (int i = 0; i < n; i++) {
                if (flowm) { Hll[i] = p->info_prior; bl[2 * i] = -p->info_prior * (f[2 * i] - p->flow0[2 * i]); bl[2 * i + 1] = -p->info_prior * (f[2 * i + 1] - p->flow0[2 * i + 1]); memset(Hpl + 12 * i, 0, sizeof(double) * 12); }
                if (outlier[i]) continue;
                double e[2], J[12]; edge_eval(p, &T, i, f + 2 * i, e, J);
                const double c2 = p->info_edge * (e[0] * e[0] + e[1] * e[1]);
                double r0 = c2, w = 1; if (has_kernel[i]) huber(c2, p->huber_delta, &r0, &w);
                const double wo = w * p->info_edge;
                for (int a = 0; a < 6; a++) {
                    b6[a] -= wo * (J[a] * e[0] + J[6 + a] * e[1]);
                    for (int c = 0; c < 6; c++) H[a * 6 + c] += wo * (J[a] * J[c] + J[6 + a] * J[6 + c]);
                }
                if (flowm) {
                    Hll[i] += wo; bl[2 * i] -= wo * e[0]; bl[2 * i + 1] -= wo * e[1];
                    for (int a = 0; a < 6; a++) { Hpl[12 * i + 2 * a] = wo * J[a]; Hpl[12 * i + 2 * a + 1] = wo * J[6 + a]; }
                }
            }
            if (it == 0) {                                                  /* computeLambdaInit: tau * max diagonal */
                double md = 0; for (int a = 0; a < 6; a++) md = fmax(md, fabs(H[a * 6 + a]));
                if (flowm) for (int i = 0; i < n; i++) md = fmax(md, fabs(Hll[i]));
                lambda = 1e-5 * md; ni = 2; nBad = 0;
            }
            double rho = 0; int qmax = 0;
            do {
                const se3 Tsave = T; if (flowm) memcpy(fsave, f, sizeof(double) * 2 * n);
                /* solve (H + lambda I) x = b, Schur over the flow vertices */
                double S[36], bs[6], xp[6];
                memcpy(S, H, sizeof S); memcpy(bs, b6, sizeof bs);
                for (int a = 0; a < 6; a++) S[a * 6 + a] += lambda;
                if (flowm) for (int i = 0; i < n; i++) {
                    const double dinv = 1.0 / (Hll[i] + lambda); const double* B = Hpl + 12 * i;
                    for (int a = 0; a < 6; a++) {
                        bs[a] -= dinv * (B[2 * a] * bl[2 * i] + B[2 * a + 1] * bl[2 * i + 1]);
                        for (int c = 0; c < 6; c++) S[a * 6 + c] -= dinv * (B[2 * a] * B[2 * c] + B[2 * a + 1] * B[2 * c + 1]);
                    }
                }
                const int ok2 = vo_ldlt_solve(S, bs, xp, 6);
                double scale = 0;
                if (ok2) {
                    if (flowm) for (int i = 0; i < n; i++) {
                        const double dinv = 1.0 / (Hll[i] + lambda); const double* B = Hpl + 12 * i;
                        double c0 = bl[2 * i], c1 = bl[2 * i + 1];
                        for (int a = 0; a < 6; a++) { c0 -= B[2 * a] * xp[a]; c1 -= B[2 * a + 1] * xp[a]; }
                        xl[2 * i] = dinv * c0; xl[2 * i + 1] = dinv * c1;
                    }
                    se3_oplus_left(&T, xp);
                    if (flowm) for (int i = 0; i < 2 * n; i++) f[i] += xl[i];
                    for (int a = 0; a < 6; a++) scale += xp[a] * (lambda * xp[a] + b6[a]);
                    if (flowm) for (int i = 0; i < 2 * n; i++) scale += xl[i] * (lambda * xl[i] + bl[i]);
                } /* !ok2: g2o still applies the stale x; with a failed LDLT the trial is rejected via tempChi=max anyway */
                double tempChi = 0;
                for (int i = 0; i < n; i++) {
                    if (!outlier[i]) {
                        edge_eval(p, &T, i, f + 2 * i, err + 2 * i, NULL);
                        const double c2 = p->info_edge * (err[2 * i] * err[2 * i] + err[2 * i + 1] * err[2 * i + 1]);
                        double r0 = c2, r1 = 1; if (has_kernel[i]) huber(c2, p->huber_delta, &r0, &r1);
                        tempChi += r0;
                    }
                    if (flowm) { const double a = f[2 * i] - p->flow0[2 * i], b = f[2 * i + 1] - p->flow0[2 * i + 1]; tempChi += p->info_prior * (a * a + b * b); }
                }
                if (!ok2) tempChi = DBL_MAX;
                rho = (currentChi - tempChi) / (scale + 1e-3);
                if (rho > 0 && isfinite(tempChi)) {
                    double alpha = 1. - pow((2 * rho - 1), 3);
                    alpha = fmin(alpha, 2. / 3.);
                    lambda *= fmax(1. / 3., alpha); ni = 2; currentChi = tempChi;
                } else { lambda *= ni; ni *= 2; T = Tsave; if (flowm) memcpy(f, fsave, sizeof(double) * 2 * n); }
                qmax++;
            } while (rho < 0 && qmax < 10);
            total_iters++;
            int terminate = (qmax == 10 || rho == 0);
            if (!terminate) { if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0; if (nBad >= 3) terminate = 1; }
            /* sparse_optimizer.cpp:393-396: activeRobustChi2() here is over the errors of the last trial */
            double lastChi = 0;
            for (int i = 0; i < n; i++) {
                if (!outlier[i]) { const double c2 = p->info_edge * (err[2 * i] * err[2 * i] + err[2 * i + 1] * err[2 * i + 1]); double r0 = c2, r1; if (has_kernel[i]) huber(c2, p->huber_delta, &r0, &r1); lastChi += r0; }
                if (flowm) { const double a = f[2 * i] - p->flow0[2 * i], b = f[2 * i + 1] - p->flow0[2 * i + 1]; lastChi += p->info_prior * (a * a + b * b); }
            }
            if (chi2_check < lastChi && it > 0) terminate = 1;
            chi2_check = lastChi;
            res->chi2_final = currentChi;
            if (terminate) break;
        }
        /* inlier / outlier classification, e.g. Optimizer.cc:2277-2312 */
        int nbad = 0;
        for (int i = 0; i < n; i++) {
            if (outlier[i]) edge_eval(p, &T, i, f + 2 * i, err + 2 * i, NULL);
            const float chi2 = (float)(p->info_edge * (err[2 * i] * err[2 * i] + err[2 * i + 1] * err[2 * i + 1]));
            if (chi2 > p->chi2_th[round]) { outlier[i] = 1; nbad++; } else outlier[i] = 0;
            if (round == p->drop_kernel_after_round) has_kernel[i] = 0;
        }
        res->n_inliers = n - nbad;
    }
    se3_to_mat(&T, res->T);
    res->lm_iterations = total_iters;
done:
    if (flow_out) { if (flowm) memcpy(flow_out, f, sizeof(double) * 2 * n); else memset(flow_out, 0, sizeof(double) * 2 * n); }
    free(f); free(err); free(fsave); free(Hpl); free(Hll); free(bl); free(xl); free(has_kernel);
    return 0;
}
