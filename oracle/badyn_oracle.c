/* badyn_oracle.c — CPU ORACLE (TEST INFRASTRUCTURE ONLY; see vido_oracle.h).
 *
 * Restates the FULL graph of Optimizer::FullBatchOptimization (vido_slam/src/Optimizer.cc:1235-2178, STATIC_ONLY = false):
 * in addition to the static factor set of ba_oracle.c
 *   - one VertexSE3 H per frame (>= 1) and object, initialised to identity            (:1583-1592)
 *   - one VertexPointXYZ per dynamic observation + its EdgeSE3PointXYZ to the camera    (:1560-1582, :1683-1726)
 *   - LandmarkMotionTernaryEdge(p_prev, p_cur, H), measurement 0, Huber                (:1728-1745)
 *       error      e = p_prev - H^-1 p_cur                       types/types_dyn_slam3d.cpp:53-61
 *       Jacobians  de/dp_prev = I, de/dp_cur = -R_H^T, de/dH = [ I | -[H^-1 p_cur]x ]  types/types_dyn_slam3d.cpp:63-85
 *                  (the rotation block is g2o's own approximation; restated as is)
 *   - EdgeSE3 smoothness between consecutive H of one object, measurement identity     (:1604-1636)
 * g2o solves the full, un-marginalised system (BlockSolverX + CSparse, :1318-1324) under LM
 * (core/optimization_algorithm_levenberg.cpp:61-189) with the gain stop (:1326-1328).  This oracle does literally that
 * with a dense LDL^T on the whole Hessian, which makes it an independent check of the HIP path's elimination scheme
 * (static points by 3x3 Schur, dynamic points by block-tridiagonal chain elimination).
 */
#include "vido_oracle.h"
#include <math.h>
#include <float.h>
#include <stdlib.h>
#include <string.h>

int vo_ldlt_solve(const double* A, const double* b, double* x, int n);
void vo_ba_edge_se3(const double* Z, const double* Xi, const double* Xj, double* e, double* Ji, double* Jj);
void vo_ba_edge_obs(const double* X, const double* p, const double* m, double* e, double* Jc, double* Jp);
void vo_iso_oplus(double* X, const double* d);

typedef struct {
    int32_t n_cam, n_pt, n_obs, n_odo, prior_cam, use_huber, max_iters, pad;
    double* cam_T; double* pt_xyz;
    const int32_t* obs_cam; const int32_t* obs_pt; const double* obs_meas;
    const int32_t* odo_i; const int32_t* odo_j; const double* odo_T;
    double prior_T[12];
    double info_obs, info_odo, info_prior, huber_obs, huber_odo, gain_threshold;
} vo_ba_problem;
typedef struct { int32_t iterations, lm_trials; double chi2_initial, chi2_final, lambda_final; } vo_ba_result;

typedef struct {
    int32_t n_H, n_dyn, n_tern, n_smooth;
    double* H_T;                 /* [n_H*12] object motions, row-major 3x4 (in/out) */
    double* dyn_xyz;             /* [n_dyn*3] dynamic points, world (in/out) */
    const int32_t* dyn_cam; const double* dyn_meas;                               /* [n_dyn], [n_dyn*3] */
    const int32_t* tern_prev; const int32_t* tern_cur; const int32_t* tern_H;     /* [n_tern] */
    const int32_t* sm_i; const int32_t* sm_j;                                     /* [n_smooth] H indices */
    double info_dyn, info_tern, info_smooth, huber_dyn, huber_tern, huber_smooth;
} vo_ba_dynamic;

static void huber_w(double e2, double delta, int use, double* rho0, double* rho1)
{
    if (!use || e2 <= delta * delta) { *rho0 = e2; *rho1 = 1.0; return; }
    const double s = sqrt(e2); *rho0 = 2 * s * delta - delta * delta; *rho1 = delta / s;
}

/* LandmarkMotionTernaryEdge */
static void edge_tern(const double* H, const double* pp, const double* pc, double* e, double* Jc /*3x3*/, double* JH /*3x6*/)
{
    double v[3];
    for (int r = 0; r < 3; r++) v[r] = H[0 * 4 + r] * (pc[0] - H[3]) + H[1 * 4 + r] * (pc[1] - H[7]) + H[2 * 4 + r] * (pc[2] - H[11]);
    for (int r = 0; r < 3; r++) e[r] = pp[r] - v[r];
    if (!Jc) return;
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) Jc[r * 3 + c] = -H[c * 4 + r];
    memset(JH, 0, sizeof(double) * 18);
    JH[0] = JH[7] = JH[14] = 1;
    JH[0 * 6 + 4] = v[2]; JH[0 * 6 + 5] = -v[1];
    JH[1 * 6 + 3] = -v[2]; JH[1 * 6 + 5] = v[0];
    JH[2 * 6 + 3] = v[1]; JH[2 * 6 + 4] = -v[0];
}

typedef struct { int nv; int off[3]; int dim[3]; int ne; double e[6]; double J[3][36]; double info, delta; int robust; } factor_t;

/* Sparse form of the same system (vo_badyn_optimize_sparse): while `coo.on` is set, add_factor appends (row, col, value) triplets of the factor's Hessian blocks
 * instead of adding into a dense matrix (duplicates are summed by the solver, like g2o's block matrix feeding CSparse, core/block_solver.hpp:104-176 +
 * solvers/linear_solver_csparse.h).  Single-threaded test code: one static sink. */
static struct { int on; size_t n, cap; int32_t* r; int32_t* c; double* v; } coo;
static void coo_push(int r, int c, double v)
{
    if (coo.n == coo.cap) {
        coo.cap = coo.cap ? coo.cap * 2 : (1u << 20);
        coo.r = (int32_t*)realloc(coo.r, sizeof(int32_t) * coo.cap); coo.c = (int32_t*)realloc(coo.c, sizeof(int32_t) * coo.cap); coo.v = (double*)realloc(coo.v, sizeof(double) * coo.cap);
    }
    coo.r[coo.n] = r; coo.c[coo.n] = c; coo.v[coo.n] = v; coo.n++;
}

/* accumulate one factor into the dense system; returns its robust chi2 */
static double add_factor(const factor_t* f, int use_huber, double* Hm, double* b, int N)
{
    double s2 = 0; for (int r = 0; r < f->ne; r++) s2 += f->e[r] * f->e[r];
    double r0 = f->info * s2, w = 1;
    if (f->robust) huber_w(f->info * s2, f->delta, use_huber, &r0, &w);
    if (!Hm) return r0;
    const double wo = w * f->info;
    for (int A = 0; A < f->nv; A++) {
        for (int a = 0; a < f->dim[A]; a++) {
            double s = 0; for (int r = 0; r < f->ne; r++) s += f->J[A][r * f->dim[A] + a] * f->e[r];
            b[f->off[A] + a] -= wo * s;
            for (int B = 0; B < f->nv; B++) for (int c = 0; c < f->dim[B]; c++) {
                double h = 0; for (int r = 0; r < f->ne; r++) h += f->J[A][r * f->dim[A] + a] * f->J[B][r * f->dim[B] + c];
                if (coo.on) coo_push(f->off[A] + a, f->off[B] + c, wo * h);
                else Hm[(size_t)(f->off[A] + a) * N + f->off[B] + c] += wo * h;
            }
        }
    }
    return r0;
}

/* builds (or only evaluates, Hm == NULL) the whole graph at the current estimates */
static double build(const vo_ba_problem* p, const vo_ba_dynamic* d, double* Hm, double* b, int N)
{
    const int oP = 0, oS = 6 * (p->n_cam + d->n_H), oD = oS + 3 * p->n_pt;
    double chi = 0; factor_t f;
    if (Hm && coo.on) { coo.n = 0; memset(b, 0, sizeof(double) * N); }
    else if (Hm) { memset(Hm, 0, sizeof(double) * (size_t)N * N); memset(b, 0, sizeof(double) * N); }
    for (int k = 0; k < p->n_obs; k++) {
        const int c = p->obs_cam[k], l = p->obs_pt[k];
        f.nv = 2; f.off[0] = oP + 6 * c; f.dim[0] = 6; f.off[1] = oS + 3 * l; f.dim[1] = 3; f.ne = 3; f.info = p->info_obs; f.delta = p->huber_obs; f.robust = 1;
        vo_ba_edge_obs(p->cam_T + 12 * c, p->pt_xyz + 3 * l, p->obs_meas + 3 * k, f.e, Hm ? f.J[0] : NULL, Hm ? f.J[1] : NULL);
        chi += add_factor(&f, p->use_huber, Hm, b, N);
    }
    for (int k = 0; k < d->n_dyn; k++) {
        const int c = d->dyn_cam[k];
        f.nv = 2; f.off[0] = oP + 6 * c; f.dim[0] = 6; f.off[1] = oD + 3 * k; f.dim[1] = 3; f.ne = 3; f.info = d->info_dyn; f.delta = d->huber_dyn; f.robust = 1;
        vo_ba_edge_obs(p->cam_T + 12 * c, d->dyn_xyz + 3 * k, d->dyn_meas + 3 * k, f.e, Hm ? f.J[0] : NULL, Hm ? f.J[1] : NULL);
        chi += add_factor(&f, p->use_huber, Hm, b, N);
    }
    for (int k = 0; k < d->n_tern; k++) {
        const int a = d->tern_prev[k], c = d->tern_cur[k], h = d->tern_H[k];
        f.nv = 3; f.off[0] = oD + 3 * a; f.dim[0] = 3; f.off[1] = oD + 3 * c; f.dim[1] = 3; f.off[2] = oP + 6 * (p->n_cam + h); f.dim[2] = 6;
        f.ne = 3; f.info = d->info_tern; f.delta = d->huber_tern; f.robust = 1;
        edge_tern(d->H_T + 12 * h, d->dyn_xyz + 3 * a, d->dyn_xyz + 3 * c, f.e, Hm ? f.J[1] : NULL, Hm ? f.J[2] : NULL);
        if (Hm) { memset(f.J[0], 0, sizeof(double) * 9); f.J[0][0] = f.J[0][4] = f.J[0][8] = 1; }
        chi += add_factor(&f, p->use_huber, Hm, b, N);
    }
    static const double I12[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    for (int k = 0; k < p->n_odo + d->n_smooth; k++) {
        const int sm = k >= p->n_odo;
        const int i = sm ? p->n_cam + d->sm_i[k - p->n_odo] : p->odo_i[k], j = sm ? p->n_cam + d->sm_j[k - p->n_odo] : p->odo_j[k];
        const double* Xi = sm ? d->H_T + 12 * (i - p->n_cam) : p->cam_T + 12 * i; const double* Xj = sm ? d->H_T + 12 * (j - p->n_cam) : p->cam_T + 12 * j;
        f.nv = 2; f.off[0] = oP + 6 * i; f.dim[0] = 6; f.off[1] = oP + 6 * j; f.dim[1] = 6; f.ne = 6;
        f.info = sm ? d->info_smooth : p->info_odo; f.delta = sm ? d->huber_smooth : p->huber_odo; f.robust = 1;
        vo_ba_edge_se3(sm ? I12 : p->odo_T + 12 * k, Xi, Xj, f.e, Hm ? f.J[0] : NULL, Hm ? f.J[1] : NULL);
        chi += add_factor(&f, p->use_huber, Hm, b, N);
    }
    if (p->prior_cam >= 0) {
        f.nv = 1; f.off[0] = oP + 6 * p->prior_cam; f.dim[0] = 6; f.ne = 6; f.info = p->info_prior; f.delta = 0; f.robust = 0;
        vo_ba_edge_se3(p->prior_T, NULL, p->cam_T + 12 * p->prior_cam, f.e, NULL, Hm ? f.J[0] : NULL);
        chi += add_factor(&f, p->use_huber, Hm, b, N);
    }
    return chi;
}

double vo_badyn_chi2(const vo_ba_problem* p, const vo_ba_dynamic* d) { return build(p, d, NULL, NULL, 0); }

/* dense Hessian / gradient of the whole graph (test hook): order = cams, Hs, static points, dynamic points */
double vo_badyn_system(const vo_ba_problem* p, const vo_ba_dynamic* d, double* Hm, double* b)
{
    const int N = 6 * (p->n_cam + d->n_H) + 3 * (p->n_pt + d->n_dyn);
    return build(p, d, Hm, b, N);
}

static void apply(vo_ba_problem* p, vo_ba_dynamic* d, const double* x)
{
    const int oS = 6 * (p->n_cam + d->n_H), oD = oS + 3 * p->n_pt;
    for (int c = 0; c < p->n_cam; c++) vo_iso_oplus(p->cam_T + 12 * c, x + 6 * c);
    for (int h = 0; h < d->n_H; h++) vo_iso_oplus(d->H_T + 12 * h, x + 6 * (p->n_cam + h));
    for (int a = 0; a < 3 * p->n_pt; a++) p->pt_xyz[a] += x[oS + a];
    for (int a = 0; a < 3 * d->n_dyn; a++) d->dyn_xyz[a] += x[oD + a];
}

int vo_badyn_optimize(vo_ba_problem* p, vo_ba_dynamic* d, vo_ba_result* res)
{
    const int N = 6 * (p->n_cam + d->n_H) + 3 * (p->n_pt + d->n_dyn);
    double* Hm = (double*)malloc(sizeof(double) * (size_t)N * N); double* A = (double*)malloc(sizeof(double) * (size_t)N * N);
    double* b = (double*)malloc(sizeof(double) * N); double* x = (double*)calloc(N, sizeof(double));
    const size_t nc = 12 * (size_t)p->n_cam, nh = 12 * (size_t)d->n_H, np = 3 * (size_t)p->n_pt, nd = 3 * (size_t)d->n_dyn;
    double* save = (double*)malloc(sizeof(double) * (nc + nh + np + nd + 1));
    double lambda = -1, ni = 2, lastChi = 0, chi2_check = 0; int nBad = 0, trials = 0, it;
    res->chi2_initial = build(p, d, NULL, NULL, 0);
    res->chi2_final = res->chi2_initial;
    for (it = 0; it < p->max_iters; it++) {
        double currentChi = build(p, d, Hm, b, N); const double iniChi = currentChi;
        if (it == 0) { double md = 0; for (int a = 0; a < N; a++) md = fmax(md, fabs(Hm[(size_t)a * N + a])); lambda = 1e-5 * md; ni = 2; nBad = 0; }
        double rho = 0; int qmax = 0;
        do {
            memcpy(save, p->cam_T, 8 * nc); memcpy(save + nc, d->H_T, 8 * nh); memcpy(save + nc + nh, p->pt_xyz, 8 * np); memcpy(save + nc + nh + np, d->dyn_xyz, 8 * nd);
            memcpy(A, Hm, sizeof(double) * (size_t)N * N);
            for (int a = 0; a < N; a++) A[(size_t)a * N + a] += lambda;
            const int ok2 = vo_ldlt_solve(A, b, x, N);
            double scale = 0, tempChi = DBL_MAX;
            if (ok2) {
                apply(p, d, x);
                for (int a = 0; a < N; a++) scale += x[a] * (lambda * x[a] + b[a]);
                tempChi = build(p, d, NULL, NULL, 0);
            }
            rho = (currentChi - tempChi) / (scale + 1e-3);
            if (rho > 0 && isfinite(tempChi)) {
                double alpha = 1. - pow((2 * rho - 1), 3); alpha = fmin(alpha, 2. / 3.);
                lambda *= fmax(1. / 3., alpha); ni = 2; currentChi = tempChi;
            } else {
                lambda *= ni; ni *= 2;
                memcpy(p->cam_T, save, 8 * nc); memcpy(d->H_T, save + nc, 8 * nh); memcpy(p->pt_xyz, save + nc + nh, 8 * np); memcpy(d->dyn_xyz, save + nc + nh + np, 8 * nd);
            }
            qmax++; trials++;
        } while (rho < 0 && qmax < 10);
        int terminate = (qmax == 10 || rho == 0);
        if (!terminate) { if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0; if (nBad >= 3) terminate = 1; }
        const double chiNow = build(p, d, NULL, NULL, 0);
        if (chi2_check < chiNow && it > 0) terminate = 1;
        chi2_check = chiNow;
        if (it == 0) lastChi = chiNow;
        else { const double gain = (lastChi - chiNow) / chiNow; lastChi = chiNow; if (gain >= 0 && gain < p->gain_threshold) terminate = 1; }
        res->chi2_final = chiNow;
        if (terminate) { it++; break; }
    }
    res->iterations = it; res->lm_trials = trials; res->lambda_final = lambda;
    free(Hm); free(A); free(b); free(x); free(save);
    return 0;
}

/* The same LM loop over the SPARSE un-eliminated system, for graphs the dense form cannot hold (configs[3](b): 20 keyframes x 2 000 landmarks + 5 objects x 100 points
 * = 36 690 unknowns).  The linear solve is delegated to `solve` (oracle/pyoracle.py hands in scipy's SuperLU: a sparse direct factorisation of the whole (H + lambda I),
 * which is what the reference does with CSparse, Optimizer.cc:1318-1324) — again a different algorithm from the product's Schur / chain elimination.
 * solve(N, nnz, rows, cols, vals, lambda, b, x) returns 1 on success.  LM policy identical to vo_badyn_optimize above. */
typedef int (*vo_sparse_solve_fn)(int32_t N, int64_t nnz, const int32_t* rows, const int32_t* cols, const double* vals, double lambda, const double* b, double* x);

int vo_badyn_optimize_sparse(vo_ba_problem* p, vo_ba_dynamic* d, vo_ba_result* res, vo_sparse_solve_fn solve)
{
    const int N = 6 * (p->n_cam + d->n_H) + 3 * (p->n_pt + d->n_dyn);
    double* b = (double*)malloc(sizeof(double) * N); double* x = (double*)calloc(N, sizeof(double));
    const size_t nc = 12 * (size_t)p->n_cam, nh = 12 * (size_t)d->n_H, np = 3 * (size_t)p->n_pt, nd = 3 * (size_t)d->n_dyn;
    double* save = (double*)malloc(sizeof(double) * (nc + nh + np + nd + 1));
    double lambda = -1, ni = 2, lastChi = 0, chi2_check = 0; int nBad = 0, trials = 0, it;
    double dummy = 0;
    res->chi2_initial = build(p, d, NULL, NULL, 0);
    res->chi2_final = res->chi2_initial;
    for (it = 0; it < p->max_iters; it++) {
        coo.on = 1; double currentChi = build(p, d, &dummy, b, N); coo.on = 0; const double iniChi = currentChi;
        if (it == 0) {
            double* dg = (double*)calloc(N, sizeof(double)); double md = 0;
            for (size_t k = 0; k < coo.n; k++) if (coo.r[k] == coo.c[k]) dg[coo.r[k]] += coo.v[k];
            for (int a = 0; a < N; a++) md = fmax(md, fabs(dg[a]));
            free(dg); lambda = 1e-5 * md; ni = 2; nBad = 0;
        }
        double rho = 0; int qmax = 0;
        do {
            memcpy(save, p->cam_T, 8 * nc); memcpy(save + nc, d->H_T, 8 * nh); memcpy(save + nc + nh, p->pt_xyz, 8 * np); memcpy(save + nc + nh + np, d->dyn_xyz, 8 * nd);
            const int ok2 = solve(N, (int64_t)coo.n, coo.r, coo.c, coo.v, lambda, b, x);
            double scale = 0, tempChi = DBL_MAX;
            if (ok2) {
                apply(p, d, x);
                for (int a = 0; a < N; a++) scale += x[a] * (lambda * x[a] + b[a]);
                tempChi = build(p, d, NULL, NULL, 0);
            }
            rho = (currentChi - tempChi) / (scale + 1e-3);
            if (rho > 0 && isfinite(tempChi)) {
                double alpha = 1. - pow((2 * rho - 1), 3); alpha = fmin(alpha, 2. / 3.);
                lambda *= fmax(1. / 3., alpha); ni = 2; currentChi = tempChi;
            } else {
                lambda *= ni; ni *= 2;
                memcpy(p->cam_T, save, 8 * nc); memcpy(d->H_T, save + nc, 8 * nh); memcpy(p->pt_xyz, save + nc + nh, 8 * np); memcpy(d->dyn_xyz, save + nc + nh + np, 8 * nd);
            }
            qmax++; trials++;
        } while (rho < 0 && qmax < 10);
        int terminate = (qmax == 10 || rho == 0);
        if (!terminate) { if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0; if (nBad >= 3) terminate = 1; }
        const double chiNow = build(p, d, NULL, NULL, 0);
        if (chi2_check < chiNow && it > 0) terminate = 1;
        chi2_check = chiNow;
        if (it == 0) lastChi = chiNow;
        else { const double gain = (lastChi - chiNow) / chiNow; lastChi = chiNow; if (gain >= 0 && gain < p->gain_threshold) terminate = 1; }
        res->chi2_final = chiNow;
        if (terminate) { it++; break; }
    }
    res->iterations = it; res->lm_trials = trials; res->lambda_final = lambda;
    free(b); free(x); free(save); free(coo.r); free(coo.c); free(coo.v); coo.r = coo.c = NULL; coo.v = NULL; coo.n = coo.cap = 0;
    return 0;
}

void vo_edge_tern(const double* H, const double* pp, const double* pc, double* e, double* Jc, double* JH) { edge_tern(H, pp, pc, e, Jc, JH); }
