/* pnp_oracle.c — CPU ORACLE (TEST INFRASTRUCTURE ONLY; see vido_oracle.h).
 *
 * Restates the initial-model stage of the tracker: Tracking::GetInitModelCam / GetInitModelObj
 * (vido_slam/src/Tracking.cc:1914-2028, :2030-2162) call cv::solvePnPRansac(..., 500 iterations, 0.4 px, 0.98,
 * SOLVEPNP_P3P).  OpenCV (third party, absent from /root/reference, pinned 3.4 by vido_slam/CMakeLists.txt:20) is
 * restated from its published algorithm: RANSAC over 4-point samples (3 for P3P + 1 to pick among the up to four
 * P3P solutions), squared reprojection error <= threshold^2, adaptive iteration count
 * niters = log(1-conf)/log(1-(1-eps)^4).  Deviations (PARITY UNPINNED, SURVEY.md hard part 6): OpenCV's global RNG
 * is replaced by a seeded counter-based generator (splitmix64 of seed and iteration), the P3P equations are solved
 * in Grunert's form with a Durand-Kerner quartic solver, and the final EPnP refit on the inliers is omitted (the
 * caller refines with Levenberg-Marquardt anyway: PoseOptimizationFlow2Cam / Flow2).
 */
#include "vido_oracle.h"
#include <math.h>
#include <float.h>
#include <stdlib.h>
#include <string.h>

static uint64_t splitmix64(uint64_t* s) { uint64_t z = (*s += 0x9E3779B97F4A7C15ULL); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31); }

void vo_pnp_sample4(uint64_t seed, int it, int n, int idx[4])
{
    uint64_t s = seed ^ (0xD1B54A32D192ED03ULL * (uint64_t)(it + 1));
    for (int k = 0; k < 4;) {
        int c = (int)(splitmix64(&s) % (uint64_t)n), dup = 0;
        for (int j = 0; j < k; j++) if (idx[j] == c) dup = 1;
        if (!dup) idx[k++] = c;
    }
}

/* real roots of c4 v^4 + ... + c0 by Durand-Kerner on the monic polynomial; returns count */
static int quartic_real_roots(const double* c, double* roots)
{
    if (fabs(c[4]) < 1e-300) return 0;
    const double a3 = c[3] / c[4], a2 = c[2] / c[4], a1 = c[1] / c[4], a0 = c[0] / c[4];
    double zr[4] = {1.0, 0.4, -0.65, -0.2755}, zi[4] = {0.0, 0.9, 0.72, -0.9602};    /* powers of (0.4+0.9i) */
    for (int it = 0; it < 100; it++) {
        double maxd = 0;
        for (int k = 0; k < 4; k++) {
            /* p(z) */
            double pr = 1, pi = 0, tr, ti;
            tr = pr * zr[k] - pi * zi[k] + a3; ti = pr * zi[k] + pi * zr[k]; pr = tr; pi = ti;
            tr = pr * zr[k] - pi * zi[k] + a2; ti = pr * zi[k] + pi * zr[k]; pr = tr; pi = ti;
            tr = pr * zr[k] - pi * zi[k] + a1; ti = pr * zi[k] + pi * zr[k]; pr = tr; pi = ti;
            tr = pr * zr[k] - pi * zi[k] + a0; ti = pr * zi[k] + pi * zr[k]; pr = tr; pi = ti;
            double qr = 1, qi = 0;
            for (int j = 0; j < 4; j++) if (j != k) { const double dr = zr[k] - zr[j], di = zi[k] - zi[j]; tr = qr * dr - qi * di; ti = qr * di + qi * dr; qr = tr; qi = ti; }
            const double den = qr * qr + qi * qi;
            if (den < 1e-300) continue;
            const double dr = (pr * qr + pi * qi) / den, di = (pi * qr - pr * qi) / den;
            zr[k] -= dr; zi[k] -= di;
            if (fabs(dr) + fabs(di) > maxd) maxd = fabs(dr) + fabs(di);
        }
        if (maxd < 1e-14) break;
    }
    int n = 0;
    for (int k = 0; k < 4; k++) if (fabs(zi[k]) < 1e-7 * (1.0 + fabs(zr[k]))) {
        double v = zr[k];
        for (int t = 0; t < 2; t++) {      /* Newton polish on the real axis */
            const double p = (((c[4] * v + c[3]) * v + c[2]) * v + c[1]) * v + c[0], dp = ((4 * c[4] * v + 3 * c[3]) * v + 2 * c[2]) * v + c[1];
            if (fabs(dp) > 1e-300) v -= p / dp;
        }
        roots[n++] = v;
    }
    return n;
}

static void cross3(const double* a, const double* b, double* c) { c[0] = a[1] * b[2] - a[2] * b[1]; c[1] = a[2] * b[0] - a[0] * b[2]; c[2] = a[0] * b[1] - a[1] * b[0]; }
static double norm3(const double* a) { return sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }

/* P3P (Grunert): world points P[3][3], unit bearing vectors j[3][3]; up to 4 poses (R row-major 9, t 3). */
int vo_p3p(const double P[3][3], const double j[3][3], double R[4][9], double t[4][3])
{
    double d[3];
    for (int k = 0; k < 3; k++) d[k] = P[1][k] - P[2][k]; const double a2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    for (int k = 0; k < 3; k++) d[k] = P[0][k] - P[2][k]; const double b2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    for (int k = 0; k < 3; k++) d[k] = P[0][k] - P[1][k]; const double c2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    if (a2 < 1e-20 || b2 < 1e-20 || c2 < 1e-20) return 0;
    const double ca = j[1][0] * j[2][0] + j[1][1] * j[2][1] + j[1][2] * j[2][2];
    const double cb = j[0][0] * j[2][0] + j[0][1] * j[2][1] + j[0][2] * j[2][2];
    const double cg = j[0][0] * j[1][0] + j[0][1] * j[1][1] + j[0][2] * j[1][2];
    const double K = (a2 - c2) / b2, M = c2 / b2;
    /* u = N(v)/D(v);  quartic: D^2 + N^2 - 2 cg N D - M Q D^2 = 0,  Q = 1 + v^2 - 2 cb v */
    const double N[3] = {1 + K, -2 * K * cb, K - 1}, D[2] = {2 * cg, -2 * ca}, Q[3] = {1, -2 * cb, 1};
    double D2[3] = {D[0] * D[0], 2 * D[0] * D[1], D[1] * D[1]};
    double c[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < 3; i++) c[i] += D2[i];
    for (int i = 0; i < 3; i++) for (int k = 0; k < 3; k++) c[i + k] += N[i] * N[k];
    for (int i = 0; i < 3; i++) for (int k = 0; k < 2; k++) c[i + k] -= 2 * cg * N[i] * D[k];
    for (int i = 0; i < 3; i++) for (int k = 0; k < 3; k++) c[i + k] -= M * Q[i] * D2[k];
    double roots[4]; const int nr = quartic_real_roots(c, roots);
    int ns = 0;
    for (int r = 0; r < nr; r++) {
        const double v = roots[r];
        if (!(v > 0)) continue;
        const double den = D[0] + D[1] * v;
        if (fabs(den) < 1e-12) continue;
        const double u = (N[0] + N[1] * v + N[2] * v * v) / den;
        if (!(u > 0)) continue;
        const double q = 1 + v * v - 2 * v * cb;
        if (!(q > 0)) continue;
        const double s1 = sqrt(b2 / q), s2 = u * s1, s3 = v * s1;
        double C[3][3];
        for (int k = 0; k < 3; k++) { C[0][k] = s1 * j[0][k]; C[1][k] = s2 * j[1][k]; C[2][k] = s3 * j[2][k]; }
        /* rigid alignment of the two triangles through orthonormal frames */
        double p1[3], p2[3], e1[3], e2[3], e3[3], f1[3], f2[3], f3[3], q1[3], q2[3];
        for (int k = 0; k < 3; k++) { p1[k] = P[1][k] - P[0][k]; p2[k] = P[2][k] - P[0][k]; q1[k] = C[1][k] - C[0][k]; q2[k] = C[2][k] - C[0][k]; }
        double n1 = norm3(p1), m1 = norm3(q1);
        if (n1 < 1e-12 || m1 < 1e-12) continue;
        for (int k = 0; k < 3; k++) { e1[k] = p1[k] / n1; f1[k] = q1[k] / m1; }
        cross3(e1, p2, e3); cross3(f1, q2, f3);
        double n3 = norm3(e3), m3 = norm3(f3);
        if (n3 < 1e-12 || m3 < 1e-12) continue;
        for (int k = 0; k < 3; k++) { e3[k] /= n3; f3[k] /= m3; }
        cross3(e3, e1, e2); cross3(f3, f1, f2);
        for (int rr = 0; rr < 3; rr++) for (int cc = 0; cc < 3; cc++) R[ns][rr * 3 + cc] = f1[rr] * e1[cc] + f2[rr] * e2[cc] + f3[rr] * e3[cc];
        for (int rr = 0; rr < 3; rr++) t[ns][rr] = C[0][rr] - (R[ns][rr * 3] * P[0][0] + R[ns][rr * 3 + 1] * P[0][1] + R[ns][rr * 3 + 2] * P[0][2]);
        ns++;
    }
    return ns;
}

static double reproj2(const double* R, const double* t, const float* X, const float* x, double fx, double fy, double cx, double cy)
{
    const double xc = R[0] * X[0] + R[1] * X[1] + R[2] * X[2] + t[0], yc = R[3] * X[0] + R[4] * X[1] + R[5] * X[2] + t[1], zc = R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + t[2];
    if (!(zc > 1e-9)) return 1e30;
    const double du = fx * xc / zc + cx - x[0], dv = fy * yc / zc + cy - x[1];
    return du * du + dv * dv;
}

/* one RANSAC hypothesis: returns inlier count (0 if no model), model in R,t */
int vo_pnp_hypothesis(const float* pts3d, const float* pts2d, int n, double fx, double fy, double cx, double cy, uint64_t seed, int it, double thr, double* R, double* t)
{
    int idx[4]; vo_pnp_sample4(seed, it, n, idx);
    double P[3][3], j[3][3];
    for (int k = 0; k < 3; k++) {
        for (int a = 0; a < 3; a++) P[k][a] = pts3d[3 * idx[k] + a];
        const double bx = (pts2d[2 * idx[k]] - cx) / fx, by = (pts2d[2 * idx[k] + 1] - cy) / fy, nn = sqrt(bx * bx + by * by + 1);
        j[k][0] = bx / nn; j[k][1] = by / nn; j[k][2] = 1 / nn;
    }
    double Rs[4][9], ts[4][3];
    const int ns = vo_p3p(P, j, Rs, ts);
    int best = -1; double be = 1e300;
    for (int s = 0; s < ns; s++) { const double e = reproj2(Rs[s], ts[s], pts3d + 3 * idx[3], pts2d + 2 * idx[3], fx, fy, cx, cy); if (e < be) { be = e; best = s; } }
    if (best < 0) return 0;
    memcpy(R, Rs[best], sizeof(double) * 9); memcpy(t, ts[best], sizeof(double) * 3);
    int cnt = 0; const double t2 = thr * thr;
    for (int i = 0; i < n; i++) if (reproj2(R, t, pts3d + 3 * i, pts2d + 2 * i, fx, fy, cx, cy) <= t2) cnt++;
    return cnt;
}

static int ransac_update_iters(double p, double ep, int model_points, int max_iters)
{
    p = fmax(fmin(p, 1.), 0.); ep = fmax(fmin(ep, 1.), 0.);
    double num = fmax(1. - p, DBL_MIN), denom = 1. - pow(1. - ep, model_points);
    if (denom < DBL_MIN) return 0;
    num = log(num); denom = log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)lrint(num / denom);
}

/* sequential RANSAC over the per-iteration hypotheses; returns inlier count, T (row-major 4x4), mask */
/* T: the pose cv::solvePnPRansac returns (refitted on the inliers); T_ransac (may be NULL): the winning minimal-sample model the inlier mask belongs to */
int vo_pnp_ransac_full(const float* pts3d, const float* pts2d, int n, double fx, double fy, double cx, double cy, int max_iters, double thr, double conf,
                       uint64_t seed, double* T, uint8_t* mask, double* T_ransac)
{
    for (int k = 0; k < 16; k++) T[k] = (k % 5 == 0);
    if (T_ransac) for (int k = 0; k < 16; k++) T_ransac[k] = (k % 5 == 0);
    if (mask) memset(mask, 0, n);
    if (n < 4) return 0;
    int niters = max_iters, best_cnt = 0; double bR[9], bt[3];
    for (int it = 0; it < niters; it++) {
        double R[9], t[3];
        const int cnt = vo_pnp_hypothesis(pts3d, pts2d, n, fx, fy, cx, cy, seed, it, thr, R, t);
        if (cnt > (best_cnt > 3 ? best_cnt : 3)) {
            best_cnt = cnt; memcpy(bR, R, sizeof bR); memcpy(bt, t, sizeof bt);
            niters = ransac_update_iters(conf, (double)(n - cnt) / n, 4, niters);
        }
    }
    if (best_cnt == 0) return 0;
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) T[r * 4 + c] = bR[r * 3 + c]; T[r * 4 + 3] = bt[r]; }
    int cnt = 0;
    uint8_t* in_mask = (uint8_t*)malloc((size_t)n);
    for (int i = 0; i < n; i++) { const int in = reproj2(bR, bt, pts3d + 3 * i, pts2d + 2 * i, fx, fy, cx, cy) <= thr * thr; in_mask[i] = (uint8_t)in; if (mask) mask[i] = (uint8_t)in; cnt += in; }
    if (T_ransac) memcpy(T_ransac, T, sizeof(double) * 16);
    /* cv::solvePnPRansac (OpenCV 3.4 calib3d/src/solvepnp.cpp): after RANSAC the pose is re-estimated from the inliers of the winning model with solvePnP (P3P -> EPNP, which ends
     * in Gauss-Newton on the reprojection error) and THAT pose is returned; the reported inliers stay those of the RANSAC model.  Restated as 8 Gauss-Newton steps on
     * sum |proj(R X + t) - x|^2 over the inliers from the winning model, update R <- exp(w) R, t <- exp(w) t + v; a failed solve keeps the RANSAC model. */
    {
        double R[9], t[3]; memcpy(R, bR, sizeof R); memcpy(t, bt, sizeof t); int ok = 1;
        for (int iter = 0; iter < 8 && ok; iter++) {
            double H[6][6], g[6]; memset(H, 0, sizeof H); memset(g, 0, sizeof g);
            for (int i = 0; i < n; i++) if (in_mask[i]) {
                const float* X = pts3d + 3 * i; const float* x = pts2d + 2 * i;
                const double xc = R[0] * X[0] + R[1] * X[1] + R[2] * X[2] + t[0], yc = R[3] * X[0] + R[4] * X[1] + R[5] * X[2] + t[1], zc = R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + t[2];
                const double iz = 1.0 / zc, ru = fx * xc * iz + cx - x[0], rv = fy * yc * iz + cy - x[1];
                const double au[3] = {fx * iz, 0.0, -fx * xc * iz * iz}, av[3] = {0.0, fy * iz, -fy * yc * iz * iz};
                const double ju[6] = {yc * au[2] - zc * au[1], zc * au[0] - xc * au[2], xc * au[1] - yc * au[0], au[0], au[1], au[2]};
                const double jv[6] = {yc * av[2] - zc * av[1], zc * av[0] - xc * av[2], xc * av[1] - yc * av[0], av[0], av[1], av[2]};
                for (int p = 0; p < 6; p++) { for (int c = 0; c < 6; c++) H[p][c] += ju[p] * ju[c] + jv[p] * jv[c]; g[p] += ju[p] * ru + jv[p] * rv; }
            }
            double L[6][6]; memcpy(L, H, sizeof L);
            for (int j = 0; j < 6 && ok; j++) {
                double s = L[j][j]; for (int k = 0; k < j; k++) s -= L[j][k] * L[j][k];
                if (!(s > 1e-300)) { ok = 0; break; }
                const double dj = sqrt(s); L[j][j] = dj;
                for (int i = j + 1; i < 6; i++) { double v = L[i][j]; for (int k = 0; k < j; k++) v -= L[i][k] * L[j][k]; L[i][j] = v / dj; }
            }
            if (!ok) break;
            double y[6], d[6];
            for (int i = 0; i < 6; i++) { double v = -g[i]; for (int k = 0; k < i; k++) v -= L[i][k] * y[k]; y[i] = v / L[i][i]; }
            for (int i = 5; i >= 0; i--) { double v = y[i]; for (int k = i + 1; k < 6; k++) v -= L[k][i] * d[k]; d[i] = v / L[i][i]; }
            const double th = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            double E[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
            if (th > 1e-12) {
                const double kx = d[0] / th, ky = d[1] / th, kz = d[2] / th, c = cos(th), s = sin(th), v = 1 - c;
                E[0] = c + kx * kx * v; E[1] = kx * ky * v - kz * s; E[2] = kx * kz * v + ky * s;
                E[3] = ky * kx * v + kz * s; E[4] = c + ky * ky * v; E[5] = ky * kz * v - kx * s;
                E[6] = kz * kx * v - ky * s; E[7] = kz * ky * v + kx * s; E[8] = c + kz * kz * v;
            }
            double Rn[9], tn[3];
            for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) Rn[r * 3 + c] = E[r * 3] * R[c] + E[r * 3 + 1] * R[3 + c] + E[r * 3 + 2] * R[6 + c];
                                          tn[r] = E[r * 3] * t[0] + E[r * 3 + 1] * t[1] + E[r * 3 + 2] * t[2] + d[3 + r]; }
            memcpy(R, Rn, sizeof R); memcpy(t, tn, sizeof t);
        }
        int fin = ok; for (int k = 0; k < 9; k++) fin = fin && isfinite(R[k]); for (int k = 0; k < 3; k++) fin = fin && isfinite(t[k]);
        if (fin) for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) T[r * 4 + c] = R[r * 3 + c]; T[r * 4 + 3] = t[r]; }
    }
    free(in_mask);
    return cnt;
}

int vo_pnp_ransac(const float* pts3d, const float* pts2d, int n, double fx, double fy, double cx, double cy, int max_iters, double thr, double conf,
                  uint64_t seed, double* T, uint8_t* mask)
{
    return vo_pnp_ransac_full(pts3d, pts2d, n, fx, fy, cx, cy, max_iters, thr, conf, seed, T, mask, NULL);
}
