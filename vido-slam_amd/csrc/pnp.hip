// pnp.hip — seeded P3P-RANSAC on gfx950: the initial-model stage of the tracker.
// Replaces the cv::solvePnPRansac(..., 500, 0.4, 0.98, SOLVEPNP_P3P) calls of Tracking::GetInitModelCam and
// Tracking::GetInitModelObj (reference vido_slam/src/Tracking.cc:1965-1970, :2068-2073).  OpenCV runs the
// hypotheses one after the other; here ALL max_iters hypotheses are evaluated at once — one workgroup per
// hypothesis: lane 0 solves Grunert's P3P quartic for the sampled triple (FP64, Durand-Kerner), the 4th sample picks
// among the up to four solutions, then the 256 threads count the inliers of the N correspondences with a DPP/LDS
// reduction — and a single-thread epilogue replays OpenCV's sequential bookkeeping (best-so-far, adaptive
// niters = log(1-conf)/log(1-(1-eps)^4)) over the per-iteration counts, so the outcome equals the sequential loop
// run with the same per-iteration samples.  Samples come from a counter-based generator (splitmix64 of seed and
// iteration index) instead of OpenCV's global RNG; OpenCV's final solvePnP over the inliers is the Gauss-Newton refit at the end of k_pnp_select.
#include "common.hpp"
#include <cfloat>

__host__ __device__ inline uint64_t splitmix64(uint64_t& s)
{
    uint64_t z = (s += 0x9E3779B97F4A7C15ULL); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31);
}
__device__ void sample4(uint64_t seed, int it, int n, int idx[4])
{
    uint64_t s = seed ^ (0xD1B54A32D192ED03ULL * (uint64_t)(it + 1));
    for (int k = 0; k < 4;) {
        const int c = (int)(splitmix64(s) % (uint64_t)n); bool dup = false;
        for (int j = 0; j < k; j++) if (idx[j] == c) dup = true;
        if (!dup) idx[k++] = c;
    }
}
// Real roots of c[4] z^4 + ... + c[0] by Durand-Kerner (Gauss-Seidel order) + two Newton steps; root k is valid when bit k of the returned mask is set (the roots keep
// their positions: every index below is a compile-time constant, so zr / zi / roots live in registers — with a compacting `roots[n++]` and rolled loops the arrays went to
// scratch memory and every access of the 100-iteration loop was a memory operation: 139 us per launch of k_pnp_solve)
__device__ int quartic_real_roots(const double* c, double* roots)
{
    if (fabs(c[4]) < 1e-300) return 0;
    const double a3 = c[3] / c[4], a2 = c[2] / c[4], a1 = c[1] / c[4], a0 = c[0] / c[4];
    double zr[4] = {1.0, 0.4, -0.65, -0.2755}, zi[4] = {0.0, 0.9, 0.72, -0.9602};
    for (int it = 0; it < 100; it++) {
        double maxd = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            double pr = 1, pi = 0, tr, ti;
            tr = pr * zr[k] - pi * zi[k] + a3; ti = pr * zi[k] + pi * zr[k]; pr = tr; pi = ti;
            tr = pr * zr[k] - pi * zi[k] + a2; ti = pr * zi[k] + pi * zr[k]; pr = tr; pi = ti;
            tr = pr * zr[k] - pi * zi[k] + a1; ti = pr * zi[k] + pi * zr[k]; pr = tr; pi = ti;
            tr = pr * zr[k] - pi * zi[k] + a0; ti = pr * zi[k] + pi * zr[k]; pr = tr; pi = ti;
            double qr = 1, qi = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) if (j != k) { const double dr = zr[k] - zr[j], di = zi[k] - zi[j]; tr = qr * dr - qi * di; ti = qr * di + qi * dr; qr = tr; qi = ti; }
            const double den = qr * qr + qi * qi;
            if (!(den < 1e-300)) {
                const double dr = (pr * qr + pi * qi) / den, di = (pi * qr - pr * qi) / den;
                zr[k] -= dr; zi[k] -= di;
                if (fabs(dr) + fabs(di) > maxd) maxd = fabs(dr) + fabs(di);
            }
        }
        if (maxd < 1e-14) break;
    }
    int mask = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        double v = zr[k];
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const double p = (((c[4] * v + c[3]) * v + c[2]) * v + c[1]) * v + c[0], dp = ((4 * c[4] * v + 3 * c[3]) * v + 2 * c[2]) * v + c[1];
            if (fabs(dp) > 1e-300) v -= p / dp;
        }
        roots[k] = v;
        if (fabs(zi[k]) < 1e-7 * (1.0 + fabs(zr[k]))) mask |= 1 << k;
    }
    return mask;
}
__device__ inline void cross3(const double* a, const double* b, double* c) { c[0] = a[1] * b[2] - a[2] * b[1]; c[1] = a[2] * b[0] - a[0] * b[2]; c[2] = a[0] * b[1] - a[1] * b[0]; }
__device__ inline double norm3(const double* a) { return sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }

// Grunert P3P: with s2 = u s1, s3 = v s1 the three cosine-law equations give u = N(v)/D(v) and the quartic
// D^2 + N^2 - 2 cos(gamma) N D - (c^2/b^2) Q D^2 = 0, Q = 1 + v^2 - 2 v cos(beta); poses from triangle alignment.
__device__ inline double reproj2(const double* R, const double* t, const float* X, const float* x, double fx, double fy, double cx, double cy)
{
    const double xc = R[0] * X[0] + R[1] * X[1] + R[2] * X[2] + t[0], yc = R[3] * X[0] + R[4] * X[1] + R[5] * X[2] + t[1], zc = R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + t[2];
    if (!(zc > 1e-9)) return 1e30;
    const double du = fx * xc / zc + cx - x[0], dv = fy * yc / zc + cy - x[1];
    return du * du + dv * dv;
}

// Grunert P3P + the choice among its solutions by a fourth correspondence (X4, x4): true and (R, t) = the solution with the smallest reprojection error of the fourth point
// (the first one on ties, in root order), false when the triple has no admissible solution.  One solution is alive at a time, everything has compile-time indices.
__device__ bool p3p_best(const double P[3][3], const double j[3][3], const float* X4, const float* x4, double fx, double fy, double cx, double cy, double Rb[9], double tb[3])
{
    double d[3];
    for (int k = 0; k < 3; k++) d[k] = P[1][k] - P[2][k]; const double a2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    for (int k = 0; k < 3; k++) d[k] = P[0][k] - P[2][k]; const double b2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    for (int k = 0; k < 3; k++) d[k] = P[0][k] - P[1][k]; const double c2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    if (a2 < 1e-20 || b2 < 1e-20 || c2 < 1e-20) return false;
    const double ca = j[1][0] * j[2][0] + j[1][1] * j[2][1] + j[1][2] * j[2][2];
    const double cb = j[0][0] * j[2][0] + j[0][1] * j[2][1] + j[0][2] * j[2][2];
    const double cg = j[0][0] * j[1][0] + j[0][1] * j[1][1] + j[0][2] * j[1][2];
    const double K = (a2 - c2) / b2, M = c2 / b2;
    const double N[3] = {1 + K, -2 * K * cb, K - 1}, D[2] = {2 * cg, -2 * ca}, Q[3] = {1, -2 * cb, 1};
    const double D2[3] = {D[0] * D[0], 2 * D[0] * D[1], D[1] * D[1]};
    double c[5] = {0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 3; i++) c[i] += D2[i];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int k = 0; k < 3; k++) c[i + k] += N[i] * N[k];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int k = 0; k < 2; k++) c[i + k] -= 2 * cg * N[i] * D[k];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int k = 0; k < 3; k++) c[i + k] -= M * Q[i] * D2[k];
    double roots[4]; const int mask = quartic_real_roots(c, roots);
    bool found = false; double be = 1e300;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        bool ok = (mask >> r) & 1;
        const double v = roots[r];
        ok = ok && v > 0;
        const double den = D[0] + D[1] * v;
        ok = ok && !(fabs(den) < 1e-12);
        const double u = (N[0] + N[1] * v + N[2] * v * v) / den;
        ok = ok && u > 0;
        const double q = 1 + v * v - 2 * v * cb;
        ok = ok && q > 0;
        if (!ok) continue;
        const double s1 = sqrt(b2 / q), s2 = u * s1, s3 = v * s1;
        double C0[3], p1[3], p2[3], e1[3], e2[3], e3[3], f1[3], f2[3], f3[3], q1[3], q2[3];
#pragma unroll
        for (int k = 0; k < 3; k++) { C0[k] = s1 * j[0][k]; const double C1 = s2 * j[1][k], C2 = s3 * j[2][k]; p1[k] = P[1][k] - P[0][k]; p2[k] = P[2][k] - P[0][k]; q1[k] = C1 - C0[k]; q2[k] = C2 - C0[k]; }
        const double n1 = norm3(p1), m1 = norm3(q1);
        if (n1 < 1e-12 || m1 < 1e-12) continue;
#pragma unroll
        for (int k = 0; k < 3; k++) { e1[k] = p1[k] / n1; f1[k] = q1[k] / m1; }
        cross3(e1, p2, e3); cross3(f1, q2, f3);
        const double n3 = norm3(e3), m3 = norm3(f3);
        if (n3 < 1e-12 || m3 < 1e-12) continue;
#pragma unroll
        for (int k = 0; k < 3; k++) { e3[k] /= n3; f3[k] /= m3; }
        cross3(e3, e1, e2); cross3(f3, f1, f2);
        double Rk[9], tk[3];
#pragma unroll
        for (int rr = 0; rr < 3; rr++)
#pragma unroll
            for (int cc = 0; cc < 3; cc++) Rk[rr * 3 + cc] = f1[rr] * e1[cc] + f2[rr] * e2[cc] + f3[rr] * e3[cc];
#pragma unroll
        for (int rr = 0; rr < 3; rr++) tk[rr] = C0[rr] - (Rk[rr * 3] * P[0][0] + Rk[rr * 3 + 1] * P[0][1] + Rk[rr * 3 + 2] * P[0][2]);
        const double e = reproj2(Rk, tk, X4, x4, fx, fy, cx, cy);
        if (e < be) {
            be = e; found = true;
#pragma unroll
            for (int k = 0; k < 9; k++) Rb[k] = Rk[k];
#pragma unroll
            for (int k = 0; k < 3; k++) tb[k] = tk[k];
        }
    }
    return found;
}
// problem p owns points [off, off + n) of the concatenated arrays
struct PnpProb { int off, n; unsigned long long seed; };
// Phase 1 (round 3: split off the scoring): ONE THREAD per (hypothesis, problem) samples its four points, solves Grunert's quartic and picks the solution the 4th point agrees
// with.  Round 2 ran this on lane 0 of a 256-thread workgroup per hypothesis — 3 000 workgroups per frame (camera + 5 objects x 500 hypotheses) whose other 255 lanes waited
// ~15 us for the FP64 root finder before they could score anything.
__global__ __launch_bounds__(64) void k_pnp_solve(const float* __restrict__ Xall, const float* __restrict__ xall, const PnpProb* __restrict__ prob, int max_iters,
                                                  double fx, double fy, double cx, double cy, double* __restrict__ models_all /*[prob][iters][12]*/, int* __restrict__ has_all)
{
    const int it = blockIdx.x * 64 + threadIdx.x; const PnpProb pr = prob[blockIdx.y];
    if (it >= max_iters) return;
    const int n = pr.n;
    double* m = models_all + ((size_t)blockIdx.y * max_iters + it) * 12; int* has = has_all + (size_t)blockIdx.y * max_iters + it;
    if (n < 4) { *has = 0; return; }
    const float* X = Xall + 3 * (size_t)pr.off; const float* x = xall + 2 * (size_t)pr.off;
    int idx[4]; sample4(pr.seed, it, n, idx);
    double P[3][3], j[3][3];
    for (int k = 0; k < 3; k++) {
        for (int a = 0; a < 3; a++) P[k][a] = X[3 * idx[k] + a];
        const double bx = (x[2 * idx[k]] - cx) / fx, by = (x[2 * idx[k] + 1] - cy) / fy, nn = sqrt(bx * bx + by * by + 1);
        j[k][0] = bx / nn; j[k][1] = by / nn; j[k][2] = 1 / nn;
    }
    double Rb[9], tb[3];
    const bool ok = p3p_best(P, j, X + 3 * idx[3], x + 2 * idx[3], fx, fy, cx, cy, Rb, tb);
    *has = ok;
#pragma unroll
    for (int k = 0; k < 9; k++) m[k] = ok ? Rb[k] : 0.0;
#pragma unroll
    for (int k = 0; k < 3; k++) m[9 + k] = ok ? tb[k] : 0.0;
}
// Phase 2: inlier counts.  A workgroup scores PNP_H hypotheses of one problem: every thread loads its points once and evaluates them against all PNP_H models (LDS).
#define PNP_H 4
__global__ __launch_bounds__(256) void k_pnp_score(const float* __restrict__ Xall, const float* __restrict__ xall, const PnpProb* __restrict__ prob, int max_iters,
                                                   double fx, double fy, double cx, double cy, double thr2, const double* __restrict__ models_all, const int* __restrict__ has_all,
                                                   int* __restrict__ counts_all)
{
    __shared__ double sm[PNP_H][12]; __shared__ int shas[PNP_H]; __shared__ int wcnt[4][PNP_H];
    const int it0 = blockIdx.x * PNP_H; const PnpProb pr = prob[blockIdx.y];
    const int n = pr.n;
    const float* X = Xall + 3 * (size_t)pr.off; const float* x = xall + 2 * (size_t)pr.off;
    int* counts = counts_all + (size_t)blockIdx.y * max_iters;
    if (threadIdx.x < PNP_H * 12) { const int h = threadIdx.x / 12, k = threadIdx.x - h * 12; sm[h][k] = it0 + h < max_iters ? models_all[((size_t)blockIdx.y * max_iters + it0 + h) * 12 + k] : 0.0; }
    if (threadIdx.x < PNP_H) shas[threadIdx.x] = it0 + threadIdx.x < max_iters ? has_all[(size_t)blockIdx.y * max_iters + it0 + threadIdx.x] : 0;
    __syncthreads();
    int cnt[PNP_H];
#pragma unroll
    for (int h = 0; h < PNP_H; h++) cnt[h] = 0;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float Xp[3] = {X[3 * i], X[3 * i + 1], X[3 * i + 2]}, xp[2] = {x[2 * i], x[2 * i + 1]};
#pragma unroll
        for (int h = 0; h < PNP_H; h++) if (shas[h]) cnt[h] += reproj2(sm[h], sm[h] + 9, Xp, xp, fx, fy, cx, cy) <= thr2;
    }
#pragma unroll
    for (int h = 0; h < PNP_H; h++) { int c = cnt[h]; for (int o = 32; o >= 1; o >>= 1) c += __shfl_xor(c, o, 64); if ((threadIdx.x & 63) == 0) wcnt[threadIdx.x >> 6][h] = c; }
    __syncthreads();
    if (threadIdx.x < PNP_H && it0 + threadIdx.x < max_iters) counts[it0 + threadIdx.x] = shas[threadIdx.x] ? wcnt[0][threadIdx.x] + wcnt[1][threadIdx.x] + wcnt[2][threadIdx.x] + wcnt[3][threadIdx.x] : 0;
}

__device__ int ransac_update_iters(double p, double ep, int model_points, int max_iters)
{
    p = fmax(fmin(p, 1.), 0.); ep = fmax(fmin(ep, 1.), 0.);
    double num = fmax(1. - p, DBL_MIN), denom = 1. - pow(1. - ep, (double)model_points);
    if (denom < DBL_MIN) return 0;
    num = log(num); denom = log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)llrint(num / denom);
}
// ---- all-inlier refit.  cv::solvePnPRansac does not return the winning minimal-sample model: after RANSAC it calls solvePnP on the inliers of that model
// (calib3d/src/solvepnp.cpp, OpenCV 3.4: compressElems(...mask...) + solvePnP(opoints_inliers, ipoints_inliers, ..., SOLVEPNP_P3P -> SOLVEPNP_EPNP)), returns THAT pose,
// and reports the inliers of the RANSAC model (the mask is not re-evaluated under the refitted pose).  EPnP ends in Gauss-Newton on the reprojection error; here the
// refit is REFIT_ITERS Gauss-Newton steps on the same least-squares problem (sum over the inliers of |proj(R X + t) - x|^2) started at the winning model, pose update
// R <- exp(w) R, t <- exp(w) t + v.  A failed 6x6 solve or a non-finite result keeps the RANSAC model (OpenCV keeps it when solvePnP fails).
#define REFIT_ITERS 8
__device__ inline void refit_rows(const double* R, const double* t, const float* X, const float* x, double fx, double fy, double cx, double cy, double ju[6], double jv[6], double& ru, double& rv)
{
    const double xc = R[0] * X[0] + R[1] * X[1] + R[2] * X[2] + t[0], yc = R[3] * X[0] + R[4] * X[1] + R[5] * X[2] + t[1], zc = R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + t[2];
    const double iz = 1.0 / zc;
    ru = fx * xc * iz + cx - x[0]; rv = fy * yc * iz + cy - x[1];
    const double au[3] = {fx * iz, 0.0, -fx * xc * iz * iz}, av[3] = {0.0, fy * iz, -fy * yc * iz * iz};      // d(u, v) / d(Xc)
    // Xc moves by w x Xc + v: d/dw = (Xc x a)^T, d/dv = a^T
    ju[0] = yc * au[2] - zc * au[1]; ju[1] = zc * au[0] - xc * au[2]; ju[2] = xc * au[1] - yc * au[0]; ju[3] = au[0]; ju[4] = au[1]; ju[5] = au[2];
    jv[0] = yc * av[2] - zc * av[1]; jv[1] = zc * av[0] - xc * av[2]; jv[2] = xc * av[1] - yc * av[0]; jv[3] = av[0]; jv[4] = av[1]; jv[5] = av[2];
}
// solves the symmetric 6x6 system H d = -g (H upper triangle packed row-major: 21 entries) by Cholesky; false if not positive definite
__device__ inline bool refit_solve(const double* Hp, const double* g, double d[6])
{
    double L[6][6];
    for (int i = 0, q = 0; i < 6; i++) for (int j = i; j < 6; j++, q++) { L[i][j] = Hp[q]; L[j][i] = Hp[q]; }
    for (int j = 0; j < 6; j++) {
        double s = L[j][j]; for (int k = 0; k < j; k++) s -= L[j][k] * L[j][k];
        if (!(s > 1e-300)) return false;
        const double dj = sqrt(s); L[j][j] = dj;
        for (int i = j + 1; i < 6; i++) { double v = L[i][j]; for (int k = 0; k < j; k++) v -= L[i][k] * L[j][k]; L[i][j] = v / dj; }
    }
    double y[6];
    for (int i = 0; i < 6; i++) { double v = -g[i]; for (int k = 0; k < i; k++) v -= L[i][k] * y[k]; y[i] = v / L[i][i]; }
    for (int i = 5; i >= 0; i--) { double v = y[i]; for (int k = i + 1; k < 6; k++) v -= L[k][i] * d[k]; d[i] = v / L[i][i]; }
    return true;
}
__device__ inline void refit_apply(double* R, double* t, const double* d)
{
    const double th = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    double E[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (th > 1e-12) {                                        // Rodrigues
        const double kx = d[0] / th, ky = d[1] / th, kz = d[2] / th, c = cos(th), s = sin(th), v = 1 - c;
        E[0] = c + kx * kx * v; E[1] = kx * ky * v - kz * s; E[2] = kx * kz * v + ky * s;
        E[3] = ky * kx * v + kz * s; E[4] = c + ky * ky * v; E[5] = ky * kz * v - kx * s;
        E[6] = kz * kx * v - ky * s; E[7] = kz * ky * v + kx * s; E[8] = c + kz * kz * v;
    }
    double Rn[9], tn[3];
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) Rn[r * 3 + c] = E[r * 3] * R[c] + E[r * 3 + 1] * R[3 + c] + E[r * 3 + 2] * R[6 + c];
                                  tn[r] = E[r * 3] * t[0] + E[r * 3 + 1] * t[1] + E[r * 3 + 2] * t[2] + d[3 + r]; }
    for (int k = 0; k < 9; k++) R[k] = Rn[k];
    for (int k = 0; k < 3; k++) t[k] = tn[k];
}

// sequential bookkeeping of cv::RANSAC over the precomputed hypotheses, the inlier mask of the winner, then the refit on those inliers; one workgroup per problem
__global__ __launch_bounds__(256) void k_pnp_select(const float* __restrict__ Xall, const float* __restrict__ xall, const PnpProb* __restrict__ prob, double fx, double fy, double cx, double cy,
                                                    int max_iters, double thr2, double conf, const double* __restrict__ models_all, const int* __restrict__ counts_all,
                                                    double* __restrict__ T_all /*[prob][16]*/, unsigned char* __restrict__ mask_all, int* __restrict__ n_inl_all)
{
    __shared__ int sbest;
    const PnpProb pr = prob[blockIdx.x]; const int n = pr.n;
    const float* X = Xall + 3 * (size_t)pr.off; const float* x = xall + 2 * (size_t)pr.off;
    const double* models = models_all + (size_t)blockIdx.x * max_iters * 12; const int* counts = counts_all + (size_t)blockIdx.x * max_iters;
    double* T_out = T_all + 16 * (size_t)blockIdx.x; unsigned char* mask = mask_all + pr.off; int* n_inl = n_inl_all + blockIdx.x;
    if (threadIdx.x == 0) {
        int niters = n >= 4 ? max_iters : 0, best_cnt = 0, best = -1;
        for (int it = 0; it < niters; it++) {
            const int cnt = counts[it];
            if (cnt > max(best_cnt, 3)) { best_cnt = cnt; best = it; niters = ransac_update_iters(conf, (double)(n - cnt) / n, 4, niters); }
        }
        sbest = best;
        for (int k = 0; k < 16; k++) T_out[k] = (k % 5 == 0) ? 1.0 : 0.0;
        if (best >= 0) { for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) T_out[r * 4 + c] = models[12 * best + r * 3 + c]; T_out[r * 4 + 3] = models[12 * best + 9 + r]; } }
        *n_inl = best >= 0 ? best_cnt : 0;
    }
    __syncthreads();
    const int best = sbest;
    for (int i = threadIdx.x; i < n; i += 256) mask[i] = best >= 0 ? (unsigned char)(reproj2(models + 12 * best, models + 12 * best + 9, X + 3 * i, x + 2 * i, fx, fy, cx, cy) <= thr2) : 0;
    if (best < 0) return;
    // ---- refit on the inliers (mask[] entries are re-read by the thread that wrote them: same i -> threadIdx mapping)
    __shared__ double sacc[4][27]; __shared__ double sRt[12]; __shared__ int sok;
    if (threadIdx.x < 12) sRt[threadIdx.x] = models[12 * best + threadIdx.x];
    if (threadIdx.x == 0) sok = 1;
    __syncthreads();
    for (int iter = 0; iter < REFIT_ITERS; iter++) {
        double R[9], t[3], a[27];
        for (int k = 0; k < 9; k++) R[k] = sRt[k];
        for (int k = 0; k < 3; k++) t[k] = sRt[9 + k];
        for (int k = 0; k < 27; k++) a[k] = 0.0;
        for (int i = threadIdx.x; i < n; i += 256) if (mask[i]) {
            double ju[6], jv[6], ru, rv; refit_rows(R, t, X + 3 * i, x + 2 * i, fx, fy, cx, cy, ju, jv, ru, rv);
            for (int p = 0, q = 0; p < 6; p++) { for (int c = p; c < 6; c++, q++) a[q] += ju[p] * ju[c] + jv[p] * jv[c]; a[21 + p] += ju[p] * ru + jv[p] * rv; }
        }
        for (int k = 0; k < 27; k++) { double v = a[k]; for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64); a[k] = v; }
        __syncthreads();                                      // (sRt has been read by everyone; sacc of the previous iteration consumed)
        if ((threadIdx.x & 63) == 0) for (int k = 0; k < 27; k++) sacc[threadIdx.x >> 6][k] = a[k];
        __syncthreads();
        if (threadIdx.x == 0 && sok) {
            double Hp[21], g[6], d[6];
            for (int k = 0; k < 21; k++) Hp[k] = sacc[0][k] + sacc[1][k] + sacc[2][k] + sacc[3][k];
            for (int k = 0; k < 6; k++) g[k] = sacc[0][21 + k] + sacc[1][21 + k] + sacc[2][21 + k] + sacc[3][21 + k];
            if (refit_solve(Hp, g, d)) { double Rn[9], tn[3]; for (int k = 0; k < 9; k++) Rn[k] = sRt[k]; for (int k = 0; k < 3; k++) tn[k] = sRt[9 + k];
                                         refit_apply(Rn, tn, d); for (int k = 0; k < 9; k++) sRt[k] = Rn[k]; for (int k = 0; k < 3; k++) sRt[9 + k] = tn[k]; }
            else sok = 0;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0 && sok) {
        bool fin = true; for (int k = 0; k < 12; k++) fin = fin && isfinite(sRt[k]);
        if (fin) for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) T_out[r * 4 + c] = sRt[r * 3 + c]; T_out[r * 4 + 3] = sRt[9 + r]; }
    }
}

struct PnpState { char* d = nullptr; char* h = nullptr; size_t cap = 0; };
void pnp_state_destroy(vido_ctx* ctx) { PnpState* S = ctx->pnp; if (!S) return; hipFree(S->d); hipHostFree(S->h); delete S; ctx->pnp = nullptr; }

extern "C" int vido_pnp_ransac_batch(vido_ctx* ctx, int n_prob, const float* const* pts3d, const float* const* pts2d, const int32_t* n, double fx, double fy, double cx, double cy,
                                     int max_iters, double reproj_err, double confidence, const uint64_t* seeds, double* T_out /*[n_prob][16]*/, uint8_t* const* inlier_mask,
                                     int32_t* n_inliers)
{
    if (!ctx) return VIDO_E_INVALID;
    if (n_prob < 0 || (n_prob && (!pts3d || !pts2d || !n || !seeds || !T_out || !n_inliers)) || max_iters < 1 || max_iters > 65536) return vido_set_error(ctx, VIDO_E_INVALID, "pnp_ransac_batch: bad arguments");
    if (n_prob == 0) return VIDO_OK;
    size_t tot = 0;
    for (int p = 0; p < n_prob; p++) { if (n[p] < 0 || (n[p] && (!pts3d[p] || !pts2d[p]))) return vido_set_error(ctx, VIDO_E_INVALID, "pnp_ransac_batch: problem %d: bad arguments", p); tot += (size_t)n[p]; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!ctx->pnp) ctx->pnp = new PnpState();
    PnpState* S = ctx->pnp; hipStream_t st = ctx->stream;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o_x3 = 0, o_x2 = o_x3 + al(tot * 12), o_pr = o_x2 + al(tot * 8), o_mod = o_pr + al((size_t)n_prob * sizeof(PnpProb)),
                 o_cnt = o_mod + al((size_t)n_prob * max_iters * 96), o_T = o_cnt + al((size_t)n_prob * max_iters * 4), o_n = o_T + al((size_t)n_prob * 128),
                 o_mask = o_n + al((size_t)n_prob * 4), o_has = o_mask + al(tot) + 256, total = o_has + al((size_t)n_prob * max_iters * 4) + 256;
    if (total > S->cap) {
        HIP_TRY(ctx, hipStreamSynchronize(st));
        if (S->d) { hipFree(S->d); hipHostFree(S->h); S->d = nullptr; S->h = nullptr; }
        S->cap = total + total / 2; HIP_TRY(ctx, hipMalloc((void**)&S->d, S->cap)); HIP_TRY(ctx, hipHostMalloc((void**)&S->h, S->cap));
    }
    PnpProb* hp = (PnpProb*)(S->h + o_pr); size_t off = 0;
    for (int p = 0; p < n_prob; p++) {
        if (n[p]) { memcpy(S->h + o_x3 + off * 12, pts3d[p], (size_t)n[p] * 12); memcpy(S->h + o_x2 + off * 8, pts2d[p], (size_t)n[p] * 8); }
        hp[p].off = (int)off; hp[p].n = n[p]; hp[p].seed = seeds[p]; off += (size_t)n[p];
    }
    HIP_TRY(ctx, hipMemcpyAsync(S->d, S->h, o_mod, hipMemcpyHostToDevice, st));
    const float* dX = (const float*)(S->d + o_x3); const float* dx = (const float*)(S->d + o_x2); const PnpProb* dp = (const PnpProb*)(S->d + o_pr);
    hipLaunchKernelGGL(k_pnp_solve, dim3((max_iters + 63) / 64, n_prob), dim3(64), 0, st, dX, dx, dp, max_iters, fx, fy, cx, cy, (double*)(S->d + o_mod), (int*)(S->d + o_has));
    hipLaunchKernelGGL(k_pnp_score, dim3((max_iters + PNP_H - 1) / PNP_H, n_prob), dim3(256), 0, st, dX, dx, dp, max_iters, fx, fy, cx, cy, reproj_err * reproj_err,
                       (const double*)(S->d + o_mod), (const int*)(S->d + o_has), (int*)(S->d + o_cnt));
    hipLaunchKernelGGL(k_pnp_select, dim3(n_prob), dim3(256), 0, st, dX, dx, dp, fx, fy, cx, cy, max_iters, reproj_err * reproj_err, confidence,
                       (const double*)(S->d + o_mod), (const int*)(S->d + o_cnt), (double*)(S->d + o_T), (unsigned char*)(S->d + o_mask), (int*)(S->d + o_n));
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(S->h + o_T, S->d + o_T, o_has - 256 - o_T, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    memcpy(T_out, S->h + o_T, (size_t)n_prob * 128);
    for (int p = 0; p < n_prob; p++) {
        n_inliers[p] = ((const int*)(S->h + o_n))[p];
        if (inlier_mask && inlier_mask[p] && n[p]) memcpy(inlier_mask[p], S->h + o_mask + hp[p].off, (size_t)n[p]);
    }
    return VIDO_OK;
}

extern "C" int vido_pnp_ransac(vido_ctx* ctx, const float* pts3d, const float* pts2d, int n, double fx, double fy, double cx, double cy,
                               int max_iters, double reproj_err, double confidence, uint64_t seed, double T_out[16], uint8_t* inlier_mask, int32_t* n_inliers)
{
    if (!ctx) return VIDO_E_INVALID;
    if (!T_out || !n_inliers || n < 0 || (n && (!pts3d || !pts2d)) || max_iters < 1 || max_iters > 65536) return vido_set_error(ctx, VIDO_E_INVALID, "pnp_ransac: bad arguments");
    for (int k = 0; k < 16; k++) T_out[k] = (k % 5 == 0) ? 1.0 : 0.0;
    *n_inliers = 0;
    if (inlier_mask && n) memset(inlier_mask, 0, n);
    if (n < 4) return VIDO_OK;
    const float* p3[1] = {pts3d}; const float* p2[1] = {pts2d}; uint8_t* mk[1] = {inlier_mask}; const int32_t nn[1] = {n}; const uint64_t sd[1] = {seed};
    return vido_pnp_ransac_batch(ctx, 1, p3, p2, nn, fx, fy, cx, cy, max_iters, reproj_err, confidence, sd, T_out, mk, n_inliers);
}
