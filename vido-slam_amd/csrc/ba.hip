// ba.hip — windowed / global bundle adjustment (static landmarks + odometry + prior) on gfx950.
// Replaces Optimizer::PartialBatchOptimization (reference vido_slam/src/Optimizer.cc:43-1228, STATIC_ONLY
// graph) and the static/odometry/prior part of Optimizer::FullBatchOptimization (:1235-2178), i.e. the g2o
// run they set up: LM (core/optimization_algorithm_levenberg.cpp:61-189) + gain stop
// (core/sparse_optimizer_terminate_action.cpp:49-85) over VertexSE3 / VertexPointXYZ with EdgeSE3PointXYZ
// (types/edge_se3_pointxyz.cpp:99-135), EdgeSE3 (types/edge_se3.cpp:77-104, isometry3d_gradients.h:85-189),
// EdgeSE3Prior (types/edge_se3_prior.cpp:89-102), Huber (core/robust_kernel_impl.cpp:65-91).
// g2o solves the un-eliminated pose+point system; here every LM trial eliminates the points by Schur
// complement (the same step algebraically) and factors only the reduced camera system.
//
// Kernels (all FP64, flat SoA in HBM):
//   k_ba_linearize  edge-parallel over observations sorted by camera: residual, 3x6 / 3x3 Jacobians, Huber;
//                   per-observation coupling block W (6x3) written once (144 B, contiguous, in landmark-major
//                   position); landmark sums by FP64 atomics (contention = track length); camera sums reduced
//                   inside the wave with DPP shuffles first (a wave's 64 observations share one camera).
//   k_ba_camfactors odometry / prior factors: 6x6 blocks of the camera-camera part.
//   k_ba_schur      wave-cooperative per landmark: D=(Hpp+lambda I)^-1, W D staged in LDS, the k(k+1)/2 6x6
//                   blocks W_i D W_j^T spread over the 64 lanes; accumulation into the reduced system either in
//                   an LDS-resident copy of S (<=22 cameras: the local-BA window) flushed once per workgroup,
//                   or with FP64 atomics into the dense S in HBM (global BA).
//   k_ba_chol_small single-workgroup in-LDS Cholesky solve of the reduced system (local BA);
//   k_chol_*        blocked right-looking Cholesky in HBM for the global reduced system.
//   k_ba_update_cams / k_ba_backsub / k_ba_chi2   trial state, back-substitution, robust chi2.
// Multi-GPU: landmarks are sharded by contiguous id range; every rank linearises its own observations, the
// partial reduced system (S, r) and the LM scalars are summed through the caller's all-reduce hook (RCCL via
// torch.distributed in bench.py), the reduced solve is replicated, back-substitution is local.
#include "common.hpp"
#include <atomic>
#include <sched.h>
#include "xwg.hpp"
#include <thread>
#include <mutex>
#include <condition_variable>
#include <functional>
#include <cfloat>
#include <chrono>
#include <cmath>

#define BA_LDS_MAX_N6 120          // reduced systems up to 20 cameras (the reference WINDOW_SIZE) accumulate / factor in LDS

// One slot record per observation, landmark-major: W (18 doubles) and Cp (9) side by side in 32 doubles = 256 bytes = two whole cache lines.  Round 2 kept two dense arrays
// (144- and 72-byte records): a record then shared its first and last line with its neighbours — other landmarks, written by other waves at other times — so lines left
// the L2 partially written (WRITE_SIZE 254 MB for 216 MB of stores) and the memory side had to merge them.  With line-aligned records every line is written by ONE lane.
#define BA_REC 32
struct BaDev {
    int n_cam, n_pt, n_obs, n_odo, prior_cam, use_huber, n6, pt_lo;
    double info_obs, info_odo, info_prior, huber_obs, huber_odo;
    double *cam, *cam_new, *pt, *pt_new;                    // [n_cam*12], [n_ptl*3]  (points: local shard indices)
    const int *obs_cam, *obs_pt, *obs_pos; const double* obs_meas;   // sorted by camera; obs_pos = landmark-major slot
    const int *pt_start;                                    // [n_ptl+1] CSR over landmark-major slots
    const int *slot_cam;                                    // [n_obs] camera of each landmark-major slot
    const int *odo_i, *odo_j; const double* odo_T; double prior_T[12];
    double *W;                                              // [n_obs*BA_REC] landmark-major slot RECORDS of BA_REC doubles (256 bytes, line-aligned): [0,18) W (6x3), [18,27) Cp = point-side terms (Hpp 6 | bp 3), rest padding
    double *Hpp, *bp;                                       // [n_ptl*6], [n_ptl*3]: summed from Cp by k_ba_schur (no atomics)
    double *Cp;                                             // = W + 18: the Cp part of slot s is Cp[BA_REC * s + a] (kept as a member for readability)
    double *Hcd, *bc, *Hodo;                                // [n_cam*36], [n6], [n_odo*36]
    double *S, *r, *x;                                      // [n6*n6], [n6], [n6]
    double *scal;                                           // [8]: 0 chi2 1 maxdiag 2 tempChi 3 scale 4 ok
    const double *odo_info, *odo_delta;                     // per camera-camera factor (odometry | object-motion smoothness)
    int n_cam_ord; const int *pose_ord, *ord_pose, *slot_ord;   // slot_ord: camera ordinal of every landmark-major slot                         // camera ordinal of a pose index (-1: object motion) and back: the Schur window counts CAMERAS, not poses
    int bw, ldb;                                            // bw >= 0: S is stored as a lower BAND, entry (r, c) at S[r*ldb + c - r + bw]; bw < 0: dense n6 x n6
    // ---- object part (FullBatchOptimization, STATIC_ONLY = false).  n_cam above counts ALL pose vertices: cameras first,
    // then the object motions H.  Dynamic points are stored chain-major (a chain = one dynamic tracklet).
    int n_dyn, n_chain;
    double info_dyn, info_tern, huber_dyn, huber_tern;
    double *dyn, *dyn_new;                                  // [n_dyn*3]
    const int *dyn_cam, *dyn_hin;                           // camera; pose index of the H of the ternary edge (k-1, k, H), -1 = chain head
    const double* dyn_meas; const int* chain_start;         // [n_dyn*3], [n_chain+1]
    double *Vd, *bd, *U, *Wc, *Wi, *Wo, *fac, *yb;          // [6],[3],[9],[18],[18],[18],[18: Dinv|G],[3] per dynamic point
};

// ---- small math ------------------------------------------------------------------------------------
__device__ __forceinline__ void huber_w(double e2, double delta, int use, double& r0, double& r1)
{
    if (!use || e2 <= delta * delta) { r0 = e2; r1 = 1.0; return; }
    const double s = sqrt(e2); r0 = 2 * s * delta - delta * delta; r1 = delta / s;
}
// sum over a workgroup, ONE atomic per workgroup: FP64 atomics onto one address retire at ~45 ns each whatever the grid looks like
// (a million-edge chi2 with one atomic per wave spent 175 of its 190 us queueing on them)
__device__ __forceinline__ void block_atomic_add(double* out, double v)
{
    __shared__ double wsum[16];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) { double t = 0; for (int w = 0; w < (int)(blockDim.x >> 6); w++) t += wsum[w]; atomicAdd(out, t); }
}
// address of entry (row, col) of the reduced system, or nullptr when the band layout does not store it (upper triangle)
__device__ __forceinline__ double* s_entry(const BaDev& P, int row, int col)
{
    if (P.bw < 0) return P.S + (size_t)row * P.n6 + col;
    if (col > row || row - col > P.bw) return nullptr;
    return P.S + (size_t)row * P.ldb + (col - row + P.bw);
}
__device__ __forceinline__ void iso_inv_mul(const double* A, const double* B, double* C)
{
#pragma unroll
    for (int r = 0; r < 3; r++) {
#pragma unroll
        for (int c = 0; c < 3; c++) C[r * 4 + c] = A[r] * B[c] + A[4 + r] * B[4 + c] + A[8 + r] * B[8 + c];
        C[r * 4 + 3] = A[r] * (B[3] - A[3]) + A[4 + r] * (B[7] - A[7]) + A[8 + r] * (B[11] - A[11]);
    }
}
__device__ void rot_to_quat(const double* M, double* q)      // Eigen Quaternion(R) + g2o normalize (w >= 0)
{
    const double mm[3][3] = {{M[0], M[1], M[2]}, {M[4], M[5], M[6]}, {M[8], M[9], M[10]}};
    double t = mm[0][0] + mm[1][1] + mm[2][2];
    if (t > 0) {
        t = sqrt(t + 1.0); q[3] = 0.5 * t; t = 0.5 / t;
        q[0] = (mm[2][1] - mm[1][2]) * t; q[1] = (mm[0][2] - mm[2][0]) * t; q[2] = (mm[1][0] - mm[0][1]) * t;
    } else {
        int i = 0; if (mm[1][1] > mm[0][0]) i = 1; if (mm[2][2] > mm[i][i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrt(mm[i][i] - mm[j][j] - mm[k][k] + 1.0);
        double qi = 0.5 * t; t = 0.5 / t;
        const double qw = (mm[k][j] - mm[j][k]) * t, qj = (mm[j][i] + mm[i][j]) * t, qk = (mm[k][i] + mm[i][k]) * t;
        q[3] = qw;
        q[0] = i == 0 ? qi : (j == 0 ? qj : qk); q[1] = i == 1 ? qi : (j == 1 ? qj : qk); q[2] = i == 2 ? qi : (j == 2 ? qj : qk);
    }
    const double nrm = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
#pragma unroll
    for (int a = 0; a < 4; a++) q[a] /= nrm;
    if (q[3] < 0) {
#pragma unroll
        for (int a = 0; a < 4; a++) q[a] = -q[a];
    }
}
// EdgeSE3 residual: E = Z^-1 Xi^-1 Xj, e = (t_E, q_E.xyz), and its exact Jacobians w.r.t. the right-multiplicative
// (dt, v) increments of VertexSE3::oplusImpl (R(v) ~ I + 2[v]x):
//   Jj = [ R_E 0 ; 0 Q ],  Ji = [ -R_A  2 R_A [t_B]x ; 0  -Q R_B^T ],  Q = w_E I + [q_E.xyz]x,  A = Z^-1, B = Xi^-1 Xj
// (g2o reaches the same matrices through dq/dR, isometry3d_gradients.h:85-189).  Xi == nullptr: EdgeSE3Prior.
__device__ void edge_se3(const double* Z, const double* Xi, const double* Xj, double* e, double* Ji, double* Jj, bool jac)
{
    double B[12], E[12], q[4];
    if (Xi) iso_inv_mul(Xi, Xj, B); else { for (int a = 0; a < 12; a++) B[a] = Xj[a]; }
    iso_inv_mul(Z, B, E);
    rot_to_quat(E, q);
    e[0] = E[3]; e[1] = E[7]; e[2] = E[11]; e[3] = q[0]; e[4] = q[1]; e[5] = q[2];
    if (!jac) return;
    const double Q[9] = {q[3], -q[2], q[1], q[2], q[3], -q[0], -q[1], q[0], q[3]};
    for (int a = 0; a < 36; a++) { Jj[a] = 0; if (Ji) Ji[a] = 0; }
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { Jj[r * 6 + c] = E[r * 4 + c]; Jj[(3 + r) * 6 + 3 + c] = Q[r * 3 + c]; }
    if (Ji) {
        const double S[9] = {0, -2 * B[11], 2 * B[7], 2 * B[11], 0, -2 * B[3], -2 * B[7], 2 * B[3], 0};
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) {
            Ji[r * 6 + c] = -Z[c * 4 + r];
            double s = 0, qq = 0;
            for (int k = 0; k < 3; k++) { s += Z[k * 4 + r] * S[k * 3 + c]; qq += Q[r * 3 + k] * B[c * 4 + k]; }
            Ji[r * 6 + 3 + c] = s; Ji[(3 + r) * 6 + 3 + c] = -qq;
        }
    }
}
__device__ __forceinline__ void inv3sym(const double* H6 /*00 01 02 11 12 22*/, double lambda, double* I)
{
    const double a = H6[0] + lambda, b = H6[1], c = H6[2], d = H6[3] + lambda, e = H6[4], f = H6[5] + lambda;
    const double c0 = d * f - e * e, c1 = e * c - b * f, c2 = b * e - d * c;
    const double det = 1.0 / (a * c0 + b * c1 + c * c2);
    I[0] = c0 * det; I[1] = c1 * det; I[2] = c2 * det;
    I[3] = I[1]; I[4] = (a * f - c * c) * det; I[5] = (b * c - a * e) * det;
    I[6] = I[2]; I[7] = I[5]; I[8] = (a * d - b * b) * det;
}

// ---- linearisation -----------------------------------------------------------------------------------
// FP64 atomics that land on one cache line serialise in L2 (~45 ns each), so the kernel is built to issue almost none of them:
//  * point side: the 9 terms of an observation go to its landmark-major slot (Cp, next to W) with plain stores; k_ba_schur, which walks a
//    landmark's slots anyway, sums them (a 17-observation track was 17 atomics per address before).
//  * camera side: observations are sorted by camera.  Every wave walks E consecutive groups of 64 observations and keeps the 28 sums in
//    registers across them; a group that straddles two cameras is folded camera by camera with lane masks (64 lanes issuing atomics onto the
//    same 43 addresses made one such wave the 30 us long pole of a 35k-edge launch).  On a camera change the wave sum goes to one of
//    LIN_SLOTS LDS accumulators of the workgroup (8 waves = 512 observations rarely see more than two cameras); the workgroup issues one
//    set of 43 atomics per camera it saw.
#define LIN_SLOTS 4
#define LIN_THREADS 512
#define LIN_RECP 30         // doubles per staged record in LDS (27 used; 240 bytes keeps every 16-byte piece aligned)
// Transposing wave reduction: N per-lane values -> every lane ends up holding the wave sum of ONE of them.  Each step halves the number of
// values a lane carries by trading the half it does not keep with lane ^ O: 29 double shuffles for 28 values instead of 168 for 28 butterflies
// (ds_bpermute goes through the CU's LDS pipe, which the 8 waves of the workgroup share — the butterflies were 40 % of the kernel).
template <int N, int O>
__device__ __forceinline__ void wave_halve(double* v, int lane)
{
    constexpr int H = (N + 1) / 2;
    const bool up = (lane & O) != 0;
#pragma unroll
    for (int a = 0; a < H; a++) {
        const double lo = v[a], hi = (a + H < N) ? v[a + H] : 0.0;
        const double send = up ? lo : hi, keep = up ? hi : lo;
        v[a] = keep + __shfl_xor(send, O, 64);
    }
}
// camera-side sums of a wave -> LDS accumulator of the workgroup (or HBM atomics when the camera is outside the workgroup's slots)
__device__ __forceinline__ void ba_flush_cam(const BaDev& P, int c, int cam0, double* lsum, int* lused, double* acc, int lane)
{
    wave_halve<28, 32>(acc, lane); wave_halve<14, 16>(acc, lane); wave_halve<7, 8>(acc, lane); wave_halve<4, 4>(acc, lane); wave_halve<2, 2>(acc, lane);
    const double sum = acc[0] + __shfl_xor(acc[0], 1, 64);
    const int sub = ((lane >> 3) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1);
    const int v = ((lane >> 5) & 1) * 14 + ((lane >> 4) & 1) * 7 + sub;          // which of the 28 sums this lane holds
    if (!(lane & 1) && sub < 7) {
        const int slot = c - cam0;
        if (slot >= 0 && slot < LIN_SLOTS) { if (v == 0) lused[slot] = 1; atomicAdd(lsum + slot * 28 + v, sum); }       // ds_add_f64
        else if (v < 21) { int a = 0, t = v; while (t >= 6 - a) { t -= 6 - a; a++; } const int b = a + t; atomicAdd(P.Hcd + 36 * c + a * 6 + b, sum); if (b != a) atomicAdd(P.Hcd + 36 * c + b * 6 + a, sum); }
        else if (v < 27) atomicAdd(P.bc + 6 * c + (v - 21), sum);
        else atomicAdd(P.scal + 0, sum);
    }
#pragma unroll
    for (int a = 0; a < 28; a++) acc[a] = 0;
}
// Body of the linearisation for virtual block `vb` (NT threads = NT / 64 groups of 64 consecutive observations x E).  k_ba_linearize runs it once per workgroup; the
// persistent local-window solver (k_ba_local_lm) loops it over the blocks of its grid.  smem: LIN_SMEM_DOUBLES(NT) doubles, 16-byte aligned.  MEAS: the observation's
// measurement rides along in the slot record's padding ([27, 30)) so that the trial chi2 can be formed landmark-major next to the back-substitution.
#define LIN_SMEM_DOUBLES(NT) (LIN_SLOTS * 28 + ((NT) / 64) * 32 * LIN_RECP + (LIN_SLOTS + ((NT) / 64) * 32 + 1) / 2 + 2)
template <int NT, bool MEAS>
__device__ __forceinline__ void ba_linearize_body(const BaDev& P, int E, int vb, double* smem)
{
    double* lsum = smem;                                                                       // [LIN_SLOTS * 28]
    double (*lin_stage)[32 * LIN_RECP] = (double (*)[32 * LIN_RECP])(smem + LIN_SLOTS * 28);     // [NT / 64][32 * LIN_RECP]   (LIN_SLOTS * 28 doubles = 896 bytes: 16-byte aligned)
    int* lused = (int*)(smem + LIN_SLOTS * 28 + (NT / 64) * 32 * LIN_RECP);                      // [LIN_SLOTS]
    int (*lin_sslot)[32] = (int (*)[32])(lused + LIN_SLOTS);                                     // [NT / 64][32]
    constexpr int NREC = MEAS ? 30 : 28, NPIECE = MEAS ? 15 : 14;
    const int lane = threadIdx.x & 63;
    // (round 3: giving every XCD one CONTIGUOUS eighth of the camera-sorted list — so that the partial-line writes of a landmark's W / Cp slots meet in one L2 — was
    // measured and changed nothing: 109-112 us per 1 M edges either way, profiles/r3/README.md; the plain order stays)
    const size_t wave_g = (size_t)vb * (NT / 64) + (threadIdx.x >> 6);
    if ((size_t)vb * (NT / 64) * E * 64 >= (size_t)P.n_obs) return;      // (whole workgroup: the rounded-up tail)
    if (threadIdx.x < LIN_SLOTS * 28) lsum[threadIdx.x] = 0;
    if (threadIdx.x < LIN_SLOTS) lused[threadIdx.x] = 0;
    const int cam0 = P.obs_cam[min((size_t)vb * (NT / 64) * E * 64, (size_t)P.n_obs - 1)];     // first camera of the workgroup
    __syncthreads();
    double acc[28];
#pragma unroll
    for (int a = 0; a < 28; a++) acc[a] = 0;
    int acc_cam = -1;                                            // camera the register sums belong to (wave-uniform)
    for (int j = 0; j <= E; j++) {                               // j == E: sentinel pass that flushes the last camera
        const size_t k = (wave_g * E + j) * 64 + lane;
        const bool act = j < E && k < (size_t)P.n_obs;
        int c = -1, slot_k = -1;
        double con[28], rec[NREC];
#pragma unroll
        for (int a = 0; a < 28; a++) con[a] = 0;
#pragma unroll
        for (int a = 0; a < NREC; a++) rec[a] = 0;
        if (act) {
            c = P.obs_cam[k]; const int l = P.obs_pt[k];
            const double* X = P.cam + 12 * c; const double* p = P.pt + 3 * l; const double* m = P.obs_meas + 3 * (size_t)k;
            const double R00 = X[0], R01 = X[1], R02 = X[2], R10 = X[4], R11 = X[5], R12 = X[6], R20 = X[8], R21 = X[9], R22 = X[10];
            const double d0 = p[0] - X[3], d1 = p[1] - X[7], d2 = p[2] - X[11];
            const double Z0 = R00 * d0 + R10 * d1 + R20 * d2, Z1 = R01 * d0 + R11 * d1 + R21 * d2, Z2 = R02 * d0 + R12 * d1 + R22 * d2;
            const double e0 = Z0 - m[0], e1 = Z1 - m[1], e2 = Z2 - m[2];
            double r0, w; huber_w(P.info_obs * (e0 * e0 + e1 * e1 + e2 * e2), P.huber_obs, P.use_huber, r0, w);
            const double wo = w * P.info_obs;
            // Jc = [-I | 2[Zc]x] (3x6), Jp = R^T
            const double Jc[18] = {-1, 0, 0, 0, -2 * Z2, 2 * Z1,   0, -1, 0, 2 * Z2, 0, -2 * Z0,   0, 0, -1, -2 * Z1, 2 * Z0, 0};
            const double Jp[9] = {R00, R10, R20, R01, R11, R21, R02, R12, R22};
            const double e[3] = {e0, e1, e2};
            int q = 0;
#pragma unroll
            for (int a = 0; a < 6; a++) {
                con[21 + a] = -wo * (Jc[a] * e[0] + Jc[6 + a] * e[1] + Jc[12 + a] * e[2]);
#pragma unroll
                for (int b = a; b < 6; b++) con[q++] = wo * (Jc[a] * Jc[b] + Jc[6 + a] * Jc[6 + b] + Jc[12 + a] * Jc[12 + b]);
            }
            con[27] = r0;
            slot_k = P.obs_pos[k];
#pragma unroll
            for (int a = 0; a < 6; a++)
#pragma unroll
                for (int b = 0; b < 3; b++) rec[a * 3 + b] = wo * (Jc[a] * Jp[b] + Jc[6 + a] * Jp[3 + b] + Jc[12 + a] * Jp[6 + b]);
            q = 18;
#pragma unroll
            for (int a = 0; a < 3; a++) {
                rec[24 + a] = -wo * (Jp[a] * e[0] + Jp[3 + a] * e[1] + Jp[6 + a] * e[2]);
#pragma unroll
                for (int b = a; b < 3; b++) rec[q++] = wo * (Jp[a] * Jp[b] + Jp[3 + a] * Jp[3 + b] + Jp[6 + a] * Jp[6 + b]);
            }
            if constexpr (MEAS) { rec[27] = m[0]; rec[28] = m[1]; rec[29] = m[2]; }
        }
        // ---- the slot records leave through LDS: a lane owns ONE record (W 18 | Cp 9 doubles) that goes to a scattered, line-aligned 256-byte slot.  Stored from the lane's
        // own registers that is 14 instructions of 64 sixteen-byte writes to 64 different lines — 14 M partial-line transactions per million edges, and the transaction rate
        // (not the bytes) was what bounded the kernel.  Staged, 16 consecutive lanes write the 14 sixteen-byte pieces of ONE record: every instruction covers four records
        // with 224 contiguous bytes each, 2 M full-line writes in all.  Half a wave at a time (32 records of 30 doubles: 61 KB of LDS for the 8 waves).
        if (j < E) {
            double* stg = lin_stage[threadIdx.x >> 6]; int* ssl = lin_sslot[threadIdx.x >> 6];
#pragma unroll
            for (int half = 0; half < 2; half++) {
                if ((lane >> 5) == half) {
                    ssl[lane & 31] = act ? slot_k : -1;
                    if (act) {
#pragma unroll
                        for (int a = 0; a < NPIECE; a++) *(double2*)(stg + (lane & 31) * LIN_RECP + 2 * a) = make_double2(rec[2 * a], rec[2 * a + 1]);
                    }
                }
                __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_wave_barrier();      // lgkmcnt(0): the LDS writes of this wave have landed
#pragma unroll
                for (int it = 0; it < 8; it++) {
                    const int r = it * 4 + (lane >> 4), pc = lane & 15, sl = ssl[r];
#if defined(LIN_FULL256) || defined(LIN_NT)
                    // experiments (round 6): the whole 256-byte record (pad included: no partially written line) and / or non-temporal stores
#ifdef LIN_FULL256
                    if (sl >= 0) {
                        const double2 v = pc < NPIECE ? *(const double2*)(stg + r * LIN_RECP + 2 * pc) : make_double2(0.0, 0.0);
#else
                    if (sl >= 0 && pc < NPIECE) {
                        const double2 v = *(const double2*)(stg + r * LIN_RECP + 2 * pc);
#endif
                        double* dst = P.W + BA_REC * (size_t)sl + 2 * pc;
#ifdef LIN_NT
                        __builtin_nontemporal_store(v.x, dst); __builtin_nontemporal_store(v.y, dst + 1);
#else
                        *(double2*)dst = v;
#endif
                    }
#else
                    if (sl >= 0 && pc < NPIECE) *(double2*)(P.W + BA_REC * (size_t)sl + 2 * pc) = *(const double2*)(stg + r * LIN_RECP + 2 * pc);
#endif
                }
                __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_wave_barrier();
            }
        }
        // fold the group into the register sums, one camera at a time (cameras are non-decreasing along the lanes)
        unsigned long long rem = j < E ? __ballot(act) : 1ull;
        while (rem) {
            const int cc = j < E ? __shfl(c, __ffsll((long long)rem) - 1, 64) : -1;
            if (acc_cam != cc) { if (acc_cam >= 0) ba_flush_cam(P, acc_cam, cam0, lsum, lused, acc, lane); acc_cam = cc; }
            if (j == E) break;
            const bool sel = act && c == cc;
#pragma unroll
            for (int a = 0; a < 28; a++) acc[a] += sel ? con[a] : 0.0;
            rem &= ~__ballot(sel);
        }
    }
    __syncthreads();
    // one thread per (slot, entry of the 6x6 block | bc | chi2): 43 atomics per camera the workgroup saw
    for (int t = threadIdx.x; t < LIN_SLOTS * 43; t += NT) {
        const int slot = t / 43, e = t - slot * 43, c = cam0 + slot;
        if (!lused[slot]) continue;
        if (e < 36) { const int a = e / 6, b = e - a * 6, lo = min(a, b), hi = max(a, b); atomicAdd(P.Hcd + 36 * c + e, lsum[slot * 28 + lo * 6 - lo * (lo - 1) / 2 + (hi - lo)]); }
        else if (e < 42) atomicAdd(P.bc + 6 * c + (e - 36), lsum[slot * 28 + 21 + (e - 36)]);
        else atomicAdd(P.scal + 0, lsum[slot * 28 + 27]);
    }
}
__global__ __launch_bounds__(LIN_THREADS) void k_ba_linearize(BaDev P, int E)
{
    __shared__ __attribute__((aligned(16))) double lin_smem[LIN_SMEM_DOUBLES(LIN_THREADS)];
    ba_linearize_body<LIN_THREADS, false>(P, E, blockIdx.x, lin_smem);
}

// odometry edges (k < n_odo) and the prior (k == n_odo): one wave per factor, lane a*6+b owns entry (a,b) of the
// 6x6 blocks (the residual and Jacobians are cheap and recomputed by every lane)
__device__ void ba_camfactor_body(const BaDev& P, int with_jac, const double* cam, double* chi_out, int k, int lane)
{
    const int n = P.n_odo + (P.prior_cam >= 0 ? 1 : 0);
    if (k >= n) return;
    const bool is_prior = k == P.n_odo;
    const int i = is_prior ? -1 : P.odo_i[k], j = is_prior ? P.prior_cam : P.odo_j[k];
    double e[6], Ji[36], Jj[36];
    edge_se3(is_prior ? P.prior_T : P.odo_T + 12 * k, is_prior ? nullptr : cam + 12 * i, cam + 12 * j, e, is_prior ? nullptr : Ji, Jj, with_jac != 0);
    double s2 = 0; for (int a = 0; a < 6; a++) s2 += e[a] * e[a];
    const double info = is_prior ? P.info_prior : P.odo_info[k];
    double r0 = info * s2, w = 1;
    if (!is_prior) huber_w(info * s2, P.odo_delta[k], P.use_huber, r0, w);
    if (lane == 0) atomicAdd(chi_out, r0);
    if (!with_jac || lane >= 36) return;
    const double wo = w * info;
    const int a = lane / 6, b = lane - a * 6;
    double hjj = 0, hii = 0, hij = 0, sj = 0, si = 0;
    for (int r = 0; r < 6; r++) {
        hjj += Jj[r * 6 + a] * Jj[r * 6 + b];
        if (!is_prior) { hii += Ji[r * 6 + a] * Ji[r * 6 + b]; hij += Ji[r * 6 + a] * Jj[r * 6 + b]; }
        if (b == 0) { sj += Jj[r * 6 + a] * e[r]; if (!is_prior) si += Ji[r * 6 + a] * e[r]; }
    }
    atomicAdd(P.Hcd + 36 * j + a * 6 + b, wo * hjj);
    if (!is_prior) { atomicAdd(P.Hcd + 36 * i + a * 6 + b, wo * hii); P.Hodo[36 * k + a * 6 + b] = wo * hij; }
    if (b == 0) { atomicAdd(P.bc + 6 * j + a, -wo * sj); if (!is_prior) atomicAdd(P.bc + 6 * i + a, -wo * si); }
}
__global__ __launch_bounds__(64) void k_ba_camfactors(BaDev P, int with_jac, const double* cam, double* chi_out) { ba_camfactor_body(P, with_jac, cam, chi_out, blockIdx.x, threadIdx.x); }

// max |diag| over camera blocks and landmark blocks (computeLambdaInit)
__device__ __forceinline__ void ba_maxdiag_body(const BaDev& P, int n_ptl, int tid, int nt)
{
    double m = 0;
    for (int a = tid; a < P.n6; a += nt) m = fmax(m, fabs(P.Hcd[36 * (a / 6) + 7 * (a % 6)]));
    for (int l = tid; l < n_ptl; l += nt) {          // runs before k_ba_schur has summed Hpp: diagonal of the landmark block from its slots
        double h0 = 0, h3 = 0, h5 = 0;
        for (int i = P.pt_start[l]; i < P.pt_start[l + 1]; i++) { h0 += P.Cp[BA_REC * (size_t)i]; h3 += P.Cp[BA_REC * (size_t)i + 3]; h5 += P.Cp[BA_REC * (size_t)i + 5]; }
        m = fmax(m, fmax(fabs(h0), fmax(fabs(h3), fabs(h5))));
    }
    if (n_ptl >= 0) for (int l = tid; l < P.n_dyn; l += nt) m = fmax(m, fmax(fabs(P.Vd[6 * (size_t)l]), fmax(fabs(P.Vd[6 * (size_t)l + 3]), fabs(P.Vd[6 * (size_t)l + 5]))));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmax(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) {      // non-negative doubles order like their bit patterns
        atomicMax((unsigned long long*)(P.scal + 1), (unsigned long long)__double_as_longlong(m));
    }
}
__global__ __launch_bounds__(256) void k_ba_maxdiag(BaDev P, int n_ptl) { ba_maxdiag_body(P, n_ptl, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x); }

// S <- camera-camera part (+ lambda on the diagonal when add_lambda), r <- bc   (upper AND lower filled)
// inline_odo (the dense LDS-path system of the local window, <= 64 camera-camera factors): the off-diagonal blocks of the factors are written here instead of by
// k_ba_add_odo, and the trial's two scalar accumulators are cleared here instead of by a memset — two stream operations less per LM trial, each of which queues behind the
// networks' workgroups when the tracker shares the GPU.  (An entry gets 0 + its factor's value either way: the same bits as the atomic add onto the initialised zero.)
__global__ __launch_bounds__(256) void k_ba_init_S(BaDev P, double lambda, int add_cam_part, int inline_odo)
{
    const int n6 = P.n6, ld = P.bw < 0 ? n6 : P.ldb;
    const size_t tot = (size_t)n6 * ld;
    if (inline_odo && blockIdx.x == 0 && threadIdx.x < 2) P.scal[2 + threadIdx.x] = 0.0;
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < tot; t += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(t / ld), col = P.bw < 0 ? (int)(t % ld) : row - P.bw + (int)(t % ld);
        double v = 0;
        if (add_cam_part && col >= 0 && col < n6 && row / 6 == col / 6) { v = P.Hcd[36 * (row / 6) + (row % 6) * 6 + col % 6]; if (row == col) v += lambda; }
        if (inline_odo && add_cam_part && row / 6 != col / 6) {
            const int bi = row / 6, bj = col / 6, a = row % 6, b = col % 6;
            for (int k = 0; k < P.n_odo; k++) {
                const int i = P.odo_i[k], j = P.odo_j[k];
                if (i == bi && j == bj) v += P.Hodo[36 * k + a * 6 + b];
                if (j == bi && i == bj) v += P.Hodo[36 * k + b * 6 + a];
            }
        }
        P.S[t] = v;
    }
    for (int a = blockIdx.x * blockDim.x + threadIdx.x; a < n6; a += gridDim.x * blockDim.x) P.r[a] = P.bc[a];
}
__global__ void k_ba_add_odo(BaDev P)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= P.n_odo * 36) return;
    const int k = t / 36, a = (t % 36) / 6, b = t % 6, i = P.odo_i[k], j = P.odo_j[k];
    const double v = P.Hodo[t];
    double* e0 = s_entry(P, 6 * i + a, 6 * j + b); if (e0) atomicAdd(e0, v);
    double* e1 = s_entry(P, 6 * j + b, 6 * i + a); if (e1) atomicAdd(e1, v);
}

// ---- Schur complement: one wave per landmark ------------------------------------------------------------
// LDS_S: the whole reduced system lives in LDS (n6 <= BA_LDS_MAX_N6) and is flushed once per workgroup.
// Only the LOWER triangle of S is produced (the Cholesky kernels read nothing else): slots of a landmark are in
// ascending camera order, so slot pairs (i >= j) are exactly the blocks (ci >= cj).
// MODE 0: the whole reduced system lives in LDS (n6 <= BA_LDS_MAX_N6, the local-BA window), flushed once per workgroup.
// MODE 1: FP64 atomics straight into the dense S in HBM.
// MODE 2: workgroups walk chunks of BA_CHUNK consecutive landmarks; landmark ids are contiguous in time for a SLAM
//         map, so a chunk touches a short run of cameras: a BA_WC-camera window of S is accumulated in LDS and flushed
//         once per chunk (blocks that fall outside the window go to HBM atomics directly).
#define BA_WC 16
#define SB_PITCH 37       // doubles per 6x6 block of the LDS-resident S (odd: consecutive blocks start in different bank pairs)
#ifndef BA_CHUNK
#define BA_CHUNK 256      // landmarks per window flush: every flush is a set of HBM atomics, and atomics onto one cache line serialise (~45 ns each)
#endif
// (bid, nblk) = (blockIdx.x, gridDim.x) for the kernel below; the persistent local-window solver passes its own workgroup index and grid
template <int MODE>
__device__ __forceinline__ void ba_schur_body(const BaDev& P, int n_ptl, double lambda, int kcap, double* S_part /*[grid][n6*n6 + n6], MODE 0*/,
                                              const int* __restrict__ chunk_cmin /*MODE 2*/, const int* __restrict__ lorder /*MODE 2: landmarks by first camera*/,
                                              const int2* __restrict__ lbc /*MODE 2: (first slot, slot count) of lorder[lp]*/, int bid, int nblk, double* lds)
{
    const int n6 = P.n6, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    const int wn = MODE == 0 ? n6 : BA_WC * 6;                 // side of the LDS-resident (window of) S
    // LDS layout of (the window of) S: the lower-triangle 6x6 blocks, block (ci >= cj) at [(ci(ci+1)/2 + cj) * SB_PITCH + 6a + b], then the rhs.
    // Lanes accumulate the SAME element (a, b) of DIFFERENT blocks at a time: with rows of the dense matrix as the layout every such address
    // differs by a multiple of 6 doubles and only 16 of the 32 bank pairs are ever hit (>= 4-way conflicts on every ds_add_f64); with one
    // block per 37 doubles the block index walks all of them.
    const int wc = wn / 6, nblk_l = wc * (wc + 1) / 2, rhs_off = nblk_l * SB_PITCH;
    double* Sl = lds;                                        // [nblk_l * SB_PITCH + wn] for MODE 0 / 2
    double* stage = lds + (MODE == 1 ? 0 : (size_t)rhs_off + wn) + (size_t)wave * (2 * kcap * 18 + kcap);   // W, WD, (pose, camera ordinal) of up to kcap obs per wave
    const int n_units = MODE == 2 ? (n_ptl + BA_CHUNK - 1) / BA_CHUNK : 1;
    for (int unit = MODE == 2 ? bid : 0; unit < n_units; unit += MODE == 2 ? nblk : 1) {
        int cbase = 0, l_beg, l_end, l_step;
        if (MODE == 2) { cbase = chunk_cmin[unit]; l_beg = unit * BA_CHUNK + wave; l_end = min(n_ptl, (unit + 1) * BA_CHUNK); l_step = nw; }
        else { l_beg = bid * nw + wave; l_end = n_ptl; l_step = nblk * nw; }
        if (MODE != 1) { for (int t = threadIdx.x; t < rhs_off + wn; t += blockDim.x) Sl[t] = 0; __syncthreads(); }
        // A landmark is a chain of dependent HBM round trips (its slot range -> its slot data -> the cameras of the slots); with ~30 landmarks
        // per wave that latency IS the kernel.  The slot range of the NEXT landmark is requested one iteration ahead (MODE 2 reads it from a
        // table in walk order, no lorder -> pt_start hop), and everything that depends on the range — Cp, W, the slots' poses and camera
        // ordinals — is requested together and parked in LDS.
        int l = 0, beg = 0, cnt = 0;
        if (l_beg < l_end) { if (MODE == 2) { l = lorder[l_beg]; const int2 bc = lbc[l_beg]; beg = bc.x; cnt = bc.y; } else { l = l_beg; beg = P.pt_start[l]; cnt = P.pt_start[l + 1] - beg; } }
        for (int lp = l_beg; lp < l_end; lp += l_step) {
            int nl = 0, nbeg = 0, ncnt = 0;
            { const int nlp = lp + l_step;
              if (nlp < l_end) { if (MODE == 2) { nl = lorder[nlp]; const int2 bc = lbc[nlp]; nbeg = bc.x; ncnt = bc.y; } else { nl = nlp; nbeg = P.pt_start[nl]; ncnt = P.pt_start[nl + 1] - nbeg; } } }
            if (cnt > 64) { l = nl; beg = nbeg; cnt = ncnt; continue; }     // wave-uniform: long tracks go through k_ba_schur_long
            const int k = min(cnt, kcap);
            double* Wl = stage; double* WDl = stage + kcap * 18; int* scam = (int*)(stage + 2 * kcap * 18); int* sord = scam + kcap;
            if (lane < k) { const int cp = P.slot_cam[beg + lane]; scam[lane] = cp; sord[lane] = MODE == 2 ? P.slot_ord[beg + lane] : cp; }
            for (int t = lane; t < k * 18; t += 64) { const int sl = t / 18; Wl[t] = P.W[BA_REC * (size_t)(beg + sl) + (t - sl * 18)]; }
            // point-side block of the landmark = sum of its slots' terms (lane i holds slot i, tracks have <= 64 observations); kept for the
            // back-substitution
            double H6[9];
            {
#pragma unroll
              for (int a = 0; a < 9; a++) H6[a] = lane < cnt ? P.Cp[BA_REC * (size_t)(beg + lane) + a] : 0.0;
#pragma unroll
              for (int a = 0; a < 9; a++) {
#pragma unroll
                  for (int o = 32; o >= 1; o >>= 1) H6[a] += __shfl_xor(H6[a], o, 64);
              }
              if (lane == 0) {
#pragma unroll
                  for (int a = 0; a < 6; a++) P.Hpp[6 * (size_t)l + a] = H6[a];
#pragma unroll
                  for (int a = 0; a < 3; a++) P.bp[3 * (size_t)l + a] = H6[6 + a];
              } }
            const double b0 = H6[6], b1 = H6[7], b2 = H6[8];
            double Di[9]; inv3sym(H6, lambda, Di);
            __builtin_amdgcn_wave_barrier();
            for (int t = lane; t < k * 6; t += 64) {              // WD = W * Di, row t of the stacked (6k x 3)
                const double w0 = Wl[t * 3], w1 = Wl[t * 3 + 1], w2 = Wl[t * 3 + 2];
                const double d0 = w0 * Di[0] + w1 * Di[3] + w2 * Di[6], d1 = w0 * Di[1] + w1 * Di[4] + w2 * Di[7], d2 = w0 * Di[2] + w1 * Di[5] + w2 * Di[8];
                WDl[t * 3] = d0; WDl[t * 3 + 1] = d1; WDl[t * 3 + 2] = d2;
                const int cpose = scam[t / 6], c = sord[t / 6] - cbase;      // MODE 2: cbase and the window count cameras
                const double rv = -(d0 * b0 + d1 * b1 + d2 * b2);
                if (MODE == 1 || (MODE == 2 && (c < 0 || c >= BA_WC))) atomicAdd(P.r + 6 * cpose + t % 6, rv);
                else atomicAdd(Sl + rhs_off + 6 * c + t % 6, rv);
            }
            __builtin_amdgcn_wave_barrier();
            // slot pairs (i >= j): block (ci, cj) -= WD_i W_j^T ; one 6x6 block per lane-iteration
            const int npair = k * (k + 1) / 2;
            for (int pq = lane; pq < npair; pq += 64) {
                int i = (int)((sqrtf(8.f * (float)pq + 1.f) - 1.f) * 0.5f);
                while (i * (i + 1) / 2 > pq) i--;
                while ((i + 1) * (i + 2) / 2 <= pq) i++;
                const int j = pq - i * (i + 1) / 2;
                const int pi = scam[i], pj = scam[j], ci = sord[i] - cbase, cj = sord[j] - cbase;
                const double* A = WDl + i * 18; const double* Bm = Wl + j * 18;
                const bool to_hbm = MODE == 1 || (MODE == 2 && (cj < 0 || ci >= BA_WC));       // ci >= cj
#pragma unroll
                for (int a = 0; a < 6; a++)
#pragma unroll
                    for (int b = 0; b < 6; b++) {
                        const double v = -(A[a * 3] * Bm[b * 3] + A[a * 3 + 1] * Bm[b * 3 + 1] + A[a * 3 + 2] * Bm[b * 3 + 2]);
                        if (to_hbm) { double* e = s_entry(P, 6 * pi + a, 6 * pj + b); if (e) atomicAdd(e, v); }
                        else atomicAdd(Sl + (size_t)(ci * (ci + 1) / 2 + cj) * SB_PITCH + a * 6 + b, v);
                    }
            }
            __builtin_amdgcn_wave_barrier();
            l = nl; beg = nbeg; cnt = ncnt;
        }
        if (MODE == 0) {
            __syncthreads();
            double* out = S_part + (size_t)bid * (rhs_off + n6);      // same block-major layout, k_ba_fold_parts maps it onto S
            for (int t = threadIdx.x; t < rhs_off + n6; t += blockDim.x) out[t] = Sl[t];
        }
        if (MODE == 2) {                                          // flush the window: one HBM atomic per touched entry
            __syncthreads();
            for (int t = threadIdx.x; t < nblk_l * 36; t += blockDim.x) {
                const int bid = t / 36, el = t - bid * 36;
                const double v = Sl[bid * SB_PITCH + el];
                if (v != 0.0) {
                    int ci = (int)((sqrtf(8.f * (float)bid + 1.f) - 1.f) * 0.5f);
                    ci -= (ci * (ci + 1) / 2 > bid); ci += ((ci + 1) * (ci + 2) / 2 <= bid);
                    const int cj = bid - ci * (ci + 1) / 2;
                    if (cbase + ci < P.n_cam_ord) { const int gr = 6 * P.ord_pose[cbase + ci] + el / 6, gc = 6 * P.ord_pose[cbase + cj] + el % 6; double* e = s_entry(P, gr, gc); if (e) atomicAdd(e, v); }
                }
            }
            for (int t = threadIdx.x; t < wn; t += blockDim.x) { const double v = Sl[rhs_off + t]; if (v != 0.0 && cbase + t / 6 < P.n_cam_ord) atomicAdd(P.r + 6 * P.ord_pose[cbase + t / 6] + t % 6, v); }
            __syncthreads();
        }
    }
}
template <int MODE>
__global__ __launch_bounds__(MODE == 2 ? 1024 : 512) void k_ba_schur(BaDev P, int n_ptl, double lambda, int kcap, double* S_part, const int* __restrict__ chunk_cmin, const int* __restrict__ lorder,
                                                  const int2* __restrict__ lbc)
{
    extern __shared__ double lds_schur_dyn[];
    ba_schur_body<MODE>(P, n_ptl, lambda, kcap, S_part, chunk_cmin, lorder, lbc, blockIdx.x, gridDim.x, lds_schur_dyn);
}
// MODE 2 on the FP64 matrix cores (round 2).  The wave-per-landmark kernel above spends ~800 VALU instructions per landmark on pair index arithmetic, 36 address
// computations and 36 LDS atomics per camera pair (0.5 ms per 100 k landmarks: 39 % of a global LM iteration).  Here a landmark contributes three ROWS of a
// matrix M instead: with Hpp + lambda I = R^T R (3x3 Cholesky) the rows are A_l = [W_s R^-1 for the landmark's slots s, placed in the columns of camera s | R^-T b], and
//     S_window -= M^T M   (camera columns),   r_window -= M^T M[:, rhs column]
// i.e. sum_l (W_i Di W_j^T) = sum_l (W_i R^-1)(W_j R^-1)^T — a dense rank-3L update of the BA_WC-camera window: v_mfma_f64_16x16x4 over 16x16 tiles of the lower
// triangle (+ one tile row for the rhs), SM_L landmarks (96 rows of M in LDS) per pass, the tiles accumulated in registers across the BA_CHUNK landmarks that share a
// window and flushed once with FP64 atomics like before.  Landmarks that do not fit their chunk's window (or have more than 64 observations) are listed by the host for
// k_ba_schur_long and skipped here (their slot count is stored negated).
#ifndef SM_L
#define SM_L 32             // landmarks per pass: 96 rows of M = 76.8 KB of LDS, two workgroups per CU
#endif
#ifndef SM_NW
#define SM_NW 8             // waves per workgroup.  16 (round 2 - 5) leaves 128 registers per lane: with the next pass's slot data held across the matrix phase the kernel spilled
#endif                      // 44 of them, and a scratch reload waits (vmcnt is in order) for every prefetched HBM load issued before it — the matrix phase took twice as long.  8 waves: 256.
#define SM_NT (64 * SM_NW)
#define SM_NTILE ((21 + SM_NW - 1) / SM_NW)      // tiles of the lower triangle per wave
#define SM_RP (SM_NT / 96)  // row parts of the rhs matrix-vector product
#define SM_NL (SM_L / SM_NW)   // landmarks per wave and pass
// M is stored in the matrix cores' operand order: element (row, col) of the 3 SM_L x 96 matrix lives at [(row / 4) * 6 + col / 16][row % 4][col % 16], so the 64 lanes of
// a wave read ONE contiguous 512-byte run per operand (a row-major M with the 4 k-rows of an operand 98 doubles apart ran into 4-way bank conflicts on every read:
// 17 of the ~35 us of a pass).  The rhs column is kept apart (G): its tile row needs only row 0 of the A operand.
#define SM_IDX(row, col) (((((row) >> 2) * 6 + ((col) >> 4)) << 6) + (((row) & 3) << 4) + ((col) & 15))
#define SM_M_DOUBLES (3 * SM_L * 96)
#define SM_LDS_BYTES ((SM_M_DOUBLES + 3 * SM_L + 9 * SM_L + 96) * sizeof(double))
__global__ __launch_bounds__(SM_NT) void k_ba_schur_mfma(BaDev P, int n_ptl, double lambda, const int* __restrict__ chunk_cmin, const int* __restrict__ lorder, const int2* __restrict__ lbc, int chunk /* landmarks per unit (window flush): a multiple of SM_L chosen by the host */)
{
    typedef double d4 __attribute__((ext_vector_type(4)));
    extern __shared__ double M[];                               // [3 SM_L / 4][6][4][16] | G [3 SM_L] | Hs [SM_L][9]
    double* G = M + SM_M_DOUBLES; double* Hs = G + 3 * SM_L; double* Rs = Hs + 9 * SM_L;      // Rs [96]: the window's rhs, summed over the thread parts at the flush
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 15, lk = lane >> 4;
    const int n_units = (n_ptl + chunk - 1) / chunk;
    // tiles of this wave: the 21 lower-triangle tiles of the 6 x 6 camera tile grid (the rhs M^T G is a matrix-vector product: vector ALUs, thread = (column, row part))
    const int rcol = threadIdx.x % 96, rpart = threadIdx.x / 96;      // SM_RP row parts (the threads past 96 SM_RP idle in that step)
    int tI[SM_NTILE], tJ[SM_NTILE]; bool tv[SM_NTILE];
#pragma unroll
    for (int u = 0; u < SM_NTILE; u++) {
        const int t = wave + SM_NW * u; tv[u] = t < 21;
        int I = 0; while ((I + 1) * (I + 2) / 2 <= t && I < 5) I++;
        tI[u] = t < 21 ? I : 6; tJ[u] = t < 21 ? t - I * (I + 1) / 2 : t - 21;
    }
#ifdef SM_PROF
    unsigned long long stp[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define SM_STAMP(k) do { if (sub == 1) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stp[k]) :: "memory"); } while (0)
#else
#define SM_STAMP(k) do { } while (0)
#endif
    // (round 6) The passes of this workgroup — eight per unit, the units blockIdx.x, blockIdx.x + gridDim.x, ... — are walked as ONE sequence, and the two dependent HBM
    // round trips of a pass (slot range -> slot data) are taken out of its critical path: the slot data of pass i + 1 is requested right after the barrier in front of pass
    // i's matrix phase, into the registers pass i has just finished with, and the slot ranges of pass i + 2 behind them.  Shader-clock stamps of the round-5 form
    // (-DSM_PROF): 8-10 k cycles from the top of a pass to its loads being issued + 7 k in the barrier behind the zero fill (= waiting for those loads), of 39 k per pass.
    // The point-side terms (6 + 3 doubles per slot) are loaded by the lane that also loads the slot's W row (lane = 6 slot + component: one or two loads) instead of nine
    // loads by one lane per slot: 5 instead of 12 live registers per landmark across the matrix phase, 2 instead of 9 LDS atomic instructions.
    const int NSUB = chunk / SM_L;
    auto advance = [&](int& uu, int& ss) { ss++; if (ss == NSUB || uu * chunk + ss * SM_L >= n_ptl) { ss = 0; uu += gridDim.x; } };
    int lm[SM_NL], beg[SM_NL], cnt[SM_NL], wcol[SM_NL]; double cpa[SM_NL], cpb[SM_NL], wv[SM_NL][3];
    int lm2[SM_NL]; int2 bc2[SM_NL];
    auto load_idx = [&](int uu, int ss) {
#pragma unroll
        for (int n = 0; n < SM_NL; n++) {
            const int q = uu * chunk + ss * SM_L + wave + SM_NW * n; const bool valid = uu < n_units && q < n_ptl;
            lm2[n] = valid ? lorder[q] : 0; bc2[n] = valid ? lbc[q] : make_int2(0, -1);      // cnt < 0: out of window / long track (k_ba_schur_long) or past the end
        }
    };
    auto issue_rec = [&]() {
#pragma unroll
        for (int n = 0; n < SM_NL; n++) {
            lm[n] = lm2[n]; beg[n] = bc2[n].x; cnt[n] = bc2[n].y;
            // (slot, component) pairs of the first 64 lanes — all of them for tracks of up to 10 observations
            const bool on = lane < cnt[n] * 6; const int sl = lane / 6, a = lane - sl * 6;
            const size_t rec = BA_REC * (size_t)(beg[n] + (on ? sl : 0));
            wcol[n] = on ? P.slot_ord[beg[n] + sl] : 0;                  // (the slot's camera ordinal as loaded: turning it into a column HERE would wait for the load in front of the matrix phase)
            const double* w = P.W + rec + 3 * a;
            wv[n][0] = on ? w[0] : 0.0; wv[n][1] = on ? w[1] : 0.0; wv[n][2] = on ? w[2] : 0.0;
            cpa[n] = on ? P.Cp[rec + a] : 0.0; cpb[n] = (on && a < 3) ? P.Cp[rec + 6 + a] : 0.0;
        }
    };
    int unit = blockIdx.x, sub = 0, nu = unit, ns = 0;
    if (unit < n_units) { load_idx(unit, 0); issue_rec(); advance(nu, ns); load_idx(nu, ns); }
    d4 acc[SM_NTILE]; double racc = 0.0;
    while (unit < n_units) {
        const int cbase = chunk_cmin[unit];
        if (sub == 0) {
            racc = 0.0;
#pragma unroll
            for (int u = 0; u < SM_NTILE; u++) acc[u] = (d4){0.0, 0.0, 0.0, 0.0};
        }
        {
            SM_STAMP(0);
            SM_STAMP(1);
            for (int t = threadIdx.x; t < SM_M_DOUBLES + 3 * SM_L + 9 * SM_L + 96; t += SM_NT) M[t] = 0.0;
            __syncthreads();
            SM_STAMP(2);
            // the landmarks' point-side sums, reduced with LDS atomics (a shuffle tree is 108 ds_bpermute per landmark: that alone saturated the CU's LDS pipe)
#pragma unroll
            for (int n = 0; n < SM_NL; n++) {
                const int sl = lane / 6, a = lane - sl * 6;
                if (lane < cnt[n] * 6) { atomicAdd(Hs + 9 * (wave + SM_NW * n) + a, cpa[n]); if (a < 3) atomicAdd(Hs + 9 * (wave + SM_NW * n) + 6 + a, cpb[n]); }
                for (int t = lane + 64; t < cnt[n] * 6; t += 64) {          // tracks longer than 10 observations
                    const int s2 = t / 6, a2 = t - s2 * 6; const size_t rec = BA_REC * (size_t)(beg[n] + s2);
                    atomicAdd(Hs + 9 * (wave + SM_NW * n) + a2, P.Cp[rec + a2]); if (a2 < 3) atomicAdd(Hs + 9 * (wave + SM_NW * n) + 6 + a2, P.Cp[rec + 6 + a2]);
                }
            }
            __builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_wave_barrier();      // the wave reads back only its own landmarks' sums
            SM_STAMP(3);
#pragma unroll
            for (int n = 0; n < SM_NL; n++) {
                if (cnt[n] < 0) continue;                       // wave-uniform
                double H6[9];
#pragma unroll
                for (int a = 0; a < 9; a++) H6[a] = Hs[9 * (wave + SM_NW * n) + a];
                if (lane == 0) {
#pragma unroll
                    for (int a = 0; a < 6; a++) P.Hpp[6 * (size_t)lm[n] + a] = H6[a];
#pragma unroll
                    for (int a = 0; a < 3; a++) P.bp[3 * (size_t)lm[n] + a] = H6[6 + a];
                }
                // Hpp + lambda I = R^T R, R upper triangular; X = R^-1 (reciprocal square roots: v_rsq_f64 + two Newton steps instead of the sqrt and division expansions)
                auto rsq = [](double x) { double y = __builtin_amdgcn_rsq(x); y = y * (1.5 - 0.5 * x * y * y); return y * (1.5 - 0.5 * x * y * y); };
                const double a00 = H6[0] + lambda, a01 = H6[1], a02 = H6[2], a11 = H6[3] + lambda, a12 = H6[4], a22 = H6[5] + lambda;
                const double i00 = rsq(a00), r01 = a01 * i00, r02 = a02 * i00;
                const double i11 = rsq(a11 - r01 * r01), r12 = (a12 - r01 * r02) * i11;
                const double i22 = rsq(a22 - r02 * r02 - r12 * r12);
                const double x01 = -r01 * i11 * i00, x12 = -r12 * i22 * i11, x02 = -(r01 * x12 + r02 * i22) * i00;
                const int row0 = 3 * (wave + SM_NW * n);
                if (lane < cnt[n] * 6) {
                    const double w0 = wv[n][0], w1 = wv[n][1], w2 = wv[n][2]; const int col = 6 * (wcol[n] - cbase) + (lane - 6 * (lane / 6));
                    M[SM_IDX(row0, col)] = w0 * i00; M[SM_IDX(row0 + 1, col)] = w0 * x01 + w1 * i11; M[SM_IDX(row0 + 2, col)] = w0 * x02 + w1 * x12 + w2 * i22;
                }
                for (int t = lane + 64; t < cnt[n] * 6; t += 64) {          // tracks longer than 10 observations
                    const int sl = t / 6, a = t - sl * 6, col = 6 * (P.slot_ord[beg[n] + sl] - cbase) + a;
                    const double* w = P.W + BA_REC * (size_t)(beg[n] + sl) + 3 * a;
                    const double w0 = w[0], w1 = w[1], w2 = w[2];
                    M[SM_IDX(row0, col)] = w0 * i00; M[SM_IDX(row0 + 1, col)] = w0 * x01 + w1 * i11; M[SM_IDX(row0 + 2, col)] = w0 * x02 + w1 * x12 + w2 * i22;
                }
                if (lane == 0) { const double b0 = H6[6], b1 = H6[7], b2 = H6[8]; G[row0] = i00 * b0; G[row0 + 1] = x01 * b0 + i11 * b1; G[row0 + 2] = x02 * b0 + x12 * b1 + i22 * b2; }
                __builtin_amdgcn_sched_barrier(0);             // one landmark's arithmetic at a time: interleaved, the two exceed the 128-VGPR budget of a 1024-thread workgroup (101 spills)
            }
            SM_STAMP(4);
            __syncthreads();
            SM_STAMP(5);
            // the next pass's slot data (its slot ranges arrived a pass ago) and the slot ranges of the pass after it: in flight beside the matrix phase
            int fu = nu, fs = ns;
            if (nu < n_units) { issue_rec(); advance(fu, fs); load_idx(fu, fs); }
#ifndef SM_NOMFMA
#pragma unroll
            for (int u = 0; u < SM_NTILE; u++) if (tv[u]) {
                const double* pb = M + 64 * tJ[u] + lane;
                const double* pa = M + 64 * tI[u] + lane;
#pragma unroll 8
                for (int ks = 0; ks < 3 * SM_L / 4; ks++) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[384 * ks], pb[384 * ks], acc[u], 0, 0, 0);
            }
            if (rpart < SM_RP) {
#pragma unroll 4
                for (int k = rpart; k < 3 * SM_L; k += SM_RP) racc = __builtin_fma(G[k], M[SM_IDX(k, rcol)], racc);
            }
#endif
            SM_STAMP(6);
            __syncthreads();
            SM_STAMP(7);
            const bool unit_done = nu != unit;
            const int cur = unit;
            unit = nu; sub = ns; nu = fu; ns = fs;
            if (!unit_done) continue;
            (void)cur;
        }
#ifdef SM_PROF
        { const int sub = 1; SM_STAMP(8); }
#endif
        // flush the window: S -= C on the lower block triangle (diagonal camera blocks in full), r -= C[rhs row]
#pragma unroll
        for (int u = 0; u < SM_NTILE; u++) if (tv[u]) {
#pragma unroll
            for (int v = 0; v < 4; v++) {
                const int row = 16 * tI[u] + lk + 4 * v, col = 16 * tJ[u] + lr;
                const double val = -acc[u][v];
#ifdef SM_NOFLUSH
                if (val != 123.456) continue;
#endif
                if (val == 0.0 || col >= 96) continue;
                const int cj = col / 6;
                if (cbase + cj >= P.n_cam_ord) continue;
                const int ci = row / 6;
                if (row >= 96 || ci < cj || cbase + ci >= P.n_cam_ord) continue;
                const int gr = 6 * P.ord_pose[cbase + ci] + row % 6, gc = 6 * P.ord_pose[cbase + cj] + col % 6;
                double* e = s_entry(P, gr, gc); if (e) atomicAdd(e, val);
                if (ci == cj && tI[u] != tJ[u]) { double* e2 = s_entry(P, gc, gr); if (e2) atomicAdd(e2, val); }      // a diagonal camera block that straddles two tiles: its upper part lives in the tile that is not computed
            }
        }
        if (rpart < SM_RP && racc != 0.0) atomicAdd(Rs + rcol, racc);      // (Rs was zeroed with M by the last pass; nothing touched it since)
        __syncthreads();
        if (threadIdx.x < 96) { const double v = -Rs[threadIdx.x]; const int cj = threadIdx.x / 6; if (v != 0.0 && cbase + cj < P.n_cam_ord) atomicAdd(P.r + 6 * P.ord_pose[cbase + cj] + threadIdx.x % 6, v); }
        __syncthreads();
#ifdef SM_PROF
        { const int sub = 1; SM_STAMP(9); }
        if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == 200))
            printf("[sm_prof wg %d] top %llu | zero+sync %llu | atomics %llu | fillM %llu | sync %llu | mfma+rhs %llu | sync %llu || flush %llu\n", blockIdx.x,
                   stp[1] - stp[0], stp[2] - stp[1], stp[3] - stp[2], stp[4] - stp[3], stp[5] - stp[4], stp[6] - stp[5], stp[7] - stp[6], stp[9] - stp[8]);
#endif
    }
}

// Landmarks with more than 64 observations (a static point watched for more than 64 keyframes: a vehicle waiting at a junction; FullBatchOptimization has no
// track-length limit, Optimizer.cc:1235ff): one 256-thread workgroup per landmark, slots streamed from HBM instead of being parked in a wave's LDS stage, every
// contribution an FP64 atomic on S / r.  O(k^2) blocks per landmark like the fast path; rare, so simple.  k_ba_schur skips these landmarks.
__global__ __launch_bounds__(256) void k_ba_schur_long(BaDev P, double lambda, const int* __restrict__ long_list)
{
    __shared__ double red[4][9]; __shared__ double sH[9], sDi[9];
    const int l = long_list[blockIdx.x], beg = P.pt_start[l], k = P.pt_start[l + 1] - beg, tid = threadIdx.x;
    double h[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = tid; i < k; i += 256) {
#pragma unroll
        for (int a = 0; a < 9; a++) h[a] += P.Cp[BA_REC * (size_t)(beg + i) + a];
    }
#pragma unroll
    for (int a = 0; a < 9; a++) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) h[a] += __shfl_xor(h[a], o, 64);
    }
    if ((tid & 63) == 0) { for (int a = 0; a < 9; a++) red[tid >> 6][a] = h[a]; }
    __syncthreads();
    if (tid == 0) {
        double H6[9]; for (int a = 0; a < 9; a++) H6[a] = red[0][a] + red[1][a] + red[2][a] + red[3][a];
        for (int a = 0; a < 6; a++) P.Hpp[6 * (size_t)l + a] = H6[a];
        for (int a = 0; a < 3; a++) P.bp[3 * (size_t)l + a] = H6[6 + a];
        double Di[9]; inv3sym(H6, lambda, Di);
        for (int a = 0; a < 9; a++) { sH[a] = H6[a]; sDi[a] = Di[a]; }
    }
    __syncthreads();
    double Di[9]; for (int a = 0; a < 9; a++) Di[a] = sDi[a];
    const double b0 = sH[6], b1 = sH[7], b2 = sH[8];
    for (int t = tid; t < k * 6; t += 256) {                 // rhs: r_c -= (W_i Di) b
        const double* w = P.W + BA_REC * (size_t)(beg + t / 6) + 3 * (t % 6);
        const double d0 = w[0] * Di[0] + w[1] * Di[3] + w[2] * Di[6], d1 = w[0] * Di[1] + w[1] * Di[4] + w[2] * Di[7], d2 = w[0] * Di[2] + w[1] * Di[5] + w[2] * Di[8];
        atomicAdd(P.r + 6 * P.slot_cam[beg + t / 6] + t % 6, -(d0 * b0 + d1 * b1 + d2 * b2));
    }
    const long long npair = (long long)k * (k + 1) / 2;
    for (long long pq = tid; pq < npair; pq += 256) {        // slot pairs i >= j (slots are in camera order): block (c_i, c_j) -= (W_i Di) W_j^T
        long long i = (long long)((sqrt(8.0 * (double)pq + 1.0) - 1.0) * 0.5);
        while (i * (i + 1) / 2 > pq) i--;
        while ((i + 1) * (i + 2) / 2 <= pq) i++;
        const int j = (int)(pq - i * (i + 1) / 2);
        const double* Wi = P.W + BA_REC * (size_t)(beg + i); const double* Wj = P.W + BA_REC * (size_t)(beg + j);
        const int pi = P.slot_cam[beg + i], pj = P.slot_cam[beg + j];
#pragma unroll
        for (int a = 0; a < 6; a++) {
            const double w0 = Wi[a * 3], w1 = Wi[a * 3 + 1], w2 = Wi[a * 3 + 2];
            const double d0 = w0 * Di[0] + w1 * Di[3] + w2 * Di[6], d1 = w0 * Di[1] + w1 * Di[4] + w2 * Di[7], d2 = w0 * Di[2] + w1 * Di[5] + w2 * Di[8];
#pragma unroll
            for (int b = 0; b < 6; b++) {
                const double v = -(d0 * Wj[b * 3] + d1 * Wj[b * 3 + 1] + d2 * Wj[b * 3 + 2]);
                double* e = s_entry(P, 6 * pi + a, 6 * pj + b); if (e) atomicAdd(e, v);
            }
        }
    }
}
__global__ __launch_bounds__(256) void k_ba_fold_parts(BaDev P, const double* S_part, int nparts)
{
    // parts are block-major (k_ba_schur<0>): lower-triangle 6x6 blocks at [bid * SB_PITCH + 6a + b], then the rhs; S itself is dense row-major
    const int nc = P.n6 / 6, nblk_l = nc * (nc + 1) / 2, rhs_off = nblk_l * SB_PITCH;
    const size_t sz = (size_t)rhs_off + P.n6;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < nblk_l * 36 + P.n6; t += gridDim.x * blockDim.x) {
        const bool is_rhs = t >= nblk_l * 36;
        const int bid = t / 36, el = t - bid * 36;
        const size_t src = is_rhs ? (size_t)rhs_off + (t - nblk_l * 36) : (size_t)bid * SB_PITCH + el;
        // blockIdx.y takes every gridDim.y-th part: a thread that walks all 256 parts serially is 64 dependent-latency steps (22 us for 60 KB);
        // the gridDim.y partial sums of an entry meet through atomics (8 per address, spread over all of S)
        double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
        int p = blockIdx.y; const int ps = gridDim.y;
        for (; p + 3 * ps < nparts; p += 4 * ps) {      // four independent load streams in flight
            s0 += S_part[(size_t)p * sz + src]; s1 += S_part[(size_t)(p + ps) * sz + src];
            s2 += S_part[(size_t)(p + 2 * ps) * sz + src]; s3 += S_part[(size_t)(p + 3 * ps) * sz + src];
        }
        for (; p < nparts; p += ps) s0 += S_part[(size_t)p * sz + src];
        const double sum = (s0 + s1) + (s2 + s3);
        if (is_rhs) atomicAdd(P.r + (t - nblk_l * 36), sum);
        else {
            int ci = (int)((sqrtf(8.f * (float)bid + 1.f) - 1.f) * 0.5f);
            ci -= (ci * (ci + 1) / 2 > bid); ci += ((ci + 1) * (ci + 2) / 2 <= bid);
            const int cj = bid - ci * (ci + 1) / 2;
            atomicAdd(P.S + (size_t)(6 * ci + el / 6) * P.n6 + 6 * cj + el % 6, sum);
        }
    }
}

// ---- reduced solve --------------------------------------------------------------------------------------
// single workgroup (1024 threads), whole system in LDS.  The right-hand side rides along as row n of the
// (n+1) x n lower factor, so the forward substitution costs nothing; rows are padded to an odd pitch (column
// walks are bank-conflict free); the scaled pivot column is mirrored into a contiguous vector so the rank-1
// trailing update reads col[i]*col[c] (broadcast) + one row-contiguous element; the backward substitution runs
// in a single wave without workgroup barriers.  scal[4] = 1 on success.
__global__ __launch_bounds__(1024) void k_ba_chol_small(BaDev P)
{
    extern __shared__ double lds[];
    const int n = P.n6, tid = threadIdx.x, nt = blockDim.x, ld = (n + 1) | 1;
    double* A = lds;                     // [(n+1)][ld]
    double* col = lds + (size_t)(n + 1) * ld;   // [n+1]
    __shared__ int ok;
    for (int t = tid; t < n * n; t += nt) { const int r = t / n, c = t - r * n; if (c <= r) A[r * ld + c] = P.S[t]; }
    for (int t = tid; t < n; t += nt) A[n * ld + t] = P.r[t];
    if (tid == 0) ok = 1;
    __syncthreads();
    for (int j = 0; j < n; j++) {
        double d = A[j * ld + j];
        if (!(d > 0) || !isfinite(d)) { if (tid == 0) ok = 0; d = 1.0; }
        const double inv = 1.0 / sqrt(d);
        for (int i = j + 1 + tid; i <= n; i += nt) { const double v = A[i * ld + j] * inv; A[i * ld + j] = v; col[i] = v; }
        __syncthreads();
        if (tid == 0) A[j * ld + j] = sqrt(d);      // after every thread has read the pivot; not touched by the update below
        // rank-1 update of rows j+1 .. n (row n = rhs), columns j+1 .. min(i, n-1); 32x32 thread tile, no divisions
        for (int i = j + 1 + (tid >> 5); i <= n; i += 32) {
            const double ci = col[i]; const int cmax = min(i, n - 1);
            for (int c = j + 1 + (tid & 31); c <= cmax; c += 32) A[i * ld + c] -= ci * col[c];
        }
        __syncthreads();
    }
    // row n now holds z = L^-1 r.  Backward substitution L^T x = z in wave 0 (LDS ops of one wave are ordered).
    if (tid < 64) {
        for (int j = n - 1; j >= 0; j--) {
            const double xj = A[n * ld + j] / A[j * ld + j];
            __builtin_amdgcn_wave_barrier();
            for (int i = tid; i < j; i += 64) A[n * ld + i] -= A[j * ld + i] * xj;
            if (tid == 0) A[n * ld + j] = xj;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
    __syncthreads();
    for (int t = tid; t < n; t += nt) P.x[t] = A[n * ld + t];
    if (tid == 0) P.scal[4] = (double)ok;
}

// blocked right-looking Cholesky in HBM, lower triangle of the row-major n x n matrix A, tile NB = 32.  The buffer
// holds one extra row (row n = right-hand side, [S | r] is contiguous): carried through the panel / update kernels it
// comes out as z = L^-1 r, so the forward substitution is free; the backward substitution L^T x = z walks the block
// columns from the last to the first with two small launches per block (rows-below dot products, 32x32 solve).
#define NB 32
__global__ __launch_bounds__(256) void k_chol_diag(double* A, int n, int k0, double* okflag)
{
    __shared__ double T[NB][NB + 1];
    const int nb = min(NB, n - k0), tid = threadIdx.x;
    for (int t = tid; t < nb * nb; t += 256) T[t / nb][t % nb] = A[(size_t)(k0 + t / nb) * n + k0 + t % nb];
    __syncthreads();
    for (int j = 0; j < nb; j++) {
        if (tid == 0) { const double d = T[j][j]; if (!(d > 0) || !isfinite(d)) { *okflag = 0; T[j][j] = 1.0; } else T[j][j] = sqrt(d); }
        __syncthreads();
        if (tid > j && tid < nb) T[tid][j] /= T[j][j];
        __syncthreads();
        for (int t = tid; t < nb * nb; t += 256) { const int i = t / nb, c = t % nb; if (i > j && c > j && c <= i) T[i][c] -= T[i][j] * T[c][j]; }
        __syncthreads();
    }
    for (int t = tid; t < nb * nb; t += 256) if (t % nb <= t / nb) A[(size_t)(k0 + t / nb) * n + k0 + t % nb] = T[t / nb][t % nb];
}
// panel: rows below the diagonal tile (including the rhs row n): X L11^T = A21  (each workgroup: NB rows)
__global__ __launch_bounds__(256) void k_chol_panel(double* A, int n, int nrows, int k0)
{
    __shared__ double L[NB][NB + 1], X[NB][NB + 1];
    const int nb = min(NB, n - k0), r0 = k0 + nb + blockIdx.x * NB, nr = min(NB, nrows - r0), tid = threadIdx.x;
    if (nr <= 0) return;
    for (int t = tid; t < nb * nb; t += 256) L[t / nb][t % nb] = A[(size_t)(k0 + t / nb) * n + k0 + t % nb];
    for (int t = tid; t < nr * nb; t += 256) X[t / nb][t % nb] = A[(size_t)(r0 + t / nb) * n + k0 + t % nb];
    __syncthreads();
    if (tid < nr) {                                       // one row per thread: forward substitution against L^T
        for (int j = 0; j < nb; j++) { double s = X[tid][j]; for (int c = 0; c < j; c++) s -= X[tid][c] * L[j][c]; X[tid][j] = s / L[j][j]; }
    }
    __syncthreads();
    for (int t = tid; t < nr * nb; t += 256) A[(size_t)(r0 + t / nb) * n + k0 + t % nb] = X[t / nb][t % nb];
}
// trailing update: A22 -= L21 L21^T on the lower triangle (+ the rhs row), one NBxNB tile per workgroup
__global__ __launch_bounds__(256) void k_chol_update(double* A, int n, int nrows, int k0)
{
    __shared__ double Pa[NB][NB + 1], Pb[NB][NB + 1];
    const int nb = min(NB, n - k0), base = k0 + nb;
    const int ti = blockIdx.y, tj = blockIdx.x;
    if (tj > ti) return;
    const int r0 = base + ti * NB, c0 = base + tj * NB, nr = min(NB, nrows - r0), nc = min(NB, n - c0), tid = threadIdx.x;
    if (nr <= 0 || nc <= 0) return;
    for (int t = tid; t < nr * nb; t += 256) Pa[t / nb][t % nb] = A[(size_t)(r0 + t / nb) * n + k0 + t % nb];
    for (int t = tid; t < nc * nb; t += 256) Pb[t / nb][t % nb] = A[(size_t)(c0 + t / nb) * n + k0 + t % nb];
    __syncthreads();
    for (int t = tid; t < nr * nc; t += 256) {
        const int i = t / nc, c = t % nc;
        if (c0 + c > r0 + i) continue;
        double s = 0;
        for (int q = 0; q < nb; q++) s += Pa[i][q] * Pb[c][q];
        A[(size_t)(r0 + i) * n + c0 + c] -= s;
    }
}
// backward substitution, block column k0: tmp[c] += sum over the rows i below the block of L[i][k0+c] * x[i]
__global__ __launch_bounds__(256) void k_chol_back_rows(const double* __restrict__ A, int n, int k0, const double* __restrict__ x, double* __restrict__ tmp)
{
    __shared__ double part[4][NB];
    const int nb = min(NB, n - k0), i = k0 + nb + blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double acc[NB];
    const bool act = i < n;
    const double xi = act ? x[i] : 0.0; const double* row = A + (size_t)(act ? i : 0) * n + k0;
#pragma unroll
    for (int c = 0; c < NB; c++) acc[c] = (act && c < nb) ? row[c] * xi : 0.0;
#pragma unroll
    for (int c = 0; c < NB; c++) { double v = acc[c];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) part[wave][c] = v; }
    __syncthreads();
    if (threadIdx.x < nb) atomicAdd(tmp + threadIdx.x, part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]);
}
__global__ __launch_bounds__(64) void k_chol_back_diag(const double* __restrict__ A, int n, int k0, double* __restrict__ x, double* __restrict__ tmp)
{
    __shared__ double T[NB][NB + 1], y[NB];
    const int nb = min(NB, n - k0), tid = threadIdx.x;
    for (int t = tid; t < nb * nb; t += 64) T[t / nb][t % nb] = A[(size_t)(k0 + t / nb) * n + k0 + t % nb];
    if (tid < nb) { y[tid] = x[k0 + tid] - tmp[tid]; tmp[tid] = 0.0; }
    __syncthreads();
    if (tid == 0) for (int j = nb - 1; j >= 0; j--) { double s = y[j]; for (int c = j + 1; c < nb; c++) s -= T[c][j] * y[c]; y[j] = s / T[j][j]; }
    __syncthreads();
    if (tid < nb) x[k0 + tid] = y[tid];
}

// ---- banded reduced solve ------------------------------------------------------------------------------------
// A SLAM map without loop closures couples a camera only to its temporal neighbours (track length, odometry), so the
// reduced camera system is block-banded: it is stored, all-reduced and factored as a band (n6 x (bw+1) doubles instead of
// n6^2 — 3 MB instead of 72 MB for 500 keyframes).  One persistent workgroup walks the diagonal in NB-column steps:
// diagonal block factored in LDS by one wave, panel rows solved one thread per row, trailing (bw x bw) window updated by
// all 1024 threads, the right-hand side carried along (forward substitution for free), then the backward sweep.
#define CB_NB 16
#define CB_MAXROWS 960                     // panel rows of one step = bw; one thread per panel row on threads 64..1023
#define CB_MAXBW 959
__global__ __launch_bounds__(1024) void k_chol_band(BaDev P)
{
    __shared__ double Ld[CB_NB + 1][CB_NB + 1];            // row CB_NB carries the right-hand side of the block (forward substitution for free)
    extern __shared__ double Pn[];                         // [bw][CB_NB + 1] panel rows (dynamic: up to 130 KB)
    __shared__ double zk[CB_NB];
    __shared__ int ok;
    const int n = P.n6, bw = P.bw, ldb = P.ldb, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    double* Sg = P.S; double* r = P.r; double* x = P.x;
#define AB(i, j) Sg[(size_t)(i) * ldb + ((j) - (i) + bw)]
    if (tid == 0) ok = 1;
    __syncthreads();
    for (int k0 = 0; k0 < n; k0 += CB_NB) {
        const int nb = min(CB_NB, n - k0), i0 = k0 + nb, i1 = min(n, k0 + nb + bw), m = i1 - i0;      // panel rows [i0, i1)
        // the block is always handled as CB_NB x CB_NB (+ the rhs row): a short last block is padded with an identity tail
        if (tid < CB_NB * CB_NB) { const int a = tid >> 4, b = tid & 15; Ld[a][b] = (a < nb && b <= a) ? AB(k0 + a, k0 + b) : (a == b ? 1.0 : 0.0); }
        if (tid >= 512 && tid < 512 + CB_NB) Ld[CB_NB][tid - 512] = tid - 512 < nb ? r[k0 + tid - 512] : 0.0;
        // panel rows are prefetched into registers while wave 0 factors the diagonal block
        double v[CB_NB];
        const bool prow = tid >= 64 && tid < 64 + m;
        const int pi = i0 + tid - 64;
        double rpi = 0;
        if (prow) {
#pragma unroll
            for (int c = 0; c < CB_NB; c++) v[c] = (c < nb && pi - (k0 + c) <= bw) ? AB(pi, k0 + c) : 0.0;
            rpi = r[pi];
        }
        __syncthreads();
        if (tid < 64) {                                   // unblocked Cholesky of the (CB_NB+1) x CB_NB block (rhs row included), all in LDS
#pragma unroll 1
            for (int c = 0; c < CB_NB; c++) {
                const double d = Ld[c][c];
                if (!(d > 0.0) || !isfinite(d)) { if (tid == 0) ok = 0; break; }
                const double sd = sqrt(d), inv = 1.0 / sd;
                __builtin_amdgcn_wave_barrier();
                if (tid > c && tid <= CB_NB) Ld[tid][c] *= inv;
                if (tid == c) Ld[c][c] = sd;
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int q = 0; q < 5; q++) {             // (CB_NB + 1) * CB_NB = 272 entries = 4.25 x 64 lanes
                    const int t = tid + 64 * q, a = t >> 4, b2 = t & 15;
                    if (t < (CB_NB + 1) * CB_NB && b2 > c && a >= b2) Ld[a][b2] -= Ld[a][c] * Ld[b2][c];
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        __syncthreads();
        if (!ok) break;
        if (tid < nb * nb) { const int a = tid / nb, b = tid - a * nb; if (b <= a) AB(k0 + a, k0 + b) = Ld[a][b]; }
        if (tid >= 512 && tid < 512 + nb) r[k0 + tid - 512] = Ld[CB_NB][tid - 512];
        if (prow) {                                       // row pi of L against the block; entries outside the band are structurally zero
            double rr = 0;
#pragma unroll
            for (int c = 0; c < CB_NB; c++) {
                if (c < nb) {
                    double a = v[c];
#pragma unroll
                    for (int e = 0; e < c; e++) a -= v[e] * Ld[c][e];
                    a /= Ld[c][c];
                    v[c] = a; rr += a * Ld[CB_NB][c];
                    if (pi - (k0 + c) <= bw) AB(pi, k0 + c) = a;
                }
                Pn[(tid - 64) * (CB_NB + 1) + c] = c < nb ? v[c] : 0.0;
            }
            r[pi] = rpi - rr;
        }
        __syncthreads();
        // trailing window: A(i, j) -= L(i, k) . L(j, k) for i0 <= j <= i < i1 (i - j <= bw holds: m <= bw); a wave per row, lanes along the
        // contiguous band storage of that row; all loads of a wave's rows are issued before the first store
        for (int a = wave; a < m; a += 16) {
            double pa[CB_NB];
#pragma unroll
            for (int c = 0; c < CB_NB; c++) pa[c] = Pn[a * (CB_NB + 1) + c];
            double* rowp = &AB(i0 + a, i0);
            for (int b = lane; b <= a; b += 64) {
                double sum = 0;
#pragma unroll
                for (int c = 0; c < CB_NB; c++) sum += pa[c] * Pn[b * (CB_NB + 1) + c];
                rowp[b] -= sum;
            }
        }
        __syncthreads();
    }
    if (ok) {                                               // backward sweep: x = L^-T z (z sits in r)
        for (int k0 = ((n - 1) / CB_NB) * CB_NB; k0 >= 0; k0 -= CB_NB) {
            const int nb = min(CB_NB, n - k0), i0 = k0 + nb, i1 = min(n, k0 + nb + bw);
            if (tid >= 512 && tid < 512 + nb * nb) { const int t = tid - 512, a = t / nb, b = t - a * nb; Ld[a][b] = b <= a ? AB(k0 + a, k0 + b) : 0.0; }
            if (wave < nb) {                                // wave c: z_c - sum_{i in panel} L(i, k0+c) x_i
                const int col = k0 + wave; double sacc = 0;
                for (int i = i0 + lane; i < i1; i += 64) if (i - col <= bw) sacc += AB(i, col) * x[i];
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) sacc += __shfl_xor(sacc, o, 64);
                if (lane == 0) zk[wave] = r[col] - sacc;
            }
            __syncthreads();
            if (tid < 64) {                                 // column-oriented triangular solve: one LDS round trip per unknown
                for (int c = nb - 1; c >= 0; c--) {
                    const double xc = zk[c] / Ld[c][c];
                    __builtin_amdgcn_wave_barrier();
                    if (tid < c) zk[tid] -= Ld[c][tid] * xc;
                    if (tid == c) { zk[c] = xc; x[k0 + c] = xc; }
                    __builtin_amdgcn_wave_barrier();
                }
            }
            __syncthreads();
        }
    }
    if (tid == 0) P.scal[4] = ok ? 1.0 : 0.0;
#undef AB
}

// Pose-block variant for short bands (the usual case: a landmark is seen from <= ~20 consecutive keyframes): pivots are the 6x6
// camera blocks, not scalars.  Every thread factors the 6x6 pivot block redundantly in registers (no broadcast, no barrier),
// panel rows are one thread each, the trailing update is one thread per (row, column block), and the (bwc+1)-block window of
// the trailing matrix lives in LDS with circular block indexing: each block row of S is read from HBM once when it enters
// the window and the factor is written once.  Two workgroup barriers per camera instead of ~10 per 16 scalar pivots.
// The backward sweep is row oriented (contiguous band rows, next row prefetched while the current one is consumed).
// Workgroup barrier that orders LDS traffic only: the factor rows streamed out to HBM inside the loop are not read again
// before the full __syncthreads() that precedes the backward sweep, so the barrier must not wait for those stores to retire
// (a plain __syncthreads() drains vmcnt: ~5 us per step, 10x the arithmetic of a step).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ bool chol6(const double* A /*21: row-major lower*/, double* L /*21*/, double* inv /*6: 1/L_cc*/)
{
    bool good = true;
#pragma unroll
    for (int c = 0; c < 6; c++) {
        double d = A[c * (c + 1) / 2 + c];
#pragma unroll
        for (int e = 0; e < c; e++) d -= L[c * (c + 1) / 2 + e] * L[c * (c + 1) / 2 + e];
        good = good && (d > 0.0) && isfinite(d);
        double iv = __builtin_amdgcn_rsq(d);
        iv = iv * (1.5 - 0.5 * d * iv * iv); iv = iv * (1.5 - 0.5 * d * iv * iv);
        inv[c] = iv; L[c * (c + 1) / 2 + c] = d * iv;
#pragma unroll
        for (int a = c + 1; a < 6; a++) {
            double v = A[a * (a + 1) / 2 + c];
#pragma unroll
            for (int e = 0; e < c; e++) v -= L[a * (a + 1) / 2 + e] * L[c * (c + 1) / 2 + e];
            L[a * (a + 1) / 2 + c] = v * iv;
        }
    }
    return good;
}
__device__ __forceinline__ double readlane_f64(double v, int src /* wave-uniform */)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
}
// -DVIDO_CHOL_PROF: per-phase shader-clock totals of the two pose-block Cholesky kernels, printed by one lane of role F and one of role T
#ifdef VIDO_CHOL_PROF
#define CH_PROF_DECL long long ch_tp[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long ch_t0 = clock64();
#define CH_TICK(i) { const long long ch_t1 = clock64(); ch_tp[i] += ch_t1 - ch_t0; ch_t0 = ch_t1; }
#define CH_PROF_PRINT(name) if (tid == 0 || tid == 130 || tid == 200) printf(name " tid %d: %lld %lld %lld %lld %lld %lld %lld %lld\n", tid, ch_tp[0], ch_tp[1], ch_tp[2], ch_tp[3], ch_tp[4], ch_tp[5], ch_tp[6], ch_tp[7]);
#else
#define CH_PROF_DECL
#define CH_TICK(i)
#define CH_PROF_PRINT(name)
#endif
// 12 waves: 168 VGPRs per lane (the pivot block, its factor and a 3x6 tile with its operands do not fit the 128 of a 1024-thread group)
#define CH_NT 768
// 3x6 half tile of the trailing update: out[a][b] = sum_c Li[a][c] * Lj[b][c] from the panel rows in LDS (pitch 7): 54 LDS reads per 108 FMAs
// (one thread per (row, column block) re-read the 36 values of Lj for every row: the update ran at the LDS bandwidth limit)
template <int UNROLL = 6>
__device__ __forceinline__ void tile36(const double* Li, const double* Lj, double (*o)[6])
{
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 6; b++) o[a][b] = 0.0;
#pragma unroll UNROLL
    for (int c = 0; c < 6; c++) {
        const double l0 = Li[c], l1 = Li[7 + c], l2 = Li[14 + c];
#pragma unroll
        for (int b = 0; b < 6; b++) { const double lj = Lj[b * 7 + c]; o[0][b] += l0 * lj; o[1][b] += l1 * lj; o[2][b] += l2 * lj; }
    }
}
__global__ __launch_bounds__(CH_NT) void k_chol_band6(BaDev P, int bwc /* block half-bandwidth: bw = 6*bwc + 5 */)
{
    extern __shared__ double cb6[];
    const int Wb = bwc + 1, Wr = 6 * Wb, ldw = Wr + 1;
    double* W = cb6;                                        // [Wr][ldw]  block (ib, jb) at physical blocks (pb(ib), pb(jb))
    double* rW = W + (size_t)Wr * ldw;                       // [Wr]  rhs of the window rows
    double* Pn = rW + Wr;                                    // [6*bwc][7]  panel rows of L of the current step (window-relative row)
    double* xs = Pn + (size_t)6 * bwc * 7;                   // [Wr]  backward sweep: x of the blocks below (circular)
    double* acc = xs + Wr;                                   // [Wr]  backward sweep: sum_i L_ik^T x_i accumulators (circular)
    __shared__ int ok;
    __shared__ double stg[27];                              // factor of the current pivot block | its z, on the way to HBM
    const int n = P.n6, nblk = n / 6, bw = P.bw, ldb = P.ldb, tid = threadIdx.x;
    double* Sg = P.S; double* r = P.r; double* x = P.x;
#define AB(i, j) Sg[(size_t)(i) * ldb + ((j) - (i) + bw)]
#define PB(ib) (((ib) - kb + boff) >= Wb ? ((ib) - kb + boff - Wb) : ((ib) - kb + boff))
    if (tid == 0) ok = 1;
    CH_PROF_DECL
    int boff = 0;
    {   // block rows [0, min(nblk, Wb)) enter
        const int kb = 0;
        for (int t = tid; t < min(nblk, Wb) * 6 * Wr; t += CH_NT) {
            const int i = t / Wr, jj = t - i * Wr, ib = i / 6, j = 6 * (ib - bwc) + jj;          // jj-th in-band column slot of row i (by block)
            if (j >= 0 && j <= i) W[(6 * PB(ib) + i % 6) * ldw + 6 * PB(j / 6) + j % 6] = AB(i, j);
        }
        for (int i = tid; i < min(nblk, Wb) * 6; i += CH_NT) rW[6 * PB(i / 6) + i % 6] = r[i];
    }
    // per-thread entry of the block row that enters the window: row a, in-band column slot jj (6*Wr <= CH_NT up to bwc = 20)
    const int e_a = tid / Wr, e_jj = tid - e_a * Wr;
    const bool e_two = 6 * Wr > CH_NT; const int e_a2 = (tid + CH_NT) / Wr, e_jj2 = tid + CH_NT - e_a2 * Wr;
    __syncthreads();
    // Look-ahead schedule with role-specialised waves (see k_ba_chol_small6): waves 0..2 (role F) own the pivot blocks, the panel rows and
    // the factor write-out; waves 3.. (role T) own the trailing tiles.  The roles run SEPARATE loops that meet at the same two LDS barriers
    // per pivot, so the 33 doubles of pivot state are live only in role F and role T keeps its tile operands in registers.  (In one shared
    // loop the allocator spilled; a spill reload waits on vmcnt(0), i.e. on the window-row prefetch and the factor stores still in flight —
    // 3000 cycles of HBM latency back on the critical path of every pivot.)
#define BAND6_STEP_BEGIN \
        const int pk = 6 * boff, nbelow = min(bwc, nblk - 1 - kb); \
        double e_val = 0, e_val2 = 0, e_r = 0; const int e_ib = kb + Wb, e_i = 6 * e_ib + e_a, e_j = 6 * (e_ib - bwc) + e_jj, e_i2 = 6 * e_ib + e_a2, e_j2 = 6 * (e_ib - bwc) + e_jj2; \
        const bool e_on = e_ib < nblk && e_a < 6 && e_j <= e_i, e_on2 = e_two && e_ib < nblk && e_a2 < 6 && e_j2 <= e_i2; \
        if (e_on) e_val = AB(e_i, e_j); \
        if (e_on2) e_val2 = AB(e_i2, e_j2); \
        if (e_ib < nblk && tid < 6) e_r = r[6 * e_ib + tid];
    // D0 above: the loads of block row kb + Wb are issued at the top of the step and land in LDS at its end (HBM latency off the critical path).
    // D below: the row reuses the storage of block row kb, which nothing in the step touches any more.
#define BAND6_STEP_END \
        if (e_on) { const int jb = e_j / 6; W[(pk + e_a) * ldw + 6 * (jb == e_ib ? boff : PB(jb)) + e_j % 6] = e_val; } \
        if (e_on2) { const int jb = e_j2 / 6; W[(pk + e_a2) * ldw + 6 * (jb == e_ib ? boff : PB(jb)) + e_j2 % 6] = e_val2; } \
        if (e_ib < nblk && tid < 6) rW[pk + tid] = e_r; \
        lds_barrier();
    if (tid < 192) {
        double Lk[21], inv[6], zk[6];
        {   // pivot block 0 (physical block 0)
            double Akk[21];
#pragma unroll
            for (int a = 0; a < 6; a++)
#pragma unroll
                for (int b = 0; b <= a; b++) Akk[a * (a + 1) / 2 + b] = W[a * ldw + b];
            const bool good = chol6(Akk, Lk, inv);
            if (!good && tid == 0) ok = 0;
#pragma unroll
            for (int c = 0; c < 6; c++) { double v = rW[c];
#pragma unroll
                for (int e = 0; e < c; e++) v -= Lk[c * (c + 1) / 2 + e] * zk[e]; zk[c] = v * inv[c]; }
        }
        lds_barrier();
        for (int kb = 0; kb < nblk && ok; kb++, boff = (boff + 1 >= Wb ? 0 : boff + 1)) {
            CH_TICK(0)
            BAND6_STEP_BEGIN
            CH_TICK(1)
            // ---- B1: panel rows, final factor + z of block kb out to HBM
            if (tid < 6 * nbelow) {
                const int ibr = tid / 6 + 1, a = tid - (ibr - 1) * 6, ib = kb + ibr, prow = 6 * PB(ib) + a, i = 6 * ib + a;
                double l[6], rr = 0, w6[6];
#pragma unroll
                for (int c = 0; c < 6; c++) w6[c] = W[prow * ldw + pk + c];
                const double r0 = rW[prow];
#pragma unroll
                for (int c = 0; c < 6; c++) {
                    double v = w6[c];
#pragma unroll
                    for (int e = 0; e < c; e++) v -= l[e] * Lk[c * (c + 1) / 2 + e];
                    v *= inv[c]; l[c] = v; rr += v * zk[c];
                }
#pragma unroll
                for (int c = 0; c < 6; c++) { Pn[tid * 7 + c] = l[c]; AB(i, 6 * kb + c) = l[c]; }
                rW[prow] = r0 - rr;
            } else if (tid == 128) {                             // factor of the pivot block and its z: one lane parks the 27 values in LDS with
#pragma unroll                                               // compile-time register indices, lanes 704..730 of role T stream them to HBM
                for (int i = 0; i < 21; i++) stg[i] = Lk[i];
#pragma unroll
                for (int c = 0; c < 6; c++) stg[c * (c + 1) / 2 + c] = inv[c];                 // diagonal: 1 / L_cc for the back sweep (no FP64 division in its chain)
#pragma unroll
                for (int c = 0; c < 6; c++) stg[21 + c] = zk[c];
            }
            CH_TICK(2)
            lds_barrier();
            CH_TICK(3)
            // ---- A: pivot block kb+1 = W block - P0 P0^T (P0 = panel rows 0..5 of this step), its factor and z
            if (nbelow > 0) {
                const int p1 = 6 * PB(kb + 1);
                // lane t < 21 of each wave forms entry t of the updated block, v_readlane hands all 21 to every lane
                double Akk[21], mine;
                { const int t = min(tid & 63, 20), a = (t >= 1) + (t >= 3) + (t >= 6) + (t >= 10) + (t >= 15), b = t - a * (a + 1) / 2;
                  double sum = 0;
#pragma unroll
                  for (int c = 0; c < 6; c++) sum += Pn[a * 7 + c] * Pn[b * 7 + c];
                  mine = W[(p1 + a) * ldw + p1 + b] - sum; }
#pragma unroll
                for (int i = 0; i < 21; i++) Akk[i] = readlane_f64(mine, i);
                const bool good = chol6(Akk, Lk, inv);
                if (!good && tid == 0) ok = 0;
#pragma unroll
                for (int c = 0; c < 6; c++) { double v = rW[p1 + c];
#pragma unroll
                    for (int e = 0; e < c; e++) v -= Lk[c * (c + 1) / 2 + e] * zk[e]; zk[c] = v * inv[c]; }
            }
            CH_TICK(4)
            BAND6_STEP_END
        }
    } else {
        lds_barrier();
        for (int kb = 0; kb < nblk && ok; kb++, boff = (boff + 1 >= Wb ? 0 : boff + 1)) {
            CH_TICK(0)
            BAND6_STEP_BEGIN
            CH_TICK(1)
            CH_TICK(2)
            lds_barrier();
            CH_TICK(3)
            if (tid >= 704 && tid < 704 + 27) {
                const int t = tid - 704, a = (t >= 1) + (t >= 3) + (t >= 6) + (t >= 10) + (t >= 15), b = t - a * (a + 1) / 2;
                double* dst = t < 21 ? &AB(6 * kb + a, 6 * kb + b) : r + 6 * kb + (t - 21);
                *dst = stg[t];
            }
            // ---- B2: trailing window (LDS only): tiles (row block ibr >= column block jc) except (0, 0), two 3-row halves, packed densely
            for (int it = tid - 192; it < bwc * (bwc + 1) - 2; it += CH_NT - 192) {
                const int tl = (it >> 1) + 1, h3 = 3 * (it & 1);
                int ibr = (int)((sqrtf(8.f * (float)tl + 1.f) - 1.f) * 0.5f);
                ibr -= (ibr * (ibr + 1) / 2 > tl); ibr += ((ibr + 1) * (ibr + 2) / 2 <= tl);
                const int jc = tl - ibr * (ibr + 1) / 2;
                if (ibr >= nbelow) continue;
                double o[3][6];
                tile36(Pn + (6 * ibr + h3) * 7, Pn + 6 * jc * 7, o);
                double* Wt = W + (size_t)(6 * PB(kb + 1 + ibr) + h3) * ldw + 6 * PB(kb + 1 + jc);
                double w[3][6];                  // all 18 loads first: a load issued after a store waits a full LDS round trip for that store's operand
#pragma unroll
                for (int a = 0; a < 3; a++)
#pragma unroll
                    for (int b = 0; b < 6; b++) w[a][b] = Wt[a * ldw + b];
#pragma unroll
                for (int a = 0; a < 3; a++)
#pragma unroll
                    for (int b = 0; b < 6; b++) Wt[a * ldw + b] = (jc == ibr && b > h3 + a) ? w[a][b] : w[a][b] - o[a][b];     // select, not a branch
            }
            CH_TICK(4)
            BAND6_STEP_END
        }
    }
#undef BAND6_STEP_BEGIN
#undef BAND6_STEP_END
    CH_TICK(5)
    __syncthreads();
    if (ok) {   // ---- backward sweep, row oriented: x_k = L_kk^-T (z_k - acc_k); then acc_j += L_kj^T x_k for the blocks j < k of row k.
        // Everything a step reads from HBM was requested one step earlier: the band row of a thread's column stays in registers, the diagonal
        // block and z (27 values, one per lane of wave 0) are parked in LDS.
        __shared__ double dz[27];
        for (int t = tid; t < Wr; t += CH_NT) acc[t] = 0.0;
        double row[6], nrow[6], nrow2[6], nval = 0, nval2 = 0;      // two steps of prefetch in flight: one step (~1000 cycles) is shorter than an L2 round trip under load
        const int dz_a = (tid >= 1) + (tid >= 3) + (tid >= 6) + (tid >= 10) + (tid >= 15), dz_b = tid - dz_a * (dz_a + 1) / 2;
        auto fetch = [&](int kb, double* frow, double& fval) {
            if (tid < 27) fval = tid < 21 ? AB(6 * kb + dz_a, 6 * kb + dz_b) : r[6 * kb + tid - 21];
            const int ncols = 6 * min(bwc, kb);
            if (tid < ncols) {
                const int j = 6 * kb - ncols + tid;
#pragma unroll
                for (int a = 0; a < 6; a++) frow[a] = AB(6 * kb + a, j);
            }
        };
        fetch(nblk - 1, row, nval);
        if (tid < 27) dz[tid] = nval;
        if (nblk > 1) fetch(nblk - 2, nrow, nval);
        __syncthreads();
        int boff2 = 0;                                       // physical slot of block kb in the circular acc / xs arrays
        for (int kb = nblk - 1; kb >= 0; kb--, boff2 = (boff2 + 1 >= Wb ? 0 : boff2 + 1)) {
            if (kb > 1) fetch(kb - 2, nrow2, nval2);
            if (tid < 64) {                                  // wave 0: x_k from the diagonal block of row-block kb
                double Lk[21], t6[6];
#pragma unroll
                for (int i = 0; i < 21; i++) Lk[i] = dz[i];
#pragma unroll
                for (int c = 0; c < 6; c++) t6[c] = dz[21 + c] - acc[6 * boff2 + c];
#pragma unroll
                for (int c = 5; c >= 0; c--) { double v = t6[c];
#pragma unroll
                    for (int e = c + 1; e < 6; e++) v -= Lk[e * (e + 1) / 2 + c] * t6[e]; t6[c] = v * Lk[c * (c + 1) / 2 + c]; }
                if (tid < 6) { double v = t6[5];
#pragma unroll
                    for (int c = 0; c < 5; c++) v = (tid == c) ? t6[c] : v;
                    x[6 * kb + tid] = v; xs[tid] = v; }
                if (tid < 27) dz[tid] = nval;                // block kb-1 for the next step (every lane of this wave has read dz above)
            }
            lds_barrier();
            // acc_j += L(kb-row a, col) * x_kb[a] for the in-band columns left of the diagonal block; slot of block j = boff2 + (kb - j)
            const int ncols = 6 * min(bwc, kb);
            if (tid < ncols) {
                const int j = 6 * kb - ncols + tid, jb = j / 6;
                double sum = 0;
#pragma unroll
                for (int a = 0; a < 6; a++) sum += row[a] * xs[a];
                int slot = boff2 + (kb - jb); if (slot >= Wb) slot -= Wb;
                acc[6 * slot + j % 6] += sum;
            }
            if (tid >= 512 && tid < 518) acc[6 * boff2 + tid - 512] = 0.0;     // this slot becomes block kb - Wb
            lds_barrier();
#pragma unroll
            for (int i = 0; i < 6; i++) { row[i] = nrow[i]; nrow[i] = nrow2[i]; }
            nval = nval2;
        }
    }
    CH_TICK(6)
    CH_PROF_PRINT("band6 [0 D+bar | 1 D0 | 2 B1 | 3 bar | 4 A or B2 | 6 back]")
    if (tid == 0) P.scal[4] = ok ? 1.0 : 0.0;
#undef AB
#undef PB
}

#define CG_NT 768
// Supernodal form of the in-place band factorisation: SNB pivot blocks at a time.  The 6*SNB panel columns (rows down to the end of the last
// pivot's band) are staged in LDS, factored there with the look-ahead scheme of k_ba_chol_small6 (LDS barriers only), written back once, and
// the trailing window in HBM gets ONE rank-6*SNB update per supernode: the window's loads, stores and the store drain (one HBM
// round trip each; a per-pivot version of this kernel spent 26k cycles per pivot on them) are paid once per SNB pivots.
template <int SNB>
__global__ __launch_bounds__(CG_NT) void k_chol_band6s(BaDev P, int bwc /* block half-bandwidth: bw = 6*bwc + 5 */)
{
    extern __shared__ double cs6s[];
    constexpr int PC = 6 * SNB + 1;                           // panel pitch (odd)
    const int Wb = bwc + 1, Wr = 6 * Wb, R = 6 * (bwc + SNB);
    double* Pp = cs6s;                                       // [R][PC]  panel: row i0 + row, column 6*k0 + col
    double* rWs = Pp + (size_t)R * PC;                       // [R]      rhs of the panel rows
    double* xs = rWs + R;                                    // [Wr]     backward sweep
    double* acc = xs + Wr;                                   // [Wr]
    __shared__ int ok;
    const int n = P.n6, nblk = n / 6, bw = P.bw, ldb = P.ldb, tid = threadIdx.x;
    double* Sg = P.S; double* r = P.r; double* x = P.x;
#define AB(i, j) Sg[(size_t)(i) * ldb + ((j) - (i) + bw)]
    if (tid == 0) ok = 1;
    const int f_nt = (6 * bwc + 63) & ~63;                   // threads of role F (whole waves): one per panel row of a pivot
    __syncthreads();
    for (int k0 = 0; k0 < nblk && ok; k0 += SNB) {
        const int nb = min(SNB, nblk - k0), i0 = 6 * k0, nrb = min(nblk, k0 + nb + bwc) - k0, m = 6 * nrb, nc = 6 * nb;   // nrb row blocks, nc panel columns
        // ---- (1) panel and its rhs: HBM -> LDS (everything an earlier supernode did to them reached L2 before the barrier that ended it)
        for (int t = tid; t < m * nc; t += CG_NT) {
            const int row = t / nc, col = t - row * nc, i = i0 + row, j = i0 + col;
            Pp[row * PC + col] = (j <= i && i - j <= bw) ? AB(i, j) : 0.0;
        }
        for (int t = tid; t < m; t += CG_NT) rWs[t] = r[i0 + t];
        __syncthreads();
        // ---- (2) the nb pivots of the supernode, in LDS
        if (tid < f_nt) {
            double Lk[21], inv[6], zk[6];
            {   // first pivot block of the supernode straight from the panel
                double Akk[21];
#pragma unroll
                for (int a = 0; a < 6; a++)
#pragma unroll
                    for (int b = 0; b <= a; b++) Akk[a * (a + 1) / 2 + b] = Pp[a * PC + b];
                const bool good = chol6(Akk, Lk, inv);
                if (!good && tid == 0) ok = 0;
#pragma unroll
                for (int c = 0; c < 6; c++) { double v = rWs[c];
#pragma unroll
                    for (int e = 0; e < c; e++) v -= Lk[c * (c + 1) / 2 + e] * zk[e]; zk[c] = v * inv[c]; }
            }
            for (int p = 0; p < nb; p++) {
                const int nbelow = min(bwc, nrb - 1 - p), cp = 6 * p;
                if (tid < 6 * nbelow) {                      // B1: panel rows of pivot p
                    const int row = cp + 6 + tid;
                    double l[6], rr = 0, w6[6];
#pragma unroll
                    for (int c = 0; c < 6; c++) w6[c] = Pp[row * PC + cp + c];
                    const double r0 = rWs[row];
#pragma unroll
                    for (int c = 0; c < 6; c++) {
                        double v = w6[c];
#pragma unroll
                        for (int e = 0; e < c; e++) v -= l[e] * Lk[c * (c + 1) / 2 + e];
                        v *= inv[c]; l[c] = v; rr += v * zk[c];
                    }
#pragma unroll
                    for (int c = 0; c < 6; c++) Pp[row * PC + cp + c] = l[c];
                    rWs[row] = r0 - rr;
                }
                if (tid == f_nt - 1) {                       // factor of the pivot block (diagonal: 1 / L_cc for the back sweep) and its z; a lane that owns no panel row
#pragma unroll
                    for (int a = 0; a < 6; a++) {
#pragma unroll
                        for (int b = 0; b <= a; b++) Pp[(cp + a) * PC + cp + b] = b == a ? inv[a] : Lk[a * (a + 1) / 2 + b];
                        rWs[cp + a] = zk[a];
                    }
                }
                lds_barrier();
                if (p + 1 < nb) {                            // A: pivot block p+1 = its panel entries - P0 P0^T
                    const int c1 = cp + 6;
                    double Akk[21], mine;
                    { const int t = min(tid & 63, 20), a = (t >= 1) + (t >= 3) + (t >= 6) + (t >= 10) + (t >= 15), b = t - a * (a + 1) / 2;
                      double sum = 0;
#pragma unroll
                      for (int c = 0; c < 6; c++) sum += Pp[(c1 + a) * PC + cp + c] * Pp[(c1 + b) * PC + cp + c];
                      mine = Pp[(c1 + a) * PC + c1 + b] - sum; }
#pragma unroll
                    for (int i = 0; i < 21; i++) Akk[i] = readlane_f64(mine, i);
                    const bool good = chol6(Akk, Lk, inv);
                    if (!good && tid == 0) ok = 0;
#pragma unroll
                    for (int c = 0; c < 6; c++) { double v = rWs[c1 + c];
#pragma unroll
                        for (int e = 0; e < c; e++) v -= Lk[c * (c + 1) / 2 + e] * zk[e]; zk[c] = v * inv[c]; }
                }
                lds_barrier();
            }
        } else {
            for (int p = 0; p < nb; p++) {
                const int nbelow = min(bwc, nrb - 1 - p), cp = 6 * p, nq = nb - 1 - p;
                lds_barrier();
                // trailing update inside the panel: column blocks q = p+1 .. nb-1, row blocks ib = q .. p+nbelow, two 3-row halves; not (p+1, p+1)
                const float inv_rows = 1.0f / (float)max(2 * nbelow, 1);
                for (int it = tid - f_nt; it < nq * 2 * nbelow; it += CG_NT - f_nt) {
                    const int qi = (int)(((float)it + 0.5f) * inv_rows), rem = it - qi * 2 * nbelow, ib = p + 1 + (rem >> 1), h3 = 3 * (rem & 1), q = p + 1 + qi;
                    if (ib < q || (ib == p + 1 && q == p + 1)) continue;
                    const double* Li = Pp + (6 * ib + h3) * PC + cp; const double* Lj = Pp + 6 * q * PC + cp;
                    double o[3][6], w[3][6];
#pragma unroll
                    for (int a = 0; a < 3; a++)
#pragma unroll
                        for (int b = 0; b < 6; b++) { o[a][b] = 0.0; w[a][b] = Pp[(6 * ib + h3 + a) * PC + 6 * q + b]; }
#pragma unroll
                    for (int c = 0; c < 6; c++) {
                        const double l0 = Li[c], l1 = Li[PC + c], l2 = Li[2 * PC + c];
#pragma unroll
                        for (int b = 0; b < 6; b++) { const double lj = Lj[b * PC + c]; o[0][b] += l0 * lj; o[1][b] += l1 * lj; o[2][b] += l2 * lj; }
                    }
#pragma unroll
                    for (int a = 0; a < 3; a++)
#pragma unroll
                        for (int b = 0; b < 6; b++) Pp[(6 * ib + h3 + a) * PC + 6 * q + b] = (q == ib && b > h3 + a) ? w[a][b] : w[a][b] - o[a][b];
                }
                lds_barrier();
            }
        }
        if (!ok) break;
        // ---- (3) the factored panel and the rhs back to HBM
        for (int t = tid; t < m * nc; t += CG_NT) {
            const int row = t / nc, col = t - row * nc, i = i0 + row, j = i0 + col;
            if (j <= i && i - j <= bw) AB(i, j) = Pp[row * PC + col];
        }
        for (int t = tid; t < m; t += CG_NT) r[i0 + t] = rWs[t];
        // ---- (4) rank-6*nb update of the window beyond the panel columns, in place in HBM, on the FP64 matrix cores: C -= A A^T over 16x16 tiles
        // of the lower triangle (v_mfma_f64_16x16x4: lane l feeds A[l & 15][l >> 4] and B[l >> 4][l & 15], holds C[(l >> 4) + 4 v][l & 15],
        // v = 0..3).  The C layout is what matters here: a register of a wave is 4 rows x 16 consecutive doubles of the band — 3x6 register
        // tiles on the vector ALUs touched one cache line per lane per load (the TA line-request rate, not arithmetic, set their pace) and
        // read each panel value from LDS once per 2 FMAs instead of once per 8.
        {
            typedef double d4 __attribute__((ext_vector_type(4)));
            const int D0 = nc, Dn = m - D0, nT = (Dn + 15) >> 4;
            const int wave = tid >> 6, lane = tid & 63, lr = lane & 15, lk = lane >> 4;
            // a wave takes units of up to 4 tiles of one tile row (same A operand, 4 independent accumulators: the C loads of a unit are one
            // HBM round trip, the MFMAs of a k step do not wait for each other); unit u -> (tile row I, tiles 4g .. 4g+3 of it)
            int n_units = 0; for (int I = 0; I < nT; I++) n_units += (I + 4) >> 2;
            for (int u = wave; u < n_units; u += CG_NT / 64) {
                int I = 0, first = 0;
                while (u >= first + ((I + 4) >> 2)) { first += (I + 4) >> 2; I++; }
                const int J0 = 4 * (u - first), ntl = min(4, I + 1 - J0), rI = D0 + 16 * I;
                d4 c[4];
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const int col = D0 + 16 * (J0 + t) + lr;
#pragma unroll
                    for (int v = 0; v < 4; v++) { const int row = rI + lk + 4 * v; c[t][v] = (t < ntl && row < m && col <= row) ? AB(i0 + row, i0 + col) : 0.0; }
                }
                const double* pa = Pp + (size_t)min(rI + lr, m - 1) * PC + lk;       // rows / columns past the window are masked on the way out; the clamp keeps the reads inside the panel
                const double* pb[4];
#pragma unroll
                for (int t = 0; t < 4; t++) pb[t] = Pp + (size_t)min(D0 + 16 * (J0 + t) + lr, m - 1) * PC + lk;
                for (int ks = 0; ks < nc; ks += 4) {
                    const bool kin = ks + lk < nc;                                   // nc = 6, 12, 18 in the last supernode: not a multiple of 4
                    const double av = kin ? -pa[ks] : 0.0;
                    double bv[4];
#pragma unroll
                    for (int t = 0; t < 4; t++) bv[t] = kin ? pb[t][ks] : 0.0;
#pragma unroll
                    for (int t = 0; t < 4; t++) c[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv[t], c[t], 0, 0, 0);
                }
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const int col = D0 + 16 * (J0 + t) + lr;
#pragma unroll
                    for (int v = 0; v < 4; v++) { const int row = rI + lk + 4 * v; if (t < ntl && row < m && col <= row) AB(i0 + row, i0 + col) = c[t][v]; }
                }
            }
        }
        __syncthreads();                                      // drains the window stores: the next supernode loads its panel from them
    }
    __syncthreads();
    if (ok) {   // ---- backward sweep (as in k_chol_band6)
        __shared__ double dz[27];
        for (int t = tid; t < Wr; t += CG_NT) acc[t] = 0.0;
        double row[6], nrow[6], nrow2[6], nval = 0, nval2 = 0;      // two steps of prefetch in flight
        const int dz_a = (tid >= 1) + (tid >= 3) + (tid >= 6) + (tid >= 10) + (tid >= 15), dz_b = tid - dz_a * (dz_a + 1) / 2;
        auto fetch = [&](int kb, double* frow, double& fval) {
            if (tid < 27) fval = tid < 21 ? AB(6 * kb + dz_a, 6 * kb + dz_b) : r[6 * kb + tid - 21];
            const int ncols = 6 * min(bwc, kb);
            if (tid < ncols) {
                const int j = 6 * kb - ncols + tid;
#pragma unroll
                for (int a = 0; a < 6; a++) frow[a] = AB(6 * kb + a, j);
            }
        };
        fetch(nblk - 1, row, nval);
        if (tid < 27) dz[tid] = nval;
        if (nblk > 1) fetch(nblk - 2, nrow, nval);
        __syncthreads();
        int boff2 = 0;
        for (int kb = nblk - 1; kb >= 0; kb--, boff2 = (boff2 + 1 >= Wb ? 0 : boff2 + 1)) {
            if (kb > 1) fetch(kb - 2, nrow2, nval2);
            if (tid < 64) {
                double Lk[21], t6[6];
#pragma unroll
                for (int i = 0; i < 21; i++) Lk[i] = dz[i];
#pragma unroll
                for (int c = 0; c < 6; c++) t6[c] = dz[21 + c] - acc[6 * boff2 + c];
#pragma unroll
                for (int c = 5; c >= 0; c--) { double v = t6[c];
#pragma unroll
                    for (int e = c + 1; e < 6; e++) v -= Lk[e * (e + 1) / 2 + c] * t6[e]; t6[c] = v * Lk[c * (c + 1) / 2 + c]; }
                if (tid < 6) { double v = t6[5];
#pragma unroll
                    for (int c = 0; c < 5; c++) v = (tid == c) ? t6[c] : v;
                    x[6 * kb + tid] = v; xs[tid] = v; }
                if (tid < 27) dz[tid] = nval;
            }
            lds_barrier();
            const int ncols = 6 * min(bwc, kb);
            if (tid < ncols) {
                const int j = 6 * kb - ncols + tid, jb = j / 6;
                double sum = 0;
#pragma unroll
                for (int a = 0; a < 6; a++) sum += row[a] * xs[a];
                int slot = boff2 + (kb - jb); if (slot >= Wb) slot -= Wb;
                acc[6 * slot + j % 6] += sum;
            }
            if (tid >= 704 && tid < 710) acc[6 * boff2 + tid - 704] = 0.0;       // this slot becomes block kb - Wb
            lds_barrier();
#pragma unroll
            for (int i = 0; i < 6; i++) { row[i] = nrow[i]; nrow[i] = nrow2[i]; }
            nval = nval2;
        }
    }
    if (tid == 0) P.scal[4] = ok ? 1.0 : 0.0;
#undef AB
}

// Local-BA reduced solve (n6 <= 120, dense): the same pose-block scheme as k_chol_band6 on the whole matrix in LDS —
// 20 block pivots instead of 120 scalar pivots.  (Factoring the pivot block redundantly on the three waves that use it is
// faster than one wave + publish: the dependent FP64 chain is latency-bound and the copies run on different SIMDs.)
// trial pose of camera c: cam_new = cam (+) d (the step's six entries), returns the camera's part of computeScale
__device__ double ba_update_cam(const BaDev& P, int c, const double* d, double lambda)
{
    const double* X = P.cam + 12 * c; double* N = P.cam_new + 12 * c;
    double w = 1 - (d[3] * d[3] + d[4] * d[4] + d[5] * d[5]);
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (!(w < 0)) {
        w = sqrt(w);
        const double x = d[3], y = d[4], z = d[5];
        R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w); R[2] = 2 * (x * z + y * w);
        R[3] = 2 * (x * y + z * w); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
        R[6] = 2 * (x * z - y * w); R[7] = 2 * (y * z + x * w); R[8] = 1 - 2 * (x * x + y * y);
    }
    for (int r = 0; r < 3; r++) {
        for (int q = 0; q < 3; q++) N[r * 4 + q] = X[r * 4] * R[q] + X[r * 4 + 1] * R[3 + q] + X[r * 4 + 2] * R[6 + q];
        N[r * 4 + 3] = X[r * 4] * d[0] + X[r * 4 + 1] * d[1] + X[r * 4 + 2] * d[2] + X[r * 4 + 3];
    }
    double sc = 0;
    for (int a = 0; a < 6; a++) sc += d[a] * (lambda * d[a] + P.bc[6 * c + a]);
    return sc;
}

// NT threads (>= 256): 768 in k_ba_chol_small6 (every tile of the first trailing update has its own thread), 512 inside the persistent solver (the first two steps take two rounds)
template <int NT>
__device__ __forceinline__ void ba_chol_small6_body(const BaDev& P, int fuse_update, double lambda, double* cs6)
{
    const int n = P.n6, nblk = n / 6, ldw = n + 1, tid = threadIdx.x;
    double* W = cs6;                                        // [n][ldw] lower triangle
    double* rW = W + (size_t)n * ldw;                        // [n] rhs -> z -> x
    double* Pn = rW + n;                                     // [n][7] panel rows of the current step
    __shared__ int ok;
    for (int t = tid; t < n * n; t += NT) { const int i = t / n, j = t - i * n; if (j <= i) W[i * ldw + j] = P.S[t]; }
    for (int t = tid; t < n; t += NT) rW[t] = P.r[t];
    if (tid == 0) ok = 1;
    CH_PROF_DECL
    __syncthreads();
    // Look-ahead schedule, two LDS barriers per pivot: while waves 3.. (role T) run the trailing update of step kb, waves 0..2 (role F: panel
    // rows tid < 114, writer lanes 128..154) already factor pivot block kb+1 — its entries are the block in W minus the first panel row block
    // times itself, which is all the trailing update would have done to it.  The latency-bound 6x6 factorisation (dependent FP64 chain,
    // ~1500 cycles) runs beside the tile work instead of before it.  The roles run separate loops that meet at the same barriers, so the pivot
    // state is live only in role F and the tile operands of role T stay in registers.
    if (tid < 192) {
        double Lk[21], inv[6], zk[6];
        {   // pivot block 0 straight from W
            double Akk[21];
#pragma unroll
            for (int a = 0; a < 6; a++)
#pragma unroll
                for (int b = 0; b <= a; b++) Akk[a * (a + 1) / 2 + b] = W[a * ldw + b];
            const bool good = chol6(Akk, Lk, inv);
#pragma unroll
            for (int c = 0; c < 6; c++) { double v = rW[c];
#pragma unroll
                for (int e = 0; e < c; e++) v -= Lk[c * (c + 1) / 2 + e] * zk[e]; zk[c] = v * inv[c]; }
            if (!good && tid == 0) ok = 0;
        }
        lds_barrier();
        for (int kb = 0; kb < nblk && ok; kb++) {
            const int pk = 6 * kb, nbelow = nblk - 1 - kb;
            CH_TICK(0)
            // ---- B1: panel rows of step kb; factor of the pivot block and its z into W / rW
            if (tid < 6 * nbelow) {
                const int prow = pk + 6 + tid;
                double l[6], rr = 0, w6[6];
#pragma unroll
                for (int c = 0; c < 6; c++) w6[c] = W[prow * ldw + pk + c];
                const double r0 = rW[prow];
#pragma unroll
                for (int c = 0; c < 6; c++) {
                    double v = w6[c];
#pragma unroll
                    for (int e = 0; e < c; e++) v -= l[e] * Lk[c * (c + 1) / 2 + e];
                    v *= inv[c]; l[c] = v; rr += v * zk[c];
                }
#pragma unroll
                for (int c = 0; c < 6; c++) { Pn[tid * 7 + c] = l[c]; W[prow * ldw + pk + c] = l[c]; }
                rW[prow] = r0 - rr;
            } else if (tid == 128) {                             // one lane, 27 stores with compile-time register indices (a 27-way select chain
#pragma unroll                                               // per lane cost the writer wave ~1000 cycles: it was the long pole of this phase)
                for (int a = 0; a < 6; a++) {
#pragma unroll
                    for (int b = 0; b <= a; b++) W[(pk + a) * ldw + pk + b] = b == a ? inv[a] : Lk[a * (a + 1) / 2 + b];      // diagonal: 1 / L_aa for the back sweep
                    rW[pk + a] = zk[a];
                }
            }
            CH_TICK(2)
            lds_barrier();
            CH_TICK(3)
            // ---- A: pivot block kb+1 = W block - P0 P0^T (P0 = panel rows 0..5 of this step), then its factor and z
            if (nbelow > 0) {
                // lane t < 21 of each wave forms entry t of the updated block, v_readlane hands all 21 to every lane
                double Akk[21], mine;
                { const int t = min(tid & 63, 20), a = (t >= 1) + (t >= 3) + (t >= 6) + (t >= 10) + (t >= 15), b = t - a * (a + 1) / 2;
                  double sum = 0;
#pragma unroll
                  for (int c = 0; c < 6; c++) sum += Pn[a * 7 + c] * Pn[b * 7 + c];
                  mine = W[(pk + 6 + a) * ldw + pk + 6 + b] - sum; }
#pragma unroll
                for (int i = 0; i < 21; i++) Akk[i] = readlane_f64(mine, i);
                const bool good = chol6(Akk, Lk, inv);
#pragma unroll
                for (int c = 0; c < 6; c++) { double v = rW[pk + 6 + c];
#pragma unroll
                    for (int e = 0; e < c; e++) v -= Lk[c * (c + 1) / 2 + e] * zk[e]; zk[c] = v * inv[c]; }
                if (!good && tid == 0) ok = 0;
            }
            CH_TICK(4)
            lds_barrier();
        }
    } else {
        lds_barrier();
        for (int kb = 0; kb < nblk && ok; kb++) {
            const int pk = 6 * kb, nbelow = nblk - 1 - kb;
            CH_TICK(0)
            CH_TICK(2)
            lds_barrier();
            CH_TICK(3)
            // ---- B2: trailing update: tiles (row block ibr >= column block jc) except (0, 0), two 3-row halves each, packed densely
            for (int it = tid - 192; it < nbelow * (nbelow + 1) - 2; it += NT - 192) {      // 2 half tiles per block of the lower triangle of the trailing matrix, block (0, 0) excepted
            const int tl = (it >> 1) + 1, h3 = 3 * (it & 1);
            int ibr = (int)((sqrtf(8.f * (float)tl + 1.f) - 1.f) * 0.5f);
            ibr -= (ibr * (ibr + 1) / 2 > tl); ibr += ((ibr + 1) * (ibr + 2) / 2 <= tl);
            const int jc = tl - ibr * (ibr + 1) / 2;
            if (ibr < nbelow) {
                double o[3][6];
                tile36(Pn + (6 * ibr + h3) * 7, Pn + 6 * jc * 7, o);
                double* Wt = W + (size_t)(pk + 6 + 6 * ibr + h3) * ldw + pk + 6 + 6 * jc;
                double w[3][6];                  // all 18 loads first: a load issued after a store waits a full LDS round trip for that store's operand
#pragma unroll
                for (int a = 0; a < 3; a++)
#pragma unroll
                    for (int b = 0; b < 6; b++) w[a][b] = Wt[a * ldw + b];
#pragma unroll
                for (int a = 0; a < 3; a++)
#pragma unroll
                    for (int b = 0; b < 6; b++) Wt[a * ldw + b] = (jc == ibr && b > h3 + a) ? w[a][b] : w[a][b] - o[a][b];     // select, not a branch
            }
            }
            CH_TICK(4)
            lds_barrier();
        }
    }
    CH_TICK(5)
    __syncthreads();
    if (ok) {                                                // backward sweep in LDS, row oriented: x_k = L_kk^-T (z_k - acc_k), acc_j += L_kj^T x_k
        for (int kb = nblk - 1; kb >= 0; kb--) {
            const int pk = 6 * kb;
            if (tid < 64) {
                double Lk[21], t6[6];
#pragma unroll
                for (int a = 0; a < 6; a++)
#pragma unroll
                    for (int b = 0; b <= a; b++) Lk[a * (a + 1) / 2 + b] = W[(pk + a) * ldw + pk + b];
#pragma unroll
                for (int c = 0; c < 6; c++) t6[c] = rW[pk + c];
#pragma unroll
                for (int c = 5; c >= 0; c--) { double v = t6[c];
#pragma unroll
                    for (int e = c + 1; e < 6; e++) v -= Lk[e * (e + 1) / 2 + c] * t6[e]; t6[c] = v * Lk[c * (c + 1) / 2 + c]; }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int c = 0; c < 6; c++) if (tid == c) rW[pk + c] = t6[c];
            }
            lds_barrier();
            for (int j = tid; j < pk; j += NT) {
                double sum = 0;
#pragma unroll
                for (int a = 0; a < 6; a++) sum += W[(pk + a) * ldw + j] * rW[pk + a];
                rW[j] -= sum;
            }
            lds_barrier();
        }
        for (int t = tid; t < n; t += NT) P.x[t] = rW[t];
    }
    CH_TICK(6)
    CH_PROF_PRINT("small6 [0 bar | 2 B1 | 3 bar | 4 A or B2 | 6 back]")
    if (tid == 0) P.scal[4] = ok ? 1.0 : 0.0;
    // the trial poses, straight from the solution in LDS (k_ba_update_cams as the tail of this workgroup: one launch less per LM trial; <= 20 cameras here, one wave)
    if (fuse_update && tid < 64) {
        double sc = (ok && tid < P.n_cam) ? ba_update_cam(P, tid, rW + 6 * tid, lambda) : 0.0;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) sc += __shfl_xor(sc, o, 64);
        if (tid == 0 && sc != 0) atomicAdd(P.scal + 3, sc);
    }
}
__global__ __launch_bounds__(CH_NT) void k_ba_chol_small6(BaDev P, int fuse_update, double lambda)
{
    extern __shared__ double cs6_dyn[];
    ba_chol_small6_body<CH_NT>(P, fuse_update, lambda, cs6_dyn);
}

// ---- reduced solve by block cyclic reduction (round 2) ---------------------------------------------------------------------------------------
// The banded reduced camera system of a sequential map (n6 = 3000, half-bandwidth 65 at 500 keyframes with 10-frame tracks) is tiny in FLOPs (~11 MFLOP
// to factor) but a band Cholesky is a chain of n6/6 = 500 dependent pivot steps: k_chol_band6 spends 1.55 ms on ONE workgroup, replicated on every rank of
// a sharded solve (64 % of a global-BA iteration).  Grouping the unknowns into superblocks of m = 6 (bwc + 1) >= bw + 1 scalars makes the matrix block
// TRIDIAGONAL (nb = ceil(n6 / m) superblocks: 46 of 66), and cyclic reduction eliminates every other superblock of the current chain in parallel:
//   level s (stride): blocks p = s mod 2s are eliminated — D_p = C C^T factored in LDS, GL_p = D_p^-1 L_p, GR_p = D_p^-1 L_{p+s}^T, y_p = D_p^-1 b_p
//   (k_bcr_elim, one workgroup per eliminated block); the survivors a = 0 mod 2s take the Schur complements
//   D_a -= L_a GR_{a-s} + L_{a+s}^T GL_{a+s},  b_a -= L_a y_{a-s} + L_{a+s}^T y_{a+s},  L_a' = -L_a GL_{a-s}   (k_bcr_update),
// log2(nb) levels of two launches instead of 500 dependent steps; the back-substitution x_p = y_p - GL_p x_{p-s} - GR_p x_{p+s} walks the levels back.
// Schur complements of an SPD matrix are SPD, so no pivoting is needed (block cyclic reduction is backward stable for SPD systems).  L_k is the block
// (k, left neighbour of k in the current chain); symmetric storage is not exploited (the work is latency, not FLOPs).
struct BcrDev { int m, nb, n6, bw, ldb; double *D, *L0, *L1, *GL, *GR, *b, *y; const double* S; const double* r; double* x; double* ok; };
__global__ __launch_bounds__(256) void k_bcr_pack(BcrDev B)
{
    const int m = B.m, k = blockIdx.x, tid = threadIdx.x;
    double* Dk = B.D + (size_t)k * m * m; double* Lk = B.L0 + (size_t)k * m * m;
    for (int t = tid; t < m * m; t += 256) {
        const int i = t / m, j = t - i * m; const int gr = k * m + i, gc = k * m + j;
        double d = 0.0;
        if (gr < B.n6 && gc < B.n6) { const int hi = max(gr, gc), lo = min(gr, gc); if (hi - lo <= B.bw) d = B.S[(size_t)hi * B.ldb + (lo - hi + B.bw)]; }
        else if (gr == gc) d = 1.0;                                  // padding rows of the last superblock: identity
        Dk[t] = d;
        double l = 0.0;
        if (k > 0) { const int lc = (k - 1) * m + j; if (gr < B.n6 && gr - lc <= B.bw) l = B.S[(size_t)gr * B.ldb + (lc - gr + B.bw)]; }
        Lk[t] = l;
    }
    for (int t = tid; t < m; t += 256) B.b[(size_t)k * m + t] = k * m + t < B.n6 ? B.r[k * m + t] : 0.0;
    if (k == 0 && tid == 0) *B.ok = 1.0;
}
// One elimination of block cyclic reduction: X = D_p^-1 [L_p | L_{p+s}^T | b_p] (mode 0: GL_p, GR_p, y_p) or x_0 = D_0^-1 b_0 (mode 1, the last block of the chain).
// The first version (in-LDS Cholesky, then triangular solves on 8-lane groups) spent 92 us per level on ~200 dependent steps of ~0.45 us.  This one is Gaussian
// elimination by COLUMNS IN REGISTERS: thread c owns column c of [D | L | L'^T | b] (M doubles, every loop fully unrolled so that the register indices are compile-time
// constants); step k: the threads of D publish their row-k element U_kc (c >= k) in row k of an LDS array — by the symmetry of the trailing Schur complement that row IS
// the multiplier column — one barrier, every thread takes col[i] -= U_ki (col[k] / U_kk) for i > k with broadcast b128 LDS reads.  The rows of U stay in LDS, and the back
// substitution x_k = (y_k - sum_{j>k} U_kj x_j) / U_kk needs no barrier at all (every right-hand side is a thread).  M steps of ~0.1 us instead of 3M of 0.45 us.
// D is SPD (Schur complements of an SPD matrix), so elimination without pivoting is backward stable, like the Cholesky it replaces.
// The 3M + 1 columns of an elimination are split over TWO workgroups (blockIdx.y): the first takes D and as many right-hand sides as fill whole waves (M = 66: 192 of the
// 199 columns), the second D again and the rest (7).  One workgroup of four waves spent a quarter of its LDS broadcast reads — what bounds the kernel — on a wave with 7 live lanes.
template <int M> struct BcrGeom {
    static constexpr int R0 = ((3 * M + 1) / 64) * 64 - M;                          // right-hand sides (of the 2M + 1: GL columns, GR columns, b) of workgroup 0
    static constexpr int NT = M + R0;                                               // a multiple of 64; workgroup 1 has M + (2M + 1 - R0) <= NT live threads
    static constexpr size_t LDS = ((size_t)M * M + M + NT) * sizeof(double);
    static_assert(R0 > 0 && 2 * M + 1 - R0 <= R0, "column split");
};
// compile-time loops: the step index must be a constant in every register index of col[] (a run-time index would send the array to scratch memory)
template <typename F, int... Ks> __device__ __forceinline__ void bcr_static_for_impl(F& f, std::integer_sequence<int, Ks...>) { (f(std::integral_constant<int, Ks>{}), ...); }
template <int N, typename F> __device__ __forceinline__ void bcr_static_for(F&& f) { bcr_static_for_impl(f, std::make_integer_sequence<int, N>{}); }
// col[2 jp], col[2 jp + 1] (jp0 <= jp < M / 2, entries with index <= K skipped) op= row pair * scalar, eight b128 LDS reads in flight at a time: the scheduling barrier
// keeps the machine scheduler from hoisting all 33 reads of a step above the FMAs (132 more live registers: spills)
typedef double bcr_d2 __attribute__((ext_vector_type(2)));
typedef const volatile __attribute__((address_space(3))) bcr_d2* bcr_lds_row;      // volatile: the reads stay where they are written (in chunks of eight, next to their FMAs)
// (round 6) the chunks are software-pipelined: chunk J + 8 is requested before the FMAs of chunk J — with one wave per SIMD nothing else hides the ~130 cycles an LDS
// read takes, and a step of four chunks that waited for each of them spent more time waiting than computing.  VIDO-internal switch: -DBCR_NO_PREFETCH restores round 5's form.
template <int M, int J0, int NE, typename OP> __device__ __forceinline__ void bcr_row_chunk_pf(bcr_lds_row row, OP&& op, const bcr_d2 (&u)[NE])
{
    constexpr int J1 = J0 + NE;
    if constexpr (J1 < M / 2) {
        constexpr int NN = (M / 2 - J1) < 8 ? (M / 2 - J1) : 8;
        bcr_d2 un[NN];
#pragma unroll
        for (int e = 0; e < NN; e++) un[e] = row[J1 + e];
        __builtin_amdgcn_sched_barrier(0);
        bcr_static_for<NE>([&](auto ec) { constexpr int e = decltype(ec)::value; op(std::integral_constant<int, J0 + e>{}, u[e]); });
        __builtin_amdgcn_sched_barrier(0);
        bcr_row_chunk_pf<M, J1, NN>(row, op, un);
    } else {
        bcr_static_for<NE>([&](auto ec) { constexpr int e = decltype(ec)::value; op(std::integral_constant<int, J0 + e>{}, u[e]); });
        __builtin_amdgcn_sched_barrier(0);
    }
}
template <int M, int K, int J0, typename OP> __device__ __forceinline__ void bcr_row_chunk(bcr_lds_row row, OP&& op)
{
#ifdef BCR_NO_PREFETCH
    if constexpr (J0 < M / 2) {
        constexpr int NE = (M / 2 - J0) < 8 ? (M / 2 - J0) : 8;
        bcr_d2 u[NE];
#pragma unroll
        for (int e = 0; e < NE; e++) u[e] = row[J0 + e];
        __builtin_amdgcn_sched_barrier(0);
        bcr_static_for<NE>([&](auto ec) { constexpr int e = decltype(ec)::value; op(std::integral_constant<int, J0 + e>{}, u[e]); });
        __builtin_amdgcn_sched_barrier(0);
        bcr_row_chunk<M, K, J0 + 8>(row, op);
    }
#else
    if constexpr (J0 < M / 2) {
        constexpr int NE = (M / 2 - J0) < 8 ? (M / 2 - J0) : 8;
        bcr_d2 u[NE];
#pragma unroll
        for (int e = 0; e < NE; e++) u[e] = row[J0 + e];
        __builtin_amdgcn_sched_barrier(0);
        bcr_row_chunk_pf<M, J0, NE>(row, op, u);
    }
#endif
}
template <int M> __global__ __launch_bounds__(BcrGeom<M>::NT) void k_bcr_elim(BcrDev B, int s, const double* __restrict__ Lcur, int mode)
{
    extern __shared__ __attribute__((aligned(16))) double bcr_lds[];
    double* U = bcr_lds;                  // [M][M] row k = (U_k0 .. ) only entries c >= k are written / read
    double* dinv = bcr_lds + M * M;       // [M]
    const int c = threadIdx.x;
    const int p = mode ? 0 : s + 2 * s * blockIdx.x;
    // right-hand side index ri of this thread in [GL columns 0..M-1 | GR columns M..2M-1 | b = 2M]: workgroup (blockIdx.y) 0 takes [0, R0), workgroup 1 the rest; mode 1: b only
    const int r_lo = mode ? 2 * M : (blockIdx.y ? BcrGeom<M>::R0 : 0), r_hi = mode ? 2 * M + 1 : (blockIdx.y ? 2 * M + 1 : BcrGeom<M>::R0);
    const int ri = r_lo + (c - M);
    const bool has_r = !mode && p + s < B.nb, is_d = c < M, live = is_d || ri < r_hi, is_b = !is_d && ri == 2 * M;
    const double* Dp = B.D + (size_t)p * M * M;
    const double* Lp = Lcur + (size_t)p * M * M; const double* Lq = Lcur + (size_t)(p + s) * M * M;
    double col[M];
    {
        const double* src = Dp + c; size_t stride = M;                             // D column c / L_p column: coalesced over the lanes
        bool ld = live;
        if (!is_d && ri < M) src = Lp + ri;
        else if (!is_d && ri < 2 * M) { src = Lq + (size_t)(ri - M) * M; stride = 1; ld = live && has_r; }      // column of L_{p+s}^T = row of L_{p+s}
        else if (is_b) { src = B.b + (size_t)p * M; stride = 1; }
        bcr_static_for<M>([&](auto ic) { constexpr int i = decltype(ic)::value; col[i] = ld ? src[(size_t)i * stride] : 0.0; });
    }
    int bad = 0;
    bcr_static_for<M>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        // (branch-free steps: with predicated stores every step is several basic blocks, and the compiler sinks the FMAs of row i into the block of step i — keeping every
        //  multiplier row it has read alive in scratch memory; stores that must not happen go to a per-thread dump slot instead)
        U[(is_d && c >= k) ? k * M + c : M * M + M + c] = col[k];
        __syncthreads();
        const double piv = U[k * M + k];
        bad |= !(piv > 0.0 && piv < 1.0e300);
        double inv = __builtin_amdgcn_rcp(piv);
        inv = inv * (2.0 - piv * inv); inv = inv * (2.0 - piv * inv);
        U[c == k ? M * M + k : M * M + M + c] = inv;                               // dinv[k]
        const double nt = -(col[k] * inv);
        bcr_row_chunk<M, k, (k + 1) / 2>((bcr_lds_row)(U + k * M), [&](auto jc, const bcr_d2& u) {      // M even: rows are 16-byte aligned
            constexpr int jp = decltype(jc)::value;
            if constexpr (2 * jp > k) col[2 * jp] = __builtin_fma(u.x, nt, col[2 * jp]);
            col[2 * jp + 1] = __builtin_fma(u.y, nt, col[2 * jp + 1]);
        });
    });
    if (bad && c == 0) *B.ok = 0.0;
    if (is_d || !live) return;                                                     // (no barrier below)
    // back substitution of this thread's right-hand side; dinv[] was written before the last barrier except dinv[M-1] (own copy below)
    {
        const double pl = U[(M - 1) * M + (M - 1)];
        double il = __builtin_amdgcn_rcp(pl); il = il * (2.0 - pl * il); il = il * (2.0 - pl * il);
        col[M - 1] *= il;
    }
    bcr_static_for<M - 1>([&](auto rc) {
        constexpr int k = M - 2 - decltype(rc)::value;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        bcr_row_chunk<M, k, (k + 1) / 2>((bcr_lds_row)(U + k * M), [&](auto jc, const bcr_d2& u) {
            constexpr int jp = decltype(jc)::value;
            if constexpr (jp & 1) { if constexpr (2 * jp > k) a2 = __builtin_fma(u.x, col[2 * jp], a2); a3 = __builtin_fma(u.y, col[2 * jp + 1], a3); }
            else                  { if constexpr (2 * jp > k) a0 = __builtin_fma(u.x, col[2 * jp], a0); a1 = __builtin_fma(u.y, col[2 * jp + 1], a1); }
        });
        col[k] = (col[k] - ((a0 + a1) + (a2 + a3))) * dinv[k];
    });
    if (mode) {
        bcr_static_for<M>([&](auto ic) { constexpr int i = decltype(ic)::value; if (i < B.n6) B.x[i] = col[i]; });
    } else if (is_b) {
        bcr_static_for<M>([&](auto ic) { constexpr int i = decltype(ic)::value; B.y[(size_t)p * M + i] = col[i]; });
    } else {
        double* dst = (ri < M ? B.GL + ri : B.GR + (ri - M)) + (size_t)p * M * M;
        bcr_static_for<M>([&](auto ic) { constexpr int i = decltype(ic)::value; dst[(size_t)i * M] = col[i]; });
    }
}
// ---- what else was tried for this elimination in round 6 (built, measured on configs[4] size, removed again: profiles/r6/global_ba_reduced_solve.txt) ----------------
//   * 6 x 6 block pivots, a thread per column, fully unrolled (11 block steps of one barrier instead of 66): 34.0 us per level against 36.1 — the barriers were never the cost;
//   * the same as a loop over the block steps (the column as a cyclic register buffer rotated by six per step; 20 KB of code instead of 78): 45.4 us — nor is instruction fetch;
//   * the whole 66 x 199 matrix in LDS, 1024 threads, 3 x 4 register tiles for the trailing update, four lanes per right-hand side in the back substitution: 55.9 us —
//     shader-clock stamps: the tile update moves 54 LDS reads per 72 FMAs and is bound by LDS bandwidth (2.9 k cycles per block step), the back substitution by LDS latency
//     (40 - 60 k cycles).  The thread-per-column form feeds 64 lanes x 2 FMAs from ONE broadcast read: that ratio is why it stays.
//   * what did help: requesting the next chunk of a row before the FMAs of the current one (36.1 -> 34.5 us).
// surviving block a = 2 s blockIdx.x takes D_a -= L_a GR_{a-s} + L_{a+s}^T GL_{a+s}, L_a' = -L_a GL_{a-s}, b_a -= L_a y_{a-s} + L_{a+s}^T y_{a+s}.
// blockIdx.y = a slice of BCR_RB rows: the slice's rows of L_a and columns of L_{a+s} are staged in LDS (broadcast operands), thread j owns output column j of the
// slice (BCR_RB + BCR_RB accumulators) and streams the rows of GR / GL with coalesced loads.
#define BCR_RB 6
__global__ __launch_bounds__(128) void k_bcr_update(BcrDev B, int s, const double* __restrict__ Lcur, double* __restrict__ Lnext)
{
    __shared__ double la[BCR_RB * 96], lq[BCR_RB * 96];
    const int m = B.m, a = 2 * s * blockIdx.x, p = a - s, q = a + s, i0 = blockIdx.y * BCR_RB, j = threadIdx.x;
    const bool has_p = p >= 0, has_q = q < B.nb;
    const double* La = Lcur + (size_t)a * m * m; const double* Lq = Lcur + (size_t)q * m * m;
    for (int t = j; t < BCR_RB * m; t += 128) {
        const int r = t / m, k = t - r * m;
        la[r * 96 + k] = has_p ? La[(size_t)(i0 + r) * m + k] : 0.0;          // L_a[i0 + r][k]
        lq[r * 96 + k] = has_q ? Lq[(size_t)k * m + i0 + r] : 0.0;            // L_{a+s}^T[i0 + r][k]
    }
    __syncthreads();
    if (j < m) {
        const double* GRp = B.GR + (size_t)p * m * m + j; const double* GLp = B.GL + (size_t)p * m * m + j; const double* GLq = B.GL + (size_t)q * m * m + j;
        double d[BCR_RB], ln[BCR_RB];
#pragma unroll
        for (int r = 0; r < BCR_RB; r++) { d[r] = 0.0; ln[r] = 0.0; }
        if (has_p) {
#pragma unroll 22
            for (int k = 0; k < m; k++) {
                const double gr = GRp[(size_t)k * m], gl = -GLp[(size_t)k * m];
#pragma unroll
                for (int r = 0; r < BCR_RB; r++) { const double l = la[r * 96 + k]; d[r] = __builtin_fma(l, gr, d[r]); ln[r] = __builtin_fma(l, gl, ln[r]); }
            }
        }
        if (has_q) {
#pragma unroll 22
            for (int k = 0; k < m; k++) {
                const double gl = GLq[(size_t)k * m];
#pragma unroll
                for (int r = 0; r < BCR_RB; r++) d[r] = __builtin_fma(lq[r * 96 + k], gl, d[r]);
            }
        }
#pragma unroll
        for (int r = 0; r < BCR_RB; r++) {
            B.D[(size_t)a * m * m + (size_t)(i0 + r) * m + j] -= d[r];
            Lnext[(size_t)a * m * m + (size_t)(i0 + r) * m + j] = ln[r];        // block (a, a - 2s)
        }
    } else if (j >= 96 && j < 96 + BCR_RB) {
        const int r = j - 96; double dv = 0.0;
        if (has_p) for (int k = 0; k < m; k++) dv += la[r * 96 + k] * B.y[(size_t)p * m + k];
        if (has_q) for (int k = 0; k < m; k++) dv += lq[r * 96 + k] * B.y[(size_t)q * m + k];
        B.b[(size_t)a * m + i0 + r] -= dv;
    }
}
// x_p = y_p - GL_p x_{p-s} - GR_p x_{p+s} for the blocks eliminated at level s; four lanes per row, each a quarter of the two dot products
__global__ __launch_bounds__(384) void k_bcr_back(BcrDev B, int s)
{
    const int m = B.m, p = s + 2 * s * blockIdx.x, i = threadIdx.x >> 2, sub = threadIdx.x & 3;
    double v = 0.0;
    const int gi = p * m + i;
    if (i < m) {
        const double* gl = B.GL + (size_t)p * m * m + (size_t)i * m; const double* gr = B.GR + (size_t)p * m * m + (size_t)i * m;
        const int l0 = (p - s) * m, r0 = (p + s) * m;
        for (int k = sub; k < m; k += 4) { if (l0 + k < B.n6) v += gl[k] * B.x[l0 + k]; }
        if (p + s < B.nb) for (int k = sub; k < m; k += 4) { if (r0 + k < B.n6) v += gr[k] * B.x[r0 + k]; }
    }
    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64);
    if (i < m && sub == 0 && gi < B.n6) B.x[gi] = B.y[(size_t)p * m + i] - v;
}

// ---- round 6: the Schur update with the contraction split over four lanes, and the whole back substitution as ONE launch ---------------------------------------
// k_bcr_update walks three 66-term contractions per thread with two dependent-latency-bound global loads per term (23 us per level for 1.7 MFLOP).  Here thread (j, kh)
// takes the terms k = kh mod 4 of output column j: every load of a thread is in flight at once (17 + 17 + 17), the four partial sums meet in LDS.
__global__ __launch_bounds__(320) void k_bcr_update4(BcrDev B, int s, const double* __restrict__ Lcur, double* __restrict__ Lnext)
{
    __shared__ double la[BCR_RB * 96], lq[BCR_RB * 96], part[3][2 * BCR_RB][96];
    const int m = B.m, a = 2 * s * blockIdx.x, p = a - s, q = a + s, i0 = blockIdx.y * BCR_RB, tid = threadIdx.x;
    const bool has_p = p >= 0, has_q = q < B.nb;
    const double* La = Lcur + (size_t)a * m * m; const double* Lq = Lcur + (size_t)q * m * m;
    for (int t = tid; t < BCR_RB * m; t += 320) {
        const int r = t / m, k = t - r * m;
        la[r * 96 + k] = has_p ? La[(size_t)(i0 + r) * m + k] : 0.0;          // L_a[i0 + r][k]
        lq[r * 96 + k] = has_q ? Lq[(size_t)k * m + i0 + r] : 0.0;            // L_{a+s}^T[i0 + r][k]
    }
    __syncthreads();
    const int kh = tid / m, j = tid - kh * m;                                 // (m <= 80: four slices of the contraction fit 320 threads; m = 96 takes three and a tail, see below)
    const int nkh = 320 / m > 4 ? 4 : 320 / m;
    double d[BCR_RB], ln[BCR_RB];
#pragma unroll
    for (int r = 0; r < BCR_RB; r++) { d[r] = 0.0; ln[r] = 0.0; }
    if (kh < nkh) {
        const double* GRp = B.GR + (size_t)p * m * m + j; const double* GLp = B.GL + (size_t)p * m * m + j; const double* GLq = B.GL + (size_t)q * m * m + j;
        if (has_p) {
#pragma unroll 8
            for (int k = kh; k < m; k += nkh) {
                const double gr = GRp[(size_t)k * m], gl = -GLp[(size_t)k * m];
#pragma unroll
                for (int r = 0; r < BCR_RB; r++) { const double l = la[r * 96 + k]; d[r] = __builtin_fma(l, gr, d[r]); ln[r] = __builtin_fma(l, gl, ln[r]); }
            }
        }
        if (has_q) {
#pragma unroll 8
            for (int k = kh; k < m; k += nkh) {
                const double gl = GLq[(size_t)k * m];
#pragma unroll
                for (int r = 0; r < BCR_RB; r++) d[r] = __builtin_fma(lq[r * 96 + k], gl, d[r]);
            }
        }
        if (kh > 0) {
#pragma unroll
            for (int r = 0; r < BCR_RB; r++) { part[kh - 1][r][j] = d[r]; part[kh - 1][BCR_RB + r][j] = ln[r]; }
        }
    }
    __syncthreads();
    if (kh == 0) {
#pragma unroll
        for (int r = 0; r < BCR_RB; r++) {
            double dv = d[r], lv = ln[r];
            for (int h = 1; h < nkh; h++) { dv += part[h - 1][r][j]; lv += part[h - 1][BCR_RB + r][j]; }      // fixed order: slices 0, 1, 2, 3
            B.D[(size_t)a * m * m + (size_t)(i0 + r) * m + j] -= dv;
            Lnext[(size_t)a * m * m + (size_t)(i0 + r) * m + j] = lv;           // block (a, a - 2s)
        }
    } else if (tid >= 320 - BCR_RB) {                                         // the right-hand side rows of the slice: threads of the last wave that hold no output column
        const int r = tid - (320 - BCR_RB); double dv = 0.0;
        if (has_p) for (int k = 0; k < m; k++) dv += la[r * 96 + k] * B.y[(size_t)p * m + k];
        if (has_q) for (int k = 0; k < m; k++) dv += lq[r * 96 + k] * B.y[(size_t)q * m + k];
        B.b[(size_t)a * m + i0 + r] -= dv;
    }
}
// The back substitution x_p = y_p - GL_p x_{p-s} - GR_p x_{p+s} walked the levels back with one launch each (6 x 9 us, each a 17-term dependent chain behind two cold
// loads).  One launch for all levels: one workgroup per eliminated block, ordered by level (the top of the tree first); a workgroup first pulls its rows of GL / GR into
// registers — they do not depend on x —, then waits for its two neighbours' x, which arrive as tagged granules (xwg.hpp: the data is its own flag) from the workgroups of
// the levels above, or — the root block 0 — were written by the launch before.  A level costs one granule hop + 17 FMAs instead of a launch.
// blocks[]: (p, s) per workgroup; xt: [nb][m][2] tagged words; every spin is bounded (xwg_wait_word), a timeout clears *B.ok.
__global__ __launch_bounds__(384) void k_bcr_back_chain(BcrDev B, const int2* __restrict__ blocks, unsigned long long* xt, unsigned epoch, unsigned* abort_word)
{
    __shared__ double xs[2][96]; __shared__ int okf;
    const int m = B.m, p = blocks[blockIdx.x].x, s = blocks[blockIdx.x].y, i = threadIdx.x >> 2, sub = threadIdx.x & 3, tid = threadIdx.x;
    const bool has_r = p + s < B.nb;
    double gl[24], gr[24];                                                    // rows of GL_p / GR_p: terms k = sub mod 4 (m <= 96)
    const int l0 = (p - s) * m, r0 = (p + s) * m;
    if (i < m) {
        const double* glp = B.GL + (size_t)p * m * m + (size_t)i * m; const double* grp = B.GR + (size_t)p * m * m + (size_t)i * m;
#pragma unroll
        for (int t = 0; t < 24; t++) { const int k = sub + 4 * t; gl[t] = k < m ? glp[k] : 0.0; gr[t] = (k < m && has_r) ? grp[k] : 0.0; }
    }
    if (tid == 0) okf = 1;
    __syncthreads();
    // neighbours' x into LDS: threads 0 .. m-1 the left one, 96 .. 96+m-1 the right one
    {
        const int side = tid >= 96 ? 1 : 0, k = tid - 96 * side;
        if (tid < 192 && k < m) {
            const int nbk = side ? p + s : p - s;
            double v = 0.0;
            if (side && !has_r) v = 0.0;
            else if (nbk == 0) { const int gi = k; v = gi < B.n6 ? B.x[gi] : 0.0; }                 // the root: solved by the launch before this one
            else {
                unsigned lo = 0, hi = 0; const unsigned long long* g = xt + ((size_t)nbk * m + k) * 2;
                const bool ok = xwg_wait_word(g, epoch, abort_word, &lo) && xwg_wait_word(g + 1, epoch, abort_word, &hi);
                if (!ok) { okf = 0; xwg_store32(abort_word, 1u); }
                v = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
            }
            xs[side][k] = v;
        }
    }
    __syncthreads();
    double v = 0.0;
    if (i < m) {
        // (the order of k_bcr_back: the left neighbour's terms, then the right one's, then the four lanes)
#pragma unroll
        for (int t = 0; t < 24; t++) { const int k = sub + 4 * t; if (k < m && l0 + k < B.n6) v += gl[t] * xs[0][k]; }
        if (has_r) {
#pragma unroll
            for (int t = 0; t < 24; t++) { const int k = sub + 4 * t; if (k < m && r0 + k < B.n6) v += gr[t] * xs[1][k]; }
        }
    }
    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64);
    const int gi = p * m + i;
    if (i < m && sub == 0) {
        const double x = B.y[(size_t)p * m + i] - v;
        if (gi < B.n6) B.x[gi] = x;
        xwg_publish_f64(xt + ((size_t)p * m + i) * 2, epoch, x);
    }
    if (tid == 0 && !okf) *B.ok = 0.0;
}

// ---- trial state ------------------------------------------------------------------------------------------
__global__ void k_ba_update_cams(BaDev P, double lambda)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    double sc = c < P.n_cam ? ba_update_cam(P, c, P.x + 6 * c, lambda) : 0.0;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) sc += __shfl_xor(sc, o, 64);
    if ((threadIdx.x & 63) == 0 && sc != 0) atomicAdd(P.scal + 3, sc);
}
// x_l = D^-1 (b_l - sum_i W_i^T x_ci), p_new = p + x_l; landmark part of computeScale
// 8 lanes per landmark: lane `sub` walks slots beg+sub, beg+sub+8, ... (a 17-observation track is 3 trips of the dependent slot_cam -> x
// load chain instead of 17), the three partial sums meet through xor 1|2|4 shuffles.
__global__ __launch_bounds__(256) void k_ba_backsub(BaDev P, int n_ptl, double lambda)
{
    const int sub = threadIdx.x & 7;
    double sc = 0;
    for (int l0 = (blockIdx.x * blockDim.x + threadIdx.x) >> 3; l0 - (int)(threadIdx.x >> 3) < n_ptl; l0 += (gridDim.x * blockDim.x) >> 3) {   // uniform trip count per workgroup
        const int l = l0;
        const bool on = l < n_ptl;
        double t0 = 0, t1 = 0, t2 = 0;
        if (on) {
            for (int s = P.pt_start[l] + sub; s < P.pt_start[l + 1]; s += 8) {
                const double* W = P.W + BA_REC * (size_t)s; const double* xc = P.x + 6 * P.slot_cam[s];
#pragma unroll
                for (int a = 0; a < 6; a++) { t0 -= W[a * 3] * xc[a]; t1 -= W[a * 3 + 1] * xc[a]; t2 -= W[a * 3 + 2] * xc[a]; }
            }
        }
#pragma unroll
        for (int o = 4; o >= 1; o >>= 1) { t0 += __shfl_xor(t0, o, 64); t1 += __shfl_xor(t1, o, 64); t2 += __shfl_xor(t2, o, 64); }
        if (on && sub == 0) {
            const double b0 = P.bp[3 * (size_t)l], b1 = P.bp[3 * (size_t)l + 1], b2 = P.bp[3 * (size_t)l + 2];
            t0 += b0; t1 += b1; t2 += b2;
            double Di[9]; inv3sym(P.Hpp + 6 * (size_t)l, lambda, Di);
            const double x0 = Di[0] * t0 + Di[1] * t1 + Di[2] * t2, x1 = Di[3] * t0 + Di[4] * t1 + Di[5] * t2, x2 = Di[6] * t0 + Di[7] * t1 + Di[8] * t2;
            P.pt_new[3 * (size_t)l] = P.pt[3 * (size_t)l] + x0; P.pt_new[3 * (size_t)l + 1] = P.pt[3 * (size_t)l + 1] + x1; P.pt_new[3 * (size_t)l + 2] = P.pt[3 * (size_t)l + 2] + x2;
            sc += x0 * (lambda * x0 + b0) + x1 * (lambda * x1 + b1) + x2 * (lambda * x2 + b2);
        }
    }
    block_atomic_add(P.scal + 3, sc);
}
__global__ __launch_bounds__(256) void k_ba_chi2(BaDev P, const double* cam, const double* pt, double* out)
{
    double acc = 0;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < P.n_obs; k += gridDim.x * blockDim.x) {      // grid-stride: <= 512 workgroups
        const double* X = cam + 12 * P.obs_cam[k]; const double* p = pt + 3 * (size_t)P.obs_pt[k]; const double* m = P.obs_meas + 3 * (size_t)k;
        const double d0 = p[0] - X[3], d1 = p[1] - X[7], d2 = p[2] - X[11];
        const double e0 = X[0] * d0 + X[4] * d1 + X[8] * d2 - m[0], e1 = X[1] * d0 + X[5] * d1 + X[9] * d2 - m[1], e2 = X[2] * d0 + X[6] * d1 + X[10] * d2 - m[2];
        double v, w; huber_w(P.info_obs * (e0 * e0 + e1 * e1 + e2 * e2), P.huber_obs, P.use_huber, v, w);
        acc += v;
    }
    block_atomic_add(out, acc);
}

// ---- the local window as ONE persistent launch (round 4) ------------------------------------------------------------------------------------------------------
// Round 3 drove the window's Levenberg-Marquardt loop from the host: ~10 launches and one read-back per trial, ~9 trials per frame — about 100 dependent stream operations
// per solve, each of which queues behind the networks' convolution workgroups when the tracker shares the GPU (2.0 ms alone, 4.6 ms inside the pipeline).  Here the whole
// solve (core/optimization_algorithm_levenberg.cpp:61-189 around block_solver.hpp:354-486) is one launch of a few dozen resident workgroups that walk the phases of an
// iteration together, separated by grid-wide barriers (xwg.hpp), and take the accept / reject / stop decisions from the same reduced scalars — every thread of every
// workgroup runs the same control flow, no broadcast, no host:
//   L   linearise (ba_linearize_body over the grid's share of the observations, the measurement stored next to W) + camera-camera factors
//   M   first iteration only: max |diagonal| for the initial lambda
//   S   Schur complement of this workgroup's landmarks into its LDS copy of the reduced system -> one partial per workgroup
//   F   fold: S = camera part + lambda I + sum of the partials (plain stores: nothing to clear, no atomics), r likewise
//   C   workgroup 0: pose-block Cholesky + back sweep in LDS, trial cameras, camera part of computeScale; the others wait
//   B   back-substitution with the trial's chi2 formed per landmark from the slot records (8 lanes per landmark), camera-camera factors at the trial state; the
//       accumulators of the NEXT linearisation (the other of two copies) are cleared here
// and then decide.  5 barriers per iteration; the accepted state flips between two buffers and is copied home at the end.
struct BaLmCtl { unsigned bar_count, abort_word; int status, iterations, trials, n_lin, barriers, pad; double lambda_final, chi2_initial, chi2_final, lin_ticks; };
struct BaLocalArgs {
    BaLmCtl* ctl; double *red0, *red1, *S_part, *cam_out, *pt_home;
    int n_ptl, kcap, max_iters; unsigned bar_base; double gain_threshold;
};
#define BAL_NT 512

__device__ __forceinline__ double bal_ld(const double* p) { return __longlong_as_double((long long)xwg_load64((const unsigned long long*)p)); }
__device__ __forceinline__ void bal_st(double* p, double v) { xwg_store64((unsigned long long*)p, (unsigned long long)__double_as_longlong(v)); }

// F: the reduced system from the camera part and the workgroups' partial Schur sums.  parts are block-major (ba_schur_body<0>): lower-triangle 6x6 blocks at
// [bid * SB_PITCH + 6a + b], then the rhs; S is dense row-major, only its lower block triangle is written (all the factorisation reads).
__device__ __forceinline__ void ba_fold_local_body(const BaDev& P, const double* __restrict__ S_part, int nparts, double lambda, int gt, int gnt)
{
    const int nc = P.n6 / 6, nblk_l = nc * (nc + 1) / 2, rhs_off = nblk_l * SB_PITCH;
    const size_t sz = (size_t)rhs_off + P.n6;
    if (gt < 2) bal_st(P.scal + 2 + gt, 0.0);                  // tempChi / scale of the trial that follows
    for (int t = gt; t < nblk_l * 36 + P.n6; t += gnt) {
        const bool is_rhs = t >= nblk_l * 36;
        const int bid = t / 36, el = t - bid * 36;
        const size_t src = is_rhs ? (size_t)rhs_off + (t - nblk_l * 36) : (size_t)bid * SB_PITCH + el;
        double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
        int q = 0;
        for (; q + 3 < nparts; q += 4) {                        // four independent load streams in flight
            s0 += S_part[(size_t)q * sz + src]; s1 += S_part[(size_t)(q + 1) * sz + src];
            s2 += S_part[(size_t)(q + 2) * sz + src]; s3 += S_part[(size_t)(q + 3) * sz + src];
        }
        for (; q < nparts; q++) s0 += S_part[(size_t)q * sz + src];
        double v = (s0 + s1) + (s2 + s3);
        if (is_rhs) { const int a = t - nblk_l * 36; P.r[a] = P.bc[a] + v; continue; }
        int ci = (int)((sqrtf(8.f * (float)bid + 1.f) - 1.f) * 0.5f);
        ci -= (ci * (ci + 1) / 2 > bid); ci += ((ci + 1) * (ci + 2) / 2 <= bid);
        const int cj = bid - ci * (ci + 1) / 2, a = el / 6, b = el - a * 6;
        if (ci == cj) { v += P.Hcd[36 * ci + el]; if (a == b) v += lambda; }
        else for (int k = 0; k < P.n_odo; k++) {                 // camera-camera factors between the two poses (k_ba_init_S, inline_odo)
            const int i = P.odo_i[k], j = P.odo_j[k];
            if (i == ci && j == cj) v += P.Hodo[36 * k + a * 6 + b];
            if (j == ci && i == cj) v += P.Hodo[36 * k + b * 6 + a];
        }
        P.S[(size_t)(6 * ci + a) * P.n6 + 6 * cj + b] = v;
    }
}
// B: k_ba_backsub with the robust chi2 of the trial state formed on the way.  8 lanes per landmark; after the xor shuffles all 8 hold the landmark's sums, so every lane has
// the new point and walks its share of the slots a second time: e = R_new^T (p_new - t_new) - m with the measurement the linearisation left in the slot record.
__device__ __forceinline__ void ba_backsub_chi2_body(const BaDev& P, int n_ptl, double lambda, int gt, int gnt, double* wsum /*[32] LDS*/)
{
    const int sub = threadIdx.x & 7;
    double sc = 0, chi = 0;
    for (int l0 = gt >> 3; l0 - (int)(threadIdx.x >> 3) < n_ptl; l0 += gnt >> 3) {   // uniform trip count per workgroup
        const int l = l0;
        const bool on = l < n_ptl;
        double t0 = 0, t1 = 0, t2 = 0;
        int beg = 0, end = 0;
        if (on) {
            beg = P.pt_start[l]; end = P.pt_start[l + 1];
            for (int s = beg + sub; s < end; s += 8) {
                const double* W = P.W + BA_REC * (size_t)s; const double* xc = P.x + 6 * P.slot_cam[s];
#pragma unroll
                for (int a = 0; a < 6; a++) { t0 -= W[a * 3] * xc[a]; t1 -= W[a * 3 + 1] * xc[a]; t2 -= W[a * 3 + 2] * xc[a]; }
            }
        }
#pragma unroll
        for (int o = 4; o >= 1; o >>= 1) { t0 += __shfl_xor(t0, o, 64); t1 += __shfl_xor(t1, o, 64); t2 += __shfl_xor(t2, o, 64); }
        if (on) {
            const double b0 = P.bp[3 * (size_t)l], b1 = P.bp[3 * (size_t)l + 1], b2 = P.bp[3 * (size_t)l + 2];
            t0 += b0; t1 += b1; t2 += b2;
            double Di[9]; inv3sym(P.Hpp + 6 * (size_t)l, lambda, Di);
            const double x0 = Di[0] * t0 + Di[1] * t1 + Di[2] * t2, x1 = Di[3] * t0 + Di[4] * t1 + Di[5] * t2, x2 = Di[6] * t0 + Di[7] * t1 + Di[8] * t2;
            const double p0 = P.pt[3 * (size_t)l] + x0, p1 = P.pt[3 * (size_t)l + 1] + x1, p2 = P.pt[3 * (size_t)l + 2] + x2;
            if (sub == 0) {
                P.pt_new[3 * (size_t)l] = p0; P.pt_new[3 * (size_t)l + 1] = p1; P.pt_new[3 * (size_t)l + 2] = p2;
                sc += x0 * (lambda * x0 + b0) + x1 * (lambda * x1 + b1) + x2 * (lambda * x2 + b2);
            }
            for (int s = beg + sub; s < end; s += 8) {
                const double* X = P.cam_new + 12 * P.slot_cam[s]; const double* m = P.W + BA_REC * (size_t)s + 27;
                const double d0 = p0 - X[3], d1 = p1 - X[7], d2 = p2 - X[11];
                const double e0 = X[0] * d0 + X[4] * d1 + X[8] * d2 - m[0], e1 = X[1] * d0 + X[5] * d1 + X[9] * d2 - m[1], e2 = X[2] * d0 + X[6] * d1 + X[10] * d2 - m[2];
                double v, w; huber_w(P.info_obs * (e0 * e0 + e1 * e1 + e2 * e2), P.huber_obs, P.use_huber, v, w);
                chi += v;
            }
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { sc += __shfl_xor(sc, o, 64); chi += __shfl_xor(chi, o, 64); }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { wsum[threadIdx.x >> 6] = sc; wsum[16 + (threadIdx.x >> 6)] = chi; }
    __syncthreads();
    if (threadIdx.x < 2) { double t = 0; for (int w = 0; w < (int)(blockDim.x >> 6); w++) t += wsum[16 * threadIdx.x + w]; if (t != 0) atomicAdd(P.scal + (threadIdx.x == 0 ? 3 : 2), t); }
}

// ---- the same bodies as stand-alone launches of the HOST-driven local-window loop (the default: see DESIGN.md section 9 for why the persistent form is opt-in).  Round 3
// spent 11 stream operations per LM iteration on the window (memset, linearise, camera factors | init S, Schur, fold, memset, Cholesky, back-substitution, chi2, camera
// factors, read-back); with these it is 6: the camera factors ride in the linearisation / back-substitution launches as extra workgroups, the fold writes S outright (no
// init, no atomics), the trial chi2 is formed landmark-major inside the back-substitution, the accumulators are double-buffered and cleared by the previous iteration.
__global__ __launch_bounds__(LIN_THREADS) void k_ba_lin_local(BaDev P, int nvb)
{
    __shared__ __attribute__((aligned(16))) double lin_smem[LIN_SMEM_DOUBLES(LIN_THREADS)];
    if ((int)blockIdx.x < nvb) { ba_linearize_body<LIN_THREADS, true>(P, 1, blockIdx.x, lin_smem); return; }
    ba_camfactor_body(P, 1, P.cam, P.scal + 0, ((int)blockIdx.x - nvb) * (LIN_THREADS / 64) + (int)(threadIdx.x >> 6), threadIdx.x & 63);      // (a factor index past the list returns)
}
__global__ __launch_bounds__(256) void k_ba_fold_local(BaDev P, const double* S_part, int nparts, double lambda) { ba_fold_local_body(P, S_part, nparts, lambda, blockIdx.x * 256 + threadIdx.x, gridDim.x * 256); }
// (the LAST workgroup to finish — ticket counter, the classic threadfence reduction — copies the trial's five scalars straight into the host's pinned block: the host waits for
//  the launch and reads them there, one copy per trial less on the stream)
__global__ __launch_bounds__(256) void k_ba_backsub_chi2_local(BaDev P, int n_ptl, double lambda, int n_bs, double* zero_buf, int zero_len, int* ticket, double* host_scal)
{
    __shared__ double wsum[32];
    __shared__ int is_last;
    if ((int)blockIdx.x < n_bs) {
        ba_backsub_chi2_body(P, n_ptl, lambda, blockIdx.x * 256 + threadIdx.x, n_bs * 256, wsum);
        if (zero_buf) for (int t = blockIdx.x * 256 + threadIdx.x; t < zero_len; t += n_bs * 256) zero_buf[t] = 0.0;      // the NEXT linearisation's accumulators (the other copy)
    } else ba_camfactor_body(P, 0, P.cam_new, P.scal + 2, ((int)blockIdx.x - n_bs) * 4 + (int)(threadIdx.x >> 6), threadIdx.x & 63);
    __threadfence();                                            // this workgroup's atomics have been performed before its ticket is drawn
    __syncthreads();
    if (threadIdx.x == 0) is_last = atomicAdd(ticket, 1) == (int)gridDim.x - 1;
    __syncthreads();
    if (is_last) {
        if (threadIdx.x < 5) host_scal[threadIdx.x] = bal_ld(P.scal + threadIdx.x);
        if (threadIdx.x == 0) *ticket = 0;
    }
}

// The solver's state flips between two buffers (accepted / trial) and the linearisation's accumulators between two copies: four variants of the problem descriptor, all
// in the kernel arguments (constant memory, read with scalar loads where a field is used); `sel` picks one.  A descriptor mutated in place would live in registers.
struct BaDev4 { BaDev v[4]; };      // [accumulator copy (iteration parity)][state flip]
static_assert(sizeof(BaDev4) + sizeof(BaLocalArgs) <= 4000, "kernel arguments of k_ba_local_lm exceed the 4 KB kernarg segment");
// ---- the local window ENQUEUED AHEAD (the default): the phases of the fused host-driven loop as launches that take the LM state from DEVICE memory (BaSpecCtl) and the
// problem descriptor as the four variants of BaDev4, so the host can put a whole solve's trials on the stream without waiting for any of them.  A trial slot is
// S, F, C, B (+ the LM policy, run by the last workgroup of B) and L, which only runs when the policy accepted the trial and asked for a new linearisation; every launch
// returns at once when the policy has stopped the solve.  The policy is g2o's (optimization_algorithm_levenberg.cpp:61-189 + the facade's stop rules), the same
// arithmetic as the host loop below and k_ba_local_lm: identical LM traces (tests/test_ba_gpu.py).
struct BaSpecCtl {
    double lambda, ni, currentChi, iniChi, chi2_check, lastChi, chi_init, chi_final, gain_threshold;
    int it, qmax, nBad, flip, stop, need_lin, trials, max_iters;
};
__device__ __forceinline__ const BaDev& bas_sel(const BaDev4& V, const BaSpecCtl* c) { return V.v[(c->it & 1) * 2 + c->flip]; }
__global__ __launch_bounds__(LIN_THREADS) void k_bas_lin(BaDev4 V, const BaSpecCtl* c, int nvb)
{
    __shared__ __attribute__((aligned(16))) double lin_smem[LIN_SMEM_DOUBLES(LIN_THREADS)];
    if (c->stop || !c->need_lin) return;
    const BaDev& P = bas_sel(V, c);
    if ((int)blockIdx.x < nvb) { ba_linearize_body<LIN_THREADS, true>(P, 1, blockIdx.x, lin_smem); return; }
    ba_camfactor_body(P, 1, P.cam, P.scal + 0, ((int)blockIdx.x - nvb) * (LIN_THREADS / 64) + (int)(threadIdx.x >> 6), threadIdx.x & 63);
}
__global__ void k_bas_init(BaDev4 V, BaSpecCtl* c)       // after the first linearisation + max diagonal: lambda_0 = tau * max diag (tau = 1e-5), chi2 at the initial state
{
    if (threadIdx.x || blockIdx.x) return;
    const double* scal = V.v[0].scal;
    c->lambda = 1e-5 * bal_ld(scal + 1); c->ni = 2; c->nBad = 0; c->chi_init = c->chi_final = c->currentChi = c->iniChi = bal_ld(scal + 0);
}
__global__ __launch_bounds__(512) void k_bas_schur(BaDev4 V, const BaSpecCtl* c, int n_ptl, int kcap, double* S_part)
{
    extern __shared__ double lds_schur_dyn[];
    if (c->stop) return;
    ba_schur_body<0>(bas_sel(V, c), n_ptl, c->lambda, kcap, S_part, (const int*)nullptr, (const int*)nullptr, (const int2*)nullptr, blockIdx.x, gridDim.x, lds_schur_dyn);
}
__global__ __launch_bounds__(256) void k_bas_fold(BaDev4 V, const BaSpecCtl* c, const double* S_part, int nparts)
{
    if (c->stop) return;
    ba_fold_local_body(bas_sel(V, c), S_part, nparts, c->lambda, blockIdx.x * 256 + threadIdx.x, gridDim.x * 256);
}
__global__ __launch_bounds__(CH_NT) void k_bas_chol(BaDev4 V, const BaSpecCtl* c)
{
    extern __shared__ double cs6_dyn[];
    if (c->stop) return;
    ba_chol_small6_body<CH_NT>(bas_sel(V, c), 1, c->lambda, cs6_dyn);
}
__global__ __launch_bounds__(256) void k_bas_backsub(BaDev4 V, BaSpecCtl* c, int n_ptl, int n_bs, double* red0, double* red1, int red_len, int* ticket, BaSpecCtl* host_ctl)
{
    __shared__ double wsum[32];
    __shared__ int is_last;
    if (c->stop) return;
    const BaDev& P = bas_sel(V, c);
    const int it = c->it, qmax = c->qmax;
    if ((int)blockIdx.x < n_bs) {
        ba_backsub_chi2_body(P, n_ptl, c->lambda, blockIdx.x * 256 + threadIdx.x, n_bs * 256, wsum);
        if (qmax == 0) { double* zb = (it & 1) ? red0 : red1; for (int t = blockIdx.x * 256 + threadIdx.x; t < red_len; t += n_bs * 256) zb[t] = 0.0; }      // the NEXT linearisation's accumulators (the other copy)
    } else ba_camfactor_body(P, 0, P.cam_new, P.scal + 2, ((int)blockIdx.x - n_bs) * 4 + (int)(threadIdx.x >> 6), threadIdx.x & 63);
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) is_last = atomicAdd(ticket, 1) == (int)gridDim.x - 1;
    __syncthreads();
    if (!is_last || threadIdx.x) return;
    *ticket = 0;
    // ---- the LM policy (one thread, after every workgroup of the trial has finished)
    BaSpecCtl s = *c;
    const double* scal = P.scal;
    if (s.qmax == 0) s.currentChi = s.iniChi = bal_ld(scal + 0);                 // chi2 at this iteration's linearisation point
    const bool ok2 = bal_ld(scal + 4) > 0.5;
    const double tempChi = ok2 ? bal_ld(scal + 2) : DBL_MAX, scale = ok2 ? bal_ld(scal + 3) : 0.0;
    const double rho = (s.currentChi - tempChi) / (scale + 1e-3);
    if (rho > 0 && isfinite(tempChi)) {
        const double tr = 2 * rho - 1; double alpha = 1. - tr * tr * tr; alpha = fmin(alpha, 2. / 3.);
        s.lambda *= fmax(1. / 3., alpha); s.ni = 2; s.currentChi = tempChi;
        s.flip ^= 1;                                                             // the trial state becomes the accepted one
    } else { s.lambda *= s.ni; s.ni *= 2; }
    s.qmax++; s.trials++;
    if (rho < 0 && s.qmax < 10) s.need_lin = 0;                                  // another trial at the same linearisation with the larger lambda
    else {
        bool terminate = (s.qmax == 10 || rho == 0);
        if (!terminate) { if ((s.iniChi - s.currentChi) * 1e3 < s.iniChi) s.nBad++; else s.nBad = 0; if (s.nBad >= 3) terminate = true; }
        const double chiNow = s.currentChi;
        if (s.chi2_check < chiNow && s.it > 0) terminate = true;
        s.chi2_check = chiNow;
        if (s.it == 0) s.lastChi = chiNow;
        else { const double gain = (s.lastChi - chiNow) / chiNow; s.lastChi = chiNow; if (gain >= 0 && gain < s.gain_threshold) terminate = true; }
        s.chi_final = chiNow;
        s.it++;
        if (terminate || s.it >= s.max_iters) s.stop = 1; else { s.need_lin = 1; s.qmax = 0; }
    }
    *c = s;
    __threadfence_system();
    *host_ctl = s;                                                               // the host polls / reads the pinned mirror
}

__global__ __launch_bounds__(BAL_NT) void k_ba_local_lm(BaDev4 V, BaLocalArgs A)
{
    extern __shared__ __attribute__((aligned(16))) double bal_smem[];
    __shared__ int bar_flag;
    __shared__ double bal_wsum[32];
    const int bid = blockIdx.x, nblk = gridDim.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, nw = BAL_NT / 64, gt = bid * BAL_NT + tid, gnt = nblk * BAL_NT;
    unsigned phase = 0; bool dead = false;
    auto barrier = [&]() { if (!dead && !xwg_grid_barrier(&A.ctl->bar_count, &A.ctl->abort_word, (unsigned)nblk, A.bar_base, ++phase, &bar_flag)) dead = true; };
    const int n6 = V.v[0].n6, n_cam = V.v[0].n_cam, n_obs = V.v[0].n_obs, ncf = V.v[0].n_odo + (V.v[0].prior_cam >= 0 ? 1 : 0);
    const size_t red_len = (size_t)n_cam * 36 + n6 + 8;
    for (size_t t = gt; t < red_len; t += gnt) { A.red0[t] = 0.0; A.red1[t] = 0.0; }
    barrier();
    double lambda = -1, ni = 2, lastChi = 0, chi2_check = 0, chi_init = 0, chi_final = 0, lin_ticks = 0; int nBad = 0, trials = 0, it = 0, n_lin = 0, flip = 0;
    for (it = 0; it < A.max_iters && !dead; it++) {
        double* red_next = (it & 1) ? A.red0 : A.red1;
        // ---- L
        const long long tk0 = wall_clock64();
        { const BaDev& P = V.v[(it & 1) * 2 + flip];
          const int nvb = (n_obs + BAL_NT - 1) / BAL_NT;
          for (int vb = bid; vb < nvb; vb += nblk) { ba_linearize_body<BAL_NT, true>(P, 1, vb, bal_smem); __syncthreads(); }
          for (int f = bid * nw + wave; f < ncf; f += nblk * nw) ba_camfactor_body(P, 1, P.cam, P.scal + 0, f, lane); }
        barrier();
        lin_ticks += (double)(wall_clock64() - tk0); n_lin++;
        if (it == 0) { ba_maxdiag_body(V.v[flip], A.n_ptl, gt, gnt); barrier(); }
        const double* scal = V.v[(it & 1) * 2].scal;
        double currentChi = bal_ld(scal + 0); const double iniChi = currentChi;
        if (it == 0) { lambda = 1e-5 * bal_ld(scal + 1); ni = 2; nBad = 0; chi_init = currentChi; chi_final = currentChi; }
        double rho = 0; int qmax = 0;
        do {
            const BaDev& P = V.v[(it & 1) * 2 + flip];
            // ---- S
            ba_schur_body<0>(P, A.n_ptl, lambda, A.kcap, A.S_part, (const int*)nullptr, (const int*)nullptr, (const int2*)nullptr, bid, nblk, bal_smem);
            barrier();
            // ---- F
            ba_fold_local_body(P, A.S_part, nblk, lambda, gt, gnt);
            barrier();
            // ---- C
            if (bid == 0) ba_chol_small6_body<BAL_NT>(P, 1, lambda, bal_smem);
            barrier();
            // ---- B
            ba_backsub_chi2_body(P, A.n_ptl, lambda, gt, gnt, bal_wsum);
            for (int f = bid * nw + wave; f < ncf; f += nblk * nw) ba_camfactor_body(P, 0, P.cam_new, P.scal + 2, f, lane);
            if (qmax == 0) for (size_t t = gt; t < red_len; t += gnt) red_next[t] = 0.0;      // (the other copy was last read in the previous iteration)
            barrier();
            const bool ok2 = bal_ld(scal + 4) > 0.5;
            const double tempChi = ok2 ? bal_ld(scal + 2) : DBL_MAX, scale = ok2 ? bal_ld(scal + 3) : 0.0;
            rho = (currentChi - tempChi) / (scale + 1e-3);
            if (rho > 0 && isfinite(tempChi)) {
                const double tr = 2 * rho - 1; double alpha = 1. - tr * tr * tr; alpha = fmin(alpha, 2. / 3.);
                lambda *= fmax(1. / 3., alpha); ni = 2; currentChi = tempChi;
                flip ^= 1;                                   // the trial state becomes the accepted one
            } else { lambda *= ni; ni *= 2; }
            qmax++; trials++;
        } while (rho < 0 && qmax < 10 && !dead);
        bool terminate = (qmax == 10 || rho == 0);
        if (!terminate) { if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0; if (nBad >= 3) terminate = true; }
        const double chiNow = currentChi;
        if (chi2_check < chiNow && it > 0) terminate = true;
        chi2_check = chiNow;
        if (it == 0) lastChi = chiNow;
        else { const double gain = (lastChi - chiNow) / chiNow; lastChi = chiNow; if (gain >= 0 && gain < A.gain_threshold) terminate = true; }
        chi_final = chiNow;
        if (terminate) { it++; break; }
    }
    // the accepted state goes home: cameras into cam_out, landmarks into the caller's array when the last accepted trial left them in the other buffer
    { const BaDev& P = V.v[flip];
      for (int t = gt; t < n_cam * 12; t += gnt) A.cam_out[t] = P.cam[t];
      if (P.pt != A.pt_home) for (size_t t = gt; t < (size_t)A.n_ptl * 3; t += gnt) A.pt_home[t] = P.pt[t]; }
    if (gt == 0) {
        BaLmCtl* c = A.ctl;
        c->status = dead ? 1 : 0; c->iterations = it; c->trials = trials; c->n_lin = n_lin; c->barriers = (int)phase;
        c->lambda_final = lambda; c->chi2_initial = chi_init; c->chi2_final = chi_final; c->lin_ticks = lin_ticks;
    }
}

// ---- object part: dynamic-point chains -------------------------------------------------------------------------
// A dynamic tracklet contributes one VertexPointXYZ per frame, tied to its camera by an EdgeSE3PointXYZ and to its
// predecessor by LandmarkMotionTernaryEdge(p_prev, p_cur, H): e = p_prev - H^-1 p_cur, de/dp_prev = I,
// de/dp_cur = -R_H^T, de/dH = [I | -[H^-1 p_cur]x] (types/types_dyn_slam3d.cpp:53-85).  The point-point Hessian of a
// tracklet is therefore block tridiagonal; the chain is eliminated with a block LDL^T (3x3 blocks) instead of the
// 3x3 inverse used for static landmarks, which is the same LM step g2o takes on the un-eliminated system.
__device__ __forceinline__ void tern_eval(const double* H, const double* pp, const double* pc, double* e, double* v)
{
    const double d0 = pc[0] - H[3], d1 = pc[1] - H[7], d2 = pc[2] - H[11];
    v[0] = H[0] * d0 + H[4] * d1 + H[8] * d2; v[1] = H[1] * d0 + H[5] * d1 + H[9] * d2; v[2] = H[2] * d0 + H[6] * d1 + H[10] * d2;
    e[0] = pp[0] - v[0]; e[1] = pp[1] - v[1]; e[2] = pp[2] - v[2];
}
__device__ __forceinline__ void tern_JH(const double* v, double* J /*3x6 row-major*/)
{
#pragma unroll
    for (int a = 0; a < 18; a++) J[a] = 0;
    J[0] = J[7] = J[14] = 1;
    J[4] = v[2]; J[5] = -v[1]; J[6 + 3] = -v[2]; J[6 + 5] = v[0]; J[12 + 3] = v[1]; J[12 + 4] = -v[0];
}
__device__ __forceinline__ void pose_accum(const BaDev& P, int c, double wo, const double* J /*3x6*/, const double* e)
{
    for (int a = 0; a < 6; a++) {
        atomicAdd(P.bc + 6 * c + a, -wo * (J[a] * e[0] + J[6 + a] * e[1] + J[12 + a] * e[2]));
        for (int b = a; b < 6; b++) {
            const double h = wo * (J[a] * J[b] + J[6 + a] * J[6 + b] + J[12 + a] * J[12 + b]);
            if (h != 0.0) { atomicAdd(P.Hcd + 36 * c + a * 6 + b, h); if (b != a) atomicAdd(P.Hcd + 36 * c + b * 6 + a, h); }
        }
    }
}
// one thread per dynamic point: its camera edge, the ternary edge it closes (as p_cur) and the one it opens (as p_prev)
__global__ __launch_bounds__(256) void k_badyn_linearize(BaDev P)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    double chi = 0;
    if (k < P.n_dyn) {
        double V[6] = {0, 0, 0, 0, 0, 0}, b[3] = {0, 0, 0};
        const double* p = P.dyn + 3 * (size_t)k;
        {   // EdgeSE3PointXYZ to the camera
            const int c = P.dyn_cam[k];
            const double* X = P.cam + 12 * c; const double* m = P.dyn_meas + 3 * (size_t)k;
            const double d0 = p[0] - X[3], d1 = p[1] - X[7], d2 = p[2] - X[11];
            const double Z0 = X[0] * d0 + X[4] * d1 + X[8] * d2, Z1 = X[1] * d0 + X[5] * d1 + X[9] * d2, Z2 = X[2] * d0 + X[6] * d1 + X[10] * d2;
            const double e[3] = {Z0 - m[0], Z1 - m[1], Z2 - m[2]};
            double r0, w; huber_w(P.info_dyn * (e[0] * e[0] + e[1] * e[1] + e[2] * e[2]), P.huber_dyn, P.use_huber, r0, w);
            chi += r0;
            const double wo = w * P.info_dyn;
            const double Jc[18] = {-1, 0, 0, 0, -2 * Z2, 2 * Z1,   0, -1, 0, 2 * Z2, 0, -2 * Z0,   0, 0, -1, -2 * Z1, 2 * Z0, 0};
            const double Jp[9] = {X[0], X[4], X[8], X[1], X[5], X[9], X[2], X[6], X[10]};
            int q = 0;
            for (int a = 0; a < 3; a++) {
                b[a] -= wo * (Jp[a] * e[0] + Jp[3 + a] * e[1] + Jp[6 + a] * e[2]);
                for (int bb = a; bb < 3; bb++) V[q++] += wo * (Jp[a] * Jp[bb] + Jp[3 + a] * Jp[3 + bb] + Jp[6 + a] * Jp[6 + bb]);
            }
            double* Wk = P.Wc + 18 * (size_t)k;
            for (int a = 0; a < 6; a++) for (int bb = 0; bb < 3; bb++) Wk[a * 3 + bb] = wo * (Jc[a] * Jp[bb] + Jc[6 + a] * Jp[3 + bb] + Jc[12 + a] * Jp[6 + bb]);
            pose_accum(P, c, wo, Jc, e);
        }
        const int hin = P.dyn_hin[k];
        double* Uk = P.U + 9 * (size_t)k; double* Wik = P.Wi + 18 * (size_t)k; double* Wok = P.Wo + 18 * (size_t)k;
        if (hin >= 0) {   // ternary edge (k-1, k, H): this point is p_cur
            const double* H = P.cam + 12 * hin;
            double e[3], v[3], JH[18]; tern_eval(H, p - 3, p, e, v); tern_JH(v, JH);
            double r0, w; huber_w(P.info_tern * (e[0] * e[0] + e[1] * e[1] + e[2] * e[2]), P.huber_tern, P.use_huber, r0, w);
            chi += r0;
            const double wt = w * P.info_tern;
            const double Jc[9] = {-H[0], -H[4], -H[8], -H[1], -H[5], -H[9], -H[2], -H[6], -H[10]};      // -R^T
            int q = 0;
            for (int a = 0; a < 3; a++) {
                b[a] -= wt * (Jc[a] * e[0] + Jc[3 + a] * e[1] + Jc[6 + a] * e[2]);
                for (int bb = a; bb < 3; bb++) V[q++] += wt * (Jc[a] * Jc[bb] + Jc[3 + a] * Jc[3 + bb] + Jc[6 + a] * Jc[6 + bb]);
            }
            for (int a = 0; a < 9; a++) Uk[a] = wt * Jc[a];                                               // J_prev^T Omega J_cur, J_prev = I
            for (int a = 0; a < 6; a++) for (int bb = 0; bb < 3; bb++) Wik[a * 3 + bb] = wt * (JH[a] * Jc[bb] + JH[6 + a] * Jc[3 + bb] + JH[12 + a] * Jc[6 + bb]);
            pose_accum(P, hin, wt, JH, e);
        } else {
            for (int a = 0; a < 9; a++) Uk[a] = 0;
            for (int a = 0; a < 18; a++) Wik[a] = 0;
        }
        const int hout = (k + 1 < P.n_dyn) ? P.dyn_hin[k + 1] : -1;
        if (hout >= 0) {  // ternary edge (k, k+1, H'): this point is p_prev (J = I)
            const double* H = P.cam + 12 * hout;
            double e[3], v[3], JH[18]; tern_eval(H, p, p + 3, e, v); tern_JH(v, JH);
            double r0, w; huber_w(P.info_tern * (e[0] * e[0] + e[1] * e[1] + e[2] * e[2]), P.huber_tern, P.use_huber, r0, w);
            const double wt = w * P.info_tern;
            V[0] += wt; V[3] += wt; V[5] += wt;
            for (int a = 0; a < 3; a++) b[a] -= wt * e[a];
            for (int a = 0; a < 6; a++) for (int bb = 0; bb < 3; bb++) Wok[a * 3 + bb] = wt * JH[bb * 6 + a];
        } else {
            for (int a = 0; a < 18; a++) Wok[a] = 0;
        }
        for (int a = 0; a < 6; a++) P.Vd[6 * (size_t)k + a] = V[a];
        for (int a = 0; a < 3; a++) P.bd[3 * (size_t)k + a] = b[a];
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) chi += __shfl_xor(chi, o, 64);
    if ((threadIdx.x & 63) == 0 && chi != 0) atomicAdd(P.scal + 0, chi);
}

__device__ __forceinline__ void mat3_mul(const double* A, const double* B, double* C, bool transA)
{
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++)
            C[r * 3 + c] = transA ? (A[r] * B[c] + A[3 + r] * B[3 + c] + A[6 + r] * B[6 + c]) : (A[r * 3] * B[c] + A[r * 3 + 1] * B[3 + c] + A[r * 3 + 2] * B[6 + c]);
}
__device__ __forceinline__ void inv3full(const double* A, double* I)
{
    const double c0 = A[4] * A[8] - A[5] * A[7], c1 = A[5] * A[6] - A[3] * A[8], c2 = A[3] * A[7] - A[4] * A[6];
    const double d = 1.0 / (A[0] * c0 + A[1] * c1 + A[2] * c2);
    I[0] = c0 * d; I[1] = (A[2] * A[7] - A[1] * A[8]) * d; I[2] = (A[1] * A[5] - A[2] * A[4]) * d;
    I[3] = c1 * d; I[4] = (A[0] * A[8] - A[2] * A[6]) * d; I[5] = (A[2] * A[3] - A[0] * A[5]) * d;
    I[6] = c2 * d; I[7] = (A[1] * A[6] - A[0] * A[7]) * d; I[8] = (A[0] * A[4] - A[1] * A[3]) * d;
}
// block LDL^T of one chain per thread: D_0 = A_0, G_k = D_{k-1}^-1 U_k, D_k = A_k - U_k^T G_k (A = Vd + lambda I);
// stores D_k^-1 | G_k and yb = V^-1 bd (the chain's part of the reduced right-hand side)
__global__ __launch_bounds__(64) void k_badyn_factor(BaDev P, double lambda)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= P.n_chain) return;
    const int k0 = P.chain_start[c], k1 = P.chain_start[c + 1];
    double Dinv[9], z[3] = {0, 0, 0};
    for (int k = k0; k < k1; k++) {
        const double* V = P.Vd + 6 * (size_t)k; const double* U = P.U + 9 * (size_t)k; double* F = P.fac + 18 * (size_t)k;
        double D[9] = {V[0] + lambda, V[1], V[2], V[1], V[3] + lambda, V[4], V[2], V[4], V[5] + lambda};
        double G[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        double w[3] = {P.bd[3 * (size_t)k], P.bd[3 * (size_t)k + 1], P.bd[3 * (size_t)k + 2]};
        if (k > k0) {
            mat3_mul(Dinv, U, G, false);
            double T[9]; mat3_mul(U, G, T, true);
            for (int a = 0; a < 9; a++) D[a] -= T[a];
            for (int a = 0; a < 3; a++) w[a] -= G[a] * z[0] + G[3 + a] * z[1] + G[6 + a] * z[2];       // z_k = w_k - G_k^T z_{k-1}
        }
        inv3full(D, Dinv);
        for (int a = 0; a < 9; a++) { F[a] = Dinv[a]; F[9 + a] = G[a]; }
        for (int a = 0; a < 3; a++) { z[a] = w[a]; P.yb[3 * (size_t)k + a] = w[a]; }
    }
    double y[3] = {0, 0, 0};
    for (int k = k1 - 1; k >= k0; k--) {
        const double* F = P.fac + 18 * (size_t)k; double* yk = P.yb + 3 * (size_t)k;
        double u[3];
        for (int a = 0; a < 3; a++) u[a] = F[a * 3] * yk[0] + F[a * 3 + 1] * yk[1] + F[a * 3 + 2] * yk[2];
        if (k + 1 < k1) { const double* Gn = P.fac + 18 * (size_t)(k + 1) + 9; for (int a = 0; a < 3; a++) u[a] -= Gn[a * 3] * y[0] + Gn[a * 3 + 1] * y[1] + Gn[a * 3 + 2] * y[2]; }
        for (int a = 0; a < 3; a++) { y[a] = u[a]; yk[a] = u[a]; }
    }
}
// Reduced-system contribution of the chains: one wave per chain, one lane per column of the chain's coupling block
// W_c^T (6 columns for each camera of the chain and each H of the chain, + one column for the right-hand side): the lane
// solves V y = w with the stored factor (y kept in this wave's scratch slice, [3L][64] so that lanes coalesce) and
// subtracts W y from the lower triangle of S (r for the last column).
// zs = per-lane column scratch (3 doubles per chain point): in LDS when 64 * 3 * lmax doubles fit (use_lds), else in HBM.  Every step of the two
// substitution sweeps reads what the previous step wrote: through HBM that is a memory round trip per chain point (the whole kernel was that
// latency: 3.3 ms for 4000 chains of up to 40 points).
__global__ __launch_bounds__(64) void k_badyn_schur(BaDev P, double* __restrict__ scratch, int lmax, int use_lds)
{
    extern __shared__ double zs_lds[];
    const int lane = threadIdx.x;
    double* zs = (use_lds ? zs_lds : scratch + (size_t)blockIdx.x * 64 * 3 * lmax) + lane;
    for (int c = blockIdx.x; c < P.n_chain; c += gridDim.x) {
        const int k0 = P.chain_start[c], L = P.chain_start[c + 1] - k0;
        const int nslot = 2 * L - 1, ncol = 6 * nslot + 1;
        for (int col = lane; col < ncol; col += 64) {
            const bool rhs = col == ncol - 1;
            const int slot = col / 6, q = col - slot * 6;
            // support of the column: point j (and j-1 for an H column)
            const int j = rhs ? 0 : (slot < L ? slot : slot - L + 1);
            const bool isH = !rhs && slot >= L;
            const int first = rhs ? 0 : (isH ? j - 1 : j);
            const int pose_b = rhs ? -1 : (isH ? P.dyn_hin[k0 + j] : P.dyn_cam[k0 + j]);
            if (!rhs) {
                double z[3] = {0, 0, 0};
                for (int k = first; k < L; k++) {
                    double w[3] = {0, 0, 0};
                    if (isH) { if (k == j) { const double* W = P.Wi + 18 * (size_t)(k0 + k) + q * 3; w[0] = W[0]; w[1] = W[1]; w[2] = W[2]; }
                               else if (k == j - 1) { const double* W = P.Wo + 18 * (size_t)(k0 + k) + q * 3; w[0] = W[0]; w[1] = W[1]; w[2] = W[2]; } }
                    else if (k == j) { const double* W = P.Wc + 18 * (size_t)(k0 + k) + q * 3; w[0] = W[0]; w[1] = W[1]; w[2] = W[2]; }
                    if (k > first) { const double* G = P.fac + 18 * (size_t)(k0 + k) + 9; for (int a = 0; a < 3; a++) w[a] -= G[a] * z[0] + G[3 + a] * z[1] + G[6 + a] * z[2]; }
                    for (int a = 0; a < 3; a++) { z[a] = w[a]; zs[(size_t)(3 * k + a) * 64] = w[a]; }
                }
                double y[3] = {0, 0, 0};
                for (int k = L - 1; k >= 0; k--) {
                    double u[3] = {0, 0, 0};
                    if (k >= first) { const double* F = P.fac + 18 * (size_t)(k0 + k); const double z0 = zs[(size_t)(3 * k) * 64], z1 = zs[(size_t)(3 * k + 1) * 64], z2 = zs[(size_t)(3 * k + 2) * 64];
                                      for (int a = 0; a < 3; a++) u[a] = F[a * 3] * z0 + F[a * 3 + 1] * z1 + F[a * 3 + 2] * z2; }
                    if (k + 1 < L) { const double* Gn = P.fac + 18 * (size_t)(k0 + k + 1) + 9; for (int a = 0; a < 3; a++) u[a] -= Gn[a * 3] * y[0] + Gn[a * 3 + 1] * y[1] + Gn[a * 3 + 2] * y[2]; }
                    for (int a = 0; a < 3; a++) { y[a] = u[a]; zs[(size_t)(3 * k + a) * 64] = u[a]; }
                }
            }
            // S[(pose_a, p), (pose_b, q)] -= sum_pts W_{a,pt}[p,:] . y[pt]    (pose_a >= pose_b only)
            for (int k = 0; k < L; k++) {
                double y[3];
                if (rhs) { y[0] = P.yb[3 * (size_t)(k0 + k)]; y[1] = P.yb[3 * (size_t)(k0 + k) + 1]; y[2] = P.yb[3 * (size_t)(k0 + k) + 2]; }
                else { y[0] = zs[(size_t)(3 * k) * 64]; y[1] = zs[(size_t)(3 * k + 1) * 64]; y[2] = zs[(size_t)(3 * k + 2) * 64]; }
                const int hk = P.dyn_hin[k0 + k], hn = (k + 1 < L) ? P.dyn_hin[k0 + k + 1] : -1;
                const int poses[3] = {P.dyn_cam[k0 + k], hk, hn};
                const double* Ws[3] = {P.Wc + 18 * (size_t)(k0 + k), P.Wi + 18 * (size_t)(k0 + k), P.Wo + 18 * (size_t)(k0 + k)};
                for (int t = 0; t < 3; t++) {
                    const int pa = poses[t];
                    if (pa < 0 || (!rhs && pa < pose_b)) continue;
                    const double* W = Ws[t];
                    for (int pp = 0; pp < 6; pp++) {
                        const double v = W[pp * 3] * y[0] + W[pp * 3 + 1] * y[1] + W[pp * 3 + 2] * y[2];
                        if (v == 0.0) continue;
                        if (rhs) atomicAdd(P.r + 6 * pa + pp, -v);
                        else { double* e = s_entry(P, 6 * pa + pp, 6 * pose_b + q); if (e) atomicAdd(e, -v); }
                    }
                }
            }
        }
    }
}
// back-substitution of one chain per thread: V dx = bd - W^T x_pose, p_new = p + dx, and the chain's part of computeScale
__global__ __launch_bounds__(64) void k_badyn_backsub(BaDev P, double lambda)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    double sc = 0;
    if (c < P.n_chain) {
        const int k0 = P.chain_start[c], k1 = P.chain_start[c + 1];
        double z[3] = {0, 0, 0};
        for (int k = k0; k < k1; k++) {
            double w[3] = {P.bd[3 * (size_t)k], P.bd[3 * (size_t)k + 1], P.bd[3 * (size_t)k + 2]};
            const int hk = P.dyn_hin[k], hn = (k + 1 < k1) ? P.dyn_hin[k + 1] : -1;
            const int poses[3] = {P.dyn_cam[k], hk, hn};
            const double* Ws[3] = {P.Wc + 18 * (size_t)k, P.Wi + 18 * (size_t)k, P.Wo + 18 * (size_t)k};
            for (int t = 0; t < 3; t++) {
                if (poses[t] < 0) continue;
                const double* x = P.x + 6 * poses[t]; const double* W = Ws[t];
                for (int a = 0; a < 6; a++) { w[0] -= W[a * 3] * x[a]; w[1] -= W[a * 3 + 1] * x[a]; w[2] -= W[a * 3 + 2] * x[a]; }
            }
            if (k > k0) { const double* G = P.fac + 18 * (size_t)k + 9; for (int a = 0; a < 3; a++) w[a] -= G[a] * z[0] + G[3 + a] * z[1] + G[6 + a] * z[2]; }
            for (int a = 0; a < 3; a++) { z[a] = w[a]; P.dyn_new[3 * (size_t)k + a] = w[a]; }
        }
        double y[3] = {0, 0, 0};
        for (int k = k1 - 1; k >= k0; k--) {
            const double* F = P.fac + 18 * (size_t)k; double* zk = P.dyn_new + 3 * (size_t)k;
            double u[3];
            for (int a = 0; a < 3; a++) u[a] = F[a * 3] * zk[0] + F[a * 3 + 1] * zk[1] + F[a * 3 + 2] * zk[2];
            if (k + 1 < k1) { const double* Gn = P.fac + 18 * (size_t)(k + 1) + 9; for (int a = 0; a < 3; a++) u[a] -= Gn[a * 3] * y[0] + Gn[a * 3 + 1] * y[1] + Gn[a * 3 + 2] * y[2]; }
            for (int a = 0; a < 3; a++) { y[a] = u[a]; sc += u[a] * (lambda * u[a] + P.bd[3 * (size_t)k + a]); zk[a] = P.dyn[3 * (size_t)k + a] + u[a]; }
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) sc += __shfl_xor(sc, o, 64);
    if ((threadIdx.x & 63) == 0 && sc != 0) atomicAdd(P.scal + 3, sc);
}
__global__ __launch_bounds__(256) void k_badyn_chi2(BaDev P, const double* cam, const double* dyn, double* out)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    double v = 0;
    if (k < P.n_dyn) {
        const double* X = cam + 12 * P.dyn_cam[k]; const double* p = dyn + 3 * (size_t)k; const double* m = P.dyn_meas + 3 * (size_t)k;
        const double d0 = p[0] - X[3], d1 = p[1] - X[7], d2 = p[2] - X[11];
        const double e0 = X[0] * d0 + X[4] * d1 + X[8] * d2 - m[0], e1 = X[1] * d0 + X[5] * d1 + X[9] * d2 - m[1], e2 = X[2] * d0 + X[6] * d1 + X[10] * d2 - m[2];
        double w; huber_w(P.info_dyn * (e0 * e0 + e1 * e1 + e2 * e2), P.huber_dyn, P.use_huber, v, w);
        const int hin = P.dyn_hin[k];
        if (hin >= 0) {
            double e[3], vv[3], r0; tern_eval(cam + 12 * hin, p - 3, p, e, vv);
            huber_w(P.info_tern * (e[0] * e[0] + e[1] * e[1] + e[2] * e[2]), P.huber_tern, P.use_huber, r0, w);
            v += r0;
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0 && v != 0) atomicAdd(out, v);
}

// ---- host driver ---------------------------------------------------------------------------------------------
struct BaState {
    std::vector<void*> allocs;
    double* h_scal = nullptr;      // pinned [8]
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr;      // around the linearisation launch; around the Schur launch (global path)
    double* d_parts = nullptr; size_t parts_cap = 0;
    double* d_scratch = nullptr; size_t scratch_cap = 0;
    char* pool = nullptr; char* h_pool = nullptr; size_t pool_cap = 0, hpool_cap = 0;
    int* d_long = nullptr; size_t long_cap = 0;      // landmarks with > 64 observations
    double* d_bcr = nullptr; size_t bcr_cap = 0;     // block-cyclic-reduction workspace (superblocks D, L x2, GL, GR, b, y)
    unsigned long long* d_bcr_xt = nullptr; size_t bcr_xt_cap = 0; int2* d_bcr_blocks = nullptr; int bcr_blocks_nb = 0, bcr_blocks_n = 0; unsigned bcr_epoch = 0; unsigned* d_bcr_abort = nullptr;      // k_bcr_back_chain: tagged granules of x, its (block, level) list, the solve counter
    int* d_ticket = nullptr;                        // k_ba_backsub_chi2_local: workgroups finished so far (reset by the last one)
    std::vector<int> hv_i[8]; std::vector<double> hv_d;      // host scratch of the set-up (observation tables of the call): kept across calls — as fresh vectors they were ~40 MB of
                                                              // mmap + page faults + zero fill per global solve, the largest and most variable part of its host time
    struct BaSpecCtl* d_spec = nullptr; struct BaSpecCtl* h_spec = nullptr; int spec_last_trials = 12;      // the enqueued-ahead local solve: LM state on the device, its pinned mirror, the previous solve's trial count
    struct BaLmCtl* d_ctl = nullptr; struct BaLmCtl* h_ctl = nullptr; unsigned bar_base = 0;   // persistent local-window solver: control block (device + pinned mirror), barrier count so far
};
void ba_state_destroy(vido_ctx* ctx)
{
    BaState* S = ctx->ba; if (!S) return;
    for (void* p : S->allocs) hipFree(p);
    hipFree(S->d_parts); hipFree(S->d_scratch); hipHostFree(S->h_scal); hipFree(S->pool); hipHostFree(S->h_pool); hipFree(S->d_long); hipFree(S->d_bcr); hipFree(S->d_bcr_xt); hipFree(S->d_bcr_blocks); hipFree(S->d_bcr_abort);
    hipFree(S->d_ctl); hipHostFree(S->h_ctl); hipFree(S->d_ticket); hipFree(S->d_spec); hipHostFree(S->h_spec);
    if (S->ev0) hipEventDestroy(S->ev0);
    if (S->ev1) hipEventDestroy(S->ev1);
    if (S->ev2) hipEventDestroy(S->ev2);
    if (S->ev3) hipEventDestroy(S->ev3);
    delete S; ctx->ba = nullptr;
}

namespace {
// Worker threads for the host set-up of LARGE graphs (>= 200 k observations: the global BA; the per-frame window stays on the calling thread).  The set-up is a dozen
// passes over the observation list (validate, stable sort by camera, gather, slot tables, window tables: 12 ms for 1 M edges on one core — more than the LM loop it
// prepares); each pass is split into contiguous chunks, one per thread, and the two sorts are stable counting sorts with per-thread histograms.  Threads are created once
// per process (a spawn per pass would cost what the pass saves).
class HostPool {
public:
    static HostPool& get() { static HostPool p; return p; }
    // processes that share this host's cores: the ranks of the first solve of the process, or what the launcher says (torchrun's LOCAL_WORLD_SIZE) — the pool is created
    // by the first bundle adjustment of the process, which may be a single-rank local window
    static int& world_hint() { static int w = [] { const char* e = getenv("LOCAL_WORLD_SIZE"); const int v = e ? atoi(e) : 1; return v > 0 ? v : 1; }(); return w; }
    static int usable_cpus() {
        int n = (int)std::thread::hardware_concurrency();
        cpu_set_t cs; if (sched_getaffinity(0, sizeof cs, &cs) == 0) n = std::min(n, CPU_COUNT(&cs));
        if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) { long long q = 0, per = 0; char buf[64] = {0};
            if (fscanf(f, "%63s %lld", buf, &per) == 2 && strcmp(buf, "max") != 0 && per > 0) { q = atoll(buf); if (q > 0) n = std::min(n, (int)std::max(1ll, q / per)); } fclose(f); }
        return std::max(1, n);
    }
    int size() const { return n_; }
    static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#elif defined(__aarch64__)
        asm volatile("yield" ::: "memory");
#else
        std::this_thread::yield();
#endif
    }
    // f(tid) on n_ threads (the caller is thread 0); returns when all are done.  A set-up is ~30 passes of 30-300 us back to back: a worker that went to sleep on the
    // condition variable after every pass cost a futex wake-up (30-60 us) per pass and thread — more than a third of the set-up.  Workers therefore SPIN for the next pass
    // for a few hundred microseconds after finishing one (they sleep between calls), and the caller spins for their completion.
    void run(const std::function<void(int)>& f) {
        if (n_ == 1) { f(0); return; }
        std::lock_guard<std::mutex> one(run_m_);                 // (several contexts may set up at once — the sharded solve in threads: their passes take turns)
        { std::lock_guard<std::mutex> g(m_); job_ = &f; pending_.store(n_ - 1, std::memory_order_relaxed); gen_.fetch_add(1, std::memory_order_release); }
        if (sleepers_.load(std::memory_order_acquire) > 0) cv_.notify_all();
        f(0);
        // bounded spin, then yield: with more runnable threads than CPUs (a quota change, several ranks without LOCAL_WORLD_SIZE) a pure spin would hold the very core a
        // worker needs for a scheduler quantum
        for (int spin = 0; pending_.load(std::memory_order_acquire) != 0; spin++) { if (spin < 4000) cpu_relax(); else std::this_thread::yield(); }
        job_ = nullptr;
    }
    // chunks of [0, n): f(lo, hi, tid)
    template <class F> void chunks(size_t n, F f) { const int T = n_; run([&](int t) { const size_t lo = n * t / T, hi = n * (t + 1) / T; if (hi > lo) f(lo, hi, t); }); }
private:
    HostPool() {
        // default: the CPUs this process may use (cgroup quota, affinity mask) shared among the ranks of the solve, at most 16.  With workers that spin between passes a
        // 1 M-edge call takes 9.3 / 8.7 / 8.1 / 7.8 ms on 4 / 8 / 12 / 16 threads of a 16-CPU container (sleeping workers: 8.7 / 9.3 / - / 8.8).
        const char* e = getenv("VIDO_BA_HOST_THREADS"); int want = e ? atoi(e) : std::max(1, std::min(16, usable_cpus() / std::max(1, world_hint())));
        n_ = std::max(1, std::min(want, (int)std::thread::hardware_concurrency()));
        for (int t = 1; t < n_; t++) th_.emplace_back([this, t] { int seen = 0; for (;;) {
            bool got = false;
            for (int i = 0; i < 20000 && !got; i++) { got = stop_.load(std::memory_order_relaxed) || gen_.load(std::memory_order_acquire) != seen; if (!got) cpu_relax(); }
            if (!got) { std::unique_lock<std::mutex> g(m_); sleepers_.fetch_add(1, std::memory_order_release);
                        cv_.wait(g, [&] { return stop_.load(std::memory_order_relaxed) || gen_.load(std::memory_order_acquire) != seen; }); sleepers_.fetch_sub(1, std::memory_order_release); }
            if (stop_.load(std::memory_order_relaxed)) return;
            seen = gen_.load(std::memory_order_acquire);
            const std::function<void(int)>* j = job_;
            (*j)(t);
            pending_.fetch_sub(1, std::memory_order_release); } });
    }
    ~HostPool() { { std::lock_guard<std::mutex> g(m_); stop_.store(true); } cv_.notify_all(); for (auto& t : th_) t.join(); }
    int n_ = 1; std::atomic<int> gen_{0}, pending_{0}, sleepers_{0}; std::atomic<bool> stop_{false}; const std::function<void(int)>* volatile job_ = nullptr;
    std::mutex m_, run_m_; std::condition_variable cv_; std::vector<std::thread> th_;
};
// stable counting sort on the pool: pos[i] = rank of element i among the n elements ordered by (key, i); bin_start[b] = first rank of bin b (nbins + 1 entries)
template <class KeyFn> static void par_counting_rank(HostPool& P, size_t n, int nbins, KeyFn key, int* pos, std::vector<int>& bin_start)
{
    const int T = P.size();
    std::vector<int> hist((size_t)T * nbins, 0);
    P.chunks(n, [&](size_t lo, size_t hi, int t) { int* h = hist.data() + (size_t)t * nbins; for (size_t i = lo; i < hi; i++) h[key(i)]++; });
    bin_start.assign(nbins + 1, 0);
    std::vector<long> part(T + 1, 0);                          // bins split over the threads: totals per range, then the offsets inside
    P.chunks((size_t)nbins, [&](size_t lo, size_t hi, int t) { long s2 = 0; for (size_t b = lo; b < hi; b++) for (int u = 0; u < T; u++) s2 += hist[(size_t)u * nbins + b]; part[t + 1] = s2; });
    for (int t = 0; t < T; t++) part[t + 1] += part[t];
    P.chunks((size_t)nbins, [&](size_t lo, size_t hi, int t) { long run = part[t]; for (size_t b = lo; b < hi; b++) { bin_start[b] = (int)run; for (int u = 0; u < T; u++) { const int c = hist[(size_t)u * nbins + b]; hist[(size_t)u * nbins + b] = (int)run; run += c; } } });
    bin_start[nbins] = (int)n;
    P.chunks(n, [&](size_t lo, size_t hi, int t) { int* h = hist.data() + (size_t)t * nbins; for (size_t i = lo; i < hi; i++) pos[i] = h[key(i)]++; });
}
// the same ranking for MANY bins (landmark ids: 100 k bins — per-thread histograms of all bins would be T x nbins counters to clear and to scan, more than the n elements
// at 16 threads): two levels.  Level 1 ranks by the high bits (<= 1024 buckets, per-thread histograms) and lists the elements bucket by bucket in their original order;
// level 2 gives every bucket to one thread, which counts the low bits (<= a few hundred values) of its few thousand elements and hands out the ranks in list order — stable.
template <class KeyFn> static void par_counting_rank_large(HostPool& P, size_t n, int nbins, KeyFn key, int* pos, std::vector<int>& bin_start, std::vector<int>& scratch)
{
    int shift = 0; while (((nbins - 1) >> shift) >= 1024) shift++;
    const int nb1 = ((nbins - 1) >> shift) + 1, nlow = 1 << shift;
    if (shift == 0) { par_counting_rank(P, n, nbins, key, pos, bin_start); return; }
    std::vector<int> b1;
    par_counting_rank(P, n, nb1, [&](size_t i) { return key(i) >> shift; }, pos, b1);      // pos = level-1 rank for now
    scratch.resize(n); int* order = scratch.data();
    P.chunks(n, [&](size_t lo, size_t hi, int) { for (size_t i = lo; i < hi; i++) order[pos[i]] = (int)i; });
    bin_start.assign(nbins + 1, 0);
    P.chunks((size_t)nb1, [&](size_t blo, size_t bhi, int) {
        std::vector<int> cnt(nlow + 1);
        for (size_t b = blo; b < bhi; b++) {
            const int s0 = b1[b], e0 = b1[b + 1], k0 = (int)b << shift, kn = std::min(nlow, nbins - k0);
            std::fill(cnt.begin(), cnt.end(), 0);
            for (int q = s0; q < e0; q++) cnt[(key((size_t)order[q]) - k0) + 1]++;
            for (int k = 0; k < kn; k++) { bin_start[k0 + k] = s0 + cnt[k]; cnt[k + 1] += cnt[k]; }
            for (int q = s0; q < e0; q++) { const int i = order[q]; pos[i] = s0 + cnt[key((size_t)i) - k0]++; }
        } });
    bin_start[nbins] = (int)n;
}
static void par_memcpy(HostPool& P, void* dst, const void* src, size_t bytes)
{
    if (bytes < (1u << 20) || P.size() == 1) { memcpy(dst, src, bytes); return; }
    P.chunks(bytes >> 6, [&](size_t lo, size_t hi, int) { memcpy((char*)dst + (lo << 6), (const char*)src + (lo << 6), (hi - lo) << 6); });
    if (bytes & 63) memcpy((char*)dst + (bytes & ~(size_t)63), (const char*)src + (bytes & ~(size_t)63), bytes & 63);
}

}  // namespace
/* HOST-only test hook (tests/test_trackhost_cpu.py): the stable ranking the BA set-up sorts its observation list with — pos[i] = rank of element i among the n elements
 * ordered by (key, i), bin_start[b] = first rank of key b (nbins + 1 entries) — on the set-up's own thread pool; two-level when nbins > 1024. */
extern "C" int vido_debug_stable_rank(const int* keys, int n, int nbins, int* pos, int* bin_start)
{
    if (!keys || !pos || !bin_start || n < 0 || nbins < 1) return VIDO_E_INVALID;
    for (int i = 0; i < n; i++) if (keys[i] < 0 || keys[i] >= nbins) return VIDO_E_INVALID;
    std::vector<int> bs, scratch;
    par_counting_rank_large(HostPool::get(), (size_t)n, nbins, [&](size_t i) { return keys[i]; }, pos, bs, scratch);
    memcpy(bin_start, bs.data(), sizeof(int) * (size_t)(nbins + 1));
    return VIDO_OK;
}
namespace {
struct Arena {                       // bump allocator over the ctx's persistent BA pool (no hipMalloc per call).  The pool's first `up_cap` bytes MIRROR the pinned stage:
    char* base; size_t cap; size_t off = 0; bool failed = false;   // an uploaded array lives at the same offset on both sides, so the uploads of a call leave as ONE copy (flush) —
    char* hbase; size_t hcap = 0, hoff = 0;                        // the local window used to enqueue nine 100-byte copies per solve, each a stream operation that queues behind the
    size_t up_cap = 0, flushed = 0;                                // networks — or, for the multi-megabyte tables of a global solve, as 4 MB pieces while the host fills the next ones
    template <class T> T* get(size_t n) { const size_t b = (std::max<size_t>(n, 1) * sizeof(T) + 255) & ~(size_t)255; if (off + b > cap) { failed = true; return nullptr; } T* p = (T*)(base + off); off += b; return p; }
    template <class T> T* put(const T* src, size_t n, hipStream_t st) {
        const size_t b = (std::max<size_t>(n, 1) * sizeof(T) + 255) & ~(size_t)255;
        if (hoff + b > up_cap || hoff + b > hcap) { failed = true; return nullptr; }
        T* d = (T*)(base + hoff);
        if (n) par_memcpy(HostPool::get(), hbase + hoff, src, n * sizeof(T));
        hoff += b;
        if (hoff - flushed >= ((size_t)4 << 20)) flush(st);
        return d; }
    // a region of the stage the caller fills IN PLACE (no copy); nothing is flushed until commit()
    bool hold = false;
    template <class T> T* stage(size_t n, const T** dev) { const size_t b = (std::max<size_t>(n, 1) * sizeof(T) + 255) & ~(size_t)255;
        if (hoff + b > up_cap || hoff + b > hcap) { failed = true; *dev = nullptr; return nullptr; }
        *dev = (const T*)(base + hoff); T* h = (T*)(hbase + hoff); hoff += b; hold = true; return h; }
    void commit(hipStream_t st) { hold = false; if (hoff - flushed >= ((size_t)4 << 20)) flush(st); }
    void flush(hipStream_t st) { if (hold) return; if (hoff > flushed) { if (hipMemcpyAsync(base + flushed, hbase + flushed, hoff - flushed, hipMemcpyHostToDevice, st) != hipSuccess) failed = true; flushed = hoff; } }
};
}

// A = [S | r] (n+1 rows of n doubles); on return x = S^-1 r.  tmp: NB doubles of zeroed device scratch.
static int chol_large(vido_ctx* ctx, double* A, int n, double* x, double* okflag, double* tmp, hipStream_t st)
{
    const int nrows = n + 1;
    for (int k0 = 0; k0 < n; k0 += NB) {
        hipLaunchKernelGGL(k_chol_diag, dim3(1), dim3(256), 0, st, A, n, k0, okflag);
        const int nbk = std::min(NB, n - k0), rem_r = nrows - k0 - nbk, rem_c = n - k0 - nbk;
        if (rem_r > 0) {
            const int ntr = (rem_r + NB - 1) / NB, ntc = (rem_c + NB - 1) / NB;
            hipLaunchKernelGGL(k_chol_panel, dim3(ntr), dim3(256), 0, st, A, n, nrows, k0);
            if (ntc > 0) hipLaunchKernelGGL(k_chol_update, dim3(ntc, ntr), dim3(256), 0, st, A, n, nrows, k0);
        }
    }
    double* z = A + (size_t)n * n;                        // the rhs row now holds z = L^-1 r; solved in place
    HIP_TRY(ctx, hipMemsetAsync(tmp, 0, NB * sizeof(double), st));
    for (int k0 = ((n - 1) / NB) * NB; k0 >= 0; k0 -= NB) {
        const int below = n - k0 - std::min(NB, n - k0);
        if (below > 0) hipLaunchKernelGGL(k_chol_back_rows, dim3((below + 255) / 256), dim3(256), 0, st, A, n, k0, z, tmp);
        hipLaunchKernelGGL(k_chol_back_diag, dim3(1), dim3(64), 0, st, A, n, k0, z, tmp);
    }
    HIP_TRY(ctx, hipMemcpyAsync(x, z, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, st));
    HIP_TRY(ctx, hipGetLastError());
    return VIDO_OK;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) only when a kernel needs MORE than it has been given on this device: the local window asks for the same sizes every
// frame, and the call is not free — it takes the runtime's lock (sporadically 6-13 ms when another thread of the process holds it: measured as the largest and the most
// variable part of the global solve's host set-up).
static hipError_t ba_lds_attr(int device, const void* fn, size_t bytes)
{
    static std::mutex mu; static std::vector<std::pair<std::pair<int, const void*>, size_t>> have;
    std::lock_guard<std::mutex> g(mu);
    for (auto& h : have) if (h.first.first == device && h.first.second == fn) { if (h.second >= bytes) return hipSuccess; const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes); if (e == hipSuccess) h.second = bytes; return e; }
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) have.push_back({{device, fn}, bytes});
    return e;
}
static int ba_run(vido_ctx* ctx, vido_ba_problem* prob, vido_ba_dynamic* dynp, vido_ba_result* res, vido_allreduce_fn allreduce, void* user, const BaDevInputs* DI = nullptr);
// the local-window solve on observation / landmark arrays that already live on the device (csrc/bawin.hip): prob carries the cameras, the camera-camera factors and the
// parameters; its observation and point fields are ignored
int ba_run_device_inputs(vido_ctx* ctx, vido_ba_problem* prob, vido_ba_result* res, const BaDevInputs* DI) { return ba_run(ctx, prob, nullptr, res, nullptr, nullptr, DI); }
extern "C" int vido_ba_optimize(vido_ctx* ctx, vido_ba_problem* prob, vido_ba_result* res, vido_allreduce_fn allreduce, void* user)
{
    return ba_run(ctx, prob, nullptr, res, allreduce, user);
}
extern "C" int vido_ba_optimize_dynamic(vido_ctx* ctx, vido_ba_problem* prob, vido_ba_dynamic* dyn, vido_ba_result* res, vido_allreduce_fn allreduce, void* user)
{
    return ba_run(ctx, prob, dyn, res, allreduce, user);
}

static int ba_run(vido_ctx* ctx, vido_ba_problem* prob, vido_ba_dynamic* dynp, vido_ba_result* res, vido_allreduce_fn allreduce, void* user, const BaDevInputs* DI)
{
    if (!ctx) return VIDO_E_INVALID;
    if (!prob || !res) return vido_set_error(ctx, VIDO_E_INVALID, "ba: null problem/result");
    vido_ba_problem pcopy = *prob;
    if (DI) { pcopy.n_pt = DI->n_ptl; pcopy.n_obs = 0; pcopy.pt_lo = pcopy.pt_hi = 0; pcopy.rank = 0; pcopy.world = 1;
              if (dynp || allreduce || 6 * pcopy.n_cam > BA_LDS_MAX_N6) return vido_set_error(ctx, VIDO_E_INVALID, "ba: device-resident inputs are for the static local window (<= %d cameras)", BA_LDS_MAX_N6 / 6); }
    const vido_ba_problem& p = pcopy;
    static const vido_ba_dynamic no_dyn{};
    const vido_ba_dynamic& dy = (dynp && p.rank == 0) ? *dynp : no_dyn;      // rank 0 owns the object part (like the camera-camera factors)
    if (dy.n_H < 0 || dy.n_dyn < 0 || dy.n_tern < 0 || dy.n_smooth < 0 || (dy.n_H && !dy.H_T) || (dy.n_dyn && (!dy.dyn_xyz || !dy.dyn_cam || !dy.dyn_meas)) ||
        (dy.n_tern && (!dy.tern_prev || !dy.tern_cur || !dy.tern_H)) || (dy.n_smooth && (!dy.sm_i || !dy.sm_j)))
        return vido_set_error(ctx, VIDO_E_INVALID, "ba: malformed dynamic part");
    const int n_H = dynp ? dynp->n_H : 0;                                    // every rank carries the H vertices (replicated reduced solve)
    const int n_pose = p.n_cam + n_H;
    if (p.n_cam < 1 || p.n_pt < 0 || p.n_obs < 0 || p.n_odo < 0 || !p.cam_T || (p.n_pt && !p.pt_xyz && !DI) ||
        (p.n_obs && (!p.obs_cam || !p.obs_pt || !p.obs_meas)) || (p.n_odo && (!p.odo_i || !p.odo_j || !p.odo_T)) || p.prior_cam >= p.n_cam)
        return vido_set_error(ctx, VIDO_E_INVALID, "ba: malformed problem");
    const int pt_lo = p.pt_hi > p.pt_lo ? p.pt_lo : 0, pt_hi = p.pt_hi > p.pt_lo ? p.pt_hi : p.n_pt;
    if (pt_lo < 0 || pt_hi > p.n_pt) return vido_set_error(ctx, VIDO_E_INVALID, "ba: landmark shard [%d,%d) outside [0,%d)", pt_lo, pt_hi, p.n_pt);
    const bool owns_cam_factors = (p.rank == 0);
    const auto t_begin = std::chrono::steady_clock::now();
    static const bool setup_verbose = getenv("VIDO_BA_VERBOSE") != nullptr;
    auto t_mark = t_begin;
    auto phase = [&](const char* what) { if (!setup_verbose) return; const auto t = std::chrono::steady_clock::now();
                                         fprintf(stderr, "[ba setup] %-28s %7.3f ms\n", what, std::chrono::duration<double, std::milli>(t - t_mark).count()); t_mark = t; };
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!ctx->ba) { ctx->ba = new BaState(); HIP_TRY(ctx, hipHostMalloc((void**)&ctx->ba->h_scal, 8 * sizeof(double)));
                    HIP_TRY(ctx, hipEventCreate(&ctx->ba->ev0)); HIP_TRY(ctx, hipEventCreate(&ctx->ba->ev1)); HIP_TRY(ctx, hipEventCreate(&ctx->ba->ev2)); HIP_TRY(ctx, hipEventCreate(&ctx->ba->ev3)); }
    BaState* BS = ctx->ba;
    hipStream_t st = ctx->stream;
    const int n6 = 6 * n_pose, n_ptl = pt_hi - pt_lo;
    // ---- object part: order the dynamic points chain-major (a chain = points linked by ternary edges)
    const int nd = dy.n_dyn;
    std::vector<int> d_order, d_new(nd, -1), d_cam(nd), d_hin(nd, -1), chain_start;
    std::vector<double> d_xyz((size_t)nd * 3), d_meas((size_t)nd * 3);
    int lmax = 0;
    if (nd) {
        std::vector<int> tin(nd, -1), tout(nd, -1);
        for (int k = 0; k < dy.n_tern; k++) {
            const int a = dy.tern_prev[k], c = dy.tern_cur[k], h = dy.tern_H[k];
            if (a < 0 || a >= nd || c < 0 || c >= nd || a == c || h < 0 || h >= n_H) return vido_set_error(ctx, VIDO_E_INVALID, "ba: ternary edge %d has a bad index", k);
            if (tin[c] >= 0 || tout[a] >= 0) return vido_set_error(ctx, VIDO_E_INVALID, "ba: ternary edge %d: dynamic points must form chains (one predecessor, one successor)", k);
            tin[c] = k; tout[a] = k;
        }
        d_order.reserve(nd);
        for (int s = 0; s < nd; s++) {
            if (tin[s] >= 0) continue;
            chain_start.push_back((int)d_order.size());
            for (int k = s; ; k = dy.tern_cur[tout[k]]) {
                d_new[k] = (int)d_order.size(); d_order.push_back(k);
                if (tout[k] < 0) break;
                if ((int)d_order.size() > nd) break;
            }
            lmax = std::max(lmax, (int)d_order.size() - chain_start.back());
        }
        if ((int)d_order.size() != nd) return vido_set_error(ctx, VIDO_E_INVALID, "ba: ternary edges form a cycle");
        chain_start.push_back(nd);
        for (int t = 0; t < nd; t++) {
            const int k = d_order[t];
            if (dy.dyn_cam[k] < 0 || dy.dyn_cam[k] >= p.n_cam) return vido_set_error(ctx, VIDO_E_INVALID, "ba: dynamic point %d has a bad camera", k);
            d_cam[t] = dy.dyn_cam[k]; d_hin[t] = tin[k] >= 0 ? p.n_cam + dy.tern_H[tin[k]] : -1;
            for (int a = 0; a < 3; a++) { d_xyz[3 * (size_t)t + a] = dy.dyn_xyz[3 * (size_t)k + a]; d_meas[3 * (size_t)t + a] = dy.dyn_meas[3 * (size_t)k + a]; }
        }
    }
    const int n_chain = nd ? (int)chain_start.size() - 1 : 0;
    // camera-camera factor list: odometry edges, then the object-motion smoothness edges (EdgeSE3 between H vertices, Z = I)
    const int n_cc = p.n_odo + dy.n_smooth;
    std::vector<int> cc_i(n_cc), cc_j(n_cc); std::vector<double> cc_T((size_t)n_cc * 12), cc_info(n_cc), cc_delta(n_cc);
    for (int k = 0; k < p.n_odo; k++) {
        if (p.odo_i[k] < 0 || p.odo_i[k] >= p.n_cam || p.odo_j[k] < 0 || p.odo_j[k] >= p.n_cam) return vido_set_error(ctx, VIDO_E_INVALID, "ba: odometry edge %d has a bad index", k);
        cc_i[k] = p.odo_i[k]; cc_j[k] = p.odo_j[k]; memcpy(&cc_T[12 * (size_t)k], p.odo_T + 12 * (size_t)k, 12 * sizeof(double)); cc_info[k] = p.info_odo; cc_delta[k] = p.huber_odo;
    }
    for (int k = 0; k < dy.n_smooth; k++) {
        if (dy.sm_i[k] < 0 || dy.sm_i[k] >= n_H || dy.sm_j[k] < 0 || dy.sm_j[k] >= n_H) return vido_set_error(ctx, VIDO_E_INVALID, "ba: smoothness edge %d has a bad index", k);
        const int t = p.n_odo + k; static const double I12[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
        cc_i[t] = p.n_cam + dy.sm_i[k]; cc_j[t] = p.n_cam + dy.sm_j[k]; memcpy(&cc_T[12 * (size_t)t], I12, sizeof I12); cc_info[t] = dy.info_smooth; cc_delta[t] = dy.huber_smooth;
    }
    // Pose order inside the solver.  Without object vertices: the cameras as given.  With them: interleaved by frame (camera f, then the object
    // motions of frame f) so that every factor couples poses that are close in the order and the reduced system stays banded.  The frame of an
    // object motion is the camera of the dynamic point it moves INTO (LandmarkMotionTernaryEdge(p_prev, p_cur, H)); unused H go last.
    std::vector<int> perm(n_pose);
    for (int i = 0; i < n_pose; i++) perm[i] = i;
    if (n_H) {
        std::vector<int> hframe(n_H, p.n_cam);
        for (int k = 0; k < dynp->n_tern; k++) {
            const int h = dynp->tern_H[k], c = dynp->tern_cur[k];
            if (h >= 0 && h < n_H && c >= 0 && c < dynp->n_dyn) { const int f = dynp->dyn_cam[c]; if (f >= 0 && f < p.n_cam) hframe[h] = std::min(hframe[h], f); }
        }
        // motions that no ternary edge uses (the object was tracked but none of its points survived into a tracklet) get their frame from the
        // smoothness edges (consecutive frames of one object); anything still unknown goes last
        for (bool changed = true; changed;) {
            changed = false;
            for (int k = 0; k < dynp->n_smooth; k++) {
                const int a = dynp->sm_i[k], b = dynp->sm_j[k];
                if (a < 0 || a >= n_H || b < 0 || b >= n_H) continue;
                if (hframe[a] == p.n_cam && hframe[b] != p.n_cam) { hframe[a] = std::max(hframe[b] - 1, 0); changed = true; }
                else if (hframe[b] == p.n_cam && hframe[a] != p.n_cam) { hframe[b] = std::min(hframe[a] + 1, p.n_cam - 1); changed = true; }
            }
        }
        std::vector<int> cnt(p.n_cam + 2, 0);                              // counting sort on the frame: camera f first, then its H in input order
        for (int h = 0; h < n_H; h++) cnt[hframe[h] + 1]++;
        for (int f = 0; f <= p.n_cam; f++) cnt[f + 1] += cnt[f];            // cnt[f] = number of H with frame < f
        std::vector<int> fillh(cnt.begin(), cnt.end() - 1);
        for (int f = 0; f < p.n_cam; f++) perm[f] = f + cnt[f];
        for (int h = 0; h < n_H; h++) { const int f = hframe[h]; perm[p.n_cam + h] = std::min(f + 1, p.n_cam) + fillh[f]++; }
    }
    for (int k = 0; k < n_cc; k++) { cc_i[k] = perm[cc_i[k]]; cc_j[k] = perm[cc_j[k]]; }
    for (int t = 0; t < nd; t++) { d_cam[t] = perm[d_cam[t]]; if (d_hin[t] >= 0) d_hin[t] = perm[d_hin[t]]; }
    std::vector<double> poses((size_t)n_pose * 12);
    for (int i = 0; i < p.n_cam; i++) memcpy(poses.data() + (size_t)perm[i] * 12, p.cam_T + (size_t)i * 12, 12 * sizeof(double));
    for (int h = 0; h < n_H; h++) memcpy(poses.data() + (size_t)perm[p.n_cam + h] * 12, dynp->H_T + (size_t)h * 12, 12 * sizeof(double));
    phase("factors, pose order");
    // ---- host preprocessing: keep this shard's observations, sort by camera, build the landmark-major slots
    HostPool::world_hint() = std::max(HostPool::world_hint(), p.world);
    HostPool& HP = HostPool::get();
    const bool par = !DI && p.n_obs >= 200000 && HP.size() > 1;      // the same tables either way; see HostPool
    // A SLAM map appends its observations frame by frame: the list usually arrives sorted by camera already, and an unsharded solve keeps all of it.  The validation pass
    // notes both (`in_order`: cameras never decrease along the list; `all_kept`: every observation belongs to this shard), and the sort — and, with all kept, the list of
    // kept indices itself — is skipped: `kidx(t)` is then the identity.
    std::vector<int>& keep = BS->hv_i[0]; keep.clear();
    bool in_order = true, all_kept = false;
    if (par) {
        std::vector<int> bad(HP.size(), -1), cntk(HP.size() + 1, 0), unord(HP.size(), 0);
        HP.chunks((size_t)p.n_obs, [&](size_t lo, size_t hi, int t) { int c = 0, u = 0; int prev = lo > 0 ? p.obs_cam[lo - 1] : -1; prev = (prev >= 0 && prev < p.n_cam) ? perm[prev] : -1;
            for (size_t k = lo; k < hi; k++) {
            if (p.obs_cam[k] < 0 || p.obs_cam[k] >= p.n_cam || p.obs_pt[k] < 0 || p.obs_pt[k] >= p.n_pt) { if (bad[t] < 0) bad[t] = (int)k; continue; }
            const int pc = perm[p.obs_cam[k]]; u |= pc < prev; prev = pc;
            if (p.obs_pt[k] >= pt_lo && p.obs_pt[k] < pt_hi) c++; } cntk[t + 1] = c; unord[t] = u; });
        for (int t = 0; t < HP.size(); t++) { if (bad[t] >= 0) return vido_set_error(ctx, VIDO_E_INVALID, "ba: observation %d has a bad index", bad[t]); cntk[t + 1] += cntk[t]; in_order = in_order && !unord[t]; }
        all_kept = cntk[HP.size()] == p.n_obs;
        if (!(in_order && all_kept)) {
            keep.resize(cntk[HP.size()]);
            HP.chunks((size_t)p.n_obs, [&](size_t lo, size_t hi, int t) { int w = cntk[t]; for (size_t k = lo; k < hi; k++) if (p.obs_pt[k] >= pt_lo && p.obs_pt[k] < pt_hi) keep[w++] = (int)k; });
        }
    } else {
        keep.reserve(p.n_obs);
        int prev = -1;
        for (int k = 0; k < p.n_obs; k++) {
            if (p.obs_cam[k] < 0 || p.obs_cam[k] >= p.n_cam || p.obs_pt[k] < 0 || p.obs_pt[k] >= p.n_pt) return vido_set_error(ctx, VIDO_E_INVALID, "ba: observation %d has a bad index", k);
            const int pc = perm[p.obs_cam[k]]; in_order = in_order && pc >= prev; prev = pc;
            if (p.obs_pt[k] >= pt_lo && p.obs_pt[k] < pt_hi) keep.push_back(k);
        }
        all_kept = (int)keep.size() == p.n_obs;
    }
    const bool identity = !DI && in_order && all_kept && par;          // (the serial path always has its list)
    const int no = DI ? DI->no : (identity ? p.n_obs : (int)keep.size());
    auto kidx = [&](size_t t) -> int { return identity ? (int)t : keep[t]; };
    if (!DI && !in_order) {   // stable counting sort by camera (O(n); a comparison sort of 1M observations costs more than the whole LM loop)
        std::vector<int>& sorted = BS->hv_i[1]; sorted.resize(no);
        if (par) {
            std::vector<int>& rank = BS->hv_i[2]; rank.resize(no); std::vector<int> bs;
            par_counting_rank(HP, (size_t)no, n_pose, [&](size_t t) { return perm[p.obs_cam[keep[t]]]; }, rank.data(), bs);
            HP.chunks((size_t)no, [&](size_t lo, size_t hi, int) { for (size_t t = lo; t < hi; t++) sorted[rank[t]] = keep[t]; });
        } else {
            std::vector<int> cstart(n_pose + 1, 0);
            for (int t = 0; t < no; t++) cstart[perm[p.obs_cam[keep[t]]] + 1]++;
            for (int c = 0; c < n_pose; c++) cstart[c + 1] += cstart[c];
            for (int t = 0; t < no; t++) sorted[cstart[perm[p.obs_cam[keep[t]]]]++] = keep[t];
        }
        keep.swap(sorted);
    }
    phase("filter + sort by camera");
    const int nh = DI ? 0 : no;                              // host-side observation arrays (none when the inputs are device-resident)
    // ---- device buffers
    size_t up_this = 0;                                      // upload bytes of THIS call (estimate with slack): the mirrored region of the arena
    {   // size the persistent pool (device + pinned mirror for the uploads) for this problem
        const size_t ndb = (size_t)n_pose * (24 + 36 + 36 + 36 + 12) + (size_t)n_ptl * (6 + 6 + 3) + (size_t)no * (3 + BA_REC) + (size_t)n_cc * (12 + 36 + 2) + (size_t)n6 * n6 + 6 * (size_t)n6 + 128 +
                           (size_t)nd * (3 + 3 + 3 + 6 + 3 + 9 + 18 * 4 + 3);
        const size_t ni32 = 2 * (size_t)n_pose + 5 * (size_t)no + 4 * (size_t)n_ptl + (size_t)n_ptl / 32 + 2 * (size_t)n_cc + 2 * (size_t)nd + (size_t)n_chain + 256;
        const size_t need = ndb * 8 + ni32 * 4 + 96 * 256 + 64 * 256 + 8192;      // (+ the slack of the upload region's estimate below)
        if (need > BS->pool_cap) {
            HIP_TRY(ctx, hipStreamSynchronize(st));
            if (BS->pool) { hipFree(BS->pool); BS->pool = nullptr; }
            BS->pool_cap = need + need / 4;
            HIP_TRY(ctx, hipMalloc((void**)&BS->pool, BS->pool_cap));
        }
        // pinned stage: only the arrays that are uploaded (poses, points, observation lists, factors, index tables) — a quarter of the pool;
        // pinning memory is the slow part of a cold call (~0.1 ms per MB)
        const size_t up = ((size_t)n_pose * 12 + (size_t)n_ptl * 3 + (size_t)no * 3 + (size_t)n_cc * 14 + (size_t)nd * 6) * 8 +
                          ((size_t)2 * n_pose + 5 * (size_t)no + 5 * (size_t)n_ptl + 2 * (size_t)n_cc + 2 * (size_t)nd + (size_t)n_chain + 64) * 4 + 64 * 256;
        up_this = (up + 255) & ~(size_t)255;
        if (up > BS->hpool_cap) {
            HIP_TRY(ctx, hipStreamSynchronize(st));
            if (BS->h_pool) { hipHostFree(BS->h_pool); BS->h_pool = nullptr; }
            BS->hpool_cap = up + up / 4;
            HIP_TRY(ctx, hipHostMalloc((void**)&BS->h_pool, BS->hpool_cap));
        }
    }
    phase("pool sizing");
    Arena A; A.base = BS->pool; A.cap = BS->pool_cap; A.hbase = BS->h_pool; A.hcap = BS->hpool_cap;
    A.up_cap = up_this; A.off = A.up_cap;                      // uploads in [0, up_cap), everything else behind
    BaDev D{};
    D.n_cam = n_pose; D.n_pt = p.n_pt; D.n_obs = no; D.n_odo = owns_cam_factors ? n_cc : 0; D.prior_cam = (owns_cam_factors && p.prior_cam >= 0) ? perm[p.prior_cam] : -1;
    D.use_huber = p.use_huber; D.n6 = n6; D.pt_lo = pt_lo;
    D.info_obs = p.info_obs; D.info_odo = p.info_odo; D.info_prior = p.info_prior; D.huber_obs = p.huber_obs; D.huber_odo = p.huber_odo;
    memcpy(D.prior_T, p.prior_T, sizeof D.prior_T);
    D.cam = A.put(poses.data(), (size_t)n_pose * 12, st); D.cam_new = A.get<double>((size_t)n_pose * 12);
    std::vector<int>&ocam = BS->hv_i[3], &opt = BS->hv_i[4], &opos = BS->hv_i[5], &slotcam = BS->hv_i[6]; std::vector<int> pstart(n_ptl + 1, 0);
    ocam.resize(nh); opt.resize(nh); opos.resize(nh); slotcam.resize(nh);      // (every element is written below)
    // the measurements (24 MB at 1 M observations) are gathered straight into the pinned stage: as a vector of their own they were written once and copied once more
    const double* d_omeas = nullptr; double* omeas = DI ? nullptr : A.stage<double>((size_t)nh * 3, &d_omeas);
    if (A.failed) return vido_set_error(ctx, VIDO_E_NOMEM, "ba: upload staging exhausted");
    int maxk = 0;
    if (par) {
        HP.chunks((size_t)nh, [&](size_t lo, size_t hi, int) { for (size_t t = lo; t < hi; t++) { const int k = kidx(t); ocam[t] = perm[p.obs_cam[k]]; opt[t] = p.obs_pt[k] - pt_lo;
                                                                 for (int a = 0; a < 3; a++) omeas[3 * t + a] = p.obs_meas[3 * (size_t)k + a]; } });
        A.commit(st);                                          // (the measurements leave for the device while the slot tables are built)
        // the slots of a landmark in ascending camera order = the stable order of the camera-sorted list by landmark
        par_counting_rank_large(HP, (size_t)nh, n_ptl, [&](size_t t) { return opt[t]; }, opos.data(), pstart, BS->hv_i[2]);      // (hv_i[2]: the camera sort's rank array, free again)
        HP.chunks((size_t)nh, [&](size_t lo, size_t hi, int) { for (size_t t = lo; t < hi; t++) slotcam[opos[t]] = ocam[t]; });
        std::vector<int> mk(HP.size(), 0);
        HP.chunks((size_t)n_ptl, [&](size_t lo, size_t hi, int t) { int m = 0; for (size_t l = lo; l < hi; l++) m = std::max(m, pstart[l + 1] - pstart[l]); mk[t] = m; });
        for (int m : mk) maxk = std::max(maxk, m);
    } else {
        for (int t = 0; t < nh; t++) { const int k = kidx((size_t)t); ocam[t] = perm[p.obs_cam[k]]; opt[t] = p.obs_pt[k] - pt_lo; for (int a = 0; a < 3; a++) omeas[3 * (size_t)t + a] = p.obs_meas[3 * (size_t)k + a]; pstart[opt[t] + 1]++; }
        for (int l = 0; l < n_ptl; l++) { maxk = std::max(maxk, pstart[l + 1]); pstart[l + 1] += pstart[l]; }
        { std::vector<int> fill(pstart.begin(), pstart.end() - 1); for (int t = 0; t < nh; t++) { opos[t] = fill[opt[t]]++; slotcam[opos[t]] = ocam[t]; } }
    }
    if (DI) maxk = DI->kcap;                                 // (a landmark of the window has at most one observation per keyframe)
    // the Schur kernels add WD_i W_j^T for slot pairs i >= j into the LOWER triangle and assume the slots of a landmark belong to distinct cameras (for two
    // slots of one camera the transposed term would be missing); the observations are sorted by camera, so duplicates are adjacent slots
    std::vector<int> long_list;                             // landmarks with more than 64 observations: k_ba_schur_long
    if (!DI) {
        std::vector<int> dup(HP.size(), -1); std::vector<std::vector<int> > longs(HP.size());
        auto scan = [&](size_t lo, size_t hi, int t) { for (size_t l = lo; l < hi; l++) {
            for (int q = pstart[l] + 1; q < pstart[l + 1]; q++) if (slotcam[q] == slotcam[q - 1] && dup[t] < 0) dup[t] = (int)l;
            if (pstart[l + 1] - pstart[l] > 64) longs[t].push_back((int)l); } };
        if (par) HP.chunks((size_t)n_ptl, scan); else scan(0, (size_t)n_ptl, 0);
        for (int t = 0; t < HP.size(); t++) if (dup[t] >= 0) { const int l = dup[t]; int q = pstart[l] + 1; while (slotcam[q] != slotcam[q - 1]) q++;
            return vido_set_error(ctx, VIDO_E_INVALID, "ba: landmark %d is observed twice from camera %d (merge duplicate observations first)", l + pt_lo, slotcam[q]); }
        for (auto& v : longs) long_list.insert(long_list.end(), v.begin(), v.end());
    }
    maxk = std::min(maxk, 64);
    phase("slot tables, checks");
    if (DI) {                                                // observations, landmarks and their index tables are already on the device
        D.pt = DI->pt; D.pt_new = A.get<double>((size_t)n_ptl * 3);
        D.obs_cam = DI->obs_cam; D.obs_pt = DI->obs_pt; D.obs_pos = DI->obs_pos; D.obs_meas = DI->obs_meas; D.pt_start = DI->pt_start; D.slot_cam = DI->slot_cam;
    } else {
    D.pt = A.put(p.pt_xyz + 3 * (size_t)pt_lo, (size_t)n_ptl * 3, st); D.pt_new = A.get<double>((size_t)n_ptl * 3);
    D.obs_cam = A.put(ocam.data(), no, st); D.obs_pt = A.put(opt.data(), no, st); D.obs_pos = A.put(opos.data(), no, st);
    A.commit(st); D.obs_meas = d_omeas; D.pt_start = A.put(pstart.data(), n_ptl + 1, st); D.slot_cam = A.put(slotcam.data(), no, st);
    }
    phase("staging copies (obs, points)");
    D.odo_i = A.put(cc_i.data(), n_cc, st); D.odo_j = A.put(cc_j.data(), n_cc, st); D.odo_T = A.put(cc_T.data(), (size_t)n_cc * 12, st);
    D.odo_info = A.put(cc_info.data(), n_cc, st); D.odo_delta = A.put(cc_delta.data(), n_cc, st);
    D.W = A.get<double>((size_t)no * BA_REC); D.Cp = D.W + 18;      /* the arena hands out 256-byte aligned blocks: records are line-aligned */
    D.Hpp = A.get<double>((size_t)n_ptl * 6); D.bp = A.get<double>((size_t)n_ptl * 3);
    D.Hodo = A.get<double>((size_t)n_cc * 36);
    // object part
    D.n_dyn = nd; D.n_chain = n_chain; D.info_dyn = dy.info_dyn; D.info_tern = dy.info_tern; D.huber_dyn = dy.huber_dyn; D.huber_tern = dy.huber_tern;
    if (nd) {
        D.dyn = A.put(d_xyz.data(), (size_t)nd * 3, st); D.dyn_new = A.get<double>((size_t)nd * 3); D.dyn_meas = A.put(d_meas.data(), (size_t)nd * 3, st);
        D.dyn_cam = A.put(d_cam.data(), nd, st); D.dyn_hin = A.put(d_hin.data(), nd, st); D.chain_start = A.put(chain_start.data(), n_chain + 1, st);
        D.Vd = A.get<double>((size_t)nd * 6); D.bd = A.get<double>((size_t)nd * 3); D.U = A.get<double>((size_t)nd * 9);
        D.Wc = A.get<double>((size_t)nd * 18); D.Wi = A.get<double>((size_t)nd * 18); D.Wo = A.get<double>((size_t)nd * 18); D.fac = A.get<double>((size_t)nd * 18);
        D.yb = A.get<double>((size_t)nd * 3);
    }
    const int dyn_grid = std::min(n_chain, 1024);
    const size_t dyn_lds = (size_t)64 * 3 * lmax * sizeof(double) <= 150 * 1024 ? (size_t)64 * 3 * std::max(lmax, 1) * sizeof(double) : 0;
    if (nd && dyn_lds) HIP_TRY(ctx, ba_lds_attr(ctx->device, (const void*)k_badyn_schur, (size_t)(dyn_lds)));
    if (nd && (size_t)dyn_grid * 64 * 3 * lmax > BS->scratch_cap) {      // per-wave column scratch of k_badyn_schur (device only, not mirrored)
        HIP_TRY(ctx, hipStreamSynchronize(st)); if (BS->d_scratch) hipFree(BS->d_scratch);
        BS->scratch_cap = (size_t)1024 * 64 * 3 * lmax; HIP_TRY(ctx, hipMalloc((void**)&BS->d_scratch, BS->scratch_cap * sizeof(double)));
    }
    // S and r are contiguous so that one all-reduce covers both
    // band layout of the reduced system when the map is sequential (every landmark / odometry edge spans few keyframes)
    const bool lds_path = n6 <= BA_LDS_MAX_N6;
    D.bw = -1; D.ldb = n6;
    if (!lds_path) {
        int bwc = 0;
        if (par && pt_lo == 0 && pt_hi == p.n_pt) {                // unsharded: the slot tables already hold every landmark's cameras in ascending order
            std::vector<int> bw_t(HP.size(), 0);
            HP.chunks((size_t)n_ptl, [&](size_t lo, size_t hi, int t) { int m = 0; for (size_t l = lo; l < hi; l++) if (pstart[l + 1] > pstart[l]) m = std::max(m, slotcam[pstart[l + 1] - 1] - slotcam[pstart[l]]); bw_t[t] = m; });
            for (int m : bw_t) bwc = std::max(bwc, m);
        } else if (par) {                                          // a shard: the layout must be the one every rank picks, so all observations count (per-thread tables, merged)
            const int T = HP.size(); std::vector<int> cmn((size_t)T * p.n_pt, n_pose), cmx((size_t)T * p.n_pt, -1);
            HP.chunks((size_t)p.n_obs, [&](size_t lo, size_t hi, int t) { int* a = cmn.data() + (size_t)t * p.n_pt; int* b2 = cmx.data() + (size_t)t * p.n_pt;
                for (size_t k = lo; k < hi; k++) { const int l = p.obs_pt[k], c = perm[p.obs_cam[k]]; a[l] = std::min(a[l], c); b2[l] = std::max(b2[l], c); } });
            std::vector<int> bw_t(T, 0);
            HP.chunks((size_t)p.n_pt, [&](size_t lo, size_t hi, int t) { int m = 0; for (size_t l = lo; l < hi; l++) { int a = n_pose, b2 = -1;
                for (int u = 0; u < T; u++) { a = std::min(a, cmn[(size_t)u * p.n_pt + l]); b2 = std::max(b2, cmx[(size_t)u * p.n_pt + l]); } if (b2 >= 0) m = std::max(m, b2 - a); } bw_t[t] = m; });
            for (int m : bw_t) bwc = std::max(bwc, m);
        } else {
            std::vector<int> cmin(p.n_pt, n_pose), cmax(p.n_pt, -1);
            for (int k = 0; k < p.n_obs; k++) { const int l = p.obs_pt[k], c = perm[p.obs_cam[k]]; cmin[l] = std::min(cmin[l], c); cmax[l] = std::max(cmax[l], c); }   // ALL shards: every rank must pick the same layout
            for (int l = 0; l < p.n_pt; l++) if (cmax[l] >= 0) bwc = std::max(bwc, cmax[l] - cmin[l]);
        }
        for (int k = 0; k < p.n_odo; k++) bwc = std::max(bwc, std::abs(perm[p.odo_i[k]] - perm[p.odo_j[k]]));
        if (dynp) {   // a dynamic tracklet couples every pose it touches with every other one (its point block is eliminated as a whole)
            for (int k = 0; k < dynp->n_smooth; k++) bwc = std::max(bwc, std::abs(perm[p.n_cam + dynp->sm_i[k]] - perm[p.n_cam + dynp->sm_j[k]]));
            std::vector<int> root(dynp->n_dyn), lo(dynp->n_dyn, n_pose), hi(dynp->n_dyn, -1);
            for (int k = 0; k < dynp->n_dyn; k++) root[k] = k;
            auto find = [&](int a) { while (root[a] != a) { root[a] = root[root[a]]; a = root[a]; } return a; };
            for (int k = 0; k < dynp->n_tern; k++) { const int a = find(dynp->tern_prev[k]), c = find(dynp->tern_cur[k]); if (a != c) root[a] = c; }
            for (int k = 0; k < dynp->n_dyn; k++) { const int r0 = find(k), c = perm[dynp->dyn_cam[k]]; lo[r0] = std::min(lo[r0], c); hi[r0] = std::max(hi[r0], c); }
            for (int k = 0; k < dynp->n_tern; k++) { const int r0 = find(dynp->tern_cur[k]), h = perm[p.n_cam + dynp->tern_H[k]]; lo[r0] = std::min(lo[r0], h); hi[r0] = std::max(hi[r0], h); }
            for (int k = 0; k < dynp->n_dyn; k++) if (hi[k] >= 0) bwc = std::max(bwc, hi[k] - lo[k]);
        }
        if (6 * bwc + 5 <= CB_MAXBW && 6 * bwc + 6 < n6 / 2) { D.bw = 6 * bwc + 5; D.ldb = (D.bw + 2) & ~1; }
    }
    const size_t sz_S = D.bw >= 0 ? (size_t)n6 * D.ldb : (size_t)n6 * n6;
    if (D.bw >= 0) HIP_TRY(ctx, ba_lds_attr(ctx->device, (const void*)k_chol_band, (size_t)(((size_t)D.bw * (CB_NB + 1) * sizeof(double)))));
    size_t band6_lds = 0;                                   // pose-block LDS-window factorisation when the window fits
    if (D.bw >= 0) { const size_t bwc = (D.bw - 5) / 6, wr = 6 * (bwc + 1), need = (wr * (wr + 1) + 3 * wr + 6 * bwc * 7 + 8) * sizeof(double);
                     if (bwc >= 1 && need <= 150 * 1024) { band6_lds = need; HIP_TRY(ctx, ba_lds_attr(ctx->device, (const void*)k_chol_band6, (size_t)(need))); } }
    size_t band6s_lds = 0;                                  // pose-block factorisation in place on the band (window in L2) when the LDS window does not fit
    int band6s_nb = 0;                                      // pivot blocks per supernode: 8 when the panel fits LDS, else 4
    if (D.bw >= 0 && !band6_lds) { const size_t bwc = (D.bw - 5) / 6, wr = 6 * (bwc + 1);
                     auto need = [&](size_t nb) { const size_t rr = 6 * (bwc + nb); return (rr * (6 * nb + 1) + rr + 2 * wr + 8) * sizeof(double); };
                     if (bwc >= 1 && bwc <= 96) { band6s_nb = need(8) <= 150 * 1024 ? 8 : 4; band6s_lds = need(band6s_nb);
                     HIP_TRY(ctx, ba_lds_attr(ctx->device, band6s_nb == 8 ? (const void*)k_chol_band6s<8> : (const void*)k_chol_band6s<4>, (size_t)(band6s_lds))); } }
    double* Sr = A.get<double>(sz_S + n6); D.S = Sr; D.r = Sr + sz_S; D.x = A.get<double>(n6);
    // block cyclic reduction of the band system when it is long enough to pay (>= 4 superblocks); the superblock is 66 or 96 unknowns (>= half-bandwidth + 1), the two
    // sizes the register-resident elimination kernel is instantiated for
    BcrDev Bc{};
    if (D.bw >= 0 && D.bw + 1 <= 96 && !getenv("VIDO_BA_NO_BCR")) {
        const int m = D.bw + 1 <= 66 ? 66 : 96, nb = (n6 + m - 1) / m;
        if (nb >= 4) {
            const size_t mm = (size_t)nb * m * m, need = 5 * mm + 2 * (size_t)nb * m;
            if (need > BS->bcr_cap) { HIP_TRY(ctx, hipStreamSynchronize(st)); if (BS->d_bcr) hipFree(BS->d_bcr); BS->bcr_cap = need + need / 4; HIP_TRY(ctx, hipMalloc((void**)&BS->d_bcr, BS->bcr_cap * sizeof(double))); }
            Bc.m = m; Bc.nb = nb; Bc.n6 = n6; Bc.bw = D.bw; Bc.ldb = D.ldb;
            Bc.D = BS->d_bcr; Bc.L0 = Bc.D + mm; Bc.L1 = Bc.L0 + mm; Bc.GL = Bc.L1 + mm; Bc.GR = Bc.GL + mm; Bc.b = Bc.GR + mm; Bc.y = Bc.b + (size_t)nb * m;
            Bc.S = D.S; Bc.r = D.r; Bc.x = D.x;
            HIP_TRY(ctx, ba_lds_attr(ctx->device, (const void*)k_bcr_elim<66>, (size_t)(BcrGeom<66>::LDS)));
            HIP_TRY(ctx, ba_lds_attr(ctx->device, (const void*)k_bcr_elim<96>, (size_t)(BcrGeom<96>::LDS)));
            // the one-launch back substitution: granule buffer (zeroed once: epoch 0 is never used), the (block, level) list with the top of the tree first
            const size_t xt_need = (size_t)nb * m * 2;
            if (xt_need > BS->bcr_xt_cap) { HIP_TRY(ctx, hipStreamSynchronize(st)); if (BS->d_bcr_xt) hipFree(BS->d_bcr_xt); BS->bcr_xt_cap = xt_need; HIP_TRY(ctx, hipMalloc((void**)&BS->d_bcr_xt, xt_need * 8)); HIP_TRY(ctx, hipMemset(BS->d_bcr_xt, 0, xt_need * 8)); BS->bcr_epoch = 0; }
            if (!BS->d_bcr_abort) { HIP_TRY(ctx, hipMalloc((void**)&BS->d_bcr_abort, 4)); HIP_TRY(ctx, hipMemset(BS->d_bcr_abort, 0, 4)); }
            if (BS->bcr_blocks_nb != nb) {
                std::vector<int2> bl; int smax = 0;
                for (int sft = 1; sft < nb; sft *= 2) smax = sft;
                for (int sft = smax; sft >= 1; sft /= 2) for (int pb = sft; pb < nb; pb += 2 * sft) bl.push_back(make_int2(pb, sft));
                HIP_TRY(ctx, hipStreamSynchronize(st)); if (BS->d_bcr_blocks) hipFree(BS->d_bcr_blocks);
                HIP_TRY(ctx, hipMalloc((void**)&BS->d_bcr_blocks, bl.size() * sizeof(int2))); HIP_TRY(ctx, hipMemcpy(BS->d_bcr_blocks, bl.data(), bl.size() * sizeof(int2), hipMemcpyHostToDevice));
                BS->bcr_blocks_nb = nb; BS->bcr_blocks_n = (int)bl.size();
            }
        }
    }
    if (getenv("VIDO_BA_VERBOSE")) fprintf(stderr, "[ba] poses %d (cams %d + H %d) landmarks %d obs %d dyn %d | bw %d (block half-width %d) %s\n", n_pose, p.n_cam, n_H, n_ptl, no, nd, D.bw, D.bw >= 0 ? (D.bw - 5) / 6 : -1,
                                             lds_path ? "LDS-resident system" : (D.bw < 0 ? "dense" : (band6_lds ? "pose-block band, LDS window" : (band6s_lds ? "pose-block band, window in L2" : "scalar band"))));
    double* chol_tmp = A.get<double>(64);
    double* red = A.get<double>((size_t)n_pose * 36 + n6 + 8);      // [Hcd | bc | scal] contiguous for the linearisation all-reduce
    D.Hcd = red; D.bc = red + (size_t)n_pose * 36; D.scal = D.bc + n6;
    Bc.ok = D.scal + 4;
    double* red1 = A.get<double>((size_t)n_pose * 36 + n6 + 8);     // second copy of the linearisation accumulators + the accepted cameras (persistent local-window solver)
    double* cam_out = A.get<double>((size_t)n_pose * 12);
    if (A.failed) return vido_set_error(ctx, VIDO_E_NOMEM, "ba: device allocation failed (n6=%d, n_obs=%d)", n6, no);
#ifndef BA_SCHUR0_GRID
#define BA_SCHUR0_GRID 256
#endif
    const int schur_grid = lds_path ? std::min(BA_SCHUR0_GRID, std::max(1, (n_ptl + 3) / 4)) : std::min(4096, std::max(1, (n_ptl + 3) / 4));
    const size_t sz_sr = sz_S + n6;
    if (lds_path && (size_t)schur_grid * sz_sr > BS->parts_cap) {
        HIP_TRY(ctx, hipStreamSynchronize(st)); if (BS->d_parts) hipFree(BS->d_parts);
        BS->parts_cap = (size_t)BA_SCHUR0_GRID * sz_sr; HIP_TRY(ctx, hipMalloc((void**)&BS->d_parts, BS->parts_cap * sizeof(double)));
    }
    const int kcap = std::max(maxk, 1);
    const size_t lds_chol = ((size_t)(n6 + 1) * ((n6 + 1) | 1) + n6 + 2) * sizeof(double);
    const size_t lds_chol6 = ((size_t)n6 * (n6 + 1) + n6 + (size_t)n6 * 7 + 8) * sizeof(double);
    const size_t win_sz = (size_t)(BA_WC * (BA_WC + 1) / 2) * SB_PITCH + BA_WC * 6;
    const size_t loc_sz = (size_t)(n_pose * (n_pose + 1) / 2) * SB_PITCH + n6;          // MODE 0: every lower block of S + rhs
    // MODE 2 runs up to 16 waves per workgroup (one landmark per wave at a time, BA_CHUNK landmarks per window flush); as many as the per-wave staging leaves room for
    int schur2_waves = 16; while (schur2_waves > 4 && (win_sz + (size_t)schur2_waves * (2 * kcap * 18 + kcap)) * sizeof(double) > 150 * 1024) schur2_waves >>= 1;
    int schur0_waves = 8; while (schur0_waves > 2 && (loc_sz + (size_t)schur0_waves * (2 * kcap * 18 + kcap)) * sizeof(double) > 150 * 1024) schur0_waves >>= 1;     // MODE 0: 8 waves measured best (2.03 -> 1.91 ms per local solve)
    const size_t lds_schur = ((lds_path ? loc_sz : win_sz) + (size_t)(lds_path ? schur0_waves : schur2_waves) * (2 * kcap * 18 + kcap)) * sizeof(double);
    // MODE 2 walks the landmarks ordered by their first camera, so that a chunk of BA_CHUNK of them touches a short run of cameras
    // (the BA_WC-camera window of S held in LDS); window base of a chunk = lowest first camera in it
    // camera ordinals: static landmarks are seen by cameras only, and with the frame-interleaved pose order a 10-frame track spans ~27 POSE
    // indices but still 10 cameras — the LDS window of k_ba_schur<2> is indexed by camera ordinal so that it keeps catching them
    std::vector<int> pose_ord_h(n_pose, -1), ord_pose_h(p.n_cam);
    { std::vector<int> inv(n_pose, -1); for (int i = 0; i < p.n_cam; i++) inv[perm[i]] = i;
      int o = 0; for (int q = 0; q < n_pose; q++) if (inv[q] >= 0) { pose_ord_h[q] = o; ord_pose_h[o] = q; o++; } }
    D.n_cam_ord = p.n_cam; D.pose_ord = A.put(pose_ord_h.data(), n_pose, st); D.ord_pose = A.put(ord_pose_h.data(), p.n_cam, st);
    if (DI) D.slot_ord = DI->slot_cam;                      // (no object motions in the local window: the camera ordinal IS the pose index)
    else { std::vector<int>& so = BS->hv_i[7]; so.resize(no);
           if (par) HP.chunks((size_t)no, [&](size_t lo, size_t hi, int) { for (size_t t = lo; t < hi; t++) so[t] = pose_ord_h[slotcam[t]]; }); else for (int t = 0; t < no; t++) so[t] = pose_ord_h[slotcam[t]];
           D.slot_ord = A.put(so.data(), no, st); }
    // landmarks per unit of the Schur kernels.  k_ba_schur<2> has BA_CHUNK compiled in; k_ba_schur_mfma (round 6) takes it as an argument and the host sizes it so that the
    // launch is ONE workgroup per CU where the map allows: 100 k landmarks in units of 256 are 391 workgroups on 256 CUs — 135 CUs carry two (which share the CU's matrix
    // pipes and finish together, late), 121 carry one; units of 416 are 241 workgroups of 13 passes each, every one alone on its CU with its own loads in flight beside
    // its matrix phase
    int chunk = BA_CHUNK;
    const bool mfma_schur_wanted = nd == 0 && !getenv("VIDO_BA_SCHUR_OLD");
    if (mfma_schur_wanted && !lds_path && n_ptl) {
        static const int force_chunk = [] { const char* e = getenv("VIDO_BA_SCHUR_CHUNK"); return e ? atoi(e) : 0; }();
        int ncu = 256; (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, ctx->device); if (ncu < 1) ncu = 256;
        const int per_cu = (n_ptl + ncu - 1) / ncu;
        chunk = std::max(64, std::min(512, ((per_cu + SM_L - 1) / SM_L) * SM_L));
        if (force_chunk >= SM_L && force_chunk % SM_L == 0) chunk = force_chunk;
    }
    int* d_chunk_cmin = nullptr; int* d_lorder = nullptr; int2* d_lbc = nullptr; int n_chunks = (n_ptl + chunk - 1) / chunk; bool use_mfma_schur = false;
    if (!lds_path && n_ptl) {
        std::vector<int> lorder(n_ptl);
        auto first_cam = [&](int l) { return pstart[l + 1] > pstart[l] ? slotcam[pstart[l]] : n_pose; };
        if (par) {   // stable counting sort by first camera, on the pool (serial it was 0.2 ms of the 0.7 ms this phase took at 100 k landmarks)
            std::vector<int>& rk = BS->hv_i[2]; rk.resize(n_ptl); std::vector<int> bsu;
            par_counting_rank(HP, (size_t)n_ptl, n_pose + 1, [&](size_t l) { return first_cam((int)l); }, rk.data(), bsu);
            HP.chunks((size_t)n_ptl, [&](size_t lo, size_t hi, int) { for (size_t l = lo; l < hi; l++) lorder[rk[l]] = (int)l; });
        } else {
            std::vector<int> cs(n_pose + 2, 0);
            for (int l = 0; l < n_ptl; l++) cs[first_cam(l) + 1]++;
            for (int c = 0; c <= n_pose; c++) cs[c + 1] += cs[c];
            for (int l = 0; l < n_ptl; l++) lorder[cs[first_cam(l)]++] = l;
        }
        std::vector<int> cmin(n_chunks, 0);
        for (int c = 0; c < n_chunks; c++) { const int m = first_cam(lorder[c * chunk]); cmin[c] = m == n_pose ? 0 : pose_ord_h[m]; }
        d_lorder = A.put(lorder.data(), n_ptl, st);
        use_mfma_schur = mfma_schur_wanted;      // (the dynamic-object graphs keep the wave-per-landmark kernel: their pose order interleaves object motions)
        { std::vector<int>& bc = BS->hv_i[1]; bc.resize(2 * (size_t)n_ptl);      // (hv_i[1]: the camera sort's output list, swapped into `keep` and free since)
          auto fill_bc = [&](size_t qlo, size_t qhi, int) { for (size_t q = qlo; q < qhi; q++) { const int l = lorder[q]; bc[2 * q] = pstart[l]; bc[2 * q + 1] = pstart[l + 1] - pstart[l]; } };
          if (par) HP.chunks((size_t)n_ptl, fill_bc); else fill_bc(0, (size_t)n_ptl, 0);
          if (use_mfma_schur) {      // k_ba_schur_mfma takes the landmarks whose cameras all fall inside their chunk's window; the others (and tracks > 64) go to k_ba_schur_long
              std::vector<char> is_long(n_ptl, 0); for (int l : long_list) is_long[l] = 1;
              std::vector<int> outside;                                  // positions q whose landmark leaves its chunk's window (loop closures, revisits, long gaps)
              { std::vector<std::vector<int> > outs(HP.size());
                auto scan_q = [&](size_t qlo, size_t qhi, int t) { for (size_t q = qlo; q < qhi; q++) {
                  const int l = lorder[q], cnt = pstart[l + 1] - pstart[l]; if (!cnt) continue;
                  const int cb = cmin[q / chunk], lo = pose_ord_h[slotcam[pstart[l]]], hi = pose_ord_h[slotcam[pstart[l + 1] - 1]];
                  bool out = cnt > 64 || lo < cb || hi >= cb + BA_WC || lo < 0 || hi < 0;
                  for (int t2 = pstart[l]; t2 < pstart[l + 1] && !out; t2++) { const int o = pose_ord_h[slotcam[t2]]; out = o < cb || o >= cb + BA_WC; }
                  if (out) outs[t].push_back((int)q); } };
                if (par) HP.chunks((size_t)n_ptl, scan_q); else scan_q(0, (size_t)n_ptl, 0);
                for (auto& v : outs) outside.insert(outside.end(), v.begin(), v.end()); }
              // k_ba_schur_long is one workgroup and O(36 k^2) HBM atomics per landmark: right for the rare track of > 64 keyframes, a cliff when a map with many revisits
              // sends a sizeable share of its landmarks there.  Above 3 % the wave-per-landmark kernel takes the whole graph instead (it spills out-of-window pairs itself).
              const size_t n_out_short = outside.size() - std::min(outside.size(), long_list.size());
              if (getenv("VIDO_BA_VERBOSE")) fprintf(stderr, "[ba] landmarks outside their chunk's camera window: %zu of %d (%.2f %%), tracks > 64: %zu\n", outside.size(), n_ptl, 100.0 * outside.size() / std::max(n_ptl, 1), long_list.size());
              if (n_out_short * 100 > (size_t)n_ptl * 3) use_mfma_schur = false;
              else for (int q : outside) { const int l = lorder[q]; bc[2 * (size_t)q + 1] = -(pstart[l + 1] - pstart[l]); if (!is_long[l]) { long_list.push_back(l); is_long[l] = 1; } }
          }
          if (!use_mfma_schur && chunk != BA_CHUNK) {      // the wave-per-landmark kernel takes over: its unit size is compiled in
              chunk = BA_CHUNK; n_chunks = (n_ptl + chunk - 1) / chunk; cmin.assign(n_chunks, 0);
              for (int c = 0; c < n_chunks; c++) { const int m = first_cam(lorder[c * chunk]); cmin[c] = m == n_pose ? 0 : pose_ord_h[m]; }
          }
          d_chunk_cmin = A.put(cmin.data(), n_chunks, st);
          d_lbc = (int2*)A.put(bc.data(), 2 * (size_t)n_ptl, st); }
        if (A.failed) return vido_set_error(ctx, VIDO_E_NOMEM, "ba: device pool exhausted");
    }
    phase("chunk order, window tables");
    const int n_long = (int)long_list.size();
    int* d_long = nullptr;
    if (n_long) {      // rare: its own small allocation instead of a slice of the arena (whose size estimate does not count it)
        if ((size_t)n_long > BS->long_cap) { HIP_TRY(ctx, hipStreamSynchronize(st)); if (BS->d_long) hipFree(BS->d_long); BS->long_cap = (size_t)n_long * 2; HIP_TRY(ctx, hipMalloc((void**)&BS->d_long, BS->long_cap * sizeof(int))); }
        d_long = BS->d_long;
        HIP_TRY(ctx, hipMemcpyAsync(d_long, long_list.data(), (size_t)n_long * sizeof(int), hipMemcpyHostToDevice, st)); HIP_TRY(ctx, hipStreamSynchronize(st));
    }
    if (lds_schur > 160 * 1024) return vido_set_error(ctx, VIDO_E_CAPACITY, "ba: LDS budget exceeded (n6=%d, max track %d)", n6, maxk);
    if (lds_path) { HIP_TRY(ctx, ba_lds_attr(ctx->device, (const void*)k_ba_schur<0>, (size_t)(lds_schur)));
                    HIP_TRY(ctx, ba_lds_attr(ctx->device, (const void*)k_ba_chol_small, (size_t)(lds_chol)));
                    HIP_TRY(ctx, ba_lds_attr(ctx->device, (const void*)k_ba_chol_small6, (size_t)(lds_chol6)));
                    HIP_TRY(ctx, ba_lds_attr(ctx->device, (const void*)k_bas_schur, (size_t)(lds_schur)));
                    HIP_TRY(ctx, ba_lds_attr(ctx->device, (const void*)k_bas_chol, (size_t)(lds_chol6))); }
    else { HIP_TRY(ctx, ba_lds_attr(ctx->device, (const void*)k_ba_schur<2>, (size_t)(lds_schur)));
           HIP_TRY(ctx, ba_lds_attr(ctx->device, (const void*)k_ba_schur_mfma, (size_t)(SM_LDS_BYTES))); }

    const int lin_E = 1;     // groups of 64 observations per wave in k_ba_linearize: with the LDS camera accumulators one group is fastest at every size measured (35 k .. 1 M edges)
    if (allreduce == vido_rccl_allreduce && user != (void*)ctx)      // the built-in path enqueues on user->stream: another context's stream would lose all ordering with this solve
        return vido_set_error(ctx, VIDO_E_INVALID, "ba: vido_rccl_allreduce must be passed with user = the context the solve runs on");
    auto AR = [&](double* dptr, size_t cnt, int op) -> int {
        if (!allreduce) return VIDO_OK;
        if (allreduce != vido_rccl_allreduce) HIP_TRY(ctx, hipStreamSynchronize(st));      // a host-side hook reads the buffer; the built-in RCCL path is ordered by the stream
        if (allreduce(user, dptr, cnt, op) != 0) return vido_set_error(ctx, VIDO_E_INVALID, "ba: all-reduce hook failed");
        return VIDO_OK;
    };
    auto read_scal = [&]() -> int { HIP_TRY(ctx, hipMemcpyAsync(BS->h_scal, D.scal, 8 * sizeof(double), hipMemcpyDeviceToHost, st)); HIP_TRY(ctx, hipStreamSynchronize(st)); return VIDO_OK; };
    // robust chi2 of the whole graph at (cam, pt): shard part on the device, summed over ranks
    auto chi2_at = [&](const double* cam, const double* pt, int slot, double* out) -> int {
        HIP_TRY(ctx, hipMemsetAsync(D.scal + slot, 0, sizeof(double), st));
        if (no) hipLaunchKernelGGL(k_ba_chi2, dim3(std::min((no + 255) / 256, 512)), dim3(256), 0, st, D, cam, pt, D.scal + slot);
        if (nd) hipLaunchKernelGGL(k_badyn_chi2, dim3((nd + 255) / 256), dim3(256), 0, st, D, cam, (const double*)D.dyn, D.scal + slot);
        const int ncf = D.n_odo + (D.prior_cam >= 0 ? 1 : 0);
        if (ncf) hipLaunchKernelGGL(k_ba_camfactors, dim3(ncf), dim3(64), 0, st, D, 0, cam, D.scal + slot);
        int rc = AR(D.scal + slot, 1, 0); if (rc) return rc;
        rc = read_scal(); if (rc) return rc;
        *out = BS->h_scal[slot];
        return VIDO_OK;
    };
    int rc;
    A.flush(st);                                                // the uploads of this call (one copy for the local window)
    if (A.failed) return vido_set_error(ctx, VIDO_E_NOMEM, "ba: upload staging exhausted");
    HIP_TRY(ctx, hipStreamSynchronize(st));
    phase("attributes, launch set-up");
    res->ms_setup = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    const auto t_loop = std::chrono::steady_clock::now();
    double lambda = -1, ni = 2, lastChi = 0, chi2_check = 0, ms_lin = 0, ms_schur = 0; int nBad = 0, trials = 0, it = 0, n_lin = 0, n_schur_timed = 0, n_schur_read = 0;
    // ---- the local window as one persistent launch (k_ba_local_lm): static graph, LDS-resident reduced system, one GPU
    // (opt-in, VIDO_BA_PERSIST=1: measured SLOWER than the host-driven loop below — 3.0 against 2.1 ms per solve alone on the GPU — and its 32 full-CU workgroups never
    //  all become resident while the networks keep the chip busy; DESIGN.md section 9.  The default local-window path is the enqueued-ahead form of the fused loop below (`spec`); VIDO_BA_NO_SPEC=1 keeps the host-driven trial loop `fl`.)
    static const bool want_persist = getenv("VIDO_BA_PERSIST") != nullptr, no_fused_local = getenv("VIDO_BA_NO_FUSED_LOCAL") != nullptr;
    const bool local_static = lds_path && !allreduce && nd == 0 && n6 % 6 == 0 && n6 >= 12 && n_pose <= 64 && D.n_odo <= 64 && n_long == 0 && p.max_iters >= 1 && n_ptl > 0 && no > 0;
    bool persist = local_static && want_persist;
    const double* cam_final = nullptr;
    if (persist) {
        static const int persist_wgs = [] { const char* e = getenv("VIDO_BA_PERSIST_WGS"); const int v = e ? atoi(e) : 32; return std::max(1, std::min(v, 64)); }();
        const size_t lds_bal = std::max(std::max(lds_chol6, (loc_sz + (size_t)(BAL_NT / 64) * (2 * kcap * 18 + kcap)) * sizeof(double)), (size_t)LIN_SMEM_DOUBLES(BAL_NT) * sizeof(double));
        if (lds_bal > 159 * 1024) persist = false;
        else {
            if (!BS->d_ctl) { HIP_TRY(ctx, hipMalloc((void**)&BS->d_ctl, sizeof(BaLmCtl))); HIP_TRY(ctx, hipMemset(BS->d_ctl, 0, sizeof(BaLmCtl))); HIP_TRY(ctx, hipHostMalloc((void**)&BS->h_ctl, sizeof(BaLmCtl))); BS->bar_base = 0; }
            if ((size_t)persist_wgs * loc_sz > BS->parts_cap) {
                HIP_TRY(ctx, hipStreamSynchronize(st)); if (BS->d_parts) hipFree(BS->d_parts);
                BS->parts_cap = (size_t)BA_SCHUR0_GRID * sz_sr; HIP_TRY(ctx, hipMalloc((void**)&BS->d_parts, BS->parts_cap * sizeof(double)));
            }
            HIP_TRY(ctx, ba_lds_attr(ctx->device, (const void*)k_ba_local_lm, (size_t)(lds_bal)));
            BaDev4 V;
            for (int par = 0; par < 2; par++) for (int flip = 0; flip < 2; flip++) {
                BaDev& Q = V.v[par * 2 + flip]; Q = D;
                double* rp = par ? red1 : red; Q.Hcd = rp; Q.bc = rp + (size_t)n_pose * 36; Q.scal = Q.bc + n6;
                if (flip) { std::swap(Q.cam, Q.cam_new); std::swap(Q.pt, Q.pt_new); }
            }
            BaLocalArgs LA{BS->d_ctl, red, red1, BS->d_parts, cam_out, D.pt, n_ptl, kcap, p.max_iters, BS->bar_base, p.gain_threshold};
            hipLaunchKernelGGL(k_ba_local_lm, dim3(persist_wgs), dim3(BAL_NT), lds_bal, st, V, LA);
            HIP_TRY(ctx, hipGetLastError());
            HIP_TRY(ctx, hipMemcpyAsync(BS->h_ctl, BS->d_ctl, sizeof(BaLmCtl), hipMemcpyDeviceToHost, st));
            HIP_TRY(ctx, hipMemcpyAsync(poses.data(), cam_out, (size_t)n_pose * 12 * sizeof(double), hipMemcpyDeviceToHost, st));      // (pageable destination: the copy is complete on return)
            HIP_TRY(ctx, hipStreamSynchronize(st));
            const BaLmCtl& C = *BS->h_ctl;
            if (C.status != 0) {      // a workgroup never arrived at a barrier (never seen; the spins are bounded so that this is an error, not a hang)
                HIP_TRY(ctx, hipMemset(BS->d_ctl, 0, sizeof(BaLmCtl))); BS->bar_base = 0;
                return vido_set_error(ctx, VIDO_E_HIP, "ba: the persistent local-window solver lost a workgroup at a grid barrier (%d barriers in)", C.barriers);
            }
            BS->bar_base += (unsigned)C.barriers * (unsigned)persist_wgs;
            it = C.iterations; trials = C.trials; lambda = C.lambda_final; n_lin = C.n_lin; ms_lin = C.lin_ticks * 1e-5;      // wall_clock64: 100 MHz
            res->chi2_initial = C.chi2_initial; res->chi2_final = C.chi2_final;
            cam_final = cam_out;
        }
    }
    // the fused host-driven loop of the local window (k_ba_lin_local / k_ba_fold_local / k_ba_backsub_chi2_local): 6 stream operations per LM iteration instead of 11
    const bool fl = local_static && !persist && !no_fused_local;
    if (fl && !BS->d_ticket) { HIP_TRY(ctx, hipMalloc((void**)&BS->d_ticket, sizeof(int))); HIP_TRY(ctx, hipMemset(BS->d_ticket, 0, sizeof(int))); }
    static const int fl_schur_grid_env = [] { const char* e = getenv("VIDO_BA_SCHUR0_GRID"); return e ? std::max(1, std::min(atoi(e), BA_SCHUR0_GRID)) : 128; }();
    const int fl_grid = std::min(fl_schur_grid_env, std::max(1, (n_ptl + 3) / 4));       // partial reduced systems: every one is summed by the fold; tracker alone: 32 / 64 / 128 / 256 partials -> 2.49 / 2.13 / 1.95 / 1.95 ms per solve
    const size_t red_len = (size_t)n_pose * 36 + n6 + 8;
    // ---- the local window enqueued ahead (default; VIDO_BA_NO_SPEC=1 keeps the host-driven trial loop below): the LM state lives on the device, the host puts
    // (previous solve's trials + 2) trial slots on the stream in one go and waits ONCE; a solve that needs more gets four more slots at a time.  No host round trip per trial:
    // inside the pipeline each one cost a stream drain behind the networks' workgroups.
    static const bool no_spec = getenv("VIDO_BA_NO_SPEC") != nullptr;
    const bool spec = fl && !no_spec;
    if (spec) {
        if (!BS->d_spec) { HIP_TRY(ctx, hipMalloc((void**)&BS->d_spec, sizeof(BaSpecCtl))); HIP_TRY(ctx, hipHostMalloc((void**)&BS->h_spec, sizeof(BaSpecCtl))); }
        BaDev4 V;
        for (int par = 0; par < 2; par++) for (int flip = 0; flip < 2; flip++) {
            BaDev& Q = V.v[par * 2 + flip]; Q = D;
            double* rp = par ? red1 : red; Q.Hcd = rp; Q.bc = rp + (size_t)n_pose * 36; Q.scal = Q.bc + n6;
            if (flip) { std::swap(Q.cam, Q.cam_new); std::swap(Q.pt, Q.pt_new); }
        }
        BaSpecCtl init; memset(&init, 0, sizeof init); init.max_iters = p.max_iters; init.gain_threshold = p.gain_threshold; init.need_lin = 1;
        *BS->h_spec = init;
        HIP_TRY(ctx, hipMemcpyAsync(BS->d_spec, BS->h_spec, sizeof(BaSpecCtl), hipMemcpyHostToDevice, st));
        HIP_TRY(ctx, hipMemsetAsync(red, 0, (size_t)((char*)(red1 + red_len) - (char*)red), st));
        const int ncf = D.n_odo + (D.prior_cam >= 0 ? 1 : 0), nvb = (no + LIN_THREADS - 1) / LIN_THREADS, lin_grid = nvb + (ncf + LIN_THREADS / 64 - 1) / (LIN_THREADS / 64);
        const int n_bs = std::min((n_ptl + 31) / 32, 1024), fold_grid = std::min(64, (int)((loc_sz + 255) / 256));
        HIP_TRY(ctx, hipEventRecord(BS->ev0, st));
        hipLaunchKernelGGL(k_bas_lin, dim3(lin_grid), dim3(LIN_THREADS), 0, st, V, (const BaSpecCtl*)BS->d_spec, nvb);
        HIP_TRY(ctx, hipEventRecord(BS->ev1, st)); n_lin = 1;
        hipLaunchKernelGGL(k_ba_maxdiag, dim3(64), dim3(256), 0, st, V.v[0], n_ptl);
        hipLaunchKernelGGL(k_bas_init, dim3(1), dim3(64), 0, st, V, BS->d_spec);
        int budget = std::max(4, std::min(BS->spec_last_trials + 2, 48)), enq = 0;
        for (;;) {
            for (int sl = 0; sl < budget; sl++) {
                hipLaunchKernelGGL(k_bas_schur, dim3(fl_grid), dim3(64 * schur0_waves), lds_schur, st, V, (const BaSpecCtl*)BS->d_spec, n_ptl, kcap, BS->d_parts);
                hipLaunchKernelGGL(k_bas_fold, dim3(fold_grid), dim3(256), 0, st, V, (const BaSpecCtl*)BS->d_spec, (const double*)BS->d_parts, fl_grid);
                hipLaunchKernelGGL(k_bas_chol, dim3(1), dim3(CH_NT), lds_chol6, st, V, (const BaSpecCtl*)BS->d_spec);
                hipLaunchKernelGGL(k_bas_backsub, dim3(n_bs + (ncf + 3) / 4), dim3(256), 0, st, V, BS->d_spec, n_ptl, n_bs, red, red1, (int)red_len, BS->d_ticket, BS->h_spec);
                hipLaunchKernelGGL(k_bas_lin, dim3(lin_grid), dim3(LIN_THREADS), 0, st, V, (const BaSpecCtl*)BS->d_spec, nvb);
            }
            enq += budget;
            HIP_TRY(ctx, hipGetLastError());
            HIP_TRY(ctx, hipStreamSynchronize(st));
            if (BS->h_spec->stop || enq > 10 * p.max_iters + 16) break;
            budget = 4;
        }
        const BaSpecCtl hs = *BS->h_spec;
        if (!hs.stop) return vido_set_error(ctx, VIDO_E_HIP, "ba: the enqueued-ahead local solve did not stop after %d trial slots", enq);
        BS->spec_last_trials = hs.trials;
        it = hs.it; trials = hs.trials; lambda = hs.lambda; res->chi2_initial = hs.chi_init; res->chi2_final = hs.chi_final;
        n_lin = hs.it;
        { float ms = 0; if (hipEventElapsedTime(&ms, BS->ev0, BS->ev1) == hipSuccess) ms_lin = ms * n_lin; }
        if (hs.flip) { std::swap(D.cam, D.cam_new); std::swap(D.pt, D.pt_new); }
    }
    if (!persist && !spec) {
    if (!fl) { if ((rc = chi2_at(D.cam, D.pt, 2, &res->chi2_initial))) return rc; res->chi2_final = res->chi2_initial; }
    for (it = 0; it < p.max_iters; it++) {
        // ---- linearise
        const int ncf = D.n_odo + (D.prior_cam >= 0 ? 1 : 0);
        if (fl) {
            // accumulators double-buffered by iteration parity: this iteration's copy was cleared by the previous iteration's back-substitution launch (both copies — they are
            // neighbours in the arena — by one memset before the first)
            if (it == 0) HIP_TRY(ctx, hipMemsetAsync(red, 0, (size_t)((char*)(red1 + red_len) - (char*)red), st));
            double* rp = (it & 1) ? red1 : red; D.Hcd = rp; D.bc = rp + (size_t)n_pose * 36; D.scal = D.bc + n6;
            const int nvb = (no + LIN_THREADS - 1) / LIN_THREADS;
            if (it == 0) HIP_TRY(ctx, hipEventRecord(BS->ev0, st));
            hipLaunchKernelGGL(k_ba_lin_local, dim3(nvb + (ncf + LIN_THREADS / 64 - 1) / (LIN_THREADS / 64)), dim3(LIN_THREADS), 0, st, D, nvb);
            if (it == 0) { HIP_TRY(ctx, hipEventRecord(BS->ev1, st)); n_lin++; }
        } else {
        HIP_TRY(ctx, hipMemsetAsync(red, 0, ((size_t)n_pose * 36 + n6 + 8) * sizeof(double), st));
        HIP_TRY(ctx, hipEventRecord(BS->ev0, st));
        if (no) hipLaunchKernelGGL(k_ba_linearize, dim3((no + LIN_THREADS * lin_E - 1) / (LIN_THREADS * lin_E)), dim3(LIN_THREADS), 0, st, D, lin_E);
        HIP_TRY(ctx, hipEventRecord(BS->ev1, st));
        if (nd) hipLaunchKernelGGL(k_badyn_linearize, dim3((nd + 255) / 256), dim3(256), 0, st, D);
        n_lin++;
        if (ncf) hipLaunchKernelGGL(k_ba_camfactors, dim3(ncf), dim3(64), 0, st, D, 1, D.cam, D.scal + 0);
        }
        if (it == 0) hipLaunchKernelGGL(k_ba_maxdiag, dim3(64), dim3(256), 0, st, D, n_ptl);
        HIP_TRY(ctx, hipGetLastError());
        if (allreduce) {     // camera diagonal blocks, bc, chi2 (sum) — then the max-diagonal (max) on its own
            if (it == 0) { HIP_TRY(ctx, hipStreamSynchronize(st)); double md; HIP_TRY(ctx, hipMemcpy(&md, D.scal + 1, 8, hipMemcpyDeviceToHost)); HIP_TRY(ctx, hipMemsetAsync(D.scal + 1, 0, 8, st));
                           if ((rc = AR(red, (size_t)n_pose * 36 + n6 + 1, 0))) return rc;
                           // the stream-ordered RCCL path does not synchronise inside AR(): order the blocking copy below behind the memset and the sum all-reduce on `st`
                           // (it == 0 only; ADVICE r2: without this the landmark part of the max-diagonal could be zeroed after it had been written)
                           HIP_TRY(ctx, hipStreamSynchronize(st));
                           HIP_TRY(ctx, hipMemcpy(D.scal + 1, &md, 8, hipMemcpyHostToDevice)); if ((rc = AR(D.scal + 1, 1, 1))) return rc;
                           // the camera part of the max must see the SUMMED camera diagonals
                           hipLaunchKernelGGL(k_ba_maxdiag, dim3(64), dim3(256), 0, st, D, 0); if ((rc = AR(D.scal + 1, 1, 1))) return rc; }
            else if ((rc = AR(red, (size_t)n_pose * 36 + n6 + 1, 0))) return rc;
        }
        // chi2 at the linearisation point (scal[0]) is only needed once the first trial of this iteration has been evaluated, so after the
        // first iteration (whose lambda needs the max diagonal) it is read together with the trial's scalars: one host sync per trial
        bool have_chi = false; double currentChi = 0, iniChi = 0;
        if (it == 0 || allreduce) {
            if ((rc = read_scal())) return rc;
            currentChi = iniChi = BS->h_scal[0]; have_chi = true;
            if (it == 0) { lambda = 1e-5 * BS->h_scal[1]; ni = 2; nBad = 0; }
            if (it == 0 && fl) { res->chi2_initial = res->chi2_final = currentChi; float ms = 0; if (hipEventElapsedTime(&ms, BS->ev0, BS->ev1) == hipSuccess) ms_lin += ms; }      // (the linearisation point of the first iteration IS the initial state; the timed launch carries the camera factors too)
        }
        double rho = 0; int qmax = 0;
        do {
            if (fl) {
                // ---- S, F, C, B as four launches (bodies shared with k_ba_local_lm)
                hipLaunchKernelGGL(k_ba_schur<0>, dim3(fl_grid), dim3(64 * schur0_waves), lds_schur, st, D, n_ptl, lambda, kcap, BS->d_parts, (const int*)nullptr, (const int*)nullptr, (const int2*)nullptr);
                hipLaunchKernelGGL(k_ba_fold_local, dim3(std::min(64, (int)((loc_sz + 255) / 256))), dim3(256), 0, st, D, (const double*)BS->d_parts, fl_grid, lambda);
                hipLaunchKernelGGL(k_ba_chol_small6, dim3(1), dim3(CH_NT), lds_chol6, st, D, 1, lambda);
                const int n_bs = std::min((n_ptl + 31) / 32, 1024);
                hipLaunchKernelGGL(k_ba_backsub_chi2_local, dim3(n_bs + (ncf + 3) / 4), dim3(256), 0, st, D, n_ptl, lambda, n_bs, qmax == 0 ? ((it & 1) ? red : red1) : (double*)nullptr, (int)red_len,
                                   BS->d_ticket, BS->h_scal);
                HIP_TRY(ctx, hipGetLastError());
                HIP_TRY(ctx, hipStreamSynchronize(st));      // the launch's last workgroup has written the trial's scalars into h_scal
            } else {
            // ---- reduced system of this shard.  With an all-reduce every rank contributes Hcd/bc ALREADY summed,
            // so only rank 0 adds the camera-camera part (+lambda) to S; the others start from zero.
            const int add_cam = (!allreduce || p.rank == 0) ? 1 : 0;
            const int inline_odo = (lds_path && !allreduce && D.bw < 0 && D.n_odo <= 64) ? 1 : 0;
            hipLaunchKernelGGL(k_ba_init_S, dim3(std::min(2048, (int)((sz_sr + 255) / 256))), dim3(256), 0, st, D, lambda, add_cam, inline_odo);
            if (!add_cam) HIP_TRY(ctx, hipMemsetAsync(D.r, 0, n6 * sizeof(double), st));
            if (add_cam && D.n_odo && !inline_odo) hipLaunchKernelGGL(k_ba_add_odo, dim3((D.n_odo * 36 + 255) / 256), dim3(256), 0, st, D);
            if (n_ptl) {
                if (lds_path) {
                    hipLaunchKernelGGL(k_ba_schur<0>, dim3(schur_grid), dim3(64 * schur0_waves), lds_schur, st, D, n_ptl, lambda, kcap, BS->d_parts, (const int*)nullptr, (const int*)nullptr, (const int2*)nullptr);
                    hipLaunchKernelGGL(k_ba_fold_parts, dim3(std::min(256, (int)((loc_sz + 255) / 256)), std::min(8, schur_grid)), dim3(256), 0, st, D, BS->d_parts, schur_grid);
                } else if (use_mfma_schur) {
                    HIP_TRY(ctx, hipEventRecord(BS->ev2, st));
                    hipLaunchKernelGGL(k_ba_schur_mfma, dim3(std::min(n_chunks, 1024)), dim3(SM_NT), SM_LDS_BYTES, st, D, n_ptl, lambda, (const int*)d_chunk_cmin, (const int*)d_lorder, (const int2*)d_lbc, chunk);
                    HIP_TRY(ctx, hipEventRecord(BS->ev3, st)); n_schur_timed++;
                }
                else hipLaunchKernelGGL(k_ba_schur<2>, dim3(std::min(n_chunks, 1024)), dim3(64 * schur2_waves), lds_schur, st, D, n_ptl, lambda, kcap, (double*)nullptr, (const int*)d_chunk_cmin, (const int*)d_lorder, (const int2*)d_lbc);
            }
            if (n_long) hipLaunchKernelGGL(k_ba_schur_long, dim3(n_long), dim3(256), 0, st, D, lambda, (const int*)d_long);
            if (nd) {
                hipLaunchKernelGGL(k_badyn_factor, dim3((n_chain + 63) / 64), dim3(64), 0, st, D, lambda);
                hipLaunchKernelGGL(k_badyn_schur, dim3(dyn_grid), dim3(64), dyn_lds, st, D, BS->d_scratch, lmax, dyn_lds ? 1 : 0);
            }
            HIP_TRY(ctx, hipGetLastError());
            if ((rc = AR(Sr, sz_sr, 0))) return rc;
            // ---- replicated reduced solve
            if (!inline_odo) HIP_TRY(ctx, hipMemsetAsync(D.scal + 2, 0, 2 * sizeof(double), st));
            const int fuse_update = (lds_path && n6 % 6 == 0 && n6 >= 12 && !allreduce && n_pose <= 64) ? 1 : 0;
            if (lds_path && n6 % 6 == 0 && n6 >= 12) hipLaunchKernelGGL(k_ba_chol_small6, dim3(1), dim3(CH_NT), lds_chol6, st, D, fuse_update, lambda);
            else if (lds_path) hipLaunchKernelGGL(k_ba_chol_small, dim3(1), dim3(1024), lds_chol, st, D);
            else if (Bc.m) {                                   // block cyclic reduction: 2 launches per level, log2(nb) levels, then the levels back
                const int m = Bc.m, nb = Bc.nb;
                // scalar_piv: round 5's reduced solve around the same elimination kernel (k_bcr_update, one back-substitution launch per level), for comparison
                static const bool scalar_piv = getenv("VIDO_BCR_SCALAR") != nullptr;
                auto elim = [&](int n_blocks, int sft, const double* Lc, int mode) {
                    if (m == 66) hipLaunchKernelGGL(k_bcr_elim<66>, dim3(n_blocks, mode ? 1 : 2), dim3(BcrGeom<66>::NT), BcrGeom<66>::LDS, st, Bc, sft, Lc, mode);
                    else         hipLaunchKernelGGL(k_bcr_elim<96>, dim3(n_blocks, mode ? 1 : 2), dim3(BcrGeom<96>::NT), BcrGeom<96>::LDS, st, Bc, sft, Lc, mode);
                };
                hipLaunchKernelGGL(k_bcr_pack, dim3(nb), dim3(256), 0, st, Bc);
                double *Lc = Bc.L0, *Ln = Bc.L1; int smax = 0;
                for (int sft = 1; sft < nb; sft *= 2) {
                    const int n_el = (nb - sft + 2 * sft - 1) / (2 * sft), n_sv = (nb + 2 * sft - 1) / (2 * sft);
                    elim(n_el, sft, Lc, 0);
                    if (scalar_piv) hipLaunchKernelGGL(k_bcr_update, dim3(n_sv, m / BCR_RB), dim3(128), 0, st, Bc, sft, (const double*)Lc, Ln);
                    else            hipLaunchKernelGGL(k_bcr_update4, dim3(n_sv, m / BCR_RB), dim3(320), 0, st, Bc, sft, (const double*)Lc, Ln);
                    std::swap(Lc, Ln); smax = sft;
                }
                elim(1, 0, Lc, 1);
                static const bool back_levels = getenv("VIDO_BCR_BACK_LEVELS") != nullptr;      // (round 5's launch per level, for comparison)
                if (scalar_piv || back_levels) { for (int sft = smax; sft >= 1; sft /= 2) hipLaunchKernelGGL(k_bcr_back, dim3((nb - sft + 2 * sft - 1) / (2 * sft)), dim3(384), 0, st, Bc, sft); }
                else {
                    if (++BS->bcr_epoch == 0) BS->bcr_epoch = 1;                  // (a tag of 0 is what the zeroed buffer holds)
                    hipLaunchKernelGGL(k_bcr_back_chain, dim3(BS->bcr_blocks_n), dim3(384), 0, st, Bc, (const int2*)BS->d_bcr_blocks, BS->d_bcr_xt, BS->bcr_epoch, BS->d_bcr_abort);
                }
            }
            else if (D.bw >= 0 && band6_lds) hipLaunchKernelGGL(k_chol_band6, dim3(1), dim3(CH_NT), band6_lds, st, D, (D.bw - 5) / 6);
            else if (D.bw >= 0 && band6s_lds && band6s_nb == 8) hipLaunchKernelGGL(k_chol_band6s<8>, dim3(1), dim3(CG_NT), band6s_lds, st, D, (D.bw - 5) / 6);
            else if (D.bw >= 0 && band6s_lds) hipLaunchKernelGGL(k_chol_band6s<4>, dim3(1), dim3(CG_NT), band6s_lds, st, D, (D.bw - 5) / 6);
            else if (D.bw >= 0) hipLaunchKernelGGL(k_chol_band, dim3(1), dim3(1024), (size_t)D.bw * (CB_NB + 1) * sizeof(double), st, D);
            else { const double one = 1.0; HIP_TRY(ctx, hipMemcpyAsync(D.scal + 4, &one, 8, hipMemcpyHostToDevice, st)); if ((rc = chol_large(ctx, D.S, n6, D.x, D.scal + 4, chol_tmp, st))) return rc; }
            // ---- trial state + its chi2
            if (!fuse_update) hipLaunchKernelGGL(k_ba_update_cams, dim3((n_pose + 63) / 64), dim3(64), 0, st, D, (allreduce && p.rank != 0) ? 0.0 : lambda);
            if (allreduce && p.rank != 0) HIP_TRY(ctx, hipMemsetAsync(D.scal + 3, 0, sizeof(double), st));      // camera part of computeScale counted once (rank 0)
            if (n_ptl) hipLaunchKernelGGL(k_ba_backsub, dim3(std::min((n_ptl + 31) / 32, 1024)), dim3(256), 0, st, D, n_ptl, lambda);
            if (no) hipLaunchKernelGGL(k_ba_chi2, dim3((no + 255) / 256), dim3(256), 0, st, D, D.cam_new, D.pt_new, D.scal + 2);
            if (nd) { hipLaunchKernelGGL(k_badyn_backsub, dim3((n_chain + 63) / 64), dim3(64), 0, st, D, lambda);
                      hipLaunchKernelGGL(k_badyn_chi2, dim3((nd + 255) / 256), dim3(256), 0, st, D, (const double*)D.cam_new, (const double*)D.dyn_new, D.scal + 2); }
            if (ncf) hipLaunchKernelGGL(k_ba_camfactors, dim3(ncf), dim3(64), 0, st, D, 0, D.cam_new, D.scal + 2);
            HIP_TRY(ctx, hipGetLastError());
            if ((rc = AR(D.scal + 2, 2, 0))) return rc;
            if ((rc = read_scal())) return rc;
            }
            if (!have_chi) { currentChi = iniChi = BS->h_scal[0]; have_chi = true; }
            if (qmax == 0 && !fl) { float ms = 0; if (hipEventElapsedTime(&ms, BS->ev0, BS->ev1) == hipSuccess) ms_lin += ms; }
            if (!fl && n_schur_timed > n_schur_read) { float ms = 0; if (hipEventElapsedTime(&ms, BS->ev2, BS->ev3) == hipSuccess) ms_schur += ms; n_schur_read = n_schur_timed; }
            const bool ok2 = BS->h_scal[4] > 0.5;
            const double tempChi = ok2 ? BS->h_scal[2] : DBL_MAX, scale = ok2 ? BS->h_scal[3] : 0.0;
            rho = (currentChi - tempChi) / (scale + 1e-3);
            if (rho > 0 && std::isfinite(tempChi)) {
                double alpha = 1. - pow((2 * rho - 1), 3); alpha = std::min(alpha, 2. / 3.);
                lambda *= std::max(1. / 3., alpha); ni = 2; currentChi = tempChi;
                std::swap(D.cam, D.cam_new); std::swap(D.pt, D.pt_new); std::swap(D.dyn, D.dyn_new);
            } else { lambda *= ni; ni *= 2; }
            qmax++; trials++;
        } while (rho < 0 && qmax < 10);
        bool terminate = (qmax == 10 || rho == 0);
        if (!terminate) { if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0; if (nBad >= 3) terminate = true; }
        const double chiNow = currentChi;     // chi2 at the accepted state (the trial that produced it evaluated it)
        if (chi2_check < chiNow && it > 0) terminate = true;
        chi2_check = chiNow;
        if (it == 0) lastChi = chiNow;
        else { const double gain = (lastChi - chiNow) / chiNow; lastChi = chiNow; if (gain >= 0 && gain < p.gain_threshold) terminate = true; }
        res->chi2_final = chiNow;
        if (terminate) { it++; break; }
    }
    }
    res->iterations = it; res->lm_trials = trials; res->lambda_final = lambda;
    res->ms_schur_kernel = n_schur_read ? ms_schur / n_schur_read : 0.0;
    res->ms_linearize_kernel = n_lin ? ms_lin / n_lin : 0.0;      // (persistent solver: the linearisation phase up to its grid barrier, device clock)
    res->ms_solve_loop = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_loop).count();
    if (!cam_final) HIP_TRY(ctx, hipMemcpyAsync(poses.data(), D.cam, (size_t)n_pose * 12 * sizeof(double), hipMemcpyDeviceToHost, st));
    if (n_ptl && !DI) HIP_TRY(ctx, hipMemcpyAsync(prob->pt_xyz + 3 * (size_t)pt_lo, D.pt, (size_t)n_ptl * 3 * sizeof(double), hipMemcpyDeviceToHost, st));
    if (n_ptl && DI && D.pt != DI->pt) HIP_TRY(ctx, hipMemcpyAsync(DI->pt, D.pt, (size_t)n_ptl * 3 * sizeof(double), hipMemcpyDeviceToDevice, st));      // the accepted state may sit in the other buffer
    if (nd) HIP_TRY(ctx, hipMemcpyAsync(d_xyz.data(), D.dyn, (size_t)nd * 3 * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    for (int i = 0; i < p.n_cam; i++) memcpy(prob->cam_T + (size_t)i * 12, poses.data() + (size_t)perm[i] * 12, 12 * sizeof(double));
    for (int h = 0; h < n_H; h++) memcpy(dynp->H_T + (size_t)h * 12, poses.data() + (size_t)perm[p.n_cam + h] * 12, 12 * sizeof(double));
    for (int t = 0; t < nd; t++) for (int a = 0; a < 3; a++) dynp->dyn_xyz[3 * (size_t)d_order[t] + a] = d_xyz[3 * (size_t)t + a];
    return VIDO_OK;
}
