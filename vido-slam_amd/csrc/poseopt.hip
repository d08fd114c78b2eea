// poseopt.hip — the four per-frame pose / object-motion optimisers as ONE persistent gfx950 kernel.
// Replaces Optimizer::PoseOptimizationNew / PoseOptimizationFlow2Cam / PoseOptimizationObjMot /
// PoseOptimizationFlow2 (reference vido_slam/src/Optimizer.cc:2180-2334, 2622-2824, 2826-3035,
// 3037-3253) and the g2o pieces they drive (LM policy core/optimization_algorithm_levenberg.cpp:61-189,
// outer loop core/sparse_optimizer.cpp:354-427, Schur core/block_solver.hpp:354-486, Huber
// core/robust_kernel_impl.cpp:65-91, residuals/Jacobians types/types_six_dof_expmap.{h,cpp}).
//
// Mapping: a cluster of 1..8 workgroups (512 threads: one residual per thread from n = 512 on) per problem, the WHOLE
// Levenberg-Marquardt run stays on the device (no host round trip per iteration); the workgroups of a cluster
// exchange their partial sums through tagged words in HBM (xwg.hpp) and all run the same control flow; per
// linearisation the 21+6 entries of J^T W J / J^T W e (+ chi2, + max diagonal) are summed with a
// wavefront DPP/shuffle tree and one LDS pass across the waves; for the flow-coupled edges each
// thread keeps its points' 2x2 (scalar*I) landmark block, 6x2 coupling block and rhs in HBM and forms
// its Schur contribution; the reduced 6x6 system is solved redundantly by every thread in registers
// (LDL^T), so the LM accept/reject logic is wave-uniform without broadcasts.  All FP64.
// Several problems (e.g. the frame's dynamic objects, or a batch of frames) run as a grid.
#include "common.hpp"
#include "xwg.hpp"
#include <cfloat>

struct PoseProbDev {
    int mode, n;
    const double *Xw, *obs, *flow0, *depth;
    double Twl[16], P[12], fx, fy, cx, cy, T_init[16], info_edge, info_prior, huber_delta;
    int use_huber, rounds, drop_kernel_after_round, iters[4];
    float chi2_th[4];
    double *f, *err, *fsave, *Hpl, *Hll, *bl, *xl;
    unsigned char *outlier, *has_kernel;
    vido_pose_result* res;
    unsigned long long* xch; int G;      // exchange area of the problem's workgroup cluster (PO_XCH_WORDS), cluster size
};

struct Se3 { double R[9], t[3]; };

__device__ __forceinline__ void mat3_mul(const double* A, const double* B, double* C)
{
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) C[r * 3 + c] = A[r * 3] * B[c] + A[r * 3 + 1] * B[3 + c] + A[r * 3 + 2] * B[6 + c];
}

// SE3Quat::exp (se3quat.h:221-262) followed by VertexSE3Expmap::oplusImpl: T <- exp(u) * T
__device__ void se3_oplus_left(Se3& T, const double* u)
{
    const double w0 = u[0], w1 = u[1], w2 = u[2];
    const double theta = sqrt(w0 * w0 + w1 * w1 + w2 * w2);
    const double O[9] = {0, -w2, w1, w2, 0, -w0, -w1, w0, 0};
    double O2[9]; mat3_mul(O, O, O2);
    double R[9], V[9];
    if (theta < 0.00001) {
#pragma unroll
        for (int i = 0; i < 9; i++) { R[i] = (i % 4 == 0 ? 1.0 : 0.0) + O[i] + O2[i]; V[i] = R[i]; }
    } else {
        const double a = sin(theta) / theta, b = (1 - cos(theta)) / (theta * theta), c = (theta - sin(theta)) / pow(theta, 3);
#pragma unroll
        for (int i = 0; i < 9; i++) { R[i] = (i % 4 == 0 ? 1.0 : 0.0) + a * O[i] + b * O2[i]; V[i] = (i % 4 == 0 ? 1.0 : 0.0) + b * O[i] + c * O2[i]; }
    }
    double et[3];
#pragma unroll
    for (int r = 0; r < 3; r++) et[r] = V[r * 3] * u[3] + V[r * 3 + 1] * u[4] + V[r * 3 + 2] * u[5];
    Se3 N; mat3_mul(R, T.R, N.R);
#pragma unroll
    for (int r = 0; r < 3; r++) N.t[r] = R[r * 3] * T.t[0] + R[r * 3 + 1] * T.t[1] + R[r * 3 + 2] * T.t[2] + et[r];
    T = N;
}

// residual (and 2x6 Jacobian when J != nullptr) of edge i
// the inputs of edge i, fetched together (and before any branch on the outlier flag: one memory round trip per edge instead of two)
struct EdgeIn { double o0, o1, a0, a1, a2; };          // observation; depth (mode 1: a0) or the 3-D point (a0, a1, a2)
__device__ __forceinline__ EdgeIn edge_load(const PoseProbDev& p, int i)
{
    EdgeIn in; in.o0 = p.obs[2 * i]; in.o1 = p.obs[2 * i + 1]; in.a1 = in.a2 = 0;
    if (p.mode == 1) in.a0 = p.depth[i]; else { in.a0 = p.Xw[3 * i]; in.a1 = p.Xw[3 * i + 1]; in.a2 = p.Xw[3 * i + 2]; }
    return in;
}
template <bool WITH_J>
__device__ __forceinline__ void edge_eval(const PoseProbDev& p, const Se3& T, const EdgeIn& in, double f0, double f1, double* e, double* J)
{
    double X0, X1, X2;
    const double o0 = in.o0, o1 = in.o1;
    if (p.mode == 1) {
        const double d = in.a0;
        const double c0 = (o0 - p.cx) * d / p.fx, c1 = (o1 - p.cy) * d / p.fy;
        X0 = p.Twl[0] * c0 + p.Twl[1] * c1 + p.Twl[2] * d + p.Twl[3];
        X1 = p.Twl[4] * c0 + p.Twl[5] * c1 + p.Twl[6] * d + p.Twl[7];
        X2 = p.Twl[8] * c0 + p.Twl[9] * c1 + p.Twl[10] * d + p.Twl[11];
    } else { X0 = in.a0; X1 = in.a1; X2 = in.a2; }
    const double x = T.R[0] * X0 + T.R[1] * X1 + T.R[2] * X2 + T.t[0];
    const double y = T.R[3] * X0 + T.R[4] * X1 + T.R[5] * X2 + T.t[1];
    const double z = T.R[6] * X0 + T.R[7] * X1 + T.R[8] * X2 + T.t[2];
    if (p.mode == 2) {
        const double* P = p.P;
        const double m1 = P[0] * x + P[1] * y + P[2] * z + P[3], m2 = P[4] * x + P[5] * y + P[6] * z + P[7], m3 = P[8] * x + P[9] * y + P[10] * z + P[11];
        const double invm3 = 1.0 / m3;
        e[0] = o0 - m1 * invm3; e[1] = o1 - m2 * invm3;
        if (WITH_J) {
            const double i2 = invm3 * invm3;
            const double t00 = i2 * (P[0] * m3 - P[8] * m1), t01 = i2 * (P[1] * m3 - P[9] * m1), t02 = i2 * (P[2] * m3 - P[10] * m1);
            const double t10 = i2 * (P[4] * m3 - P[8] * m2), t11 = i2 * (P[5] * m3 - P[9] * m2), t12 = i2 * (P[6] * m3 - P[10] * m2);
            J[0] = -1.0 * (y * t02 - z * t01); J[1] = -1.0 * (z * t00 - x * t02); J[2] = -1.0 * (x * t01 - y * t00); J[3] = -t00; J[4] = -t01; J[5] = -t02;
            J[6] = -1.0 * (y * t12 - z * t11); J[7] = -1.0 * (z * t10 - x * t12); J[8] = -1.0 * (x * t11 - y * t10); J[9] = -t10; J[10] = -t11; J[11] = -t12;
        }
        return;
    }
    const double u = x / z * p.fx + p.cx, v = y / z * p.fy + p.cy;
    if (p.mode == 1) { e[0] = (o0 + f0) - u; e[1] = (o1 + f1) - v; } else { e[0] = o0 - u; e[1] = o1 - v; }
    if (WITH_J) {
        if (p.mode == 0) {
            const double invz = 1.0 / z, invz_2 = invz * invz;
            J[0] = x * y * invz_2 * p.fx; J[1] = -(1 + (x * x * invz_2)) * p.fx; J[2] = y * invz * p.fx; J[3] = -invz * p.fx; J[4] = 0; J[5] = x * invz_2 * p.fx;
            J[6] = (1 + y * y * invz_2) * p.fy; J[7] = -x * y * invz_2 * p.fy; J[8] = -x * invz * p.fy; J[9] = 0; J[10] = -invz * p.fy; J[11] = y * invz_2 * p.fy;
        } else {
            const double z_2 = z * z;
            J[0] = x * y / z_2 * p.fx; J[1] = -(1 + (x * x / z_2)) * p.fx; J[2] = y / z * p.fx; J[3] = -1. / z * p.fx; J[4] = 0; J[5] = x / z_2 * p.fx;
            J[6] = (1 + y * y / z_2) * p.fy; J[7] = -x * y / z_2 * p.fy; J[8] = -x / z * p.fy; J[9] = 0; J[10] = -1. / z * p.fy; J[11] = y / z_2 * p.fy;
        }
    }
}

__device__ __forceinline__ void huber(double e2, double delta, double& rho0, double& rho1)
{
    const double dsqr = delta * delta;
    if (e2 <= dsqr) { rho0 = e2; rho1 = 1.0; }
    else { const double s = sqrt(e2); rho0 = 2 * s * delta - dsqr; rho1 = delta / s; }
}

// Transposing wave reduction (see ba.hip): N per-lane values -> one wave total per lane.  Each step halves the number of values a lane carries by
// trading the half it does not keep with lane ^ O; once one value is left the remaining steps are a plain butterfly.  28 sums cost 29 double
// shuffles instead of 168 — ds_bpermute goes through the CU's LDS pipe, which the 8 waves of the workgroup share, and the butterflies were a
// quarter of an LM iteration.  idx = which of the N values the lane ends up with, valid = the lane is the one writer of that value.
template <int N, int O>
struct WaveTranspose {
    static __device__ __forceinline__ void run(double* v, int lane, int& idx, bool& valid)
    {
        if constexpr (O == 0) { idx = 0; valid = true; }
        else if constexpr (N > 1) {
            constexpr int H = (N + 1) / 2;
            const bool up = (lane & O) != 0;
#pragma unroll
            for (int a = 0; a < H; a++) {
                const double lo = v[a], hi = (a + H < N) ? v[a + H] : 0.0;
                const double send = up ? lo : hi, keep = up ? hi : lo;
                v[a] = keep + __shfl_xor(send, O, 64);
            }
            int sub; bool sv;
            WaveTranspose<H, O / 2>::run(v, lane, sub, sv);
            idx = sub + (up ? H : 0); valid = sv && idx < N;
        } else {
            v[0] += __shfl_xor(v[0], O, 64);
            WaveTranspose<1, O / 2>::run(v, lane, idx, valid);
            valid = valid && !(lane & O);
        }
    }
};
// Workgroup barrier that orders LDS traffic only.  Every per-edge array in global memory is written and read by the SAME thread in every pass
// (edge i belongs to the same thread of the same workgroup throughout), so nothing has to be visible across threads through global memory — but __syncthreads()
// also waits for the stores in flight (vmcnt(0)): 12 barriers per LM iteration x ~1.5 us of store latency was the whole n-independent part
// of an iteration (19 of 67 us at N = 3000).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- all-reduce over the workgroups of a problem (round 4) ---------------------------------------------------------------------------------------------------
// A problem is solved by a CLUSTER of G workgroups (G = ceil(n / 512) <= 8: one edge per thread) that run the same LM control flow on the same reduced numbers.
// Stage 1 is the workgroup reduction (wave transpose + one LDS pass); stage 2 exchanges the G workgroup totals through tagged granules (xwg.hpp): every workgroup
// publishes its totals, reads everybody's and adds them in rank order — the sums are bit-identical in all G workgroups, so every branch of the LM policy stays uniform
// across the cluster without a broadcast.  Two exchange buffers alternate by epoch parity: a workgroup can only be one exchange ahead of the slowest one (it needs that
// one's contribution to finish its own), so the buffer it overwrites at epoch e + 2 has been read by everybody.
#define PO_NMAX 56         // largest exchange: 29 linearisation entries + 27 Schur sums
#define PO_GMAX 8
#define PO_XCH_WORDS ((size_t)2 * PO_GMAX * PO_NMAX * 2)      // u64 words of one problem's exchange area
struct Cluster { unsigned long long* xch; unsigned* abort_word; int G, rank; unsigned epoch; bool dead; };
struct PoLds { double part[16 * 29]; double tot[PO_NMAX]; double xl[PO_GMAX * PO_NMAX]; double tsave[12]; int flag; };

// workgroup stage: entries [0, NSUM) of v are summed, [NSUM, NV) max-reduced over the workgroup; totals land in tot[0..NV) (visible to all threads on return)
template <int NV, int NSUM>
__device__ void wg_reduce(double* v, double* part /*[16][NV]*/, double* tot)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
    for (int k = NSUM; k < NV; k++) {
        double x = v[k];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) x = fmax(x, __shfl_xor(x, o, 64));
        v[k] = x;
    }
    int idx; bool valid;
    WaveTranspose<NSUM, 32>::run(v, lane, idx, valid);
    lds_barrier();
    if (valid) part[wave * NV + idx] = v[0];
    if (lane == 0) {
#pragma unroll
        for (int k = NSUM; k < NV; k++) part[wave * NV + k] = v[k];
    }
    lds_barrier();
    if (threadIdx.x < NV) {
        const int k = threadIdx.x;
        double x = part[k];
        for (int w = 1; w < nw; w++) x = k >= NSUM ? fmax(x, part[w * NV + k]) : x + part[w * NV + k];
        tot[k] = x;
    }
    lds_barrier();
}
// cluster stage on tot[0..nv): sum over the G workgroups (entry max_idx: maximum).  No-op for a single workgroup.
__device__ void cluster_sum(double* tot, int nv, int max_idx, PoLds& L, Cluster& cs)
{
    if (cs.G == 1) return;
    const unsigned ep = ++cs.epoch;
    unsigned long long* buf = cs.xch + (size_t)(ep & 1u) * PO_GMAX * PO_NMAX * 2;
    if (cs.dead) return;
    if ((int)threadIdx.x < nv) xwg_publish_f64(buf + ((size_t)cs.rank * PO_NMAX + threadIdx.x) * 2, ep, tot[threadIdx.x]);
    if (threadIdx.x == 0) L.flag = 0;
    lds_barrier();
    unsigned* xw = (unsigned*)L.xl;                       // [G][2 nv] 32-bit halves = [G][nv] doubles
    const int nwords = cs.G * nv * 2;
    for (int w = threadIdx.x; w < nwords; w += blockDim.x) {
        const int g = w / (2 * nv), j = w - g * 2 * nv;
        unsigned payload;
        if (!xwg_wait_word(buf + (size_t)g * PO_NMAX * 2 + j, ep, cs.abort_word, &payload)) L.flag = 1;
        xw[w] = payload;
    }
    lds_barrier();
    if (L.flag) { cs.dead = true; if (threadIdx.x == 0) xwg_store32(cs.abort_word, 1u); }
    if ((int)threadIdx.x < nv) {
        const int k = threadIdx.x;
        double x = L.xl[k];
        for (int g = 1; g < cs.G; g++) x = k == max_idx ? fmax(x, L.xl[g * nv + k]) : x + L.xl[g * nv + k];
        tot[k] = cs.dead ? 0.0 : x;
    }
    lds_barrier();
}

// 6x6 LDL^T solve in registers on the packed upper triangle (row-major: 00 01 .. 05 11 12 .. 55): the normal equations are wave-uniform
// values held in vector registers, and a full 6x6 copy of H plus one of S cost 144 of them (the kernel spilled, and a spill reload inside the
// edge loops waits for every outstanding global load).  Same operation order as the dense form.
#define PQ(a, c) ((a) * 6 - (a) * ((a) - 1) / 2 + ((c) - (a)))      /* c >= a */
#define LL(i, j) L[(i) * ((i) - 1) / 2 + (j)]                        /* i > j  */
__device__ bool ldlt6p(const double* A, const double* b, double* x)
{
    double L[15], D[6];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 6; j++) {
        double d = A[PQ(j, j)];
#pragma unroll
        for (int k = 0; k < 6; k++) if (k < j) d -= LL(j, k) * LL(j, k) * D[k];
        if (!(d > 0) || !isfinite(d)) { ok = false; d = 1.0; }
        D[j] = d;
#pragma unroll
        for (int i = 0; i < 6; i++) if (i > j) {
            double s = A[PQ(j, i)];
#pragma unroll
            for (int k = 0; k < 6; k++) if (k < j) s -= LL(i, k) * LL(j, k) * D[k];
            LL(i, j) = s / d;
        }
    }
#pragma unroll
    for (int i = 0; i < 6; i++) { double s = b[i];
#pragma unroll
        for (int k = 0; k < 6; k++) if (k < i) s -= LL(i, k) * x[k];
        x[i] = s; }
#pragma unroll
    for (int i = 0; i < 6; i++) x[i] /= D[i];
#pragma unroll
    for (int i = 5; i >= 0; i--) { double s = x[i];
#pragma unroll
        for (int k = 0; k < 6; k++) if (k > i) s -= LL(k, i) * x[k];
        x[i] = s; }
    return ok;
}

#define NRED 29      // 21 H + 6 b + chi + maxdiag

// One LM iteration costs two exchanges in the steady state (round 3: four workgroup reductions behind four passes over the edges):
//   pass L  residuals + Jacobians + normal equations; for the flow-coupled problems the Schur sums of the first trial are formed in the same pass from the values just
//           computed (lambda is known from the previous iteration; the first iteration of a round — lambda comes from this very linearisation — and repeated trials
//           take the separate pass S);
//   pass T  trial state: flow update, new residuals, robust chi2 of the trial (edge part and prior part apart), computeScale, and the prior part at the OLD flows — so
//           that activeRobustChi2() over "the errors left by the last trial" (sparse_optimizer.cpp:393-396), which round 3 evaluated with a fourth pass, is a sum of
//           numbers this pass already has.
// The end-of-round classification needs no reduction except the final inlier count.
__global__ __launch_bounds__(512) void k_pose_opt(const PoseProbDev* __restrict__ probs, const int2* __restrict__ wgmap, unsigned epoch0, unsigned* abort_word)
{
    __shared__ PoLds L;
    const int2 wm = wgmap[blockIdx.x];
    // a reference, not a copy: the ~130 dwords of the problem descriptor are wave-uniform and are re-read with scalar loads where they are used;
    // held in registers for the whole kernel they pushed the allocator into scratch (and a scratch reload waits on every store in flight)
    const PoseProbDev& p = probs[wm.x];
    Cluster cs{p.xch, abort_word, p.G, wm.y, epoch0, false};
    // the edge->thread mapping depends on the problem alone (not on the batch it is launched with), so results are bit-identical
    // however problems are grouped.  One workgroup: nt = ~4 edges per thread (threads beyond it only take part in the reductions); a cluster: 512 threads per workgroup
    const int n = p.n, G = p.G;
    const int nt = G > 1 ? (int)blockDim.x : min((int)blockDim.x, max(64, (((n + 3) >> 2) + 63) & ~63));
    const int tid = (int)threadIdx.x < nt ? wm.y * nt + (int)threadIdx.x : n, stride = G * nt;
    const bool flowm = p.mode == 1;
    Se3 T;
    auto reset_pose = [&]() {
#pragma unroll
        for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) T.R[r * 3 + c] = p.T_init[r * 4 + c]; T.t[r] = p.T_init[r * 4 + 3]; } };
    reset_pose();
    for (int i = tid; i < n; i += stride) {
        p.outlier[i] = 0; p.has_kernel[i] = p.use_huber ? 1 : 0;
        if (flowm) { p.f[2 * i] = p.flow0[2 * i]; p.f[2 * i + 1] = p.flow0[2 * i + 1]; }
    }
    int total_iters = 0, n_inl = 0; double chi2_final = 0;
    if (n >= 3) {
        for (int round = 0; round < p.rounds && !cs.dead; round++) {
            reset_pose();
            double lambda = -1, ni = 2, chi2_check = 0; int nBad = 0;
            const int round_iters = probs[wm.x].iters[round];        // dynamic index: read through the pointer so that the local copy `p` stays in registers
            const float round_chi2_th = probs[wm.x].chi2_th[round];
            for (int it = 0; it < round_iters && !cs.dead; it++) {
                // ---- pass L: computeActiveErrors + buildSystem (+ the first trial's Schur sums)
                const bool fused = flowm && it > 0;
                double acc[NRED], sa[27];
#pragma unroll
                for (int k = 0; k < NRED; k++) acc[k] = 0;
#pragma unroll
                for (int k = 0; k < 27; k++) sa[k] = 0;
                for (int i = tid; i < n; i += stride) {
                    double f0 = 0, f1 = 0, hll = 0, bl0 = 0, bl1 = 0;
                    if (flowm) {
                        f0 = p.f[2 * i]; f1 = p.f[2 * i + 1];
                        const double a = f0 - p.flow0[2 * i], b = f1 - p.flow0[2 * i + 1];
                        acc[27] += p.info_prior * (a * a + b * b);
                        hll = p.info_prior; bl0 = -p.info_prior * a; bl1 = -p.info_prior * b;
                    }
                    double wo = 0, e[2] = {0, 0}, J[12];
                    const EdgeIn in = edge_load(p, i);
                    const bool act = !p.outlier[i], hk = p.has_kernel[i] != 0;
                    if (act) {
                        edge_eval<true>(p, T, in, f0, f1, e, J);
                        p.err[2 * i] = e[0]; p.err[2 * i + 1] = e[1];
                        const double c2 = p.info_edge * (e[0] * e[0] + e[1] * e[1]);
                        double r0 = c2, w = 1; if (hk) huber(c2, p.huber_delta, r0, w);
                        acc[27] += r0;
                        wo = w * p.info_edge;
                        int q = 0;
#pragma unroll
                        for (int a = 0; a < 6; a++) {
                            acc[21 + a] -= wo * (J[a] * e[0] + J[6 + a] * e[1]);
#pragma unroll
                            for (int c = a; c < 6; c++) acc[q++] += wo * (J[a] * J[c] + J[6 + a] * J[6 + c]);
                        }
                    }
                    if (flowm) {
                        if (act) { hll += wo; bl0 -= wo * e[0]; bl1 -= wo * e[1]; }
                        p.Hll[i] = hll; p.bl[2 * i] = bl0; p.bl[2 * i + 1] = bl1;
                        double Bv[12];
#pragma unroll
                        for (int a = 0; a < 6; a++) { Bv[2 * a] = act ? wo * J[a] : 0.0; Bv[2 * a + 1] = act ? wo * J[6 + a] : 0.0; p.Hpl[12 * i + 2 * a] = Bv[2 * a]; p.Hpl[12 * i + 2 * a + 1] = Bv[2 * a + 1]; }
                        acc[28] = fmax(acc[28], fabs(hll));
                        if (fused) {
                            const double dinv = 1.0 / (hll + lambda);
                            int q = 0;
#pragma unroll
                            for (int a = 0; a < 6; a++) {
                                sa[21 + a] += dinv * (Bv[2 * a] * bl0 + Bv[2 * a + 1] * bl1);
#pragma unroll
                                for (int c = a; c < 6; c++) sa[q++] += dinv * (Bv[2 * a] * Bv[2 * c] + Bv[2 * a + 1] * Bv[2 * c + 1]);
                            }
                        }
                    }
                }
                wg_reduce<NRED, 28>(acc, L.part, L.tot);
                if (fused) { wg_reduce<27, 27>(sa, L.part, L.tot + NRED); cluster_sum(L.tot, NRED + 27, 28, L, cs); }
                else cluster_sum(L.tot, NRED, 28, L, cs);
                // the normal equations (packed upper triangle PQ(a, c) at tot[0..21), rhs at tot[21..27)) stay in LDS for the trials of this iteration: as wave-uniform values in
                // vector registers they cost 54 VGPRs for the whole trial loop
                const double* H = L.tot; const double* b6 = L.tot + 21;
                double currentChi = L.tot[27]; const double iniChi = L.tot[27];
                if (it == 0) {
                    double md = L.tot[28];
#pragma unroll
                    for (int a = 0; a < 6; a++) md = fmax(md, fabs(H[PQ(a, a)]));
                    lambda = 1e-5 * md; ni = 2; nBad = 0;
                }
                double rho = 0, last = 0; int qmax = 0;
                do {
                    if (threadIdx.x == 0) {                      // one lane, compile-time register indices
#pragma unroll
                        for (int k = 0; k < 9; k++) L.tsave[k] = T.R[k];
#pragma unroll
                        for (int k = 0; k < 3; k++) L.tsave[9 + k] = T.t[k];
                    }
                    double S[21], bs[6], xp[6];
#pragma unroll
                    for (int k = 0; k < 21; k++) S[k] = H[k];
#pragma unroll
                    for (int a = 0; a < 6; a++) { bs[a] = b6[a]; S[PQ(a, a)] += lambda; }
                    if (flowm) {
                        if (!(fused && qmax == 0)) {               // pass S: the Schur sums at this trial's lambda
#pragma unroll
                            for (int k = 0; k < 27; k++) sa[k] = 0;
                            for (int i = tid; i < n; i += stride) {
                                const double dinv = 1.0 / (p.Hll[i] + lambda); const double* B = p.Hpl + 12 * i;
                                const double g0 = p.bl[2 * i], g1 = p.bl[2 * i + 1];
                                double Bv[12];
#pragma unroll
                                for (int k = 0; k < 12; k++) Bv[k] = B[k];
                                int q = 0;
#pragma unroll
                                for (int a = 0; a < 6; a++) {
                                    sa[21 + a] += dinv * (Bv[2 * a] * g0 + Bv[2 * a + 1] * g1);
#pragma unroll
                                    for (int c = a; c < 6; c++) sa[q++] += dinv * (Bv[2 * a] * Bv[2 * c] + Bv[2 * a + 1] * Bv[2 * c + 1]);
                                }
                            }
                            wg_reduce<27, 27>(sa, L.part, L.tot + NRED);
                            cluster_sum(L.tot + NRED, 27, -1, L, cs);
                        }
#pragma unroll
                        for (int a = 0; a < 6; a++) bs[a] -= L.tot[NRED + 21 + a];
#pragma unroll
                        for (int q = 0; q < 21; q++) S[q] -= L.tot[NRED + q];
                    }
                    const bool ok2 = ldlt6p(S, bs, xp);
                    // pass T.  part: [0] robust chi2 of the edges at the trial state, [1] flow prior at the trial flows, [2] landmark part of computeScale, [3] flow prior at the flows before the trial
                    double part[4] = {0, 0, 0, 0};
                    if (ok2) se3_oplus_left(T, xp);
                    for (int i = tid; i < n; i += stride) {
                        double f0 = 0, f1 = 0;
                        const EdgeIn in = edge_load(p, i);
                        const bool out_i = p.outlier[i] != 0, hk = p.has_kernel[i] != 0;
                        if (flowm) {
                            f0 = p.f[2 * i]; f1 = p.f[2 * i + 1];
                            p.fsave[2 * i] = f0; p.fsave[2 * i + 1] = f1;
                            { const double a = f0 - p.flow0[2 * i], b = f1 - p.flow0[2 * i + 1]; part[3] += p.info_prior * (a * a + b * b); }
                            if (ok2) {
                                const double dinv = 1.0 / (p.Hll[i] + lambda); const double* B = p.Hpl + 12 * i;
                                double c0 = p.bl[2 * i], c1 = p.bl[2 * i + 1];
#pragma unroll
                                for (int a = 0; a < 6; a++) { c0 -= B[2 * a] * xp[a]; c1 -= B[2 * a + 1] * xp[a]; }
                                const double x0 = dinv * c0, x1 = dinv * c1;
                                part[2] += x0 * (lambda * x0 + p.bl[2 * i]) + x1 * (lambda * x1 + p.bl[2 * i + 1]);
                                f0 += x0; f1 += x1; p.f[2 * i] = f0; p.f[2 * i + 1] = f1;
                            }
                            const double a = f0 - p.flow0[2 * i], b = f1 - p.flow0[2 * i + 1];
                            part[1] += p.info_prior * (a * a + b * b);
                        }
                        if (!out_i) {
                            double e[2]; edge_eval<false>(p, T, in, f0, f1, e, nullptr);
                            p.err[2 * i] = e[0]; p.err[2 * i + 1] = e[1];
                            const double c2 = p.info_edge * (e[0] * e[0] + e[1] * e[1]);
                            double r0 = c2, w = 1; if (hk) huber(c2, p.huber_delta, r0, w);
                            part[0] += r0;
                        }
                    }
                    wg_reduce<4, 4>(part, L.part, L.tot + NRED + 27 - 27);      // (the Schur totals have been consumed: tot[NRED ..] is free again)
                    cluster_sum(L.tot + NRED, 4, -1, L, cs);
                    const double chiE = L.tot[NRED], priN = L.tot[NRED + 1], priO = L.tot[NRED + 3];
                    double tempChi = chiE + priN, scale = L.tot[NRED + 2];
                    if (ok2) {
#pragma unroll
                        for (int a = 0; a < 6; a++) scale += xp[a] * (lambda * xp[a] + b6[a]);
                    } else tempChi = DBL_MAX;
                    rho = (currentChi - tempChi) / (scale + 1e-3);
                    if (rho > 0 && isfinite(tempChi)) {
                        const double tr = 2 * rho - 1; double alpha = 1. - tr * tr * tr;      // pow(x, 3) of the reference; the libm call is ~500 instructions on every lane
                        alpha = fmin(alpha, 2. / 3.);
                        lambda *= fmax(1. / 3., alpha); ni = 2; currentChi = tempChi;
                        last = chiE + priN;                      // activeRobustChi2() after this trial: its errors, the flows it left
                    } else {
                        lambda *= ni; ni *= 2;
#pragma unroll
                        for (int k = 0; k < 9; k++) T.R[k] = L.tsave[k];       // written before the barriers of this trial's reductions; a re-write by the next trial stores the same values
#pragma unroll
                        for (int k = 0; k < 3; k++) T.t[k] = L.tsave[9 + k];
                        if (flowm) for (int i = tid; i < n; i += stride) { p.f[2 * i] = p.fsave[2 * i]; p.f[2 * i + 1] = p.fsave[2 * i + 1]; }
                        last = chiE + priO;                      // the trial's edge errors stay (g2o does not recompute them on a rejected step), the flow vertices are popped back
                    }
                    qmax++;
                    if (cs.dead) break;
                } while (rho < 0 && qmax < 10);
                total_iters++;
                bool terminate = (qmax == 10 || rho == 0);
                if (!terminate) { if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0; if (nBad >= 3) terminate = true; }
                if (chi2_check < last && it > 0) terminate = true;      // sparse_optimizer.cpp:393-396
                chi2_check = last;
                chi2_final = currentChi;
                if (terminate) break;
            }
            // ---- inlier / outlier classification
            double nb[1] = {0};
            for (int i = tid; i < n; i += stride) {
                double e0 = p.err[2 * i], e1 = p.err[2 * i + 1];
                if (p.outlier[i]) { double e[2]; edge_eval<false>(p, T, edge_load(p, i), flowm ? p.f[2 * i] : 0.0, flowm ? p.f[2 * i + 1] : 0.0, e, nullptr); e0 = e[0]; e1 = e[1]; p.err[2 * i] = e0; p.err[2 * i + 1] = e1; }
                const float chi2 = (float)(p.info_edge * (e0 * e0 + e1 * e1));
                if (chi2 > round_chi2_th) { p.outlier[i] = 1; nb[0] += 1; } else p.outlier[i] = 0;
                if (round == p.drop_kernel_after_round) p.has_kernel[i] = 0;
            }
            if (round == p.rounds - 1) {                      // (the count of the earlier rounds is never used)
                wg_reduce<1, 1>(nb, L.part, L.tot);
                cluster_sum(L.tot, 1, -1, L, cs);
                n_inl = n - (int)L.tot[0];
            }
        }
    }
    if (threadIdx.x == 0 && wm.y == 0) {
        vido_pose_result* r = p.res;
        for (int rr = 0; rr < 3; rr++) { for (int c = 0; c < 3; c++) r->T[rr * 4 + c] = T.R[rr * 3 + c]; r->T[rr * 4 + 3] = T.t[rr]; }
        r->T[12] = r->T[13] = r->T[14] = 0; r->T[15] = 1;
        r->n_inliers = cs.dead ? -1 : n_inl; r->lm_iterations = cs.dead ? -1 : total_iters; r->chi2_final = chi2_final;
    }
}

// ---- host ------------------------------------------------------------------------------------------
struct PoseState {
    double* d_arena = nullptr; size_t arena_cap = 0;       // doubles
    unsigned char* d_bytes = nullptr; size_t bytes_cap = 0;
    PoseProbDev* d_probs = nullptr; vido_pose_result* d_res = nullptr; size_t prob_cap = 0;
    double* h_stage = nullptr; size_t stage_cap = 0;       // pinned
    PoseProbDev* h_probs = nullptr; vido_pose_result* h_res = nullptr; unsigned char* h_bytes = nullptr; size_t hbytes_cap = 0;
    unsigned long long* d_xch = nullptr; int2* d_wgmap = nullptr; int2* h_wgmap = nullptr; unsigned* d_abort = nullptr;   // cluster exchange areas [prob_cap], workgroup -> (problem, rank)
    unsigned launches = 0;                                  // epoch base of a launch = launches << PO_EPOCH_SHIFT: tags of earlier launches never match
};
#define PO_EPOCH_SHIFT 14                                   // > the exchanges of one launch (4 rounds x 100 iterations x (1 + 2 x 10 trials) = 8400)
#define PO_MAX_WGS 64                                       // workgroups per launch: a quarter of the chip, so that every cluster is resident whatever else runs (xwg.hpp)
static inline int pose_cluster_size(int n) { return std::min(PO_GMAX, std::max(1, (n + 511) / 512)); }

template <class T>
static int grow_dev(vido_ctx* ctx, T** p, size_t* cap, size_t need)
{
    if (need <= *cap) return VIDO_OK;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (*p) { HIP_TRY(ctx, hipFree(*p)); *p = nullptr; }
    const size_t n = need + need / 2 + 64;
    HIP_TRY(ctx, hipMalloc((void**)p, n * sizeof(T)));
    *cap = n;
    return VIDO_OK;
}
template <class T>
static int grow_pinned(vido_ctx* ctx, T** p, size_t* cap, size_t need)
{
    if (need <= *cap) return VIDO_OK;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (*p) { HIP_TRY(ctx, hipHostFree(*p)); *p = nullptr; }
    const size_t n = need + need / 2 + 64;
    HIP_TRY(ctx, hipHostMalloc((void**)p, n * sizeof(T)));
    *cap = n;
    return VIDO_OK;
}

void pose_state_destroy(vido_ctx* ctx)
{
    PoseState* S = ctx->pose;
    if (!S) return;
    hipFree(S->d_arena); hipFree(S->d_bytes); hipFree(S->d_probs); hipFree(S->d_res); hipFree(S->d_xch); hipFree(S->d_wgmap); hipFree(S->d_abort);
    hipHostFree(S->h_stage); hipHostFree(S->h_probs); hipHostFree(S->h_res); hipHostFree(S->h_bytes); hipHostFree(S->h_wgmap);
    delete S; ctx->pose = nullptr;
}

extern "C" int vido_pose_optimize_batch(vido_ctx* ctx, const vido_pose_problem* probs, int n_prob, vido_pose_result* results,
                                        uint8_t* const* outlier_out, double* const* flow_out)
{
    if (!ctx) return VIDO_E_INVALID;
    if (!probs || !results || n_prob < 1) return vido_set_error(ctx, VIDO_E_INVALID, "pose_optimize: null/empty problem list");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!ctx->pose) ctx->pose = new PoseState();
    PoseState* S = ctx->pose;
    hipStream_t st = ctx->stream;
    size_t nd_in = 0, nd_work = 0, nb = 0; int nmax = 0;
    for (int k = 0; k < n_prob; k++) {
        const vido_pose_problem& p = probs[k];
        if (p.n < 0 || p.mode < 0 || p.mode > 2 || p.rounds < 1 || p.rounds > 4 || (p.n > 0 && !p.obs) || (p.mode != 1 && p.n > 0 && !p.Xw) ||
            (p.mode == 1 && p.n > 0 && (!p.flow0 || !p.depth)))
            return vido_set_error(ctx, VIDO_E_INVALID, "pose_optimize: problem %d is malformed (mode %d, n %d)", k, p.mode, p.n);
        {   // a clustered problem (n > 512) tags its exchanges with epochs from a window of 1 << PO_EPOCH_SHIFT per launch: at most 1 + 2 x 10 exchanges per LM iteration
            // (+ 1 per round), so the iteration counts must keep a launch inside its window or the NEXT launch would match this one's stale tags
            long long it = 0; bool neg = false;
            for (int q = 0; q < p.rounds; q++) { neg |= p.iters[q] < 0; it += p.iters[q]; }
            if (neg || 21 * it + p.rounds + 1 >= (1ll << PO_EPOCH_SHIFT))
                return vido_set_error(ctx, VIDO_E_INVALID, "pose_optimize: problem %d asks for %lld LM iterations over its rounds (allowed: 0 .. %lld)", k, it, ((1ll << PO_EPOCH_SHIFT) - 6) / 21);
        }
        const size_t n = (size_t)p.n;
        nd_in += 8 * n + 8; nd_work += 23 * n + 32; nb += 2 * n + 16; nmax = std::max(nmax, p.n);
    }
    int rc;
    if ((rc = grow_dev(ctx, &S->d_arena, &S->arena_cap, nd_in + nd_work))) return rc;
    if ((rc = grow_dev(ctx, &S->d_bytes, &S->bytes_cap, nb))) return rc;
    if ((rc = grow_pinned(ctx, &S->h_stage, &S->stage_cap, std::max(nd_in, (size_t)2 * nmax * n_prob + 16)))) return rc;
    if ((rc = grow_pinned(ctx, &S->h_bytes, &S->hbytes_cap, nb))) return rc;
    if ((size_t)n_prob > S->prob_cap) {
        HIP_TRY(ctx, hipStreamSynchronize(st));
        if (S->d_probs) { hipFree(S->d_probs); hipFree(S->d_res); hipHostFree(S->h_probs); hipHostFree(S->h_res); hipFree(S->d_xch); hipFree(S->d_wgmap); hipHostFree(S->h_wgmap); }
        S->prob_cap = (size_t)n_prob * 2 + 8;
        HIP_TRY(ctx, hipMalloc((void**)&S->d_probs, S->prob_cap * sizeof(PoseProbDev))); HIP_TRY(ctx, hipMalloc((void**)&S->d_res, S->prob_cap * sizeof(vido_pose_result)));
        HIP_TRY(ctx, hipHostMalloc((void**)&S->h_probs, S->prob_cap * sizeof(PoseProbDev))); HIP_TRY(ctx, hipHostMalloc((void**)&S->h_res, S->prob_cap * sizeof(vido_pose_result)));
        HIP_TRY(ctx, hipMalloc((void**)&S->d_xch, S->prob_cap * PO_XCH_WORDS * sizeof(unsigned long long))); HIP_TRY(ctx, hipMemset(S->d_xch, 0, S->prob_cap * PO_XCH_WORDS * sizeof(unsigned long long)));
        HIP_TRY(ctx, hipMalloc((void**)&S->d_wgmap, S->prob_cap * PO_GMAX * sizeof(int2))); HIP_TRY(ctx, hipHostMalloc((void**)&S->h_wgmap, S->prob_cap * PO_GMAX * sizeof(int2)));
        S->launches = 0;                                  // (fresh, zeroed exchange areas)
    }
    if (!S->d_abort) { HIP_TRY(ctx, hipMalloc((void**)&S->d_abort, sizeof(unsigned))); HIP_TRY(ctx, hipMemset(S->d_abort, 0, sizeof(unsigned))); }
    if (S->launches + (unsigned)n_prob + 2 >= (1u << (32 - PO_EPOCH_SHIFT)) - 1) {      // the 32-bit tags are about to wrap: start over on zeroed areas
        HIP_TRY(ctx, hipStreamSynchronize(st)); HIP_TRY(ctx, hipMemset(S->d_xch, 0, S->prob_cap * PO_XCH_WORDS * sizeof(unsigned long long))); S->launches = 0;
    }
    // pack inputs into the pinned stage and lay out the device arena
    size_t so = 0, wo = nd_in, bo = 0;
    std::vector<size_t> flow_off(n_prob), byte_off(n_prob);
    for (int k = 0; k < n_prob; k++) {
        const vido_pose_problem& p = probs[k]; PoseProbDev& d = S->h_probs[k];
        const size_t n = (size_t)p.n;
        memset(&d, 0, sizeof d);
        d.mode = p.mode; d.n = p.n;
        auto put = [&](const double* src, size_t cnt) -> const double* {
            const double* dev = S->d_arena + so;
            if (src && cnt) memcpy(S->h_stage + so, src, cnt * sizeof(double)); else if (cnt) memset(S->h_stage + so, 0, cnt * sizeof(double));
            so += cnt; return dev; };
        d.Xw = put(p.mode != 1 ? p.Xw : nullptr, p.mode != 1 ? 3 * n : 0);
        d.obs = put(p.obs, 2 * n);
        d.flow0 = put(p.mode == 1 ? p.flow0 : nullptr, p.mode == 1 ? 2 * n : 0);
        d.depth = put(p.mode == 1 ? p.depth : nullptr, p.mode == 1 ? n : 0);
        so = (so + 1) & ~(size_t)1;
        memcpy(d.Twl, p.Twl, sizeof d.Twl); memcpy(d.P, p.P, sizeof d.P); memcpy(d.T_init, p.T_init, sizeof d.T_init);
        d.fx = p.fx; d.fy = p.fy; d.cx = p.cx; d.cy = p.cy; d.info_edge = p.info_edge; d.info_prior = p.info_prior; d.huber_delta = p.huber_delta;
        d.use_huber = p.use_huber; d.rounds = p.rounds; d.drop_kernel_after_round = p.drop_kernel_after_round;
        for (int q = 0; q < 4; q++) { d.iters[q] = p.iters[q]; d.chi2_th[q] = p.chi2_th[q]; }
        double* w = S->d_arena + wo;
        d.f = w; flow_off[k] = wo; w += 2 * n; d.err = w; w += 2 * n; d.fsave = w; w += 2 * n; d.Hpl = w; w += 12 * n; d.Hll = w; w += n; d.bl = w; w += 2 * n; d.xl = w; w += 2 * n;
        wo += 23 * n + 32;
        d.outlier = S->d_bytes + bo; byte_off[k] = bo; d.has_kernel = S->d_bytes + bo + n; bo += 2 * n + 16;
        d.res = S->d_res + k;
        d.G = pose_cluster_size(p.n); d.xch = S->d_xch + (size_t)k * PO_XCH_WORDS;
    }
    if (so) HIP_TRY(ctx, hipMemcpyAsync(S->d_arena, S->h_stage, so * sizeof(double), hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(S->d_probs, S->h_probs, n_prob * sizeof(PoseProbDev), hipMemcpyHostToDevice, st));
    const int threads = nmax > 512 ? 512 : std::min(512, std::max(64, ((nmax + 3) / 4 + 63) & ~63));      // one workgroup: ~4 edges per thread; a cluster (n > 512): 512 threads, one edge each (2 waves per SIMD keeps the 256-VGPR budget)
    // workgroup table: the G workgroups of a problem are consecutive; launches of at most PO_MAX_WGS workgroups, whole clusters each (stream order between them)
    int n_wg = 0;
    for (int k = 0; k < n_prob; k++) for (int r = 0; r < S->h_probs[k].G; r++) S->h_wgmap[n_wg++] = make_int2(k, r);
    HIP_TRY(ctx, hipMemcpyAsync(S->d_wgmap, S->h_wgmap, (size_t)n_wg * sizeof(int2), hipMemcpyHostToDevice, st));
    for (int w0 = 0, k = 0; k < n_prob;) {
        int w1 = w0, k1 = k;
        while (k1 < n_prob && (w1 == w0 || w1 - w0 + S->h_probs[k1].G <= PO_MAX_WGS)) { w1 += S->h_probs[k1].G; k1++; }
        S->launches++;
        hipLaunchKernelGGL(k_pose_opt, dim3(w1 - w0), dim3(threads), 0, st, S->d_probs, (const int2*)(S->d_wgmap + w0), S->launches << PO_EPOCH_SHIFT, S->d_abort);
        w0 = w1; k = k1;
    }
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(S->h_res, S->d_res, n_prob * sizeof(vido_pose_result), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipMemcpyAsync(S->h_bytes, S->d_bytes, bo, hipMemcpyDeviceToHost, st));
    // the refined flows come back through the same pinned stage: the uploads that read it precede these copies in stream order, so no host
    // synchronisation is needed in between
    size_t fo = 0;
    std::vector<size_t> fstage(n_prob);
    for (int k = 0; k < n_prob; k++) {
        const size_t n = (size_t)probs[k].n;
        fstage[k] = fo;
        if (flow_out && flow_out[k] && probs[k].mode == 1 && n) { HIP_TRY(ctx, hipMemcpyAsync(S->h_stage + fo, S->d_arena + flow_off[k], 2 * n * sizeof(double), hipMemcpyDeviceToHost, st)); fo += 2 * n; }
    }
    HIP_TRY(ctx, hipStreamSynchronize(st));
    for (int k = 0; k < n_prob; k++) if (S->h_res[k].lm_iterations < 0) {      // a cluster gave up waiting for one of its workgroups (xwg.hpp): never seen, reported instead of hanging
        HIP_TRY(ctx, hipMemset(S->d_abort, 0, sizeof(unsigned)));
        return vido_set_error(ctx, VIDO_E_HIP, "pose_optimize: the workgroups of problem %d lost each other (exchange timed out)", k);
    }
    for (int k = 0; k < n_prob; k++) {
        const size_t n = (size_t)probs[k].n;
        results[k] = S->h_res[k];
        if (outlier_out && outlier_out[k] && n) memcpy(outlier_out[k], S->h_bytes + byte_off[k], n);
        if (flow_out && flow_out[k] && n) { if (probs[k].mode == 1) memcpy(flow_out[k], S->h_stage + fstage[k], 2 * n * sizeof(double)); else memset(flow_out[k], 0, 2 * n * sizeof(double)); }
    }
    return VIDO_OK;
}

extern "C" int vido_pose_optimize(vido_ctx* ctx, const vido_pose_problem* prob, vido_pose_result* result, uint8_t* outlier_out, double* flow_out)
{
    uint8_t* o[1] = {outlier_out}; double* f[1] = {flow_out};
    return vido_pose_optimize_batch(ctx, prob, 1, result, o, f);
}
