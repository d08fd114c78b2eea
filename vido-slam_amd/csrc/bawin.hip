// bawin.hip — the local-BA window RESIDENT on the device between frames (second half).
//
// The reference re-assembles the graph of Optimizer::PartialBatchOptimization from the Map on every call (vido_slam/src/Optimizer.cc:56-94, 276-350: a walk over the
// window's frames, features, tracklet labels and cv::Mat 3x1 points), and rounds 1-2 of this build did the same walk on the host and uploaded the flat problem every frame
// (~35 k observations, ~1.1 MB, + the host-side sort / slot tables of the solver).  Here the per-feature data of the last frames stays on the device:
//   per ring slot (frame % cap_frames):  meas [n][3] f64   Optimizer::Get3DinCamera of every static feature      (Map::vpFeatSta / vfDepSta)
//                                        xyz  [n][3] f32   its world point                                          (Map::vp3DPointSta — updated in place after every solve)
//                                        asso [n]    i32   index of the feature it continues in the previous frame  (Map::vnAssoSta)
//                                        trk / pos [n]     owning tracklet (length >= 3) and position in it        (Map::vnTrkSta / vnPosSta)
// Per frame the host sends only the NEW frame's rows (~80 KB) and the tracklet-label changes Map::UpdateTracklets made (a few thousand (frame, feature, tracklet, position)
// quads); k_bawin_assemble then builds, on the device, exactly the arrays the host walk built — observations in (frame, feature) order, landmark ids in order of first
// appearance, landmark chains followed through `asso` like the walk follows its `mak` table, chains that start before the window dropped — plus the solver's landmark-major
// slot tables, and ba.hip solves on them in place (ba_run_device_inputs).  Afterwards k_bawin_writeback stores the refined landmark into every observation slot's xyz
// (Optimizer.cc:1130-1160).  What still crosses PCIe per solve: the window's poses and odometry factors (2 KB each way) and two counters.
#include "common.hpp"
#include <chrono>
#include <vector>
#include <cstring>

struct BaWin {
    int cap_f = 0, cap_n = 0, cap_obs = 0, cap_pt = 0;
    double* d_meas = nullptr; float* d_xyz = nullptr; int *d_asso = nullptr, *d_trk = nullptr, *d_pos = nullptr, *d_pid = nullptr, *d_nfeat = nullptr;
    int *d_obs_cam = nullptr, *d_obs_pt = nullptr, *d_obs_pos = nullptr, *d_obs_src = nullptr, *d_pt_start = nullptr, *d_slot_cam = nullptr, *d_cnt = nullptr, *d_first = nullptr;
    double *d_obs_meas = nullptr, *d_pt = nullptr;
    int* d_counts = nullptr; int* h_counts = nullptr;              // [4]: n_obs, n_pt, overflow, max track
    char* h_stage = nullptr; size_t stage_cap = 0; char* d_stage = nullptr;
    std::vector<int> frame_of, nfeat;                               // per ring slot: the frame it holds (-1: empty), its feature count
    int last_start = -1, last_N = -1, last_nobs = 0;
};

#define BAWIN_MAX_FEATURES 8192      // per frame: 1024 threads x 8 features, and n < 2^16 for the packed scan of k_bawin_assemble

namespace {
struct BaWinDev {
    int cap_f, cap_n, cap_obs, cap_pt;
    const double* meas; float* xyz; const int *asso, *trk, *pos; int* pid; const int* nfeat;
    int *obs_cam, *obs_pt, *obs_pos, *obs_src, *pt_start, *slot_cam, *cnt, *first; double *obs_meas, *pt;
};

// exclusive prefix sum over the workgroup (1024 threads); *total = sum
__device__ int bw_excl_scan(int v, int* wsum /*[17]*/, int* total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
    __syncthreads();
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) { int run = 0; for (int w = 0; w < 16; w++) { const int t = wsum[w]; wsum[w] = run; run += t; } wsum[16] = run; }
    __syncthreads();
    *total = wsum[16];
    return wsum[wave] + incl - v;
}

// One frame's rows from the staging block [meas n x 3 f64 | xyz n x 3 f32 | asso n i32] into ring slot s; the slot's labels start as "no tracklet".  One upload + this
// launch instead of four copies and two fills per frame (every stream operation of the tracker queues behind the networks' workgroups).
__global__ __launch_bounds__(256) void k_bawin_ingest(const char* __restrict__ stage, int n, int cap_n, size_t o, double* __restrict__ meas, float* __restrict__ xyz,
                                                      int* __restrict__ asso, int* __restrict__ trk, int* __restrict__ pos, int* __restrict__ nfeat_s)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) *nfeat_s = n;
    if (i < cap_n) { trk[o + i] = -1; pos[o + i] = 0; }
    if (i >= n) return;
    const double* sm = (const double*)stage; const float* sx = (const float*)(stage + (size_t)n * 24); const int* sa = (const int*)(stage + (size_t)n * 36);
#pragma unroll
    for (int a = 0; a < 3; a++) { meas[3 * (o + i) + a] = sm[3 * (size_t)i + a]; xyz[3 * (o + i) + a] = sx[3 * (size_t)i + a]; }
    asso[o + i] = sa[i];
}

__global__ __launch_bounds__(256) void k_bawin_labels(const int4* __restrict__ upd, int n, int cap_f, int cap_n, int* __restrict__ trk, int* __restrict__ pos)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int4 u = upd[i];                                        // (frame, feature, tracklet, position)
    const size_t o = (size_t)(u.x % cap_f) * cap_n + u.y;
    trk[o] = u.z; pos[o] = u.w;
}

// The Map walk of batch_optimize (facade.cpp) / Optimizer.cc:276-350 for the frames [start, N), by ONE workgroup (the frames are a dependent chain: a feature's landmark is
// its predecessor's).  Per frame: landmark id of every labelled feature (new id for position 0 — ids in (frame, feature) order — else the predecessor's id, -1 when the chain
// starts before the window), the observation list in the same order, the landmark's first point and first camera; then the landmark-major slot tables.
__global__ __launch_bounds__(1024) void k_bawin_assemble(BaWinDev W, int start, int N, int* __restrict__ counts)
{
    __shared__ int wsum[17];
    const int tid = threadIdx.x;
    if (tid == 0) { counts[2] = 0; counts[3] = 0; }            // (the flags / maximum this launch accumulates below; counts[0..1] are plain stores at the end)
    __syncthreads();
    // (W.cnt is all zero on entry: cleared at creation and re-cleared after every solve by k_bawin_slots, all CUs — one workgroup clearing cap_pt counters here took 60 of
    //  this kernel's 258 us, a memset on the stream is one more operation queueing behind the networks)
    int n_pt = 0, n_obs = 0, overflow = 0;
    for (int f = start; f < N; f++) {
        const int s = f % W.cap_f, sp = (f + W.cap_f - 1) % W.cap_f, n = W.nfeat[s], np = f > start ? W.nfeat[sp] : 0;
        // a thread takes ipt CONSECUTIVE features, so that one scan over the per-thread sums gives ids and observation positions in feature order (three scans of 1024
        // features each per frame, as the first version did, are 9 barriers per frame on a chain of 20 frames)
        const int ipt = (n + 1023) >> 10;                         // <= 8 (cap_features <= 8192)
        int pm[8]; unsigned char fl[8];                            // predecessor's landmark, flags: 1 = starts a landmark, 2 = has a landmark
        int mine_new = 0, mine_has = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            pm[i] = -1; fl[i] = 0;
            const int j = tid * ipt + i;
            if (i < ipt && j < n) {
                const size_t o = (size_t)s * W.cap_n + j;
                const int lab = W.trk[o], ps = W.pos[o];
                const bool is_new = lab != -1 && ps == 0;
                if (lab != -1 && ps > 0 && f > start) { const int a = W.asso[o]; if (a >= 0 && a < np) pm[i] = W.pid[(size_t)sp * W.cap_n + a]; }
                fl[i] = (unsigned char)((is_new ? 1 : 0) | ((is_new || pm[i] != -1) ? 2 : 0));
                mine_new += fl[i] & 1; mine_has += (fl[i] >> 1) & 1;
            }
        }
        int tot; const int ex = bw_excl_scan(mine_new | (mine_has << 16), wsum, &tot);
        int pnew = n_pt + (ex & 0xffff), k = n_obs + (ex >> 16);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int j = tid * ipt + i;
            if (i < ipt && j < n) {
                const size_t o = (size_t)s * W.cap_n + j;
                const bool is_new = fl[i] & 1, has = fl[i] & 2;
                const int p = is_new ? pnew : pm[i];
                W.pid[o] = has ? p : -1;
                if (has) {
                    if (k < W.cap_obs && p < W.cap_pt) {
                        W.obs_cam[k] = f - start; W.obs_pt[k] = p; W.obs_src[k] = (int)o;
                        W.obs_meas[3 * (size_t)k] = W.meas[3 * o]; W.obs_meas[3 * (size_t)k + 1] = W.meas[3 * o + 1]; W.obs_meas[3 * (size_t)k + 2] = W.meas[3 * o + 2];
                        // a landmark is seen once per frame from its first frame on: the observation of frame f must be number (f - start) - first.  Two features of one
                        // frame that continue the same predecessor (or a gap in the chain) would share / skip a slot and leave another one unwritten -> flagged,
                        // the host reports VIDO_E_INVALID like ba_run does for host inputs
                        const int seen = atomicAdd(&W.cnt[p], 1);
                        if (seen != (is_new ? 0 : (f - start) - W.first[p])) overflow |= 2;
                        if (is_new) { W.first[p] = f - start; W.pt[3 * (size_t)p] = (double)W.xyz[3 * o]; W.pt[3 * (size_t)p + 1] = (double)W.xyz[3 * o + 1]; W.pt[3 * (size_t)p + 2] = (double)W.xyz[3 * o + 2]; }
                    } else overflow |= 1;
                    k++; pnew += is_new ? 1 : 0;
                }
            }
        }
        n_pt += tot & 0xffff; n_obs += tot >> 16;
        __threadfence_block(); __syncthreads();                  // this frame's ids are the next frame's predecessors
    }
    n_pt = min(n_pt, W.cap_pt); n_obs = min(n_obs, W.cap_obs);
    __threadfence(); __syncthreads();
    // landmark-major slots: pt_start = exclusive scan of the landmarks' observation counts; a landmark is seen once per frame from its first frame on, so the slot of an
    // observation is pt_start[landmark] + (camera - first camera) — the slots of a landmark in ascending camera order, as the solver expects
    int run = 0, maxk = 0;
    for (int p0 = 0; p0 < n_pt; p0 += 1024) {
        const int p = p0 + tid; const int c = p < n_pt ? __hip_atomic_load(&W.cnt[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        maxk = max(maxk, c);
        int tot; const int ex = bw_excl_scan(c, wsum, &tot);
        if (p < n_pt) W.pt_start[p] = run + ex;
        run += tot;
    }
    if (tid == 0) W.pt_start[n_pt] = run;
    // (obs_pos / slot_cam: k_bawin_slots, all CUs — 35 dependent three-load trips per thread of this one workgroup were 60 of its 160 us)
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { maxk = max(maxk, __shfl_xor(maxk, o, 64)); overflow |= __shfl_xor(overflow, o, 64); }
    if ((tid & 63) == 0) { atomicMax(&counts[3], maxk); if (overflow) atomicOr(&counts[2], overflow); }
    if (tid == 0) { counts[0] = n_obs; counts[1] = n_pt; if (run != n_obs) atomicOr(&counts[2], 1); }
}

// slot of an observation = pt_start[landmark] + (camera - first camera of the landmark); counts[0] = number of observations (written by k_bawin_assemble)
__global__ __launch_bounds__(256) void k_bawin_slots(BaWinDev W, const int* __restrict__ counts, int clear_n)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < clear_n) W.cnt[k] = 0;                             // the per-landmark counters of this assembly, for the next one (k_bawin_assemble is done with them)
    if (k >= counts[0]) return;
    const int p = W.obs_pt[k], c = W.obs_cam[k], sl = W.pt_start[p] + (c - W.first[p]);
    W.obs_pos[k] = sl; W.slot_cam[sl] = c;
}

// Optimizer.cc:1130-1160: the refined landmark goes into every observation slot of the window (float, like the Map's cv::Mat 3x1 CV_32F)
__global__ __launch_bounds__(256) void k_bawin_writeback(const int* __restrict__ obs_src, const int* __restrict__ obs_pt, const double* __restrict__ pt, int n_obs, float* __restrict__ xyz)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_obs) return;
    const size_t o = (size_t)obs_src[k]; const double* p = pt + 3 * (size_t)obs_pt[k];
    xyz[3 * o] = (float)p[0]; xyz[3 * o + 1] = (float)p[1]; xyz[3 * o + 2] = (float)p[2];
}
}  // namespace

void bawin_state_destroy(vido_ctx* ctx)
{
    BaWin* B = ctx->bawin; if (!B) return;
    for (void* p : {(void*)B->d_meas, (void*)B->d_xyz, (void*)B->d_asso, (void*)B->d_trk, (void*)B->d_pos, (void*)B->d_pid, (void*)B->d_nfeat, (void*)B->d_obs_cam, (void*)B->d_obs_pt, (void*)B->d_obs_pos,
                    (void*)B->d_obs_src, (void*)B->d_pt_start, (void*)B->d_slot_cam, (void*)B->d_cnt, (void*)B->d_first, (void*)B->d_obs_meas, (void*)B->d_pt, (void*)B->d_counts, (void*)B->d_stage}) if (p) hipFree(p);
    if (B->h_counts) hipHostFree(B->h_counts);
    if (B->h_stage) hipHostFree(B->h_stage);
    delete B; ctx->bawin = nullptr;
}

extern "C" {

/* The device-resident window of PartialBatchOptimization (Optimizer.cc:43-1228): a ring of cap_frames frames (> the optimisation window) with up to cap_features static
 * features each.  Re-creating it drops the stored frames. */
int vido_bawin_create(vido_ctx* ctx, int cap_frames, int cap_features)
{
    if (!ctx) return VIDO_E_INVALID;
    // k_bawin_assemble walks at most 8 features per thread of its 1024-thread workgroup and packs two counts into the 16-bit halves of one scan word
    if (cap_frames < 2 || cap_frames > 64 || cap_features < 1 || cap_features > BAWIN_MAX_FEATURES)
        return vido_set_error(ctx, VIDO_E_INVALID, "bawin_create: bad capacities (2..64 frames, 1..%d features per frame)", BAWIN_MAX_FEATURES);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    bawin_state_destroy(ctx);
    BaWin* B = new BaWin(); ctx->bawin = B;
    B->cap_f = cap_frames; B->cap_n = cap_features; B->cap_obs = cap_frames * cap_features; B->cap_pt = B->cap_obs;
    const size_t nf = (size_t)cap_frames * cap_features;
    HIP_TRY(ctx, hipMalloc((void**)&B->d_meas, nf * 24)); HIP_TRY(ctx, hipMalloc((void**)&B->d_xyz, nf * 12));
    HIP_TRY(ctx, hipMalloc((void**)&B->d_asso, nf * 4)); HIP_TRY(ctx, hipMalloc((void**)&B->d_trk, nf * 4)); HIP_TRY(ctx, hipMalloc((void**)&B->d_pos, nf * 4)); HIP_TRY(ctx, hipMalloc((void**)&B->d_pid, nf * 4));
    HIP_TRY(ctx, hipMalloc((void**)&B->d_nfeat, cap_frames * 4));
    HIP_TRY(ctx, hipMalloc((void**)&B->d_obs_cam, nf * 4)); HIP_TRY(ctx, hipMalloc((void**)&B->d_obs_pt, nf * 4)); HIP_TRY(ctx, hipMalloc((void**)&B->d_obs_pos, nf * 4)); HIP_TRY(ctx, hipMalloc((void**)&B->d_obs_src, nf * 4));
    HIP_TRY(ctx, hipMalloc((void**)&B->d_pt_start, (nf + 1) * 4)); HIP_TRY(ctx, hipMalloc((void**)&B->d_slot_cam, nf * 4)); HIP_TRY(ctx, hipMalloc((void**)&B->d_cnt, nf * 4)); HIP_TRY(ctx, hipMalloc((void**)&B->d_first, nf * 4));
    HIP_TRY(ctx, hipMalloc((void**)&B->d_obs_meas, nf * 24)); HIP_TRY(ctx, hipMalloc((void**)&B->d_pt, nf * 24));
    HIP_TRY(ctx, hipMalloc((void**)&B->d_counts, 16)); HIP_TRY(ctx, hipHostMalloc((void**)&B->h_counts, 16));
    B->stage_cap = (size_t)cap_features * (24 + 12 + 4) + (size_t)cap_features * 16 * 4 + 4096;      // one frame's rows, or 4 x cap_features label quads
    HIP_TRY(ctx, hipHostMalloc((void**)&B->h_stage, B->stage_cap)); HIP_TRY(ctx, hipMalloc((void**)&B->d_stage, B->stage_cap));
    HIP_TRY(ctx, hipMemset(B->d_trk, 0xff, nf * 4)); HIP_TRY(ctx, hipMemset(B->d_pos, 0, nf * 4)); HIP_TRY(ctx, hipMemset(B->d_nfeat, 0, cap_frames * 4));
    HIP_TRY(ctx, hipMemset(B->d_cnt, 0, nf * 4)); HIP_TRY(ctx, hipMemset(B->d_counts, 0, 16));
    B->frame_of.assign(cap_frames, -1); B->nfeat.assign(cap_frames, 0);
    return VIDO_OK;
}

/* One frame's static features enter the ring (what Tracking::Track pushes into the Map, Tracking.cc:1382-1394): meas [n][3] f64 = Get3DinCamera(feature, depth),
 * xyz [n][3] f32 = the feature's world point, asso [n] = index of the previous frame's feature it continues (-1: none; NULL for the first frame).  Labels start as "no
 * tracklet"; vido_bawin_set_labels brings them in. */
int vido_bawin_push_frame(vido_ctx* ctx, int frame, int n, const double* meas, const float* xyz, const int32_t* asso)
{
    if (!ctx || !ctx->bawin) return VIDO_E_INVALID;
    BaWin* B = ctx->bawin;
    if (frame < 0 || n < 0 || n > B->cap_n || (n && (!meas || !xyz))) return vido_set_error(ctx, n > B->cap_n ? VIDO_E_CAPACITY : VIDO_E_INVALID, "bawin_push_frame: %d features (capacity %d)", n, B->cap_n);
    hipStream_t st = ctx->stream; const int s = frame % B->cap_f;
    HIP_TRY(ctx, hipStreamSynchronize(st));                                           // the staging buffer is reused
    char* h = B->h_stage;                                                              // [meas n x 24 | xyz n x 12 | asso n x 4], 8-byte aligned pieces (n x 24 and n x 36 are multiples of 4; meas first)
    memcpy(h, meas, (size_t)n * 24); memcpy(h + (size_t)n * 24, xyz, (size_t)n * 12);
    int* ha = (int*)(h + (size_t)n * 36);
    for (int i = 0; i < n; i++) ha[i] = asso ? asso[i] : -1;
    const size_t o = (size_t)s * B->cap_n;
    static const bool trace = getenv("VIDO_LBA_TRACE") != nullptr;      // diagnosis: where the first operations of the window's stream wait
    auto stamp = [&](const char* w) { if (trace) { hipStreamSynchronize(st); fprintf(stderr, "[lba trace] %12.3f   bawin: %s\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(), w); } };
    stamp("before the upload");
    if (n) HIP_TRY(ctx, hipMemcpyAsync(B->d_stage, h, (size_t)n * 40, hipMemcpyHostToDevice, st));
    stamp("upload done");
    hipLaunchKernelGGL(k_bawin_ingest, dim3((B->cap_n + 255) / 256), dim3(256), 0, st, (const char*)B->d_stage, n, B->cap_n, o, B->d_meas, B->d_xyz, B->d_asso, B->d_trk, B->d_pos, B->d_nfeat + s);
    stamp("ingest kernel done");
    HIP_TRY(ctx, hipGetLastError());
    B->frame_of[s] = frame; B->nfeat[s] = n;
    return VIDO_OK;
}

/* Tracklet-label changes since the last call: quads (frame, feature, tracklet, position) — Map::vnTrkSta / vnPosSta entries that Map::UpdateTracklets wrote.  Quads of
 * frames that have left the ring are ignored. */
int vido_bawin_set_labels(vido_ctx* ctx, int n, const int32_t* quads)
{
    if (!ctx || !ctx->bawin) return VIDO_E_INVALID;
    BaWin* B = ctx->bawin;
    if (n < 0 || (n && !quads)) return vido_set_error(ctx, VIDO_E_INVALID, "bawin_set_labels: bad arguments");
    hipStream_t st = ctx->stream;
    const int per = (int)((B->stage_cap - 4096) / 16);
    for (int a = 0; a < n; a += per) {
        const int m = std::min(per, n - a);
        HIP_TRY(ctx, hipStreamSynchronize(st));
        int* h = (int*)B->h_stage; int k = 0;
        for (int i = 0; i < m; i++) {
            const int32_t* q = quads + 4 * (size_t)(a + i);
            if (q[0] < 0 || B->frame_of[q[0] % B->cap_f] != q[0] || q[1] < 0 || q[1] >= B->nfeat[q[0] % B->cap_f]) continue;
            h[4 * k] = q[0]; h[4 * k + 1] = q[1]; h[4 * k + 2] = q[2]; h[4 * k + 3] = q[3]; k++;
        }
        if (!k) continue;
        HIP_TRY(ctx, hipMemcpyAsync(B->d_stage, h, (size_t)k * 16, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_bawin_labels, dim3((k + 255) / 256), dim3(256), 0, st, (const int4*)B->d_stage, k, B->cap_f, B->cap_n, B->d_trk, B->d_pos);
    }
    HIP_TRY(ctx, hipGetLastError());
    return VIDO_OK;
}

/* PartialBatchOptimization over the frames [start, N) of the ring.  prob: the cameras (cam_T, n_cam = N - start), the odometry factors, prior, weights and LM parameters as
 * for vido_ba_optimize; its observation / point fields are ignored (they come from the ring).  On return prob->cam_T holds the refined cameras, the refined landmarks have
 * been written into the ring's xyz rows; n_obs_out / n_pt_out: size of the graph that was solved. */
int vido_bawin_solve(vido_ctx* ctx, int start, int N, vido_ba_problem* prob, vido_ba_result* res, int32_t* n_obs_out, int32_t* n_pt_out)
{
    if (!ctx || !ctx->bawin) return VIDO_E_INVALID;
    BaWin* B = ctx->bawin;
    if (!prob || !res || start < 0 || N <= start || N - start >= B->cap_f || prob->n_cam != N - start) return vido_set_error(ctx, VIDO_E_INVALID, "bawin_solve: bad window [%d, %d)", start, N);
    for (int f = start; f < N; f++) if (B->frame_of[f % B->cap_f] != f) return vido_set_error(ctx, VIDO_E_INVALID, "bawin_solve: frame %d is not in the ring", f);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    BaWinDev W{B->cap_f, B->cap_n, B->cap_obs, B->cap_pt, B->d_meas, B->d_xyz, B->d_asso, B->d_trk, B->d_pos, B->d_pid, B->d_nfeat,
               B->d_obs_cam, B->d_obs_pt, B->d_obs_pos, B->d_obs_src, B->d_pt_start, B->d_slot_cam, B->d_cnt, B->d_first, B->d_obs_meas, B->d_pt};
    size_t n_feat_window = 0; for (int f = start; f < N; f++) n_feat_window += (size_t)B->nfeat[f % B->cap_f];      // an upper bound of the landmark / observation counts
    const size_t bound = std::min(n_feat_window, (size_t)B->cap_obs);
    hipLaunchKernelGGL(k_bawin_assemble, dim3(1), dim3(1024), 0, st, W, start, N, B->d_counts);
    hipLaunchKernelGGL(k_bawin_slots, dim3((unsigned)std::max<size_t>(1, (bound + 255) / 256)), dim3(256), 0, st, W, (const int*)B->d_counts, (int)std::min(n_feat_window, (size_t)B->cap_pt));
    HIP_TRY(ctx, hipMemcpyAsync(B->h_counts, B->d_counts, 16, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    const int no = B->h_counts[0], np = B->h_counts[1];
    if (B->h_counts[2] & 2) return vido_set_error(ctx, VIDO_E_INVALID, "bawin_solve: a landmark is observed twice from one frame, or its chain skips a frame (obs %d, landmarks %d)", no, np);
    if (B->h_counts[2]) return vido_set_error(ctx, VIDO_E_CAPACITY, "bawin_solve: window exceeds the ring's capacity (obs %d, landmarks %d)", no, np);
    if (n_obs_out) *n_obs_out = no;
    if (n_pt_out) *n_pt_out = np;
    B->last_start = start; B->last_N = N; B->last_nobs = no;
    if (np == 0) { memset(res, 0, sizeof *res); return VIDO_OK; }
    BaDevInputs DI{no, np, std::max(1, std::min(B->h_counts[3], N - start)), B->d_obs_cam, B->d_obs_pt, B->d_obs_pos, B->d_pt_start, B->d_slot_cam, B->d_obs_meas, B->d_pt};
    int rc = ba_run_device_inputs(ctx, prob, res, &DI); if (rc) return rc;
    hipLaunchKernelGGL(k_bawin_writeback, dim3((no + 255) / 256), dim3(256), 0, st, (const int*)B->d_obs_src, (const int*)B->d_obs_pt, (const double*)B->d_pt, no, B->d_xyz);
    HIP_TRY(ctx, hipGetLastError());
    return VIDO_OK;
}

/* The world points of one stored frame (the Map's vp3DPointSta row as the solves have left it): xyz_out [n][3] f32. */
int vido_bawin_read_points(vido_ctx* ctx, int frame, int n, float* xyz_out)
{
    if (!ctx || !ctx->bawin) return VIDO_E_INVALID;
    BaWin* B = ctx->bawin;
    if (frame < 0 || B->frame_of[frame % B->cap_f] != frame || n < 0 || n > B->nfeat[frame % B->cap_f] || (n && !xyz_out)) return vido_set_error(ctx, VIDO_E_INVALID, "bawin_read_points: frame %d is not in the ring", frame);
    if (!n) return VIDO_OK;
    HIP_TRY(ctx, hipMemcpyAsync(xyz_out, B->d_xyz + 3 * (size_t)(frame % B->cap_f) * B->cap_n, (size_t)n * 12, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return VIDO_OK;
}

}  // extern "C"
