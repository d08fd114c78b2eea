// track.hip — data-parallel stages of the per-frame tracking front-end on gfx950.
// Replaces (reference, vido_slam/src): Tracking.cc:299-322 depth pre-scale; Frame.cc:72-100,165-177
// static-candidate filter + depth gather; Frame.cc:184-211 dense object sampling; Tracking.cc:369-421
// cross-frame gathers; Tracking.cc:3291-3357 UpdateMask; Frame.cc:706-771 back-projection;
// Tracking.cc:1582-1668 scene flow.
// Layout: per frame slot, depth (f32), flow (f32x2) and mask (i32) maps stay resident in HBM between
// the calls of one frame and across the frame k / k-1 pair; the list outputs are compacted in the
// reference's visiting order with wave ballots + one LDS scan per 1024-thread workgroup (the lists are
// order-sensitive: index i of frame k-1's correspondences IS feature i of frame k).
#include "common.hpp"

struct TrackState {
    int W = 0, H = 0, B = 0, max_kp = 0, max_obj = 0;
    float *d_depth = nullptr, *d_flow = nullptr; int32_t* d_mask = nullptr;      // [slots][H*W]
    vido_keypoint* d_kps = nullptr;                                              // [B][max_kp]
    int32_t *d_sidx = nullptr, *d_nstat = nullptr, *d_nobj = nullptr, *d_olabel = nullptr;
    float *d_scorr = nullptr, *d_sflow = nullptr, *d_sdepth = nullptr;
    float *d_okeys = nullptr, *d_ocorr = nullptr, *d_odepth = nullptr, *d_oflow = nullptr;
    float* d_tmpf = nullptr; int32_t* d_tmpi = nullptr; size_t tmp_cap = 0;       // scratch for gathers
    bool zc_io = true;                                                            // the small per-point calls run their kernels on h_io itself (no copies); VIDO_TRACK_IO_COPIES=1: through d_tmpf
    float* h_io = nullptr;                                                        // pinned mirror of d_tmpf, word for word: a call packs its inputs there (ONE copy up) and reads its outputs there (ONE copy down)
    int32_t* h_cnt = nullptr;
    char* h_stage = nullptr; size_t stage_cap = 0;
    char* h_maps = nullptr; size_t maps_cap = 0;          // pinned stage of host-resident depth / flow / mask (vido_frontend_batch, maps_on_device = 0)
    char* h_view = nullptr; size_t view_cap = 0;         // pinned lists of the fused front end (vido_frontend_batch)
    // where the maps of each slot live: the ctx's own buffers, or (zero-copy batches) the caller's device memory
    std::vector<float*> sdepth, sflow; std::vector<int32_t*> smask;
    hipEvent_t ev_maps = nullptr, ev_cnt = nullptr;
};

// ---- kernels -------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_depth_prescale(float* __restrict__ d, size_t n4, int mode, float factor, float bf, float scale)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = ((float4*)d)[i];
        float* e = (float*)&v;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            float x = e[k];
            if (x < 0) x = 0;
            else if (mode == 0) x = x / factor;
            else if (mode == 1) x = bf / (x / factor);
            else x = scale * bf / (x / factor);
            e[k] = x;
        }
        ((float4*)d)[i] = v;
    }
}

// exclusive position of a flagged element among all flagged elements of the workgroup so far, in thread
// order; `base` carries the running total across loop iterations (uniform).
__device__ __forceinline__ int block_ordered_slot(bool flag, int& base, int* wsum /*[17]*/)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const unsigned long long b = __ballot(flag);
    const int within = __popcll(b & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wave] = __popcll(b);
    __syncthreads();
    int before = 0, total = 0;
    for (int w = 0; w < nw; w++) { const int c = wsum[w]; if (w < wave) before += c; total += c; }
    __syncthreads();
    const int pos = base + before + within;
    base += total;
    return pos;
}

// Frame.cc:72-100 + :165-177.  One workgroup per frame.
__global__ __launch_bounds__(1024) void k_static_filter(const vido_keypoint* __restrict__ kps, const int32_t* __restrict__ n_kps, int kp_pitch, int max_kp,
                                                        const float* __restrict__ depth, const float* __restrict__ flow, const int32_t* __restrict__ mask,
                                                        int w, int h, float th_depth,
                                                        int32_t* __restrict__ out_idx, float* __restrict__ out_corr, float* __restrict__ out_flow,
                                                        float* __restrict__ out_depth, int32_t* __restrict__ n_out)
{
    __shared__ int wsum[17];
    const int f = blockIdx.x;
    const size_t px = (size_t)w * h;
    const vido_keypoint* K = kps + (size_t)f * kp_pitch;
    const float* D = depth + f * px; const float* F = flow + f * px * 2; const int32_t* M = mask + f * px;
    const int n = n_kps[f];
    int base = 0;
    for (int i0 = 0; i0 < n; i0 += blockDim.x) {
        const int i = i0 + threadIdx.x;
        bool keep = false; float kx = 0, ky = 0, fx = 0, fy = 0, dd = 0;
        if (i < n) {
            kx = K[i].x; ky = K[i].y;
            const size_t p = (size_t)(int)ky * w + (int)kx;
            dd = D[p];
            if (M[p] == 0 && !(dd > th_depth || dd <= 0)) {
                fx = F[2 * p]; fy = F[2 * p + 1];
                keep = (fx != 0 && fy != 0) && (kx + fx < (float)w && ky + fy < (float)h && kx < (float)w && ky < (float)h);
            }
        }
        const int pos = block_ordered_slot(keep, base, wsum);
        if (keep) {
            const size_t o = (size_t)f * max_kp + pos;
            out_idx[o] = i; out_corr[2 * o] = kx + fx; out_corr[2 * o + 1] = ky + fy;
            out_flow[2 * o] = fx; out_flow[2 * o + 1] = fy; out_depth[o] = dd > 0 ? dd : -1.f;
        }
    }
    if (threadIdx.x == 0) n_out[f] = base;
}

// Frame.cc:184-211.  One workgroup per frame, lattice visited row-major.
__global__ __launch_bounds__(1024) void k_dense_sample(const float* __restrict__ depth, const float* __restrict__ flow, const int32_t* __restrict__ mask,
                                                       int w, int h, float th_obj, int step, int max_obj,
                                                       float* __restrict__ keys, float* __restrict__ corr, float* __restrict__ odepth,
                                                       int32_t* __restrict__ label, float* __restrict__ oflow, int32_t* __restrict__ n_out)
{
    __shared__ int wsum[17];
    const int f = blockIdx.x;
    const size_t px = (size_t)w * h;
    const float* D = depth + f * px; const float* F = flow + f * px * 2; const int32_t* M = mask + f * px;
    const int gw = (w + step - 1) / step, gh = (h + step - 1) / step, n = gw * gh;
    int base = 0;
    for (int q0 = 0; q0 < n; q0 += blockDim.x) {
        const int q = q0 + threadIdx.x;
        bool keep = false; int i = 0, j = 0, lab = 0; float fx = 0, fy = 0, dd = 0;
        if (q < n) {
            i = (q / gw) * step; j = (q % gw) * step;
            const size_t p = (size_t)i * w + j;
            lab = M[p]; dd = D[p];
            if (lab != 0 && dd < th_obj && dd > 0) {
                fx = F[2 * p]; fy = F[2 * p + 1];
                keep = (j + fx < (float)w && j + fx > 0 && i + fy < (float)h && i + fy > 0);
            }
        }
        const int pos = block_ordered_slot(keep, base, wsum);
        if (keep && pos < max_obj) {
            const size_t o = (size_t)f * max_obj + pos;
            keys[2 * o] = (float)j; keys[2 * o + 1] = (float)i; corr[2 * o] = j + fx; corr[2 * o + 1] = i + fy;
            odepth[o] = dd; label[o] = lab; oflow[2 * o] = fx; oflow[2 * o + 1] = fy;
        }
    }
    if (threadIdx.x == 0) n_out[f] = base;
}

// Tracking.cc:369-391 / :398-421
__global__ void k_gather_static(const float* __restrict__ keys, int n, const float* __restrict__ depth, int w, int h, float* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int u = (int)keys[2 * i], v = (int)keys[2 * i + 1];
    float r = -1.f;
    if (u < (w - 1) && u > 0 && v < (h - 1) && v > 0) { const float d = depth[(size_t)v * w + u]; if (d > 0) r = d; }
    out[i] = r;
}
__global__ void k_gather_object(const float* __restrict__ keys, int n, const float* __restrict__ depth, const int32_t* __restrict__ mask,
                                int w, int h, float th_obj, float* __restrict__ out_d, int32_t* __restrict__ out_l)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int u = (int)keys[2 * i], v = (int)keys[2 * i + 1];
    float d = 0.1f; int l = 0;
    if (u < (w - 1) && u > 0 && v < (h - 1) && v > 0) {
        const float dd = depth[(size_t)v * w + u];
        if (dd < th_obj && dd > 0) { d = dd; l = mask[(size_t)v * w + u]; }
    }
    out_d[i] = d; out_l[i] = l;
}
// mask / depth / flow at ((int)x, (int)y) of a point list: what the host-side renew stages (trackhost.cpp: vido_renew_*_sampled) read from the maps
__global__ void k_gather_samples(const float* __restrict__ keys, int n, const float* __restrict__ depth, const float* __restrict__ flow, const int32_t* __restrict__ mask,
                                 int w, int h, float* __restrict__ out_d, float* __restrict__ out_f, int32_t* __restrict__ out_m)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int u = (int)keys[2 * i], v = (int)keys[2 * i + 1];
    float d = 0.f, fx = 0.f, fy = 0.f; int m = 0;
    if (u < w && u >= 0 && v < h && v >= 0) { const size_t o = (size_t)v * w + u; d = depth[o]; m = mask[o]; fx = flow[2 * o]; fy = flow[2 * o + 1]; }
    out_d[i] = d; out_f[2 * i] = fx; out_f[2 * i + 1] = fy; out_m[i] = m;
}
// UpdateMask helpers: labels of the current mask at the propagated points; scatter of one lost label
__global__ void k_mask_at(const float* __restrict__ corr, int n, const int32_t* __restrict__ mask, int w, int h, int32_t* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int u = (int)corr[2 * i], v = (int)corr[2 * i + 1];
    out[i] = (u < w && u > 0 && v < h && v > 0) ? mask[(size_t)v * w + u] : INT32_MIN;
}
__global__ __launch_bounds__(256) void k_mask_scatter(const int32_t* __restrict__ mask_last, const float* __restrict__ flow_last,
                                                      int32_t* __restrict__ mask_cur, int w, int h, int label)
{
    const size_t n = (size_t)w * h;
    for (size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x; p < n; p += (size_t)gridDim.x * blockDim.x) {
        if (mask_last[p] != label) continue;
        const int j = (int)(p / w), k = (int)(p % w);
        const int fx = (int)flow_last[2 * p], fy = (int)flow_last[2 * p + 1];
        if (k + fx < w && k + fx > 0 && j + fy < h && j + fy > 0) mask_cur[(size_t)(j + fy) * w + (k + fx)] = label;   // same value from every writer
    }
}
// Frame.cc:706-771 (addnoise = 0): camera -> world back-projection; cv::Mat float products accumulate in double
__global__ void k_unproject_world(const float* __restrict__ keys, const float* __restrict__ z, int n, float cx, float cy, float invfx, float invfy,
                                  const float* __restrict__ RT /* Rwl[9] twl[3] */, float* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float zz = z[i];
    if (!(zz > 0)) { out[3 * i] = out[3 * i + 1] = out[3 * i + 2] = 0; return; }
    const float x = (keys[2 * i] - cx) * zz * invfx, y = (keys[2 * i + 1] - cy) * zz * invfy;
#pragma unroll
    for (int r = 0; r < 3; r++) {
        const double s = (double)RT[r * 3] * x + (double)RT[r * 3 + 1] * y + (double)RT[r * 3 + 2] * zz;
        out[3 * i + r] = (float)s + RT[9 + r];
    }
}
__global__ void k_scene_flow(const float* __restrict__ Xl, const float* __restrict__ Xc, const int32_t* __restrict__ sl, const int32_t* __restrict__ sc,
                             int n, float* __restrict__ flow3d, int32_t* __restrict__ obj_label)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (sc[i] <= 0 || sl[i] <= 0) { obj_label[i] = -1; flow3d[3 * i] = flow3d[3 * i + 1] = flow3d[3 * i + 2] = 0; return; }
#pragma unroll
    for (int r = 0; r < 3; r++) flow3d[3 * i + r] = Xc[3 * i + r] - Xl[3 * i + r];
}

// ---- host ------------------------------------------------------------------------------------------
static int track_state(vido_ctx* ctx, TrackState** out)
{
    if (ctx->trk) { *out = ctx->trk; return VIDO_OK; }
    TrackState* T = new TrackState();
    ctx->trk = T;
    T->W = ctx->cfg.width; T->H = ctx->cfg.height; T->B = std::max(ctx->cfg.max_batch, 2);    // >= 2 slots: frame k and k-1
    T->max_kp = ctx->cfg.n_features * 2 + 256;
    T->max_obj = ((T->W + 3) / 4) * ((T->H + 3) / 4);
    const size_t px = (size_t)T->W * T->H, B = T->B;
    HIP_TRY(ctx, hipMalloc(&T->d_depth, B * px * 4)); HIP_TRY(ctx, hipMalloc(&T->d_flow, B * px * 8)); HIP_TRY(ctx, hipMalloc(&T->d_mask, B * px * 4));
    HIP_TRY(ctx, hipMalloc(&T->d_kps, B * T->max_kp * sizeof(vido_keypoint)));
    HIP_TRY(ctx, hipMalloc(&T->d_sidx, B * T->max_kp * 4)); HIP_TRY(ctx, hipMalloc(&T->d_scorr, B * T->max_kp * 8));
    HIP_TRY(ctx, hipMalloc(&T->d_sflow, B * T->max_kp * 8)); HIP_TRY(ctx, hipMalloc(&T->d_sdepth, B * T->max_kp * 4));
    HIP_TRY(ctx, hipMalloc(&T->d_nstat, B * 4)); HIP_TRY(ctx, hipMalloc(&T->d_nobj, B * 4));
    HIP_TRY(ctx, hipMalloc(&T->d_okeys, B * T->max_obj * 8)); HIP_TRY(ctx, hipMalloc(&T->d_ocorr, B * T->max_obj * 8));
    HIP_TRY(ctx, hipMalloc(&T->d_odepth, B * T->max_obj * 4)); HIP_TRY(ctx, hipMalloc(&T->d_olabel, B * T->max_obj * 4));
    HIP_TRY(ctx, hipMalloc(&T->d_oflow, B * T->max_obj * 8));
    T->tmp_cap = (size_t)std::max(T->max_obj, T->max_kp) * 12 + 16;      // the widest call layout (scene_flow: 12 words per point) for every n up to max_obj / max_kp
    HIP_TRY(ctx, hipMalloc(&T->d_tmpf, T->tmp_cap * 4)); HIP_TRY(ctx, hipMalloc(&T->d_tmpi, T->tmp_cap * 4));
    HIP_TRY(ctx, hipHostMalloc((void**)&T->h_io, T->tmp_cap * 4 + (size_t)T->max_kp * sizeof(vido_keypoint) + 256));
    T->zc_io = getenv("VIDO_TRACK_IO_COPIES") == nullptr;
    HIP_TRY(ctx, hipHostMalloc(&T->h_cnt, 2 * B * 4));
    T->sdepth.resize(B); T->sflow.resize(B); T->smask.resize(B);
    for (size_t b = 0; b < B; b++) { T->sdepth[b] = T->d_depth + b * px; T->sflow[b] = T->d_flow + b * px * 2; T->smask[b] = T->d_mask + b * px; }
    *out = T;
    return VIDO_OK;
}

void track_state_destroy(vido_ctx* ctx)
{
    TrackState* T = ctx->trk;
    if (!T) return;
    hipFree(T->d_depth); hipFree(T->d_flow); hipFree(T->d_mask); hipFree(T->d_kps); hipFree(T->d_sidx); hipFree(T->d_scorr); hipFree(T->d_sflow);
    hipFree(T->d_sdepth); hipFree(T->d_nstat); hipFree(T->d_nobj); hipFree(T->d_okeys); hipFree(T->d_ocorr); hipFree(T->d_odepth); hipFree(T->d_olabel);
    hipFree(T->d_oflow); hipFree(T->d_tmpf); hipFree(T->d_tmpi); hipHostFree(T->h_cnt); hipHostFree(T->h_io); hipHostFree(T->h_stage); hipHostFree(T->h_view); hipHostFree(T->h_maps);
    if (T->ev_maps) hipEventDestroy(T->ev_maps);
    if (T->ev_cnt) hipEventDestroy(T->ev_cnt);
    delete T; ctx->trk = nullptr;
}

static inline hipMemcpyKind in_kind(int on_device) { return on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice; }

__global__ __launch_bounds__(256) void k_diag_lds(int* out) { __shared__ int pad[8192]; pad[threadIdx.x] = threadIdx.x; __syncthreads(); if (threadIdx.x == 0 && blockIdx.x == 0) out[1] = pad[255]; }
extern "C" {

/* diagnosis (VIDO_DIAG_FIRSTOP): `rounds` x (a 4-byte memset on the context's stream + a host wait), each timed; the means per round are printed when the process exits */
int vido_debug_first_op(vido_ctx* ctx, int rounds)
{
    if (!ctx) return VIDO_E_INVALID;
    static double acc[8] = {0}; static long cnt = 0; static int* dword = nullptr;
    struct Printer { ~Printer() { if (cnt) for (int i = 0; i < 8; i++) if (acc[i] > 0) fprintf(stderr, "[diag first-op] round %d: %.3f ms mean over %ld frames\n", i, acc[i] / cnt, cnt); } };
    static Printer pr;
    if (!dword) HIP_TRY(ctx, hipMalloc((void**)&dword, 64));
    for (int i = 0; i < std::min(rounds, 8); i++) {
        const auto t0 = std::chrono::steady_clock::now();
        // rounds 0, 1: a 4-byte memset; 2, 3: a 16 KB device -> pinned-host copy; 4, 5: a 16 KB pinned-host -> device copy; 6, 7: a 64-workgroup kernel with 32 KB of LDS each
        TrackState* T = ctx->trk;
        if (i < 2 || !T) HIP_TRY(ctx, hipMemsetAsync(dword, 0, 4, ctx->stream));
        else if (i < 4) HIP_TRY(ctx, hipMemcpyAsync(T->h_io, T->d_tmpf, 16384, hipMemcpyDeviceToHost, ctx->stream));
        else if (i < 6) HIP_TRY(ctx, hipMemcpyAsync(T->d_tmpf, T->h_io, 16384, hipMemcpyHostToDevice, ctx->stream));
        else hipLaunchKernelGGL(k_diag_lds, dim3(64), dim3(256), 0, ctx->stream, dword);
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        acc[i] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    cnt++;
    return VIDO_OK;
}

int vido_track_slots(vido_ctx* ctx)
{
    TrackState* T; if (!ctx) return VIDO_E_INVALID;
    int rc = track_state(ctx, &T); if (rc) return rc;
    return T->B;
}

int vido_frame_upload(vido_ctx* ctx, int slot0, int n_frames, float* depth, const float* flow, const int32_t* mask, int on_device,
                      const vido_track_params* p)
{
    if (!ctx || !p) return VIDO_E_INVALID;
    TrackState* T; int rc = track_state(ctx, &T); if (rc) return rc;
    if (slot0 < 0 || n_frames < 1 || slot0 + n_frames > T->B || !depth || !flow || !mask)
        return vido_set_error(ctx, VIDO_E_INVALID, "frame_upload: slots [%d,%d) outside [0,%d) or null map", slot0, slot0 + n_frames, T->B);
    const size_t px = (size_t)T->W * T->H, n = px * n_frames;
    if ((n & 3) != 0) return vido_set_error(ctx, VIDO_E_INVALID, "frame_upload: width*height must be a multiple of 4");      // every argument is validated before any state changes
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    if (on_device == 2) {
        // zero-copy: the slots REFER to the caller's device buffers (the reference keeps shallow references to the caller's Mats, Tracking.cc:343-345, and rescales the depth in
        // the caller's buffer, :299-322).  No copy in either direction: 6.1 MB of device-to-device copies per 640 x 480 frame, which beside saturating network kernels cost the
        // tracker's first synchronisation ~1.1 ms (profiles/r6/tracker_zero_copy_maps.txt).  The caller keeps a frame's maps alive and untouched while the slot is in use (the
        // tracker reads the current and the previous frame).
        for (int f = 0; f < n_frames; f++) { T->sdepth[slot0 + f] = depth + (size_t)f * px; T->sflow[slot0 + f] = (float*)flow + (size_t)f * px * 2; T->smask[slot0 + f] = (int32_t*)mask + (size_t)f * px; }
        hipLaunchKernelGGL(k_depth_prescale, dim3((int)std::min<size_t>((n / 4 + 255) / 256, 2048)), dim3(256), 0, st, depth, n / 4, p->dataset, p->depth_map_factor, p->bf, p->kaist_scale);
        HIP_TRY(ctx, hipGetLastError());
        return VIDO_OK;
    }
    float* dd = T->d_depth + slot0 * px;
    HIP_TRY(ctx, hipMemcpyAsync(dd, depth, n * 4, in_kind(on_device), st));
    HIP_TRY(ctx, hipMemcpyAsync(T->d_flow + slot0 * px * 2, flow, n * 8, in_kind(on_device), st));
    HIP_TRY(ctx, hipMemcpyAsync(T->d_mask + slot0 * px, mask, n * 4, in_kind(on_device), st));
    // the slot tables are repointed only once the copies have been accepted (a failed enqueue leaves the previous frame's slots referenced)
    for (int f = 0; f < n_frames; f++) { T->sdepth[slot0 + f] = T->d_depth + (slot0 + f) * px; T->sflow[slot0 + f] = T->d_flow + (slot0 + f) * px * 2; T->smask[slot0 + f] = T->d_mask + (slot0 + f) * px; }
    const int grid = (int)std::min<size_t>((n / 4 + 255) / 256, 2048);
    hipLaunchKernelGGL(k_depth_prescale, dim3(grid), dim3(256), 0, st, dd, n / 4, p->dataset, p->depth_map_factor, p->bf, p->kaist_scale);
    // the reference mutates the caller's depth buffer in place (Tracking.cc:299-322): hand the scaled map back
    HIP_TRY(ctx, hipMemcpyAsync(depth, dd, n * 4, on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, st));
    if (!on_device) HIP_TRY(ctx, hipStreamSynchronize(st));
    HIP_TRY(ctx, hipGetLastError());
    return VIDO_OK;
}

// threads of the one-workgroup-per-frame list kernels.  (VIDO_LISTS_NT: experiment switch)
static int lists_nt() { static const int v = [] { const char* e = getenv("VIDO_LISTS_NT"); const int t = e ? atoi(e) : 1024; return (t == 64 || t == 128 || t == 256 || t == 512) ? t : 1024; }(); return v; }

int vido_frame_features(vido_ctx* ctx, int slot0, int n_frames, const vido_keypoint* kps, const int32_t* n_kps, int max_kp,
                        const vido_track_params* p, vido_frame_lists* out)
{
    if (!ctx || !p || !out || !kps || !n_kps) return VIDO_E_INVALID;
    TrackState* T; int rc = track_state(ctx, &T); if (rc) return rc;
    if (slot0 < 0 || n_frames < 1 || slot0 + n_frames > T->B || max_kp > T->max_kp || max_kp < 1)
        return vido_set_error(ctx, VIDO_E_INVALID, "frame_features: bad slot range or max_kp (%d > %d)", max_kp, T->max_kp);
    if (out->max_stat < max_kp || out->max_obj < 1) return vido_set_error(ctx, VIDO_E_INVALID, "frame_features: output capacity too small");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const size_t px = (size_t)T->W * T->H;
    for (int f = 0; f < n_frames; f++) if (n_kps[f] < 0 || n_kps[f] > max_kp) return vido_set_error(ctx, VIDO_E_INVALID, "frame_features: n_kps[%d]=%d", f, n_kps[f]);
    // keypoints are packed to the ctx's own pitch
    // (through the pinned stage: an asynchronous copy FROM PAGEABLE memory of this size — 100 KB of keypoints — took 3 ms inside the pipeline, 0.04 ms on an idle GPU; the runtime
    //  pins such a range on the fly and that waits on the busy device.  Only the n_kps[f] rows that exist go up.)
    { VidoProfScope ps("frame_features: uploads (keypoints + counts, pinned stage)", st, true);
    char* hk = (char*)T->h_io + T->tmp_cap * 4;                                    // [max_kp] keypoints of ONE frame at a time, then the counts
    if (n_frames == 1) {
        memcpy(hk, kps, (size_t)n_kps[0] * sizeof(vido_keypoint));
        int32_t* hc = (int32_t*)(hk + (size_t)T->max_kp * sizeof(vido_keypoint)); hc[0] = n_kps[0];
        if (n_kps[0]) HIP_TRY(ctx, hipMemcpyAsync(T->d_kps, hk, (size_t)n_kps[0] * sizeof(vido_keypoint), hipMemcpyHostToDevice, st));
        HIP_TRY(ctx, hipMemcpyAsync(T->d_nstat, hc, 4, hipMemcpyHostToDevice, st));     // reused as the input count, overwritten by the kernel
        HIP_TRY(ctx, hipMemcpyAsync(T->d_nobj, hc, 4, hipMemcpyHostToDevice, st));
    } else {
    HIP_TRY(ctx, hipMemcpy2DAsync(T->d_kps, (size_t)T->max_kp * sizeof(vido_keypoint), kps, (size_t)max_kp * sizeof(vido_keypoint),
                                  (size_t)max_kp * sizeof(vido_keypoint), n_frames, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(T->d_nstat, n_kps, n_frames * 4, hipMemcpyHostToDevice, st));     // reused as the input count, overwritten by the kernel
    HIP_TRY(ctx, hipMemcpyAsync(T->d_nobj, n_kps, n_frames * 4, hipMemcpyHostToDevice, st));
    } }
    for (int f = 1; f < n_frames; f++) if (T->sdepth[slot0 + f] != T->sdepth[slot0] + f * px) return vido_set_error(ctx, VIDO_E_INVALID, "frame_features: slots [%d,%d) do not hold one contiguous batch", slot0, slot0 + n_frames);
    { VidoProfScope ps("frame_features: k_static_filter", st, true);
    hipLaunchKernelGGL(k_static_filter, dim3(n_frames), dim3(lists_nt()), 0, st, T->d_kps, T->d_nobj, T->max_kp, T->max_kp,
                       T->sdepth[slot0], T->sflow[slot0], T->smask[slot0], T->W, T->H, p->th_depth_bg,
                       T->d_sidx, T->d_scorr, T->d_sflow, T->d_sdepth, T->d_nstat); }
    const int step = p->dense_step > 0 ? p->dense_step : 4;
    const int lattice = ((T->W + step - 1) / step) * ((T->H + step - 1) / step);
    if (lattice > T->max_obj) return vido_set_error(ctx, VIDO_E_INVALID, "frame_features: dense_step %d gives %d probes > %d", step, lattice, T->max_obj);
    { VidoProfScope ps("frame_features: k_dense_sample", st, true);
    hipLaunchKernelGGL(k_dense_sample, dim3(n_frames), dim3(lists_nt()), 0, st, T->sdepth[slot0], T->sflow[slot0], T->smask[slot0],
                       T->W, T->H, p->th_depth_obj, step, T->max_obj, T->d_okeys, T->d_ocorr, T->d_odepth, T->d_olabel, T->d_oflow, T->d_nobj); }
    { VidoProfScope ps("frame_features: counts download + wait", st, false);
    HIP_TRY(ctx, hipMemcpyAsync(T->h_cnt, T->d_nstat, n_frames * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipMemcpyAsync(T->h_cnt + T->B, T->d_nobj, n_frames * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st)); }
    HIP_TRY(ctx, hipGetLastError());
    int max_ns = 0, max_no = 0;
    for (int f = 0; f < n_frames; f++) {
        const int ns = T->h_cnt[f], no = T->h_cnt[T->B + f];
        out->n_stat[f] = ns; out->n_obj[f] = no;
        if (no > out->max_obj) return vido_set_error(ctx, VIDO_E_CAPACITY, "frame_features: %d object samples > max_obj %d", no, out->max_obj);
        max_ns = std::max(max_ns, ns); max_no = std::max(max_no, no);
    }
    // one strided copy per list (rows = frames, width = longest list) into the pinned stage, then row memcpys
    // into the caller's (pageable) arrays — nine device copies per call instead of nine per frame
    const size_t need = (size_t)n_frames * ((size_t)max_ns * 24 + (size_t)max_no * 32) + 4096;
    if (need > T->stage_cap) {
        if (T->h_stage) HIP_TRY(ctx, hipHostFree(T->h_stage));
        T->stage_cap = need + need / 2; HIP_TRY(ctx, hipHostMalloc((void**)&T->h_stage, T->stage_cap));
    }
    struct Seg { void* dst; size_t dpitch; char* stage; size_t el; int width; };
    std::vector<Seg> segs; char* cur = T->h_stage;
    auto copy2d = [&](void* dst, size_t dpitch_el, const void* src, size_t spitch_el, size_t el, int width_el) -> int {
        if (width_el <= 0) return VIDO_OK;
        if (n_frames == 1) HIP_TRY(ctx, hipMemcpyAsync(cur, src, (size_t)width_el * el, hipMemcpyDeviceToHost, st));      // (one frame: a plain copy — see orb_collect)
        else HIP_TRY(ctx, hipMemcpy2DAsync(cur, (size_t)width_el * el, src, spitch_el * el, (size_t)width_el * el, n_frames, hipMemcpyDeviceToHost, st));
        segs.push_back(Seg{dst, dpitch_el * el, cur, el, width_el}); cur += (size_t)n_frames * width_el * el;
        return VIDO_OK;
    };
    VidoProfScope ps_dl("frame_features: 9 list downloads + wait + host copies", st, false);
    if ((rc = copy2d(out->stat_idx, out->max_stat, T->d_sidx, T->max_kp, 4, max_ns))) return rc;
    if ((rc = copy2d(out->stat_corr, out->max_stat, T->d_scorr, T->max_kp, 8, max_ns))) return rc;
    if ((rc = copy2d(out->stat_flow, out->max_stat, T->d_sflow, T->max_kp, 8, max_ns))) return rc;
    if ((rc = copy2d(out->stat_depth, out->max_stat, T->d_sdepth, T->max_kp, 4, max_ns))) return rc;
    if ((rc = copy2d(out->obj_keys, out->max_obj, T->d_okeys, T->max_obj, 8, max_no))) return rc;
    if ((rc = copy2d(out->obj_corr, out->max_obj, T->d_ocorr, T->max_obj, 8, max_no))) return rc;
    if ((rc = copy2d(out->obj_depth, out->max_obj, T->d_odepth, T->max_obj, 4, max_no))) return rc;
    if ((rc = copy2d(out->obj_label, out->max_obj, T->d_olabel, T->max_obj, 4, max_no))) return rc;
    if ((rc = copy2d(out->obj_flow, out->max_obj, T->d_oflow, T->max_obj, 8, max_no))) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(st));
    for (const Seg& g : segs)
        for (int f = 0; f < n_frames; f++) {
            const int cnt = (g.dst == out->stat_idx || g.dst == out->stat_corr || g.dst == out->stat_flow || g.dst == out->stat_depth) ? out->n_stat[f] : out->n_obj[f];
            memcpy((char*)g.dst + (size_t)f * g.dpitch, g.stage + (size_t)f * g.width * g.el, (size_t)cnt * g.el);
        }
    return VIDO_OK;
}

/* Fused per-frame front end for a batch: ORB extraction + depth pre-scale + Frame::Frame lists in ONE stream of launches with
 * the keypoints handed from the extractor to the static filter on the device (no H2D of keypoints, no intermediate sync),
 * results mirrored into ctx-owned pinned host memory and returned as a view (no copies into caller arrays). */
int vido_frontend_batch(vido_ctx* ctx, const uint8_t* imgs, int imgs_on_device, int n_frames, size_t frame_stride, int stride, int width, int height,
                        float* depth, const float* flow, const int32_t* mask, int maps_on_device, int slot0, const vido_track_params* p,
                        vido_frontend_view* view)
{
    if (!ctx || !p || !view) return VIDO_E_INVALID;
    TrackState* T; int rc = track_state(ctx, &T); if (rc) return rc;
    if (slot0 < 0 || n_frames < 1 || slot0 + n_frames > T->B || !depth || !flow || !mask)
        return vido_set_error(ctx, VIDO_E_INVALID, "frontend_batch: slots [%d,%d) outside [0,%d) or null map", slot0, slot0 + n_frames, T->B);
    hipStream_t st = ctx->stream, st2 = ctx->stream2;
    const size_t px = (size_t)T->W * T->H, n = px * n_frames;
    if ((n & 3) != 0) return vido_set_error(ctx, VIDO_E_INVALID, "frontend_batch: width*height must be a multiple of 4");
    const int step = p->dense_step > 0 ? p->dense_step : 4;
    const int lattice = ((T->W + step - 1) / step) * ((T->H + step - 1) / step);
    if (lattice > T->max_obj) return vido_set_error(ctx, VIDO_E_INVALID, "frontend_batch: dense_step %d gives %d probes > %d", step, lattice, T->max_obj);
    if (!T->ev_maps) { HIP_TRY(ctx, hipEventCreateWithFlags(&T->ev_maps, hipEventDisableTiming)); HIP_TRY(ctx, hipEventCreateWithFlags(&T->ev_cnt, hipEventDisableTiming)); }
    // maps_on_device == 2: zero-copy — the slots REFER to the caller's device buffers (the reference keeps shallow references to the caller's
    // depth/flow/mask Mats until the next frame, Tracking.cc:343-345, and rescales the depth in the caller's buffer); 0/1: copied into the ctx
    const bool alias = maps_on_device == 2;
    float* dd = alias ? depth : T->d_depth + slot0 * px;
    const float* fl = alias ? flow : T->d_flow + slot0 * px * 2; const int32_t* mk = alias ? mask : T->d_mask + slot0 * px;
    for (int f = 0; f < n_frames; f++) { T->sdepth[slot0 + f] = dd + (size_t)f * px; T->sflow[slot0 + f] = (float*)fl + (size_t)f * px * 2; T->smask[slot0 + f] = (int32_t*)mk + (size_t)f * px; }
    // ---- second stream: everything that only needs the maps (pre-scale, dense object sampling) runs next to the extractor
    HIP_TRY(ctx, hipEventRecord(T->ev_maps, st)); HIP_TRY(ctx, hipStreamWaitEvent(st2, T->ev_maps, 0));      // order behind earlier work on the main stream
    // Host-resident maps (the facade's per-frame call): a copy from pageable memory blocks the calling thread for its whole duration, so the
    // extractor is enqueued FIRST and the maps are staged through pinned memory while the GPU is already busy with it; the rescaled depth comes
    // back into the same pinned stage and reaches the caller's buffer after the final synchronisation.
    const bool host_maps = maps_on_device == 0;
    float* hm_depth = nullptr;
    if (host_maps) {
        if ((rc = orb_enqueue(ctx, imgs, imgs_on_device, n_frames, frame_stride, stride, width, height))) return rc;
        if (n * 16 > T->maps_cap) { HIP_TRY(ctx, hipStreamSynchronize(st2)); if (T->h_maps) HIP_TRY(ctx, hipHostFree(T->h_maps)); T->maps_cap = n * 16; HIP_TRY(ctx, hipHostMalloc((void**)&T->h_maps, T->maps_cap)); }
        hm_depth = (float*)T->h_maps; float* hm_flow = hm_depth + n; int32_t* hm_mask = (int32_t*)(hm_flow + 2 * n);
        memcpy(hm_depth, depth, n * 4); memcpy(hm_flow, flow, n * 8); memcpy(hm_mask, mask, n * 4);
        HIP_TRY(ctx, hipMemcpyAsync(dd, hm_depth, n * 4, hipMemcpyHostToDevice, st2));
        HIP_TRY(ctx, hipMemcpyAsync(T->d_flow + slot0 * px * 2, hm_flow, n * 8, hipMemcpyHostToDevice, st2));
        HIP_TRY(ctx, hipMemcpyAsync(T->d_mask + slot0 * px, hm_mask, n * 4, hipMemcpyHostToDevice, st2));
    } else if (!alias) {
        HIP_TRY(ctx, hipMemcpyAsync(dd, depth, n * 4, hipMemcpyDeviceToDevice, st2));
        HIP_TRY(ctx, hipMemcpyAsync(T->d_flow + slot0 * px * 2, flow, n * 8, hipMemcpyDeviceToDevice, st2));
        HIP_TRY(ctx, hipMemcpyAsync(T->d_mask + slot0 * px, mask, n * 4, hipMemcpyDeviceToDevice, st2));
    }
    hipLaunchKernelGGL(k_depth_prescale, dim3((int)std::min<size_t>((n / 4 + 255) / 256, 2048)), dim3(256), 0, st2, dd, n / 4, p->dataset, p->depth_map_factor, p->bf, p->kaist_scale);
    if (host_maps) HIP_TRY(ctx, hipMemcpyAsync(hm_depth, dd, n * 4, hipMemcpyDeviceToHost, st2));                       // in-place semantics of Tracking.cc:299-322 ...
    else if (!alias) HIP_TRY(ctx, hipMemcpyAsync(depth, dd, n * 4, hipMemcpyDeviceToDevice, st2));
    hipLaunchKernelGGL(k_dense_sample, dim3(n_frames), dim3(1024), 0, st2, (const float*)dd, fl, mk,
                       T->W, T->H, p->th_depth_obj, step, T->max_obj, T->d_okeys, T->d_ocorr, T->d_odepth, T->d_olabel, T->d_oflow, T->d_nobj);
    HIP_TRY(ctx, hipMemcpyAsync(T->h_cnt + T->B, T->d_nobj, n_frames * 4, hipMemcpyDeviceToHost, st2));
    HIP_TRY(ctx, hipEventRecord(T->ev_maps, st2));                 // maps rescaled, object samples done
    HIP_TRY(ctx, hipEventRecord(T->ev_cnt, st2));
    // ---- main stream: extractor, then the static filter (needs the keypoints AND the rescaled depth)
    if (!host_maps && (rc = orb_enqueue(ctx, imgs, imgs_on_device, n_frames, frame_stride, stride, width, height))) return rc;
    // object-sample rows go to the host on the second stream while the extractor is still running (their counts are known early)
    HIP_TRY(ctx, hipEventSynchronize(T->ev_cnt));
    const size_t Bv = T->B, need = Bv * ((size_t)T->max_kp * 24 + (size_t)T->max_obj * 32);
    if (need > T->view_cap) { HIP_TRY(ctx, hipStreamSynchronize(st)); HIP_TRY(ctx, hipStreamSynchronize(st2)); if (T->h_view) HIP_TRY(ctx, hipHostFree(T->h_view)); T->view_cap = need; HIP_TRY(ctx, hipHostMalloc((void**)&T->h_view, T->view_cap)); }
    int max_no = 0;
    for (int f = 0; f < n_frames; f++) max_no = std::max(max_no, T->h_cnt[T->B + f]);
    char* cur = T->h_view;
    auto rows = [&](hipStream_t s_, const void* src, size_t pitch_el, size_t el, int width_el) -> char* {
        char* dst = cur; cur += Bv * pitch_el * el;
        if (width_el > 0) hipMemcpy2DAsync(dst, pitch_el * el, src, pitch_el * el, (size_t)width_el * el, n_frames, hipMemcpyDeviceToHost, s_);
        return dst;
    };
    view->obj_keys = (const float*)rows(st2, T->d_okeys, T->max_obj, 8, max_no); view->obj_corr = (const float*)rows(st2, T->d_ocorr, T->max_obj, 8, max_no);
    view->obj_depth = (const float*)rows(st2, T->d_odepth, T->max_obj, 4, max_no); view->obj_label = (const int32_t*)rows(st2, T->d_olabel, T->max_obj, 4, max_no);
    view->obj_flow = (const float*)rows(st2, T->d_oflow, T->max_obj, 8, max_no);
    if ((rc = orb_mirror_async(ctx, n_frames))) return rc;       // keypoint / descriptor rows stream to the host while the static filter runs
    const OrbView ov = orb_view(ctx);
    if (ov.row_cap > T->max_kp) return vido_set_error(ctx, VIDO_E_INVALID, "frontend_batch: extractor rows (%d) exceed the list capacity (%d)", ov.row_cap, T->max_kp);
    HIP_TRY(ctx, hipStreamWaitEvent(st, T->ev_maps, 0));
    hipLaunchKernelGGL(k_static_filter, dim3(n_frames), dim3(1024), 0, st, ov.d_kpf, ov.d_nkp, ov.row_cap, T->max_kp,
                       (const float*)dd, fl, mk, T->W, T->H, p->th_depth_bg,
                       T->d_sidx, T->d_scorr, T->d_sflow, T->d_sdepth, T->d_nstat);
    HIP_TRY(ctx, hipMemcpyAsync(T->h_cnt, T->d_nstat, n_frames * 4, hipMemcpyDeviceToHost, st));
    if ((rc = orb_collect(ctx, n_frames, 0))) return rc;           // waits for everything enqueued above on the main stream
    int max_ns = 0;
    for (int f = 0; f < n_frames; f++) max_ns = std::max(max_ns, T->h_cnt[f]);
    view->n_frames = n_frames; view->kp_pitch = ov.row_cap; view->stat_pitch = T->max_kp; view->obj_pitch = T->max_obj;
    view->kps = ov.h_kpf; view->desc = ov.h_descf; view->frame_beg = ov.h_frame_beg;
    view->n_stat = T->h_cnt; view->n_obj = T->h_cnt + T->B;
    view->stat_idx = (const int32_t*)rows(st, T->d_sidx, T->max_kp, 4, max_ns); view->stat_corr = (const float*)rows(st, T->d_scorr, T->max_kp, 8, max_ns);
    view->stat_flow = (const float*)rows(st, T->d_sflow, T->max_kp, 8, max_ns); view->stat_depth = (const float*)rows(st, T->d_sdepth, T->max_kp, 4, max_ns);
    HIP_TRY(ctx, hipStreamSynchronize(st));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream2));
    HIP_TRY(ctx, hipGetLastError());
    if (host_maps) memcpy(depth, hm_depth, n * 4);                 // ... the caller's depth buffer now holds the rescaled values
    return VIDO_OK;
}

// The small per-point calls below hand caller (pageable) arrays in and out.  Each packs its inputs into the pinned mirror of the device scratch (h_io <-> d_tmpf, same
// layout), sends them with ONE copy, and fetches all of its outputs with ONE copy: 3 stream operations per call instead of 3..8, none of them on pageable memory.
// (round 6) ZERO-COPY by default: the kernels read their inputs from and write their outputs to the pinned buffer itself (device-visible host memory: a few dozen KB per
// call over the bus) — ONE stream operation per call; beside saturating network kernels a small copy + kernel + copy round trip measured 1.17 ms where the kernel alone
// takes 0.1 (profiles/r6/tracker_zero_copy_io.txt).  VIDO_TRACK_IO_COPIES=1 restores the copies.
int vido_gather_static_depth(vido_ctx* ctx, int slot, const float* keys_xy, int n, float* depth_out)
{
    if (!ctx) return VIDO_E_INVALID;
    TrackState* T; int rc = track_state(ctx, &T); if (rc) return rc;
    if (slot < 0 || slot >= T->B || n < 0 || (size_t)n * 3 > T->tmp_cap) return vido_set_error(ctx, VIDO_E_INVALID, "gather_static_depth: bad slot/n");
    if (n == 0) return VIDO_OK;
    hipStream_t st = ctx->stream;
    const bool zc = T->zc_io; float* io = zc ? T->h_io : T->d_tmpf;      // zero-copy: the kernels read and write the pinned buffer itself
    memcpy(T->h_io, keys_xy, (size_t)n * 8);
    if (!zc) HIP_TRY(ctx, hipMemcpyAsync(io, T->h_io, (size_t)n * 8, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_gather_static, dim3((n + 255) / 256), dim3(256), 0, st, io, n, T->sdepth[slot], T->W, T->H, io + 2 * (size_t)n);
    if (!zc) HIP_TRY(ctx, hipMemcpyAsync(T->h_io + 2 * (size_t)n, io + 2 * (size_t)n, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    memcpy(depth_out, T->h_io + 2 * (size_t)n, (size_t)n * 4);
    return VIDO_OK;
}

int vido_gather_object_depth_label(vido_ctx* ctx, int slot, const float* keys_xy, int n, float th_depth_obj, float* depth_out, int32_t* label_out)
{
    if (!ctx) return VIDO_E_INVALID;
    TrackState* T; int rc = track_state(ctx, &T); if (rc) return rc;
    if (slot < 0 || slot >= T->B || n < 0 || (size_t)n * 4 > T->tmp_cap) return vido_set_error(ctx, VIDO_E_INVALID, "gather_object: bad slot/n");
    if (n == 0) return VIDO_OK;
    hipStream_t st = ctx->stream;
    const bool zc = T->zc_io; float* io = zc ? T->h_io : T->d_tmpf;      // zero-copy: the kernels read and write the pinned buffer itself
    float* dd = io + 2 * (size_t)n; int32_t* dl = (int32_t*)(io + 3 * (size_t)n);      // [keys 2n | depth n | label n]
    memcpy(T->h_io, keys_xy, (size_t)n * 8);
    if (!zc) HIP_TRY(ctx, hipMemcpyAsync(io, T->h_io, (size_t)n * 8, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_gather_object, dim3((n + 255) / 256), dim3(256), 0, st, io, n, T->sdepth[slot], T->smask[slot], T->W, T->H, th_depth_obj, dd, dl);
    if (!zc) HIP_TRY(ctx, hipMemcpyAsync(T->h_io + 2 * (size_t)n, dd, (size_t)n * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    memcpy(depth_out, T->h_io + 2 * (size_t)n, (size_t)n * 4); memcpy(label_out, T->h_io + 3 * (size_t)n, (size_t)n * 4);
    return VIDO_OK;
}

int vido_gather_point_samples(vido_ctx* ctx, int slot, const float* xy, int n, int32_t* mask_out, float* depth_out, float* flow_out)
{
    if (!ctx) return VIDO_E_INVALID;
    TrackState* T; int rc = track_state(ctx, &T); if (rc) return rc;
    if (slot < 0 || slot >= T->B || n < 0 || (size_t)n * 6 > T->tmp_cap || (n && (!xy || !mask_out || !depth_out || !flow_out))) return vido_set_error(ctx, VIDO_E_INVALID, "gather_point_samples: bad slot/n");
    if (n == 0) return VIDO_OK;
    hipStream_t st = ctx->stream;
    const bool zc = T->zc_io; float* io = zc ? T->h_io : T->d_tmpf;      // zero-copy: the kernels read and write the pinned buffer itself
    float* dk = io; float* dd = dk + 2 * (size_t)n; float* df = dd + n; int32_t* dm = (int32_t*)(df + 2 * (size_t)n);      // [xy 2n | depth n | flow 2n | mask n]
    memcpy(T->h_io, xy, (size_t)n * 8);
    if (!zc) HIP_TRY(ctx, hipMemcpyAsync(dk, T->h_io, (size_t)n * 8, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_gather_samples, dim3((n + 255) / 256), dim3(256), 0, st, dk, n, T->sdepth[slot], T->sflow[slot], T->smask[slot], T->W, T->H, dd, df, dm);
    if (!zc) HIP_TRY(ctx, hipMemcpyAsync(T->h_io + 2 * (size_t)n, dd, (size_t)n * 16, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    memcpy(depth_out, T->h_io + 2 * (size_t)n, (size_t)n * 4); memcpy(flow_out, T->h_io + 3 * (size_t)n, (size_t)n * 8); memcpy(mask_out, T->h_io + 5 * (size_t)n, (size_t)n * 4);
    return VIDO_OK;
}

int vido_update_mask(vido_ctx* ctx, int slot_last, int slot_cur, const int32_t* last_label, const float* last_corr_xy, int n,
                     int32_t* recovered_out, int cap, int32_t* n_recovered)
{
    if (!ctx || !n_recovered) return VIDO_E_INVALID;
    TrackState* T; int rc = track_state(ctx, &T); if (rc) return rc;
    *n_recovered = 0;
    if (slot_last < 0 || slot_last >= T->B || slot_cur < 0 || slot_cur >= T->B || n < 0 || (size_t)n * 3 > T->tmp_cap)
        return vido_set_error(ctx, VIDO_E_INVALID, "update_mask: bad slots/n");
    if (n == 0) return VIDO_OK;
    hipStream_t st = ctx->stream;
    const bool zc = T->zc_io; float* io = zc ? T->h_io : T->d_tmpf;      // zero-copy: the kernels read and write the pinned buffer itself
    std::vector<int> uni(last_label, last_label + n);
    std::sort(uni.begin(), uni.end()); uni.erase(std::unique(uni.begin(), uni.end()), uni.end());
    // The labels' samples are read in ONE gather (one upload, one launch, one download, one wait) instead of a round trip per label: five labels were 15 stream operations
    // and 5 host waits per frame.  A label that IS recovered rewrites the mask, which the labels after it must see (the reference's loop is sequential): after a scatter the
    // remaining labels are sampled again — rare (an object whose mask the detector lost in this frame).
    std::vector<float> corr; std::vector<int32_t> labs, vals; std::vector<int> g_lab, g_off;
    for (int lab : uni) {
        const size_t before = corr.size();
        for (int j = 0; j < n; j++) if (last_label[j] == lab) { corr.push_back(last_corr_xy[2 * j]); corr.push_back(last_corr_xy[2 * j + 1]); }
        if ((corr.size() - before) / 2 < 100) { corr.resize(before); continue; }      // fewer than 100 in-image samples is impossible to reach with < 100 points
        g_lab.push_back(lab); g_off.push_back((int)(before / 2));
    }
    g_off.push_back((int)(corr.size() / 2));
    for (size_t g0 = 0; g0 < g_lab.size();) {
        const int p0 = g_off[g0], m_all = g_off.back() - p0;
        VidoProfScope prof_round("update_mask: one gather round (upload + k_mask_at + download + wait)");
        memcpy(T->h_io, corr.data() + 2 * (size_t)p0, (size_t)m_all * 8);                 // [corr 2m | values m] in the pinned mirror of d_tmpf
        if (!zc) HIP_TRY(ctx, hipMemcpyAsync(io, T->h_io, (size_t)m_all * 8, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_mask_at, dim3((m_all + 255) / 256), dim3(256), 0, st, io, m_all, T->smask[slot_cur], T->W, T->H, (int32_t*)(io + 2 * (size_t)m_all));
        if (!zc) HIP_TRY(ctx, hipMemcpyAsync(T->h_io + 2 * (size_t)m_all, io + 2 * (size_t)m_all, (size_t)m_all * 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipStreamSynchronize(st));
        vals.assign((const int32_t*)(T->h_io + 2 * (size_t)m_all), (const int32_t*)(T->h_io + 2 * (size_t)m_all) + m_all);
        size_t g = g0; bool scattered = false;
        for (; g < g_lab.size() && !scattered; g++) {
            labs.assign(vals.begin() + (g_off[g] - p0), vals.begin() + (g_off[g + 1] - p0));
            labs.erase(std::remove(labs.begin(), labs.end(), INT32_MIN), labs.end());
            if (labs.size() < 100) continue;
            std::sort(labs.begin(), labs.end());
            int best = labs[0], bc = 0, run = 0;
            for (size_t j = 0; j < labs.size(); j++) { run = (j > 0 && labs[j] == labs[j - 1]) ? run + 1 : 1; if (run > bc) { bc = run; best = labs[j]; } }
            if (best != 0) continue;
            hipLaunchKernelGGL(k_mask_scatter, dim3(1024), dim3(256), 0, st, T->smask[slot_last], T->sflow[slot_last],
                               T->smask[slot_cur], T->W, T->H, g_lab[g]);
            if (*n_recovered < cap && recovered_out) recovered_out[*n_recovered] = g_lab[g];
            (*n_recovered)++;
            scattered = true;                                      // the labels after this one read the rewritten mask
        }
        g0 = g;
    }
    HIP_TRY(ctx, hipStreamSynchronize(st));
    HIP_TRY(ctx, hipGetLastError());
    return VIDO_OK;
}

int vido_read_maps(vido_ctx* ctx, int slot, float* depth_out, float* flow_out, int32_t* mask_out)
{
    if (!ctx) return VIDO_E_INVALID;
    TrackState* T; int rc = track_state(ctx, &T); if (rc) return rc;
    if (slot < 0 || slot >= T->B) return vido_set_error(ctx, VIDO_E_INVALID, "read_maps: bad slot");
    const size_t px = (size_t)T->W * T->H;
    if (depth_out) HIP_TRY(ctx, hipMemcpyAsync(depth_out, T->sdepth[slot], px * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (flow_out) HIP_TRY(ctx, hipMemcpyAsync(flow_out, T->sflow[slot], px * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (mask_out) HIP_TRY(ctx, hipMemcpyAsync(mask_out, T->smask[slot], px * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return VIDO_OK;
}

int vido_unproject_world(vido_ctx* ctx, const float* keys_xy, const float* z, int n, const vido_track_params* p, const float* Tcw, float* xyz_out)
{
    if (!ctx || !p || !Tcw) return VIDO_E_INVALID;
    TrackState* T; int rc = track_state(ctx, &T); if (rc) return rc;
    if (n < 0 || (size_t)n * 6 + 16 > T->tmp_cap) return vido_set_error(ctx, VIDO_E_INVALID, "unproject_world: n too large");
    if (n == 0) return VIDO_OK;
    hipStream_t st = ctx->stream;
    const bool zc = T->zc_io; float* io = zc ? T->h_io : T->d_tmpf;      // zero-copy: the kernels read and write the pinned buffer itself
    float RT[12];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) RT[r * 3 + c] = Tcw[c * 4 + r];
    for (int r = 0; r < 3; r++) { double s = 0; for (int c = 0; c < 3; c++) s += (double)(-RT[r * 3 + c]) * (double)Tcw[c * 4 + 3]; RT[9 + r] = (float)s; }
    float* dk = io; float* dz = dk + 2 * (size_t)n; float* drt = dz + n; float* dout = drt + 16;      // [keys 2n | z n | RT 12 (+4) | xyz 3n]
    memcpy(T->h_io, keys_xy, (size_t)n * 8); memcpy(T->h_io + 2 * (size_t)n, z, (size_t)n * 4); memcpy(T->h_io + 3 * (size_t)n, RT, sizeof RT);
    if (!zc) HIP_TRY(ctx, hipMemcpyAsync(dk, T->h_io, ((size_t)n * 3 + 16) * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_unproject_world, dim3((n + 255) / 256), dim3(256), 0, st, dk, dz, n, p->cx, p->cy, 1.0f / p->fx, 1.0f / p->fy, drt, dout);
    if (!zc) HIP_TRY(ctx, hipMemcpyAsync(T->h_io + 3 * (size_t)n + 16, dout, (size_t)n * 12, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    memcpy(xyz_out, T->h_io + 3 * (size_t)n + 16, (size_t)n * 12);
    return VIDO_OK;
}

int vido_scene_flow(vido_ctx* ctx, const float* xyz_last, const float* xyz_cur, const int32_t* sem_last, const int32_t* sem_cur, int n,
                    float* flow3d_out, int32_t* obj_label_inout)
{
    if (!ctx) return VIDO_E_INVALID;
    TrackState* T; int rc = track_state(ctx, &T); if (rc) return rc;
    if (n < 0 || (size_t)n * 12 > T->tmp_cap) return vido_set_error(ctx, VIDO_E_INVALID, "scene_flow: n too large");
    if (n == 0) return VIDO_OK;
    hipStream_t st = ctx->stream;
    const bool zc = T->zc_io; float* io = zc ? T->h_io : T->d_tmpf;      // zero-copy: the kernels read and write the pinned buffer itself
    const size_t N = (size_t)n;
    float *a = io, *b = a + 3 * N; int32_t *sl = (int32_t*)(b + 3 * N), *sc = sl + N, *ol = sc + N; float* c = (float*)(ol + N);      // [last 3n | cur 3n | sem last n | sem cur n | label n (in/out) | flow 3n]
    memcpy(T->h_io, xyz_last, N * 12); memcpy(T->h_io + 3 * N, xyz_cur, N * 12); memcpy(T->h_io + 6 * N, sem_last, N * 4); memcpy(T->h_io + 7 * N, sem_cur, N * 4); memcpy(T->h_io + 8 * N, obj_label_inout, N * 4);
    if (!zc) HIP_TRY(ctx, hipMemcpyAsync(a, T->h_io, N * 36, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_scene_flow, dim3((n + 255) / 256), dim3(256), 0, st, a, b, sl, sc, n, c, ol);
    if (!zc) HIP_TRY(ctx, hipMemcpyAsync(T->h_io + 8 * N, ol, N * 16, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    memcpy(obj_label_inout, T->h_io + 8 * N, N * 4); memcpy(flow3d_out, T->h_io + 9 * N, N * 12);
    return VIDO_OK;
}

}  // extern "C"
