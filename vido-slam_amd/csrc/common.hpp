// common.hpp — ctx layout and error plumbing shared by the HIP translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/vido_c.h"


struct LevelInfo {
    int w, h, pitch;               // level size, row pitch in bytes (multiple of 64)
    int off;                       // byte offset of the level inside one frame's pyramid slab
    int first_cell, n_cells;       // cells of this level inside the cell table (reference order)
    int xtab_off, ytab_off;        // offsets (elements) into the resize tables
    int n_budget;                  // mnFeaturesPerLevel
    float scale;                   // mvScaleFactor
    int rs_rows, rs_ndw;           // resize staging tile: max source rows / dwords per row of a 128x8 output tile
};

struct PyrDev {                    // passed by value to kernels
    int n_levels;
    int w[VIDO_MAX_LEVELS], h[VIDO_MAX_LEVELS], pitch[VIDO_MAX_LEVELS], off[VIDO_MAX_LEVELS];
};

struct CellDesc { int level, x0, y0, sw, sh, pad0, pad1, pad2; };   // FAST sub-image of one cell (level coords)
struct BlurTile { int level, tx, ty, pad; };

struct OrbState;                   // orb.hip
struct TrackState;                 // track.hip
struct BaState;                    // ba.hip

struct vido_ctx {
    vido_config cfg{};
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;
    hipStream_t ext_stream = nullptr;  // vido_set_stream: caller-owned stream for the on_device network ops
    bool has_ext_stream = false;
    std::string err;
    char dev_name[256] = {0};
    OrbState* orb = nullptr;
    TrackState* trk = nullptr;
    BaState* ba = nullptr;
    struct HamState* ham = nullptr;
    struct PoseState* pose = nullptr;
    struct NetState* net = nullptr;
    struct PnpState* pnp = nullptr;
    struct BaWin* bawin = nullptr;     // bawin.hip: the device-resident local-BA window
    void* detpost_buf = nullptr; size_t detpost_cap = 0; unsigned long long detpost_sig = 0;   // detpost.hip: scratch of the RPN selection (keys, histograms, state)
    void* rccl_comm = nullptr;         // ncclComm_t of vido_rccl_init (rccl.cpp): the sharded BA's all-reduce on this context's stream
    int rccl_rank = 0, rccl_world = 1;
    unsigned* c1_range_flag = nullptr; // conv1x1.hip: pinned host word the split-fp16 kernels raise when an activation leaves fp16's range (vido_conv1x1_range_flag)
};

int vido_set_error(vido_ctx* ctx, int code, const char* fmt, ...);
// VIDO_CALL_PROF=1: named host-side sections inside the library (ctx.cpp keeps the table, prints it at exit next to the facade's per-call table).  A section that ends with
// `sync` waits for the stream first, so that the section owns the device time of what it enqueued (diagnosis only: the waits change the overlap they measure).
bool vido_prof_on();
void vido_prof_add(const char* name, double ms);
struct VidoProfScope {
    const char* name; hipStream_t st; bool sync; std::chrono::steady_clock::time_point t0;
    VidoProfScope(const char* n, hipStream_t s = nullptr, bool sy = false) : name(n), st(s), sync(sy) { if (vido_prof_on()) t0 = std::chrono::steady_clock::now(); }
    ~VidoProfScope() { if (!vido_prof_on()) return; if (sync) (void)hipStreamSynchronize(st); vido_prof_add(name, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count()); }
};

#define HIP_TRY(ctx, expr)                                                                  \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess)                                                               \
            return vido_set_error((ctx), VIDO_E_HIP, "%s failed: %s (%s:%d)", #expr,       \
                                  hipGetErrorString(_e), __FILE__, __LINE__);               \
    } while (0)

int orb_state_create(vido_ctx* ctx);
// extractor split used by the fused front end (track.hip): enqueue everything, then wait + check (+ mirror the rows)
int orb_enqueue(vido_ctx* ctx, const uint8_t* imgs, int on_device, int nf, size_t frame_stride, int stride, int width, int height);
int orb_collect(vido_ctx* ctx, int nf, int copy);
int orb_mirror_async(vido_ctx* ctx, int nf);
struct OrbView { const vido_keypoint* d_kpf; const int* d_nkp; int row_cap; const vido_keypoint* h_kpf; const uint8_t* h_descf; const int* h_frame_beg; };
OrbView orb_view(vido_ctx* ctx);
void track_state_destroy(vido_ctx* ctx);
void ham_state_destroy(vido_ctx* ctx);
void pose_state_destroy(vido_ctx* ctx);
void ba_state_destroy(vido_ctx* ctx);
// device-resident inputs of the local-window solve (bawin.hip assembles them, ba.hip solves): observations sorted by camera, landmark-major slot tables, points (in / out)
struct BaDevInputs { int no, n_ptl, kcap; const int *obs_cam, *obs_pt, *obs_pos, *pt_start, *slot_cam; const double* obs_meas; double* pt; };
int ba_run_device_inputs(vido_ctx* ctx, vido_ba_problem* prob, vido_ba_result* res, const BaDevInputs* DI);
void bawin_state_destroy(vido_ctx* ctx);
void net_state_destroy(vido_ctx* ctx);
void pnp_state_destroy(vido_ctx* ctx);
void orb_state_destroy(vido_ctx* ctx);
