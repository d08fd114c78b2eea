/* vido_c.h — the C-ABI drop-in boundary of the MI355X-native VIDO-SLAM hot path.
 *
 * Everything behind this header is hand-written HIP for gfx950 (libvido_slam_hip.so).  There is NO
 * CPU fallback: every entry point returns VIDO_E_NO_DEVICE / VIDO_E_HIP when the GPU path cannot run.
 * Plain C types, caller-owned buffers, int status (0 = ok, <0 = VIDO_E_*), no exceptions, no exit().
 * One vido_ctx = one HIP stream + one device arena; a ctx is not thread-safe; several ctxs may coexist.
 *
 * Each entry point names the reference interface it replaces (paths relative to the reference root):
 *   vido_orb_extract*      ORBextractor::operator()            vido_slam/include/ORBextractor.h:49,  src/ORBextractor.cc:1034-1105
 *   vido_hamming_match*    (no reference call site; north_star-defined brute-force matcher)
 *   vido_frame_features*   Frame::Frame RGB-D ctor             vido_slam/src/Frame.cc:36-241 (+ depth pre-scale Tracking.cc:299-322)
 *   vido_pose_opt_*        Optimizer::PoseOptimization{New,Flow2Cam,ObjMot,Flow2}   vido_slam/include/Optimizer.h:26-29
 *   vido_local_ba          Optimizer::PartialBatchOptimization vido_slam/include/Optimizer.h:31, src/Optimizer.cc:43-1228
 *   vido_global_ba         Optimizer::FullBatchOptimization    vido_slam/include/Optimizer.h:30, src/Optimizer.cc:1235-2178
 * The C++ facade (headers under include/vido_slam/) re-exports the reference's VIDO_SLAM::System/Tracking/Optimizer
 * class surface on top of these calls; INTEGRATION.md shows the binding a maintainer adds.
 */
#ifndef VIDO_C_H
#define VIDO_C_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define VIDO_OK              0
#define VIDO_E_INVALID      -1   /* bad argument / size beyond what the ctx was created for */
#define VIDO_E_NO_DEVICE    -2   /* no gfx950 device visible: the product never falls back to the CPU */
#define VIDO_E_HIP          -3   /* a HIP runtime call failed; see vido_last_error */
#define VIDO_E_CAPACITY     -4   /* an internal fixed-capacity buffer overflowed (see message) */
#define VIDO_E_NOMEM        -5

#define VIDO_MAX_LEVELS 16

typedef struct vido_ctx vido_ctx;

/* cv::KeyPoint fields the reference reads (pt, size, angle, response, octave). */
typedef struct vido_keypoint { float x, y, size, angle, response; int32_t octave; } vido_keypoint;

typedef struct vido_config {
    int32_t device;            /* HIP device ordinal */
    int32_t width, height;     /* frame size the arena is laid out for */
    int32_t max_batch;         /* frames in flight per *_batch call (>=1) */
    /* ORBextractor ctor arguments (ORBextractor.h:39-40); YAML keys ORBextractor.* (Tracking.cc:137-141) */
    int32_t n_features; float scale_factor; int32_t n_levels; int32_t ini_th_fast; int32_t min_th_fast;
    int32_t compute_descriptors;   /* 1: blur + rBRIEF (north_star); 0: reference behaviour (call commented out) */
    int32_t host_threads;      /* threads for the serial per-(frame,level) quadtree stage; 0 = auto */
} vido_config;

void        vido_config_default(vido_config* cfg);           /* 640x480, batch 1, KAIST ORB params */
int         vido_create(const vido_config* cfg, vido_ctx** out);
void        vido_destroy(vido_ctx* ctx);
const char* vido_last_error(const vido_ctx* ctx);            /* valid until the next call on ctx (NULL ctx: global create error) */
int         vido_device_name(const vido_ctx* ctx, char* buf, int buflen);
void*       vido_stream(vido_ctx* ctx);                      /* the ctx's hipStream_t (for callers that enqueue device work) */
int         vido_synchronize(vido_ctx* ctx);

/* ---- ORB ---------------------------------------------------------------------------------------
 * Single frame, host buffers (what ORBextractor::operator() is handed): gray CV_8UC1 `stride` bytes/row.
 * kp_out[max_kp], desc_out[max_kp*32] (may be NULL); *n_out = number of keypoints.
 * Keypoints come out in the reference's order: level 0..L-1, within a level the quadtree list order. */
int vido_orb_extract(vido_ctx* ctx, const uint8_t* gray, int stride, int width, int height,
                     vido_keypoint* kp_out, int max_kp, int* n_out, uint8_t* desc_out);

/* Batch of n_frames (<= max_batch) frames.  `imgs` is a DEVICE pointer when on_device!=0 (frames
 * `frame_stride` bytes apart, rows `stride` bytes apart), else a host pointer.  Outputs are host
 * arrays: kp_out[n_frames*max_kp], desc_out[n_frames*max_kp*32] (NULL ok), n_out[n_frames]. */
int vido_orb_extract_batch(vido_ctx* ctx, const uint8_t* imgs, int on_device, int n_frames,
                           size_t frame_stride, int stride, int width, int height,
                           vido_keypoint* kp_out, int max_kp, int* n_out, uint8_t* desc_out);

/* Diagnostics for the parity tests: copy pyramid level `level` (tight lw*lh bytes) of frame `frame` of the
 * last extract call back to the host; blurred!=0 selects the 7x7-blurred copy. */
int vido_orb_level_size(const vido_ctx* ctx, int level, int* lw, int* lh);
int vido_orb_read_level(vido_ctx* ctx, int frame, int level, int blurred, uint8_t* out);
/* FAST candidates of the last call for (frame, level) in reference order: packed x | y<<12 | score<<24
 * (level coordinates).  Returns count (or <0). */
int vido_orb_read_candidates(vido_ctx* ctx, int frame, int level, uint32_t* out, int cap);
/* per-stage time of the last batch call (HIP events on the ctx stream), ms: [0] pyramid (level-0 copy + resize
 * launches) [1] k_fast_cells alone [2] host quadtree [3] blur [4] keypoint upload + orient/rBRIEF + download
 * [5] total wall [6] scan+gather [7] number of FAST candidates in the batch */
int vido_orb_last_timing(const vido_ctx* ctx, float ms[8]);

/* ---- Hamming -------------------------------------------------------------------------------------
 * For each of the na 256-bit descriptors in a: index of the closest descriptor in b (smallest Hamming
 * distance, lowest index on ties) and that distance.  on_device!=0: a, b, idx_out, dist_out are device
 * pointers and the call only enqueues on the ctx stream. */
int vido_hamming_match(vido_ctx* ctx, const uint8_t* a, int na, const uint8_t* b, int nb,
                       int32_t* idx_out, int32_t* dist_out, int on_device);

#ifdef __cplusplus
}
#endif
#endif
