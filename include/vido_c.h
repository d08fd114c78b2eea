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
    int32_t host_threads;      /* unused since the quadtree stage moved to the device (kept for ABI stability) */
} vido_config;

void        vido_config_default(vido_config* cfg);           /* 640x480, batch 1, KAIST ORB params */
int         vido_create(const vido_config* cfg, vido_ctx** out);
void        vido_destroy(vido_ctx* ctx);
const char* vido_last_error(const vido_ctx* ctx);            /* valid until the next call on ctx (NULL ctx: global create error) */
int         vido_device_name(const vido_ctx* ctx, char* buf, int buflen);
void*       vido_stream(vido_ctx* ctx);                      /* the ctx's hipStream_t (for callers that enqueue device work) */
int         vido_synchronize(vido_ctx* ctx);
/* Network ops called with on_device != 0 enqueue on `hip_stream` (a caller-owned hipStream_t, e.g. torch's current
 * stream; NULL = the legacy default stream) while enable != 0; enable == 0 restores the ctx stream. */
int         vido_set_stream(vido_ctx* ctx, void* hip_stream, int enable);
/* The ctx's own stream waits (on the device) for `hip_event` (a hipEvent_t the producer recorded on its stream): how device buffers written by another stream — the
 * network nodes' — are handed to the tracker without a host-side wait. */
int         vido_stream_wait_event(vido_ctx* ctx, void* hip_event);

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

/* cvtColor(BGR|RGB|BGRA|RGBA -> GRAY) + ORBextractor::operator() in one stream of launches: what Tracking::GrabImageRGBD does with the
 * caller's colour image before the Frame is built (Tracking.cc:327-340: cvtColor(mImGray, mImGray, CV_RGB2GRAY / CV_BGR2GRAY / ...A2GRAY),
 * then Frame.cc:62 ExtractORB).  img: n_frames interleaved u8 images, `channels` (3 or 4) bytes per pixel, rows `stride` bytes apart;
 * rgb_order != 0: the first channel is red (Camera.RGB: 1).  gray_out (may be NULL): tight width*height bytes per frame, in the same memory
 * space as img (host pointer when on_device == 0, device pointer otherwise) — the mImGray the reference keeps.  Other arguments as
 * vido_orb_extract_batch.  The converted image is written straight into level 0 of the pyramid (no host pass over the pixels). */
int vido_orb_extract_color(vido_ctx* ctx, const uint8_t* img, int channels, int rgb_order, int on_device, int n_frames,
                           size_t frame_stride, int stride, int width, int height, uint8_t* gray_out,
                           vido_keypoint* kp_out, int max_kp, int* n_out, uint8_t* desc_out);

/* Diagnostics for the parity tests: copy pyramid level `level` (tight lw*lh bytes) of frame `frame` of the
 * last extract call back to the host; blurred!=0 selects the 7x7-blurred copy. */
/* enqueue the extraction of a device-resident colour frame and return; the next vido_orb_extract_color call for the same pointer and size only collects (csrc/orb.hip) */
int vido_orb_prefetch_color(vido_ctx* ctx, const uint8_t* img_dev, int channels, int rgb_order, int stride, int width, int height, void* ready_event);
int vido_orb_level_size(const vido_ctx* ctx, int level, int* lw, int* lh);
int vido_orb_read_level(vido_ctx* ctx, int frame, int level, int blurred, uint8_t* out);
/* FAST candidates of the last call for (frame, level) in reference order: packed x | y<<12 | score<<24
 * (level coordinates).  Returns count (or <0). */
int vido_orb_read_candidates(vido_ctx* ctx, int frame, int level, uint32_t* out, int cap);
/* per-stage time of the last batch call (HIP events on the ctx stream), ms: [0] pyramid (level-0 copy + resize
 * launches) [1] k_fast_cells alone [2] device quadtree + keypoint list [3] blur [4] orient/rBRIEF
 * [5] total wall [6] scan+gather [7] number of FAST candidates in the batch */
int vido_orb_last_timing(const vido_ctx* ctx, float ms[8]);

/* ---- Hamming -------------------------------------------------------------------------------------
 * For each of the na 256-bit descriptors in a: index of the closest descriptor in b (smallest Hamming
 * distance, lowest index on ties) and that distance.  on_device!=0: a, b, idx_out, dist_out are device
 * pointers and the call only enqueues on the ctx stream. */
int vido_hamming_match(vido_ctx* ctx, const uint8_t* a, int na, const uint8_t* b, int nb,
                       int32_t* idx_out, int32_t* dist_out, int on_device);

/* ---- Tracking front-end, data-parallel stages ---------------------------------------------------------
 * Frame maps (depth f32, flow f32x2, mask i32; width*height of the ctx) live in device "slots"
 * (vido_track_slots() of them, >= 2 so that frame k and k-1 are both resident). */
typedef struct vido_track_params {
    int32_t dataset;            /* YAML ChooseData: 0 OMD (d/f), 1 KITTI (bf/(d/f)), 2 KAIST (scale*bf/(d/f))  Tracking.cc:306-319 */
    float depth_map_factor, bf, kaist_scale;
    float th_depth_bg, th_depth_obj;     /* ThDepthBG / ThDepthOBJ (Frame::mThDepth, mThDepthObj) */
    int32_t dense_step;         /* 4 (Frame.cc:184) */
    float fx, fy, cx, cy;       /* Camera.fx.. (Frame.cc:229-234) */
} vido_track_params;

/* Host-side list outputs of vido_frame_features, [n_frames][max_*] each (caller-owned). */
typedef struct vido_frame_lists {
    int32_t max_stat, max_obj;
    int32_t* n_stat; int32_t* stat_idx; float* stat_corr; float* stat_flow; float* stat_depth;      /* mvStatKeysTmp(idx into kps)/mvCorres/mvFlowNext/mvStatDepthTmp */
    int32_t* n_obj; float* obj_keys; float* obj_corr; float* obj_depth; int32_t* obj_label; float* obj_flow;  /* mvObjKeys/mvObjCorres/mvObjDepth/vSemObjLabel/mvObjFlowNext */
} vido_frame_lists;

int vido_track_slots(vido_ctx* ctx);
/* diagnosis only (VIDO_DIAG_FIRSTOP=n in the facade): n rounds of a trivial stream operation + host wait on the context's stream, timed; means printed at exit */
int vido_debug_first_op(vido_ctx* ctx, int rounds);
/* Tracking::GrabImageRGBD depth pre-scale (Tracking.cc:299-322): copies the maps of n_frames frames into
 * slots [slot0, slot0+n_frames), rescales depth on the device and writes the rescaled depth back into the
 * caller's `depth` buffer (the reference mutates it in place).  on_device: 1 = the three pointers are device pointers; 2 = device pointers that the slots ADOPT (zero-copy: no
 * copy in either direction, depth rescaled in place; the caller keeps a frame's maps alive and untouched while its slot is in use — the current and the previous frame). */
int vido_frame_upload(vido_ctx* ctx, int slot0, int n_frames, float* depth, const float* flow, const int32_t* mask,
                      int on_device, const vido_track_params* p);
/* Frame::Frame RGB-D ctor lists (Frame.cc:72-100, 165-177, 184-211) for the frames in the slots, from the
 * ORB keypoints kps[n_frames][max_kp] (host). */
int vido_frame_features(vido_ctx* ctx, int slot0, int n_frames, const vido_keypoint* kps, const int32_t* n_kps, int max_kp,
                        const vido_track_params* p, vido_frame_lists* out);
/* Fused batch front end (what Tracking::GrabImageRGBD does per frame before Track(): ORBextractor::operator() +
 * depth pre-scale + the Frame::Frame lists), one stream of launches, keypoints handed over on the device.  Results are
 * returned as a VIEW into ctx-owned pinned host memory, valid until the next call on ctx: kps/desc rows are
 * [frame][kp_pitch], list rows [frame][stat_pitch] / [frame][obj_pitch]; frame f has frame_beg[f+1]-frame_beg[f]
 * keypoints.  `depth` is rescaled in place like the reference does.  maps_on_device: 0 host buffers, 1 device buffers copied into the
 * ctx's slots, 2 device buffers used ZERO-COPY: the slots refer to them until they are overwritten (the reference keeps shallow
 * references to the caller's Mats the same way, Tracking.cc:343-345), so the caller must keep them alive and unmodified. */
typedef struct vido_frontend_view {
    int32_t n_frames, kp_pitch, stat_pitch, obj_pitch;
    const vido_keypoint* kps; const uint8_t* desc; const int32_t* frame_beg;
    const int32_t* n_stat; const int32_t* stat_idx; const float* stat_corr; const float* stat_flow; const float* stat_depth;
    const int32_t* n_obj; const float* obj_keys; const float* obj_corr; const float* obj_depth; const int32_t* obj_label; const float* obj_flow;
} vido_frontend_view;
int vido_frontend_batch(vido_ctx* ctx, const uint8_t* imgs, int imgs_on_device, int n_frames, size_t frame_stride, int stride, int width, int height,
                        float* depth, const float* flow, const int32_t* mask, int maps_on_device, int slot0, const vido_track_params* p,
                        vido_frontend_view* view);
/* Tracking.cc:369-391 / 398-421: depth (and label) of the current frame at last frame's correspondences. */
int vido_gather_static_depth(vido_ctx* ctx, int slot, const float* keys_xy, int n, float* depth_out);
int vido_gather_object_depth_label(vido_ctx* ctx, int slot, const float* keys_xy, int n, float th_depth_obj,
                                   float* depth_out, int32_t* label_out);
/* Tracking::UpdateMask (Tracking.cc:3291-3357): mask of slot_cur is patched in place on the device. */
int vido_update_mask(vido_ctx* ctx, int slot_last, int slot_cur, const int32_t* last_label, const float* last_corr_xy, int n,
                     int32_t* recovered_out, int cap, int32_t* n_recovered);
int vido_read_maps(vido_ctx* ctx, int slot, float* depth_out, float* flow_out, int32_t* mask_out);   /* any pointer may be NULL */
/* mask / depth / flow of slot `slot` at ((int)x, (int)y) of n points (host xy in, host values out; points outside the image give 0): the only map data the host-side
 * renew stages need (vido_renew_*_sampled) — a few thousand points instead of three whole maps. */
int vido_gather_point_samples(vido_ctx* ctx, int slot, const float* xy, int n, int32_t* mask_out, float* depth_out, float* flow_out);
/* Frame::UnprojectStereoStat/Object, addnoise=0 (Frame.cc:706-771): Tcw row-major 4x4 f32. */
int vido_unproject_world(vido_ctx* ctx, const float* keys_xy, const float* z, int n, const vido_track_params* p,
                         const float* Tcw, float* xyz_out);
/* Tracking::GetSceneFlowObj (Tracking.cc:1582-1668). */
int vido_scene_flow(vido_ctx* ctx, const float* xyz_last, const float* xyz_cur, const int32_t* sem_last, const int32_t* sem_cur,
                    int n, float* flow3d_out, int32_t* obj_label_inout);

/* ---- Per-frame pose / object-motion optimisers (Levenberg-Marquardt on the device, FP64) -----------------
 * One problem = one of the reference's four optimiser calls; the constants each of them hard-codes
 * (information, Huber delta, rounds, iteration caps, chi2 thresholds) are explicit fields so the caller
 * (C++ facade Optimizer::PoseOptimization*, or vido-slam_amd/problems.py) states them once.
 *   mode 0  EdgeSE3ProjectXYZOnlyPose     e = obs - pi_K(T Xw)                  PoseOptimizationNew      Optimizer.cc:2180
 *   mode 1  EdgeSE3ProjectFlow2 + prior   e = (obs+f) - pi_K(T Twl K^-1(obs,d)) PoseOptimizationFlow2Cam :2622 / Flow2 :3037
 *   mode 2  EdgeSE3ProjectXYZOnlyObjMotion e = obs - proj(P (H Xw))             PoseOptimizationObjMot   :2826
 * All matrices row-major double; T is updated as exp(delta) * T (VertexSE3Expmap). */
typedef struct vido_pose_problem {
    int32_t mode, n;
    const double* Xw;          /* [n*3] world points (modes 0, 2) */
    const double* obs;         /* [n*2] measurement: current keypoint (0, 2) or LAST-frame keypoint (1) */
    const double* flow0;       /* [n*2] initial optical flow (mode 1) */
    const double* depth;       /* [n]   depth in the last frame (mode 1) */
    double Twl[16];            /* last camera -> world (mode 1) */
    double P[12];              /* K [R|t]_cw, 3x4 (mode 2) */
    double fx, fy, cx, cy;
    double T_init[16];         /* initial estimate; every round restarts from it (Optimizer.cc:2266, 2742) */
    double info_edge, info_prior, huber_delta;
    int32_t use_huber, rounds, drop_kernel_after_round;
    int32_t iters[4];
    float chi2_th[4];
} vido_pose_problem;

typedef struct vido_pose_result { double T[16]; int32_t n_inliers, lm_iterations; double chi2_final; } vido_pose_result;

/* outlier_out[n] (1 = rejected), flow_out[n*2] refined flow (mode 1; zeros otherwise); either may be NULL. */
int vido_pose_optimize(vido_ctx* ctx, const vido_pose_problem* prob, vido_pose_result* result, uint8_t* outlier_out, double* flow_out);
/* n_prob independent problems in one launch (one workgroup each): e.g. all dynamic objects of a frame. */
int vido_pose_optimize_batch(vido_ctx* ctx, const vido_pose_problem* probs, int n_prob, vido_pose_result* results,
                             uint8_t* const* outlier_out, double* const* flow_out);

/* ---- Bundle adjustment (windowed = PartialBatchOptimization, global = FullBatchOptimization) ---------------
 * Flat SoA problem, all f64 / i32, caller-owned; mirrors what the reference assembles from Map
 * (Optimizer.cc:216-350): camera-to-world poses (Map::vmCameraPose), one 3-D point per static tracklet,
 * observations = the point in the camera frame (Get3DinCamera), odometry = Map::vmRigidMotion[k][0],
 * optional prior on one camera.  Poses are row-major 3x4 [R|t].  cam_T and pt_xyz are updated in place.
 * Factor set of this build: EdgeSE3PointXYZ + EdgeSE3 + EdgeSE3Prior (the STATIC_ONLY graph); the
 * object-motion factors of FullBatchOptimization are listed under "next" in DESIGN.md.
 * Sharding (global BA over several GPUs, one process per GPU): every rank passes the whole problem with its own
 * landmark range [pt_lo, pt_hi) and rank/world; partial reduced systems are summed through `allreduce`. */
typedef struct vido_ba_problem {
    int32_t n_cam, n_pt, n_obs, n_odo, prior_cam, use_huber, max_iters, pad;
    double* cam_T; double* pt_xyz;
    const int32_t* obs_cam; const int32_t* obs_pt; const double* obs_meas;
    const int32_t* odo_i; const int32_t* odo_j; const double* odo_T;
    double prior_T[12];
    double info_obs, info_odo, info_prior, huber_obs, huber_odo, gain_threshold;
    int32_t pt_lo, pt_hi;      /* landmark shard of this rank; pt_hi <= pt_lo means "all" */
    int32_t rank, world;       /* rank 0 owns the camera-camera factors (odometry, prior) */
} vido_ba_problem;

typedef struct vido_ba_result { int32_t iterations, lm_trials; double chi2_initial, chi2_final, lambda_final;
                                double ms_setup;        /* host preprocessing + upload (wall) */
                                double ms_solve_loop;   /* the LM loop proper, inputs resident in HBM (wall) */
                                double ms_linearize_kernel;   /* mean k_ba_linearize duration (HIP events on the ctx stream) */
                                double ms_schur_kernel;       /* mean k_ba_schur_mfma duration (global path; 0 when another Schur kernel ran) */
} vido_ba_result;

/* In-place all-reduce of `count` doubles at DEVICE address `dev_ptr` over all ranks (op 0 = sum, 1 = max);
 * return 0 on success.  The ctx stream is idle when it is called.  NULL = single GPU. */
typedef int (*vido_allreduce_fn)(void* user, void* dev_ptr, size_t count, int op);

int vido_ba_optimize(vido_ctx* ctx, vido_ba_problem* prob, vido_ba_result* result, vido_allreduce_fn allreduce, void* user);
/* The all-reduce issued by the library itself: RCCL (over xGMI) on the context's stream, in place, no host synchronisation — the role the reference gives nothing to
 * (it is single-process).  One process per GPU: rank 0 calls vido_rccl_unique_id and distributes the 128 bytes by any means (bench.py: a
 * torch.distributed broadcast), every rank calls vido_rccl_init on its context, then passes `vido_rccl_allreduce` as `allreduce` and the context as `user`.
 * librccl is resolved at run time (dlopen); a process that never calls these does not load it. */
int vido_rccl_unique_id(uint8_t id_out[128]);
int vido_rccl_init(vido_ctx* ctx, const uint8_t id[128], int rank, int world);
int vido_rccl_allreduce(void* user_ctx, void* dev_ptr, size_t count, int op);
int vido_rccl_destroy(vido_ctx* ctx);

/* Object part of Optimizer::FullBatchOptimization (Optimizer.cc:1235-2178 with STATIC_ONLY = false): per (object, frame)
 * motion vertices H (VertexSE3, the reference initialises them to identity, :1592), one dynamic point vertex per
 * observation with its EdgeSE3PointXYZ to camera dyn_cam[k] (:1560-1582, :1683-1726), LandmarkMotionTernaryEdge
 * (tern_prev, tern_cur, tern_H), e = p_prev - H^-1 p_cur (:1728-1745, types/types_dyn_slam3d.cpp:53-85) and EdgeSE3
 * smoothness edges with identity measurement between motion vertices sm_i -> sm_j (:1604-1636).  The ternary edges
 * must link the dynamic points into chains (every point has at most one predecessor and one successor), which is what
 * the reference's tracklets produce.  H_T / dyn_xyz are updated in place.  With an all-reduce hook rank 0 owns this part. */
typedef struct vido_ba_dynamic {
    int32_t n_H, n_dyn, n_tern, n_smooth;
    double* H_T;                 /* [n_H*12] row-major 3x4 */
    double* dyn_xyz;             /* [n_dyn*3] world */
    const int32_t* dyn_cam; const double* dyn_meas;                            /* [n_dyn], [n_dyn*3] (point in the camera frame) */
    const int32_t* tern_prev; const int32_t* tern_cur; const int32_t* tern_H;  /* [n_tern] */
    const int32_t* sm_i; const int32_t* sm_j;                                  /* [n_smooth] indices into H */
    double info_dyn, info_tern, info_smooth, huber_dyn, huber_tern, huber_smooth;
} vido_ba_dynamic;
int vido_ba_optimize_dynamic(vido_ctx* ctx, vido_ba_problem* prob, vido_ba_dynamic* dyn, vido_ba_result* result,
                             vido_allreduce_fn allreduce, void* user);

/* ---- Native ops of the three network nodes (the nets' conv/GEMM layers run on PyTorch-ROCm) -----------------
 * All tensors f32, NCHW contiguous.  on_device != 0: every pointer is a device pointer and the call only
 * enqueues on the ctx stream (this is how the torch modules call them); otherwise host pointers, synchronous. */
/* correlation.FunctionCorrelation(first, second, intStride) — flow_net/src/correlation/correlation.py:277-335.
 * out: [B, 49, ceil(H/stride), ceil(W/stride)]. */
int vido_correlation(vido_ctx* ctx, const float* first, const float* second, int B, int C, int H, int W, int stride,
                     float* out, int on_device);
/* the static detector head's tail (confidence test, stable descending order by score, labels of the live slots, their count) in one launch: csrc/nets.hip::k_det_order */
int vido_det_order(vido_ctx* ctx, const float* scores, const long long* labels, const int* n_det, float confidence, int cap, long long* order, long long* labels_out, long long* n_live);
/* FPN level of every box (LevelMapper, modeling/poolers.py:11-45: floor(4 + log2(sqrt(area) / 224 + 1e-6)) clamped to [k_min, k_max], minus k_min) in one launch */
int vido_roi_levels(vido_ctx* ctx, const float* boxes, int n, float k_min, float k_max, int* out);
/* The mask head's tail for the one class channel a detection needs (mask_head/roi_mask_predictors.py:27-31 + inference.py:29-47): out[n][p] = sigmoid(sum_c w[label[n]][c]
 * feat[n][c][p] + b[label[n]]); feat [n][c][hw] f32, w [classes][c], labels int64 [n], out [n][hw] (csrc/nets.hip). */
int vido_mask_logit_select(vido_ctx* ctx, const float* feat, const float* w, const float* b, const long long* labels, float* out, int n, int c, int hw, int classes);
/* Conv epilogue on a DEVICE tensor x[N,C,H,W] (f32, contiguous), in place: x = leaky_relu(x + bias[c], slope) — the bias add and the
 * LeakyReLU(0.1) that follow every convolution of flow_net/src/layers.py fused into one pass (slope = 1: plain bias add). */
int vido_bias_act(vido_ctx* ctx, float* x, const float* bias, int N, int C, int H, int W, float slope);
/* MonoDepth2's decoder glue as single passes (mono_depth2/src/networks/depth_decoder.py:51-66, layers.py ConvBlock / Conv3x3 / upsample; run_mono_depth.py:137-145), DEVICE tensors:
 * vido_bias_unary: x = f(x + bias[c]) in place, kind 1 = ELU (ConvBlock), 2 = logistic function (the disparity head);
 * vido_upcat_reflect: ReflectionPad2d(1)(cat([upsample(x, 2, nearest), skip], 1)) — upsample + cat + the next Conv3x3's pad; x [C1][h][w], skip [C2][2h][2w] or NULL, out [C1+C2][2h+2][2w+2];
 * vido_minmax_norm_u16: the node's min-max normalisation to MONO16, out = (int) clamp((d - min) / (max - min + 1e-12) * 65536, 0, 65535), mm = {min, max} on the device. */
int vido_bias_unary(vido_ctx* ctx, float* x, const float* bias, int N, int C, int H, int W, int kind);
int vido_upcat_reflect(vido_ctx* ctx, const float* x, const float* skip, int C1, int C2, int h, int w, float* out);
int vido_minmax_norm_u16(vido_ctx* ctx, const float* d, const float* mm, int64_t n, int32_t* out);
/* x = act(x + bias[c] + res), res a DEVICE tensor shaped like x (NULL: vido_bias_act): closes a ResNe(X)t bottleneck whose FrozenBatchNorm2d
 * (maskrcnn_benchmark/layers/batch_norm.py:19-31) has been folded into the convolution weights and this bias
 * (modeling/backbone/resnet.py:352-372: out = bn3(conv3(.)); out += identity; relu). */
int vido_bias_res_act(vido_ctx* ctx, float* x, const float* bias, const float* res, int N, int C, int H, int W, float slope);
/* Node pre-processing in one pass: u8 H x W x 3 interleaved BGR (DEVICE) -> f32 [3][OH][OW] planar RGB, resized with the area rule of the nodes' cv2.resize(..., INTER_AREA) /
 * torch interpolate(mode="area") (mask_rcnn/src/predictor.py:267-283, mono_depth2/src/run_mono_depth.py:113-118), divided by `div` (255 for the depth node, 1 for the detector). */
int vido_area_feed(vido_ctx* ctx, const uint8_t* bgr, int H, int W, float* out, int OH, int OW, float div);
/* flow_net/src/layers.py:25-37 `Backward`: bilinear warp of x[B,C,H,W] by flow[B,2,H,W] (pixels; grid_sample with zero padding, align_corners False on the grid
 * -1 + (2i+1)/n + flow / ((n-1)/2)); DEVICE tensors, f32, contiguous. */
int vido_backwarp(vido_ctx* ctx, const float* x, const float* flow, int B, int C, int H, int W, float* out);
/* LiteFlowNet's regularisation stage (flow_net/src/layers.py:213-262, Regularization.forward) outside its convolutions, as two passes over DEVICE tensors (f32, NCHW):
 * front: out[:, 0] = sqrt(sum_c (im1 - Backward(im2, flow * scale))^2), out[:, 1:3] = flow - mean (mean [B,2] = the spatial mean of flow, a device tensor); out has
 *        out_channels >= 3 channels, the caller copies netFeat's features behind the first three (the stage's torch.cat);
 * tail:  dist [B, K*K, H, W] = netDist's output -> out [B, 2, H, W] = (netScaleX(d * unfold(flow_x, K)), netScaleY(d * unfold(flow_y, K))) / sum_c d with
 *        d = exp(-dist^2 - max_c(-dist^2)); wx / wy [K*K] and bx / by [1] are the two 1x1 convolutions' parameters; K in {3, 5, 7}. */
/* Depthwise ConvTranspose2d(C, C, 4, stride 2, padding 1, groups = C, bias = False) on DEVICE tensors: x [B,C,H,W] -> out [B,C,2H,2W], weight [C,1,4,4]; the input passes
 * through LeakyReLU(input_slope) first (1 = none).  LiteFlowNet's netUpflow / netUpcorr (flow_net/src/layers.py:105-108, 130-135). */
int vido_deconv4s2_depthwise(vido_ctx* ctx, const float* x, const float* weight, int B, int C, int H, int W, float input_slope, float* out);

/* Grouped 3x3 convolution + bias + ReLU on the fp32 matrix cores (csrc/gconv.hip): `conv2` of BottleneckWithFixedBatchNorm with the frozen batch norm folded in
 * (maskrcnn_benchmark/modeling/backbone/resnet.py:300-372, layers/batch_norm.py:19-31) — Conv2d(width, width, 3, 1, 1, groups) + FrozenBatchNorm2d + relu_ as ONE launch.
 * x [groups * cpg_in][H][W], y [groups * cpg_out][H][W] f32 DEVICE tensors of one image (x != y), bias [groups * cpg_out]; w_packed: the folded weight rearranged into
 * matrix-core operand order (vido_gconv3x3_packed_size floats; layout in csrc/gconv.hip, built by vido_slam_amd/nets/ops.py::pack_gconv3x3).  slope: 0 = ReLU, 1 = none.
 * in_bias (NULL or [groups * cpg_in]): the convolution reads relu(x + in_bias[channel]) instead of x — conv1's folded batch norm + relu_ of the same bottleneck applied on
 * the way in, so that the 1x1 convolution in front needs no pass of its own over its output.
 * vido_gconv3x3_supported: 1 when a kernel exists for the shape (8, 16 or a multiple of 32 channels per group and a row band that fits LDS); otherwise the call returns
 * VIDO_E_INVALID and the caller keeps the library convolution. */
/* 1x1 convolution (stride 1, batch 1) + bias + residual + leaky-ReLU as one fp32 matrix-core GEMM (csrc/conv1x1.hip): conv1 / conv3 / the stride-1 shortcut of
 * BottleneckWithFixedBatchNorm with their folded FrozenBatchNorm2d (maskrcnn_benchmark/modeling/backbone/resnet.py:300-372, layers/batch_norm.py:19-31).  x [cin][hw],
 * y / residual [cout][hw], f32 DEVICE, 16-byte aligned, y != x; bias [cout] or NULL; residual NULL = none; w_packed: [cout][cin] in operand order
 * (vido_slam_amd/nets/ops.py::pack_conv1x1).  slope: 0 = ReLU, 1 = none.  vido_conv1x1_supported: cout % 128 == 0, cin % 32 == 0, hw >= 128. */
int vido_conv1x1_supported(int cin, int cout, int hw);
/* the weight packing vido_conv1x1_bias_act / _up2_act expect for a shape: 0 = [co / 32][k / 8][32 (k & 1) + co % 32][(k % 8) / 2] (128 x 128 tiles),
 * 1 = [co / 16][k / 16][16 (k & 3) + co % 16][(k % 16) / 4] (128 x 112 tiles: fewer idle CUs in the last round of workgroups) */
int vido_conv1x1_layout(int cin, int cout, int hw);
/* Arithmetic of the 1x1 GEMM (round 6): fp32-equivalent results from the 16-bit matrix instructions (error against float64 not above the fp32 instruction's:
 * tests/test_maskrcnn_gpu.py).
 *   0 (default) = split-fp16: every fp32 operand is h + 2^-11 l' with h = rne16(x), l' = rne16(2^11 (x - h)) (|x - h - 2^-11 l'| <= 2^-22 |x|, 2^-24.5 |x| rms); the three products
 *       w_h x_h, w_h x_l', w_l' x_h run on v_mfma_f32_32x32x16_f16 with fp32 accumulators.  vido_conv1x1_layout answers 3 =
 *       [co / 32][k / 16][plane 2][32 ((k % 16) / 8) + co % 32][k % 8] fp16 of the weight scaled per OUTPUT CHANNEL by the power of two that puts the channel's largest |w| into
 *       [2^14, 2^15), followed by [cout] floats: the inverse scales (4 bytes per weight + 4 per channel).  Activations: full precision for 2.5e-4 <= |x| < 65504 (smaller ones: an absolute error <= 1.5e-11); a launch
 *       that meets |x| >= 65504 raises vido_conv1x1_range_flag (its outputs are not valid then).
 *   2 = split-bf16: three bf16 planes, the six plane products with i + j <= 2 on v_mfma_f32_32x32x16_bf16; fp32's range, twice the matrix work; layout 2 =
 *       [co / 32][k / 16][plane 3][32 ((k % 16) / 8) + co % 32][k % 8] bf16 (6 bytes per weight).
 *   1 = the fp32 matrix instruction (layouts 0 / 1 above).
 * Also selected by VIDO_CONV1X1_ARITH = f16x2 | bf16x3 | f32 in the environment.  Returns the previous setting; process-wide — set it before weights are packed (a packed
 * weight carries its layout, a mismatch is refused by the caller's shape check). */
int vido_conv1x1_set_arith(int arith);
/* non-zero when a split-fp16 launch on this context met an activation outside fp16's range since the last reset; read after the stream has been waited for */
int vido_conv1x1_range_flag(vido_ctx* ctx, int reset);
int vido_conv1x1_bias_act(vido_ctx* ctx, const float* x, const float* w_packed, const float* bias, const float* residual, float* y, int cin, int cout, int hw, float slope);
/* 2 x 2 stride-2 transposed convolution + bias + leaky-ReLU of a batch as one split-fp16 GEMM with a scatter epilogue (the mask head's conv5_mask,
 * roi_heads/mask_head/roi_mask_predictors.py:17-31): x [n][cin][h][w] -> y [n][cout][2 h][2 w]; w_packed: pack_conv1x1 layout 3 of [(2 a + b) cout + co][ci] = w[ci][co][a][b]
 * (vido_slam_amd/nets/ops.py::pack_deconv2x2).  vido_deconv2x2_supported: cout % 128 == 0, cin % 32 == 0, (h w) % 4 == 0, n h w >= 128, split-fp16 arithmetic selected. */
int vido_deconv2x2_supported(int n, int cin, int cout, int h, int w);
int vido_deconv2x2_bias_act(vido_ctx* ctx, const float* x, const float* w_packed, const float* bias, float* y, int n, int cin, int cout, int h, int w, float slope);
/* ... with the residual at half the resolution [cout][h/2][w/2], added nearest-upsampled: the FPN's lateral convolution + top-down sum (backbone/fpn.py:55-66); h, w even */
int vido_conv1x1_bias_up2_act(vido_ctx* ctx, const float* x, const float* w_packed, const float* bias, const float* residual_half, float* y, int cin, int cout, int h, int w, float slope);

/* k x k convolution (k = 3, 5, 7; stride 1, padding k / 2) to TWO output channels + bias + residual for one image (csrc/convsmall.hip): the last layer of LiteFlowNet's
 * matching / sub-pixel heads with the `flow + netMain(...)` behind it (flow_net/src/layers.py:152-160, 191-199).  x [cin][h][w], w [2][cin][k][k], bias [2] or NULL,
 * residual [2][h][w] or NULL, y [2][h][w]: f32 DEVICE tensors. */
int vido_conv_kxk_c2(vido_ctx* ctx, const float* x, const float* w, const float* bias, const float* residual, float* y, int cin, int k, int h, int w_);
/* 1x1 convolution with FEW input channels (even, <= 256; cout <= 256) + bias + residual + leaky ReLU for one image, no LDS: LiteFlowNet's netFeat layers
 * (layers.py:99, 125, 140), the detector's layer1.  w_packed: element (co, k) at [co / 32][k / 2][32 * (k & 1) + co % 32], cout padded to 32 with zeros. */
int vido_conv1x1_skinny(vido_ctx* ctx, const float* x, const float* w_packed, const float* bias, const float* residual, float* y, int cin, int cout, long long hw, float slope);

/* k x k convolution (7x7, 5x5, 3x3, 7x1, 1x7, 5x1, 1x5, 1x1; strides 1-4; zero padding) + bias + leaky ReLU as a direct implicit GEMM on the fp32 matrix pipe, one launch
 * (csrc/convdirect.hip): the layers of LiteFlowNet that the library ran as im2col / transposes + GEMM + a bias pass — the 7x7 stem, the stride-2 3x3 convolutions of the
 * feature pyramid, the separable distance layers of the regularisation (flow_net/src/layers.py:39-73, 217-235).  x [n][cin][h][w], y [n][cout][ho][wo] f32 DEVICE tensors,
 * w_packed = vido_conv_direct_pack(w [cout][cin][kh][kw]) (host) copied to the device, vido_conv_direct_packed_floats floats; bias [cout] or NULL; slope 0 = ReLU, 1 = none. */
int vido_conv_direct_supported(int cin, int cout, int h, int w, int kh, int kw, int sh, int sw, int ph, int pw);
long long vido_conv_direct_packed_floats(int cin, int cout, int kh, int kw);
int vido_conv_direct_pack(const float* w, int cin, int cout, int kh, int kw, float* w_packed);
int vido_conv_direct_bias_act(vido_ctx* ctx, const float* x, const float* w_packed, const float* bias, float* y, int n, int cin, int cout, int h, int w,
                              int kh, int kw, int sh, int sw, int ph, int pw, float slope);

/* 3x3 stride-1 padding-1 convolution + bias + leaky-ReLU as Winograd F(2x2, 3x3) with its sixteen channel contractions on the fp32 matrix pipe (csrc/wino.hip): the
 * dense 3x3 convolutions of LiteFlowNet (flow_net/src/layers.py:39-315), the FPN output / RPN head / mask head convolutions of the detector
 * (maskrcnn_benchmark/modeling/backbone/fpn.py, rpn/rpn.py:74-107, roi_heads/mask_head/roi_mask_feature_extractors.py) — what the library runs as a vector-ALU Winograd
 * kernel followed by a bias + activation pass.  x [n][cin][h][w], y [n][cout][h][w] f32 DEVICE tensors (y != x), bias [cout] or NULL, slope 0 = ReLU, 1 = none.
 * u_packed: vido_wino3x3_pack(w) copied to the device (16-byte aligned), vido_wino3x3_packed_floats(cin, cout) floats.  vido_wino3x3_pack runs on the HOST:
 * w [cout][cin][3][3] -> U = G g G^T in float64, rounded once, in the kernel's operand order.  vido_wino3x3_supported: cin >= 8, cout >= 32, h, w >= 2, tensors < 1 GB.
 * The result differs from a direct fp32 convolution by rounding only (the class of the library's own Winograd kernels).  Enqueues on the adopted stream; capturable. */
int vido_wino3x3_supported(int cin, int cout, int h, int w);
int vido_wino3x3_fills_chip(int n, int cout, int h, int w, int min_wgs);   /* 1 when the launch has >= min_wgs (0: 128) workgroups: below that the library's kernels win */
long long vido_wino3x3_packed_floats(int cin, int cout);
int vido_wino3x3_pack(const float* w, int cin, int cout, float* u_packed);
int vido_wino3x3_bias_act(vido_ctx* ctx, const float* x, const float* u_packed, const float* bias, float* y, int n, int cin, int cout, int h, int w, float slope);
/* The two launch forms.  0: the tile form above (a wave walks ALL input channels of its 32 channels x 32 tiles).  1: the K-split form for launches that would leave most of
 * the chip idle (fewer than 128 workgroups of the tile form: FPN P4-P6, the flow network's levels 3-6): a workgroup = 32 channels x 32 tiles, its four waves take a quarter of
 * the input channels each, the partial sums meet in LDS in a fixed order.  2: two channel slices x two tile blocks per workgroup (same packing as 1).  vido_wino3x3_form: the
 * form the library recommends for a launch (VIDO_WINO_KSPLIT=0/4/2: never / always 1 / always 2 for under-filled launches);
 * the packed weight must be of the form the launch is given (form 1 packs 4-channel chunks for every cout).  The un-suffixed entries are form 0. */
int vido_wino3x3_form(int n, int cin, int cout, int h, int w);
/* Fully connected layer y[rows][outs] = leaky_relu(x[rows][k] w[outs][k]^T + bias, slope) in the split-fp16 arithmetic of vido_conv1x1_set_arith(0), split over k (csrc/fch.hip,
 * round 6): the box head's fc6 (roi_heads/box_head/roi_box_feature_extractors.py:50-81).  w_packed: pack_conv1x1 layout 3 of w viewed as [outs][k][1][1]; part: scratch of
 * vido_fc_h_splitk(rows, k, outs) x rows x outs floats.  vido_fc_h_splitk: 0 = shape not taken (outs % 128, k % (32 S)). */
int vido_fc_h_splitk(int rows, int k, int outs);
int vido_fc_h(vido_ctx* ctx, const float* x, const void* w_packed, const float* bias, float* part, float* y, int rows, int k, int outs, float slope);
/* Dense 3x3 stride-1 `same` convolution + bias + leaky-ReLU as a DIRECT implicit GEMM in the split-fp16 arithmetic of vido_conv1x1_set_arith(0) (csrc/conv3x3h.hip, round 6):
 * the detector's chip-filling 256 -> 256 layers (FPN outputs and RPN head on P2 / P3: backbone/fpn.py:55-66, rpn/rpn.py:74-107; the mask head: roi_mask_feature_extractors.py).
 * x [n][cin][h][w], y [n][cout][h][w] f32 DEVICE; w_packed: two fp16 planes of the output channels scaled by powers of two, plane p of element (co, ci, dy, dx) at
 * [co / 32][ci / 16][dy][dx][p][32 ((ci % 16) / 8) + co % 32][ci % 8], then [cout] floats: the inverse scales (vido_slam_amd/nets/ops.py::pack_conv3x3_h).
 * vido_conv3x3_h_supported: cout 32, 64 or a multiple of 128 (input channels are padded to a multiple of 16 with zero weights), tensors below 1 GB.  Activations must stay below 65504 in magnitude (vido_conv1x1_range_flag otherwise). */
int vido_conv3x3_h_supported(int n, int cin, int cout, int h, int w);
int vido_conv3x3_h_workgroups(int n, int cout, int h, int w);
int vido_conv3x3_h_bias_act(vido_ctx* ctx, const float* x, const void* w_packed, const float* bias, float* y, int n, int cin, int cout, int h, int w, float slope);
long long vido_wino3x3_packed_floats_form(int cin, int cout, int form);
int vido_wino3x3_pack_form(const float* w, int cin, int cout, int form, float* u_packed);
int vido_wino3x3_bias_act_form(vido_ctx* ctx, const float* x, const float* u_packed, const float* bias, float* y, int n, int cin, int cout, int h, int w, float slope, int form);
int vido_gconv3x3_supported(int H, int W, int cpg_in, int cpg_out);
int64_t vido_gconv3x3_packed_size(int groups, int cpg_in, int cpg_out);
int vido_gconv3x3_bias_act(vido_ctx* ctx, const float* x, const float* in_bias, const float* w_packed, const float* bias, float* y, int groups, int cpg_in, int cpg_out, int H, int W, float slope);
/* ... with stride 2 (padding 1): the strided `conv2` of the first bottleneck of a ResNeXt stage (resnet.py:300-372 with STRIDE_IN_1X1 = False).  y [groups * cpg_out][(H + 1) / 2][W / 2];
 * W a multiple of 4, 8 / 16 / a multiple of 32 output channels per group (same w_packed), x 16-byte aligned, no in_bias.  vido_gconv3x3_s2_supported: 1 when the shape has a kernel. */
int vido_gconv3x3_s2_supported(int H, int W, int cpg_in, int cpg_out);
int vido_gconv3x3_s2_bias_act(vido_ctx* ctx, const float* x, const float* w_packed, const float* bias, float* y, int groups, int cpg_in, int cpg_out, int H, int W, float slope);
int vido_lfn_reg_front(vido_ctx* ctx, const float* im1, const float* im2, const float* flow, const float* mean, float scale, int B, int C, int H, int W, float* out, int out_channels);
int vido_lfn_reg_tail(vido_ctx* ctx, const float* dist, const float* flow, const float* wx, const float* bx, const float* wy, const float* by, int B, int K, int H, int W, float* out);
/* layers.ROIAlign forward — mask_rcnn/maskrcnn_benchmark/csrc/cuda/ROIAlign_cuda.cu:257-299.  rois [n,5] =
 * (batch index, x1, y1, x2, y2); out [n, C, pooled_h, pooled_w]. */
int vido_roi_align(vido_ctx* ctx, const float* feat, int B, int C, int H, int W, const float* rois, int n_rois,
                   float spatial_scale, int pooled_h, int pooled_w, int sampling_ratio, float* out, int on_device);
/* layers.nms(boxes, scores, thresh) — csrc/cuda/nms.cu:70-131 (IoU with the +1 convention, suppress when > thresh).
 * Host mode returns the kept ORIGINAL indices in ascending order like the reference.  Device mode: boxes already
 * sorted by descending score, keep_out/n_keep are device pointers receiving kept positions (ascending). */
int vido_nms(vido_ctx* ctx, const float* boxes_xyxy, const float* scores, int n, float thresh, int32_t* keep_out,
             int32_t* n_keep, int on_device);
/* The per-class NMS loop of the box head (modeling/roi_heads/box_head/inference.py:96-118: layers.nms once per class) in one
 * pass: boxes with different `groups` never suppress each other.  Conventions as vido_nms. */
int vido_nms_grouped(vido_ctx* ctx, const float* boxes_xyxy, const float* scores, const int32_t* groups, int n, float thresh,
                     int32_t* keep_out, int32_t* n_keep, int on_device);
/* Device-only batched NMS (layers.nms semantics per segment): the five per-level NMS calls of the RPN (modeling/rpn/inference.py:106-113), or the per-class
 * loop of the box head with `groups`, as ONE pair of launches without a host round trip.  boxes [total,4] DEVICE, segments back to back, each sorted by descending
 * score; seg_off / seg_n DEVICE int32 [n_seg]; max_n >= every segment length.  keep_out [n_seg, max_n]: kept positions relative to the segment start, ascending,
 * padded with -1; n_keep [n_seg].  Enqueues on the adopted stream (vido_set_stream), synchronises nothing. */
int vido_nms_segments(vido_ctx* ctx, const float* boxes_xyxy, const int32_t* groups, const int32_t* seg_off, const int32_t* seg_n, int n_seg, int max_n, int total,
                      float thresh, int32_t* keep_out, int32_t* n_keep);
/* Pooler.forward (modeling/poolers.py:97-121: LevelMapper, one ROIAlign per FPN level, scatter back by index) in one launch: feat[l] DEVICE [1,C,H[l],W[l]] f32,
 * boxes [n,4] x1 y1 x2 y2, level [n] in 0..3 (the LevelMapper's result, computed by the caller), out [n,C,pooled_h,pooled_w]. */
int vido_roi_align_fpn(vido_ctx* ctx, const float* const feat[4], const int H[4], const int W[4], const float scale[4], int C, const float* boxes, const int32_t* level,
                       int n, int pooled_h, int pooled_w, int sampling_ratio, float* out);
/* The layout the ROI-Align kernel reads: [B][C][H][W] -> [B][H][W][C] (DEVICE, f32).  ROIAlign_cuda.cu:15-122 gives every thread one output element and four scattered
 * 4-byte gathers per sample from channel-planar maps; this build reads channels-last (lanes across channels: one coalesced 256-byte load per tap and 64 channels). */
int vido_nchw_to_nhwc(vido_ctx* ctx, const float* src, int B, int C, int H, int W, float* dst);
/* vido_roi_align_fpn with CHANNELS-LAST maps feat[l] = [H[l]][W[l]][C] (vido_nchw_to_nhwc once per frame; the box and the mask pooler share the copies). */
int vido_roi_align_fpn_nhwc(vido_ctx* ctx, const float* const feat[4], const int H[4], const int W[4], const float scale[4], int C, const float* boxes, const int32_t* level,
                            int n, int pooled_h, int pooled_w, int sampling_ratio, float* out);
/* ---- The local-BA window resident on the device between frames (vido-slam_amd/csrc/bawin.hip).  The reference re-assembles the graph of
 * Optimizer::PartialBatchOptimization from the Map on every call (Optimizer.cc:56-94, 276-350).  Here a ring of the last frames' static features stays on the device:
 * vido_bawin_push_frame sends one frame's rows (Get3DinCamera measurement, world point, index of the previous frame's feature it continues: Map::vpFeatSta / vfDepSta /
 * vp3DPointSta / vnAssoSta), vido_bawin_set_labels the tracklet-label changes ((frame, feature, tracklet, position) quads: Map::vnTrkSta / vnPosSta), vido_bawin_solve assembles
 * the window's graph on the device exactly like the Map walk does (observations in (frame, feature) order, landmark ids in order of first appearance, chains that start before
 * the window dropped), solves it in place and writes the refined landmarks back into the ring.  prob: cameras (n_cam = N - start), odometry factors, prior, weights and LM
 * parameters as for vido_ba_optimize; its observation / point fields are ignored.  vido_bawin_read_points: one stored frame's world points (f32 [n][3]). */
int vido_bawin_create(vido_ctx* ctx, int cap_frames, int cap_features);
int vido_bawin_push_frame(vido_ctx* ctx, int frame, int n, const double* meas, const float* xyz, const int32_t* asso);
int vido_bawin_set_labels(vido_ctx* ctx, int n, const int32_t* quads);
int vido_bawin_solve(vido_ctx* ctx, int start, int N, vido_ba_problem* prob, vido_ba_result* res, int32_t* n_obs_out, int32_t* n_pt_out);
int vido_bawin_read_points(vido_ctx* ctx, int frame, int n, float* xyz_out);
/* ---- The detector's selection logic between its convolutions, device-side with fixed shapes (vido-slam_amd/csrc/detpost.hip).  Keys are (score bits << 32) | ~index, so a
 * descending key order is the reference's stable descending sort (ties -> lower index).
 * vido_rpn_select: RPNPostProcessor.forward_for_single_feature_map without the NMS (modeling/rpn/inference.py:73-105) for all FPN levels in one launch: sigmoid(objectness),
 *   the K = pre_nms_top_n best anchors of every level (anchors counted (y, x, a)), BoxCoder(1,1,1,1).decode + clip_to_image.  logits[l] [A,h,w], deltas[l] [4A,h,w] DEVICE;
 *   cell_anchors [n_levels][A][4] HOST; boxes_out [n_levels*K, 4], scores_out [n_levels*K] (-1 = padding row), n_out [n_levels] DEVICE.
 * vido_rpn_merge: select_over_all_levels (test branch, :125-159) after vido_nms_segments: the n_final best kept boxes over all levels.
 * vido_det_class_sort / vido_det_select: PostProcessor.filter_results (modeling/roi_heads/box_head/inference.py:96-137) around the per-class NMS: score threshold + descending
 *   order per class with decoded, clipped boxes; then the detections_per_img rule (threshold = k-th largest kept score, the reference's kthvalue) and the detections in
 *   (class, proposal) order in `cap` fixed slots; n_det (DEVICE) = the count the reference returns.  scratch: DEVICE, >= (nc + 1) int32. */
int vido_rpn_select(vido_ctx* ctx, int n_levels, const float* const* logits, const float* const* deltas, const int* h, const int* w, const int* stride, const float* cell_anchors,
                    int A, int K, int img_w, int img_h, float* boxes_out, float* scores_out, int32_t* n_out);
int vido_rpn_merge(vido_ctx* ctx, const float* boxes, const float* scores, const int32_t* keep, const int32_t* cnt, int n_levels, int K, int post_nms_top_n, int n_final,
                   float* out_boxes, float* out_scores, int32_t* n_valid);
int vido_det_class_sort(vido_ctx* ctx, const float* prob, const float* deltas, const float* proposals, const float* objectness, int N, int nc, float thresh, const float weights[4],
                        int img_w, int img_h, float* seg_boxes, int32_t* order, int32_t* seg_n);
int vido_det_select(vido_ctx* ctx, const float* prob, const float* seg_boxes, const int32_t* order, const int32_t* keep, const int32_t* cnt, int N, int nc, int detections_per_img, int cap,
                    int32_t* scratch, float* out_boxes, float* out_scores, int64_t* out_labels, int32_t* n_det);
/* Masker(threshold 0.5, padding 1).forward (modeling/roi_heads/mask_head/inference.py:87-160, per detection on the host in the reference) fused with the node's
 * label image (src/run_mask_rcnn.py:112-118: blank_mask += mask * class_index): masks [n,1,M,M] f32, boxes [n,4] f32 in the output image, labels [n] i64,
 * all DEVICE, detections in the order the node adds them; out [H,W] u8 = (sum over detections of pasted mask * class index) mod 256. */
int vido_mask_label_image(vido_ctx* ctx, const float* masks, const float* boxes, const int64_t* labels, int n, int M, int padding, float thresh, int H, int W, uint8_t* out);
/* BoxCoder(weights).decode(deltas [n,4k], boxes [n,4]) — modeling/box_coder.py:52-95. */
int vido_box_decode(vido_ctx* ctx, const float* deltas, const float* boxes, int n, int k, const float weights[4],
                    float* out, int on_device);

/* ---- Initial model: seeded P3P-RANSAC (Tracking::GetInitModelCam / GetInitModelObj, Tracking.cc:1965-1970, 2068-2073:
 * cv::solvePnPRansac(pre_3d, cur_2d, K, 0, ..., 500, 0.4, 0.98, inliers, SOLVEPNP_P3P)).  pts3d [n*3] f32 (previous frame,
 * world), pts2d [n*2] f32 (current keypoints); T_out row-major 4x4 world->camera; inlier_mask[n] may be NULL.
 * All max_iters hypotheses are scored in parallel; OpenCV's adaptive early stop is replayed on the counts.
 * DEVIATION from cv::solvePnPRansac: OpenCV refits the winning model on its inlier set before returning (solvepnp.cpp, the final solvePnP over the inliers); this entry
 * returns the winning minimal-sample P3P pose itself.  Both reference call sites hand the pose straight to PoseOptimizationFlow2Cam / PoseOptimizationFlow2, which
 * re-optimise it over the same inliers, so only the optimiser's starting point differs; the sampling sequence (counter-based RNG) also differs from cv::RNG by design. */
int vido_pnp_ransac(vido_ctx* ctx, const float* pts3d, const float* pts2d, int n, double fx, double fy, double cx, double cy,
                    int max_iters, double reproj_err, double confidence, uint64_t seed, double T_out[16], uint8_t* inlier_mask,
                    int32_t* n_inliers);

/* All dynamic objects of a frame in one pair of launches and ONE synchronisation (Tracking::Track loops GetInitModelObj over the objects, Tracking.cc:1192-1228):
 * problem p has n[p] points pts3d[p] / pts2d[p], seed seeds[p]; T_out [n_prob][16], inlier_mask[p] (may be NULL) [n[p]], n_inliers [n_prob].  Results are identical to
 * n_prob calls of vido_pnp_ransac. */
int vido_pnp_ransac_batch(vido_ctx* ctx, int n_prob, const float* const* pts3d, const float* const* pts2d, const int32_t* n, double fx, double fy, double cx, double cy,
                          int max_iters, double reproj_err, double confidence, const uint64_t* seeds, double* T_out, uint8_t* const* inlier_mask, int32_t* n_inliers);

/* ---- Host-side bookkeeping stages of the tracker on flat arrays (no device work, no ctx; vido-slam_amd/csrc/trackhost.cpp) ---------------------------------
 * What the C++ facade's Tracking / Frame methods of the same names run; exported so that hosts in other languages and the parity tests can call them. */
typedef struct vido_host_maps { const int32_t* mask; const float* depth; const float* flow; int32_t width, height; } vido_host_maps;   /* mSegMap / mDepthMap / mFlowMap */
/* Frame::UndistortKeyPoints (Frame.cc:603-633): cv::undistortPoints(mat, mat, mK, mDistCoef, cv::Mat(), mK) — five fixed-point iterations of the Brown model;
 * K = (fx, fy, cx, cy), dist = (k1, k2, p1, p2, k3); k1 == 0: copy (Frame.cc:605-609). */
/* Frame::UnprojectStereo* / ObtainFlowDepth* with addnoise = 1 (Frame.cc:706-716 ...): the depth with ONE draw of a fresh cv::RNG(seed) added,
 * z + gaussian(z*z / (725*0.5) * 0.15); seed 0 = time(NULL) as the reference.  Host function (cv::RNG's multiply-with-carry + ziggurat restated; OpenCV side unpinned). */
float vido_depth_noise(float z, unsigned seed);
int vido_undistort_points(const float* xy, int n, const float K[4], const float dist[5], float* xy_out);
/* Tracking::RenewFrameInfo, static part (Tracking.cc:2973-3075): the inliers TM_sta (indices into stat_xy = mvStatKeys, -1 = rejected) that still pass the mask /
 * depth <= 40 / flow tests, then top-up from sample_xy (= mvKeys) in stride-20 passes, skipping samples closer than 1 px to a kept inlier, until max_num.
 * Element k of the result: src_out[k] = index into stat_xy (inlier_out[k] >= 0) or into sample_xy (inlier_out[k] == -1), inlier_out[k] (nStaInlierID),
 * flow_out[2k..] (mvFlowNext).  *n_out = count (VIDO_E_CAPACITY if > cap). */
int vido_renew_static(const vido_host_maps* maps, const float* stat_xy, int n_stat, const int32_t* TM_sta, int n_tm, const float* sample_xy, int n_sample,
                      int max_num, int32_t* src_out, int32_t* inlier_out, float* flow_out, int cap, int32_t* n_out);
/* Tracking::RenewFrameInfo, object part (Tracking.cc:3116-3270).  obj_xy / obj_label: mvObjKeys / vObjLabel of the current frame (n_obj_pts); per tracked object i:
 * inlier ids inl_ids[inl_off[i] .. inl_off[i+1]) (vnObjInlierID), obj_stat (bObjStat), sem_position (nSemPosition), mod_label (nModLabel); tmp_*: the dense samples
 * of this frame set aside by GrabImageRGBD (mvTmpObjKeys / Depth / SemObjLabel / FlowNext / Corres).  Outputs = the new mvObjKeys, mvObjDepth, vSemObjLabel,
 * mvObjFlowNext, mvObjCorres, nDynInlierID, vObjLabel. */
int vido_renew_objects(const vido_host_maps* maps, const float* obj_xy, const int32_t* obj_label, int n_obj_pts,
                       int n_objects, const int32_t* inl_off, const int32_t* inl_ids, const uint8_t* obj_stat, const int32_t* sem_position, const int32_t* mod_label,
                       const float* tmp_xy, const float* tmp_depth, const int32_t* tmp_sem, const float* tmp_flow, const float* tmp_corr, int n_tmp, int max_num_obj,
                       float* keys_out, float* depth_out, int32_t* sem_out, float* flow_out, float* corr_out, int32_t* inlier_out, int32_t* label_out, int cap, int32_t* n_out);
/* The same two stages for maps that live on the DEVICE (the in-process network -> tracker hand-over): instead of whole host maps they take the map
 * values AT the candidate points — mask / depth / flow at ((int)x, (int)y) of every point of the list, gathered on the device (vido_gather_point_samples).  Entries of
 * points outside the image are never read.  vido_renew_static / vido_renew_objects sample their host maps and call these. */
typedef struct vido_point_samples { const int32_t* mask; const float* depth; const float* flow; } vido_point_samples;   /* [n], [n], [2n] */
int vido_renew_static_sampled(int width, int height, const float* stat_xy, int n_stat, const vido_point_samples* stat_samples, const int32_t* TM_sta, int n_tm,
                              const float* sample_xy, int n_sample, const vido_point_samples* sample_samples, int max_num,
                              int32_t* src_out, int32_t* inlier_out, float* flow_out, int cap, int32_t* n_out);
int vido_renew_objects_sampled(int width, int height, const float* obj_xy, const int32_t* obj_label, int n_obj_pts, const vido_point_samples* obj_samples,
                               int n_objects, const int32_t* inl_off, const int32_t* inl_ids, const uint8_t* obj_stat, const int32_t* sem_position, const int32_t* mod_label,
                               const float* tmp_xy, const float* tmp_depth, const int32_t* tmp_sem, const float* tmp_flow, const float* tmp_corr, int n_tmp, int max_num_obj,
                               float* keys_out, float* depth_out, int32_t* sem_out, float* flow_out, float* corr_out, int32_t* inlier_out, int32_t* label_out, int cap, int32_t* n_out);
/* Tracking::DynObjTracking (Tracking.cc:1670-1912): groups the n object points by semantic label, drops objects mostly on the image border / static (scene flow) /
 * far / smaller than 150 points (obj_label is updated in place: -1 outlier, 0 static, else the track id), and assigns track ids from the last frame's objects
 * (last_sem_position / last_obj_stat / last_mod_label, n_last) or from *max_id.  Result: n_objects objects, points of object i = obj_ids[obj_off[i] .. obj_off[i+1]),
 * mod_label_out (nModLabel), sem_position_out (nSemPosition). */
int vido_dyn_obj_tracking(const int32_t* sem_label, int32_t* obj_label, const float* obj_xy, const float* obj_depth, const float* flow3d, const int32_t* last_sem_label, int n,
                          const int32_t* last_sem_position, const uint8_t* last_obj_stat, const int32_t* last_mod_label, int n_last, int rows, int cols,
                          float sf_mg_thres, float sf_ds_thres, float th_depth_obj, int f_id, int32_t* max_id,
                          int32_t* obj_off, int32_t* obj_ids, int32_t* mod_label_out, int32_t* sem_position_out, int max_objects, int32_t* n_objects);

/* Tracking::GetStaticTrack / GetDynamicTrackNew (Tracking.cc:2514-2720) through the facade's INCREMENTAL store (Map::UpdateTracklets, one association row per call like
 * Tracking::Track): row i belongs to frame i+1, TM[row_off[i] + j] = feature of frame i matched by its feature j (-1: none), labels (dynamic store; NULL = static store)
 * the per-feature object labels.  Tracklet t = (frame, feature) pairs pairs[2*trk_off[t] .. 2*trk_off[t+1]); obj_id[t] (dynamic).  owner_trk / owner_pos (may be NULL):
 * for every feature of every frame (concatenated), the tracklet of length >= 3 that owns it and its position in it, -1 if none. */
int vido_tracklets_incremental(int n_rows, const int32_t* row_off, const int32_t* row_n, const int32_t* TM, const int32_t* labels, int n_feat0,
                               int32_t* trk_off, int32_t* pairs, int32_t* obj_id, int32_t* owner_trk, int32_t* owner_pos, int cap_trk, int cap_pairs, int32_t* n_trk);

/* ---- The whole per-frame pipeline behind one C handle ---------------------------------------------------------------------------
 * VIDO_SLAM::System (System.h:72-114) for hosts that bind C instead of C++ (ctypes / cgo / JNI): System::System() + Init(yaml, RGBD)
 * (System.cc:23-48), TrackRGBD (System.cc:51-63 -> Tracking::GrabImageRGBD, Tracking.cc:283-782 -> Track(), :1081-1509) and
 * SaveResultsIJRR2020 (System.cc:80-240).  Errors that the C++ facade throws come back as VIDO_E_* + vido_system_last_error.
 * One live system per process, like the reference (Frame's statics, Frame.cc:26-30). */
typedef struct vido_system vido_system;
typedef struct vido_system_stats {          /* of the last vido_system_track_rgbd call */
    int32_t frame_id, n_keypoints, n_static, n_static_inliers, n_objects, n_object_points, ba_window, pad;
    float ms_total;                          /* TrackRGBD wall time */
    float ms_update_mask, ms_frame;          /* Tracking::UpdateMask; Frame::Frame (cvtColor + ORB + lists) + hand-over gathers */
    float ms_cam_pose, ms_obj_tracking, ms_obj_motion, ms_renew;   /* the reference's all_timing[1..4] (Tracking.cc:1120-1324); obj_motion = sum over objects */
    float ms_local_ba;                       /* Map::fLBA_time (Tracking.cc:1436-1451) */
    float ms_wait_inputs;                    /* vido_system_track_rgbd_device only: host time spent waiting for the producer's ready event (the networks of this frame) — inside ms_total,
                                                outside every stage time above */
    float ms_orb, ms_lists;                  /* inside ms_frame: the extractor call (cvtColor + pyramid + FAST + quadtree + rBRIEF, keypoints on the host) | Frame's static / object lists */
} vido_system_stats;
int         vido_system_create(const char* settings_yaml, vido_system** out);
void        vido_system_destroy(vido_system* sys);
const char* vido_system_last_error(const vido_system* sys);      /* NULL sys: error of the last failed vido_system_create */
/* im: u8 interleaved, `channels` 1 (gray), 3 or 4 (BGR[A], or RGB[A] when the settings say Camera.RGB: 1), tight rows; depth f32 (raw sensor
 * units; REWRITTEN IN PLACE with the pre-scaled depth like the reference does to the caller's Mat, Tracking.cc:299-322); flow f32 x2; mask i32;
 * all width*height, host memory, alive until the NEXT call returns (the reference keeps shallow references, Tracking.cc:343-345, 777-780).
 * n_image: StopFrame = n_image - 1.  Tcw_out: row-major 4x4 world->camera pose of this frame (identity for the first). */
int         vido_system_track_rgbd(vido_system* sys, const uint8_t* im, int channels, int width, int height, float* depth, const float* flow,
                                   const int32_t* mask, double timestamp, int n_image, float Tcw_out[16]);
/* The same with the image (u8, 1 / 3 / 4 interleaved channels) and the three maps already RESIDENT ON THE DEVICE (plain device pointers; depth is rescaled in place on the
 * device): the in-process replacement of the reference's three service round trips (src/realtime_demo/src/run_vido.cc:57-171 -> :229-235).  Nothing is uploaded, no map is
 * downloaded; ready_event (hipEvent_t, may be NULL) orders the tracker's stream behind the producer of the buffers. */
/* zero_copy != 0: vido_system_track_rgbd_device ADOPTS the three map buffers instead of copying them into the tracker's slots (vido_frame_upload on_device = 2): the caller
 * keeps the maps of a frame alive and untouched until the call AFTER the next one has returned (a ring of >= 3 frames; pipeline.EndToEnd's has 4).  Default 0. */
int         vido_system_set_zero_copy_maps(vido_system* sys, int zero_copy);
/* The ORB extraction of the NEXT vido_system_track_rgbd_device call's image, put on the tracker's stream now (it needs nothing but the image: a pipeline calls this while it
 * still waits for the networks of the frame); image_ready_event (hipEvent_t or NULL) orders it behind the image's upload.  The track call for the same pointer and size then
 * only collects.  Same thread as the track calls; the image stays untouched until that call has returned. */
int         vido_system_prefetch_image_device(vido_system* sys, const void* im_dev, int channels, int width, int height, void* image_ready_event);
int         vido_system_track_rgbd_device(vido_system* sys, const void* im_dev, int channels, int width, int height, float* depth_dev, const float* flow_dev,
                                          const int32_t* mask_dev, void* ready_event, double timestamp, int n_image, float Tcw_out[16]);
int         vido_system_get_stats(const vido_system* sys, vido_system_stats* out);
int         vido_system_save_results(vido_system* sys, const char* prefix);
/* the vido_ctx the system's tracker runs on (NULL before the first frame): lets a caller share the device / query timings */
vido_ctx*   vido_system_context(vido_system* sys);
/* The reference's addnoise = 1 depth noise (Frame.cc:711-716) is seeded with time(NULL): PoseOptimizationNew's result then differs from one wall-clock second to the next.
 * seed != 0 pins cv::RNG's seed for every later draw of this process (tests, reproducible runs); 0 restores the reference behaviour.  (C++ callers: detail::SetDepthNoiseSeed.) */
int         vido_system_set_depth_noise_seed(vido_system* sys, unsigned seed);

#ifdef __cplusplus
}
#endif
#endif
