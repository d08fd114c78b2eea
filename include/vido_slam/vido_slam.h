/* vido_slam.h — C++ facade with the reference's class surface (namespace VIDO_SLAM) over the C-ABI of vido_c.h.
 * Class / method names, argument order and meaning follow the reference headers so that the offline driver
 * (vido_slam/demo/run_vido_slam.cc:67-137) ports by changing includes only:
 *   System      vido_slam/include/System.h:72-114        Init, TrackRGBD, SaveResultsIJRR2020
 *   Tracking    vido_slam/include/Tracking.h:63-114       GrabImageRGBD, Track, Initialization, GetInitModelCam/Obj,
 *                                                         GetSceneFlowObj, DynObjTracking, RenewFrameInfo, UpdateMask,
 *                                                         GetStaticTrack, GetDynamicTrackNew
 *   Frame       vido_slam/include/Frame.h                 RGB-D constructor + the public per-frame lists
 *   Map         vido_slam/include/Map.h:20-104
 *   Optimizer   vido_slam/include/Optimizer.h:22-38       static PoseOptimization*, Partial/FullBatchOptimization
 *   ORBextractor vido_slam/include/ORBextractor.h:33-104
 * Out of scope here: viewer, cvplot, IMU/VIO overloads, GT metrics.  The heavy lifting (ORB, frame
 * lists, LM optimisers, BA) runs on the GPU through vido_c.h; this layer is host orchestration only and has no
 * CPU fallback for those stages. */
#ifndef VIDO_SLAM_FACADE_H
#define VIDO_SLAM_FACADE_H
#if defined(__has_include)
#if __has_include(<opencv2/core.hpp>) && !defined(VIDO_FORCE_CV_COMPAT)
#include <opencv2/core.hpp>
#else
#include "cv_compat.h"
#endif
#else
#include "cv_compat.h"
#endif
#include <map>
#include <string>
#include <utility>
#include <vector>
#include "../vido_c.h"

namespace VIDO_SLAM {

class System; class Tracking; class Map; class Frame;
/* ONE System per process: like the reference, whose Frame keeps the camera intrinsics and grid in static members (Frame.h: fx, fy, cx, cy, mnMinX ..., set by the first
 * frame), this build keeps the tracker's device context, map slot and frame parameters in process-wide state (csrc/facade.cpp: g_ctx / g_slot / g_tp) because the static
 * Optimizer:: methods and the Frame constructor of the reference's interface carry no context argument.  Two Systems grabbing frames in one process would share that state:
 * not supported.  Calls with no live System throw instead of touching a stale context; the C-ABI below the facade (vido_create + the per-stage entry points) has no such limit. */

class ORBextractor {
public:
    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST);
    ~ORBextractor();
    /* image: CV_8UC1; mask is ignored as in the reference (ORBextractor.cc:1034).  Same signature (and, against real OpenCV, the same symbol) as
       vido_slam/include/ORBextractor.h:49; a cv::Mat converts to both proxy types implicitly. */
    void operator()(cv::InputArray image, cv::InputArray mask, std::vector<cv::KeyPoint>& keypoints, cv::OutputArray descriptors);
    int GetLevels() const { return nlevels; }
    float GetScaleFactor() const { return scaleFactor; }
    std::vector<float> GetScaleFactors() const { return mvScaleFactor; }
    vido_ctx* context(int width, int height);          /* lazily created for the first image size seen */
    /* Tracking::GrabImageRGBD's cvtColor (Tracking.cc:327-340) fused into the extractor: the next operator() call whose `image` is `gray`
       converts `color` (CV_8UC3 / CV_8UC4) on the device straight into pyramid level 0 and fills `gray` with the result, instead of
       converting on the host and uploading the gray image. */
    void SetColorSource(const cv::Mat& color, bool rgb_order, const cv::Mat& gray) { color_ = &color; rgb_ = rgb_order; gray_ = gray.data; }
    /* extension: the next operator() call reads a DEVICE-resident image (u8, `channels` = 1 / 3 / 4 interleaved, rows contiguous) instead of `image`'s host pixels —
       the in-process network -> tracker hand-over (Tracking::GrabImageRGBDDevice); `image` then only carries the size. */
    void SetDeviceSource(const void* dev_pixels, int channels, bool rgb_order) { dev_src_ = dev_pixels; dev_ch_ = channels; rgb_ = rgb_order; }
    /* extension: put the extraction of a device-resident colour image on the GPU now (vido_orb_prefetch_color); the operator() call for the same image then only collects */
    void PrefetchDevice(const void* dev_pixels, int channels, bool rgb_order, int width, int height, void* ready_event);
    int nfeatures; float scaleFactor; int nlevels, iniThFAST, minThFAST;
private:
    std::vector<float> mvScaleFactor;
    vido_ctx* ctx_ = nullptr; int w_ = 0, h_ = 0;
    const cv::Mat* color_ = nullptr; bool rgb_ = false; const unsigned char* gray_ = nullptr;
    const void* dev_src_ = nullptr; int dev_ch_ = 0;
};

class Frame {
public:
    Frame() {}
    /* RGB-D constructor, Frame.cc:36-241.  UseSampleFea == 1 (Option II, Frame.cc:101-150): the static candidates come from SampleKeyPoints instead of the ORB keypoints. */
    Frame(const cv::Mat& imGray, const cv::Mat& imDepth, const cv::Mat& imFlow, const cv::Mat& maskSEM, const double& timeStamp,
          ORBextractor* extractor, cv::Mat& K, cv::Mat& distCoef, const float& bf, const float& thDepth, const float& thDepthObj, const int& UseSampleFea);
    void SetPose(cv::Mat Tcw);
    /* Frame.cc:888-956: 3000 points, one per cell of a 20 x 20 grid per round, integer coordinates, listed cell by cell.  The reference seeds cv::RNG with time(NULL)
     * (not reproducible); this build draws from a counter-based generator seeded with the frame id, so a run can be repeated. */
    std::vector<cv::KeyPoint> SampleKeyPoints(const int& rows, const int& cols);
    cv::Mat GetRotationInverse() const { return mRwc.clone(); }
    cv::Mat GetCameraCenter() const { return mOw.clone(); }
    cv::Mat UnprojectStereoStat(const int& i, const bool& addnoise);      /* Frame.cc:706-737; addnoise = 1: one cv::RNG(time(NULL)) draw on the depth (vido_depth_noise) */
    cv::Mat UnprojectStereoObject(const int& i, const bool& addnoise);    /* Frame.cc:739-771 */
    cv::Mat ObtainFlowDepthCamera(const int& i, const bool& addnoise);    /* Frame.cc:833-858: (flow_x, flow_y, depth) */
    cv::Mat ObtainFlowDepthObject(const int& i, const bool& addnoise);    /* Frame.cc:860-886 */

    static long unsigned int nNextId;
    long unsigned int mnId = 0; double mTimeStamp = 0;
    float fx = 0, fy = 0, cx = 0, cy = 0, invfx = 0, invfy = 0, mbf = 0, mThDepth = 0, mThDepthObj = 0;
    cv::Mat mK, mDistCoef, mTcw, mRcw, mtcw, mRwc, mOw, mInitModel;
    int N = 0, N_s = 0, N_s_tmp = 0;
    std::vector<cv::KeyPoint> mvKeys, mvKeysUn; cv::Mat mDescriptors;
    std::vector<cv::KeyPoint> mvStatKeysTmp, mvCorres, mvStatKeys; std::vector<cv::Point2f> mvFlowNext;
    std::vector<float> mvStatDepthTmp, mvStatDepth; std::vector<cv::Mat> mvStat3DPointTmp;
    std::vector<cv::KeyPoint> mvObjKeys, mvObjCorres; std::vector<cv::Point2f> mvObjFlowNext; std::vector<float> mvObjDepth;
    std::vector<cv::Mat> mvObj3DPoint; std::vector<int> vSemObjLabel, vObjLabel;
    std::vector<cv::Point3f> vFlow_3d;
    std::vector<int> nModLabel, nSemPosition, nStaInlierID, nDynInlierID; std::vector<bool> bObjStat;
    std::vector<cv::Mat> vObjMod, vObjCentre3D; std::vector<cv::Point2f> vSpeed;
    std::vector<std::vector<int> > vnObjID, vnObjInlierID;
    Frame *mpPrevFrame = nullptr, *mpNextFrame = nullptr;
};

class Map {
public:
    Map() {}
    void reset();
    void AddFrame(Frame* f) { vpFrames.push_back(f); }
    int GetFramesInMapSize() { return (int)vpFrames.size(); }
    std::vector<std::vector<cv::KeyPoint> > vpFeatSta, vpFeatDyn;
    std::vector<std::vector<float> > vfDepSta, vfDepDyn;
    std::vector<std::vector<cv::Mat> > vp3DPointSta, vp3DPointDyn;
    std::vector<std::vector<int> > vnAssoSta, vnAssoDyn, vnFeatLabel;
    std::vector<std::vector<std::pair<int, int> > > TrackletSta, TrackletDyn; std::vector<int> nObjID;
    /* Incremental tracklet store (extension).  UpdateTracklets() consumes the rows of vnAssoSta / vnAssoDyn added since the last
       call and appends to TrackletSta / TrackletDyn / nObjID in place - same content as the reference's per-frame full rebuild (Tracking.cc:2514-2720).
       vnTrkSta[f][k] / vnPosSta[f][k]: tracklet (of length >= 3) that owns feature k of frame f and its position in it, -1 if none; same for Dyn. */
    void UpdateTracklets();
    std::vector<std::vector<int> > vnTrkSta, vnPosSta, vnTrkDyn, vnPosDyn;
    std::vector<int> trkPreSta, trkPreDyn; size_t trkRowsSta = 0, trkRowsDyn = 0;
    /* Device-resident local-BA window (extension; SURVEY.md 8f row 2, csrc/bawin.hip): the static features of the last frames live on the device, PartialBatchOptimization
       sends only the new frame and the label changes below.  vp3DPointSta rows of the frames still in the device ring may be stale on the host until SyncPointsFromDevice()
       (called before anything on the host reads them: FullBatchOptimization, the host-walk check). */
    std::vector<int> trkChangesSta;           /* (frame, feature, tracklet, position) quads written by UpdateTracklets since the last PartialBatchOptimization */
    int devFramesPushed = 0; bool devWindow = false; bool devWindowDisabled = false;   /* disabled: this map's sequence does not fit the ring (a frame with > 8192 static features) */
    void SyncPointsFromDevice();              /* waits for a window solve still in flight (Tracking::Track runs it beside the next frame), then copies */
    void SyncPointsFromDeviceNow();           /* internal: the copy alone (the solve's own thread) */
    std::vector<cv::Mat> vmCameraPose, vmCameraPose_RF, vmCameraPose_GT;
    std::vector<std::vector<cv::Mat> > vmRigidCentre, vmRigidMotion, vmRigidMotion_RF;
    std::vector<std::vector<int> > vnRMLabel, vnSMLabel; std::vector<std::vector<bool> > vbObjStat;
    std::vector<float> fLBA_time; std::vector<std::vector<float> > vfAll_time;
protected:
    std::vector<Frame*> vpFrames;
};

class Optimizer {
public:
    static int PoseOptimizationNew(Frame* pCurFrame, Frame* pLastFrame, std::vector<int>& TemperalMatch);
    static int PoseOptimizationFlow2Cam(Frame* pCurFrame, Frame* pLastFrame, std::vector<int>& TemperalMatch);
    static cv::Mat PoseOptimizationObjMot(Frame* pCurFrame, Frame* pLastFrame, const std::vector<int>& ObjId, std::vector<int>& InlierID);
    static cv::Mat PoseOptimizationFlow2(Frame* pCurFrame, Frame* pLastFrame, const std::vector<int>& ObjId, std::vector<int>& InlierID);
    /* extension: PoseOptimizationFlow2 (joint) / PoseOptimizationObjMot for all dynamic objects of the frame in one launch; same results as the per-object calls */
    static std::vector<cv::Mat> PoseOptimizationObjectsBatch(Frame* pCurFrame, Frame* pLastFrame, const std::vector<std::vector<int> >& ObjIds, const std::vector<cv::Mat>& InitModels,
                                                             std::vector<std::vector<int> >& InlierIDs, bool joint);
    static void FullBatchOptimization(Map* pMap, const cv::Mat Calib_K);
    static void PartialBatchOptimization(Map* pMap, const cv::Mat Calib_K, const int WINDOW_SIZE);
    static cv::Mat Get3DinWorld(const cv::KeyPoint& Feats2d, const float& Dpts, const cv::Mat& Calib_K, const cv::Mat& CameraPose);
    static cv::Mat Get3DinCamera(const cv::KeyPoint& Feats2d, const float& Dpts, const cv::Mat& Calib_K);
};

class Converter {
public:
    static cv::Mat toInvMatrix(const cv::Mat& T);       /* Converter.cc:155-170: rigid inverse of a 4x4 CV_32F */
};

class Tracking {
public:
    Tracking(System* pSys, Map* pMap, const std::string& strSettingPath, const int sensor);
    ~Tracking();
    cv::Mat GrabImageRGBD(const cv::Mat& imRGB, cv::Mat& imD, const cv::Mat& imFlow, const cv::Mat& maskSEM, const cv::Mat& mTcw_gt,
                          const std::vector<std::vector<float> >& vObjPose_gt, const double& timestamp, cv::Mat& imTraj, const int& nImage);
    /* extension (replaces the three service round trips of src/realtime_demo/src/run_vido.cc:57-171): the same call with the image and the three
       maps RESIDENT ON THE DEVICE — u8 image (1 / 3 / 4 channels), depth CV_32F (raw sensor units, rescaled in place like the host form), flow CV_32FC2, mask CV_32SC1 as
       plain device pointers of a width x height frame.  Nothing is uploaded and no map is downloaded: the host-side stages read the map values at their few thousand
       candidate points through device gathers.  ready_event: a hipEvent_t the producer recorded after writing the buffers (the tracker's stream waits for it; may be null). */
    void PrefetchImageDevice(const void* im_dev, int channels, int width, int height, void* image_ready_event);
    cv::Mat GrabImageRGBDDevice(const void* im_dev, int channels, int width, int height, float* depth_dev, const float* flow_dev, const int* mask_dev, void* ready_event,
                                const double& timestamp, const int& nImage);
    void Track();
    void Initialization();
    void GetSceneFlowObj();
    std::vector<std::vector<int> > DynObjTracking();
    cv::Mat GetInitModelCam(const std::vector<int>& MatchId, std::vector<int>& MatchId_sub);
    cv::Mat GetInitModelObj(const std::vector<int>& ObjId, std::vector<int>& ObjId_sub, const int objid);
    /* extension: GetInitModelObj for all objects of the frame with one batched RANSAC launch (same seeds and results) */
    std::vector<cv::Mat> GetInitModelObjBatch(const std::vector<std::vector<int> >& ObjIds, std::vector<std::vector<int> >& ObjIds_sub);
    std::vector<std::vector<std::pair<int, int> > > GetStaticTrack();
    std::vector<std::vector<std::pair<int, int> > > GetDynamicTrackNew();
    void RenewFrameInfo(const std::vector<int>& TM_sta);
    void UpdateMask();

    enum eTrackingState { NO_IMAGES_YET = 0, NOT_INITIALIZED = 1, OK = 2 };
    enum eDataState { OMD = 1, KITTI = 2, KAIST = 3 };
    eTrackingState mState; eDataState mTestData; int mSensor;
    bool bJoint;                      /* never initialised in the reference; explicit here, default true */
    int f_id, max_id, StopFrame;
    Frame *mpCurrentFrame = nullptr, *mpLastFrame = nullptr;
    cv::Mat mImGray, mImGrayLast, mDepthMap, mFlowMap, mFlowMapLast, mSegMap, mSegMapLast, mK, mDistCoef, mVelocity;
    float mbf, mThDepth, mThDepthObj, mDepthMapFactor, mScale, fSFMgThres, fSFDsThres;
    int nMaxTrackPointBG, nMaxTrackPointOBJ, nWINDOW_SIZE, nOVERLAP_SIZE, nUseSampleFea; bool mbRGB;
    std::vector<int> TemperalMatch, TemperalMatch_subset;
    std::vector<cv::KeyPoint> mvTmpObjKeys, mvTmpObjCorres; std::vector<float> mvTmpObjDepth; std::vector<int> mvTmpSemObjLabel; std::vector<cv::Point2f> mvTmpObjFlowNext;
    std::vector<float> all_timing;
    float ms_total = 0, ms_update_mask = 0, ms_frame = 0, ms_obj_motion_sum = 0, ms_wait_inputs = 0;     /* wall-clock stage times of the last GrabImageRGBD */
    unsigned ransac_seed;             /* solvePnPRansac uses OpenCV's global RNG; seeded explicitly here */
    ORBextractor* mpORBextractorLeft = nullptr;
protected:
    System* mpSystem; Map* mpMap;
    int slot_cur_ = 0;
    cv::Mat GrabCommon(const double& timestamp, const int& nImage, void* t_grab);      /* everything after the maps are in the slot and mImGray is set */
};

class System {
public:
    enum eSensor { MONOCULAR = 0, STEREO = 1, RGBD = 2, IMU_RGBD = 3 };
    System() {}
    ~System();
    void Init(const std::string& strSettingsFile, const eSensor sensor);
    cv::Mat TrackRGBD(const cv::Mat& im, cv::Mat& depthmap, const cv::Mat& flowmap, const cv::Mat& masksem, const cv::Mat& mTcw_gt,
                      const std::vector<std::vector<float> >& vObjPose_gt, const double& timestamp, cv::Mat& imTraj, const int& nImage);
    /* extension: TrackRGBD on device-resident inputs (Tracking::GrabImageRGBDDevice) */
    cv::Mat TrackRGBDDevice(const void* im_dev, int channels, int width, int height, float* depth_dev, const float* flow_dev, const int* mask_dev, void* ready_event,
                            const double& timestamp, const int& nImage);
    /* extension: the ORB extraction of the NEXT frame's device-resident image, enqueued ahead of TrackRGBDDevice (it needs nothing but the image: a pipeline calls this while it
       still waits for the frame's depth / flow / mask); image_ready_event (hipEvent_t or NULL) orders it behind the image's producer */
    void PrefetchImageDevice(const void* im_dev, int channels, int width, int height, void* image_ready_event);
    void SaveResultsIJRR2020(const std::string& filename);
    Map* GetMap() { return mpMap; }
    Tracking* GetTracker() { return mpTracker; }
private:
    eSensor mSensor = RGBD; Map* mpMap = nullptr; Tracking* mpTracker = nullptr;
};

namespace detail {
void SetZeroCopyMaps(bool on);                        /* TrackRGBDDevice adopts the caller's map buffers instead of copying them (the caller keeps a frame's maps alive for two more calls) */
void SetDepthNoiseSeed(unsigned seed);                 /* addnoise = 1 draws: seed != 0 pins cv::RNG's seed (tests, reproducible runs); 0 = (unsigned)time(NULL) as the reference (Frame.cc:711) */
float LastFrameStageMs(int which);                     /* wall time inside the last Frame constructor: 0 = extractor call, 1 = static / object lists */
void ResidentCheckStats(int* checks, int* mismatches);  /* VIDO_BA_RESIDENT_CHECK=1: windows solved both ways (device-resident window / Map walk) and how many disagreed */
vido_ctx* Context();                                   /* the process-wide ctx of the live System (one System per process, as in the reference) */
std::map<std::string, std::string> ParseSettings(const std::string& path);    /* OpenCV-YAML 1.0 `key: value` subset */
}

}  // namespace VIDO_SLAM
#endif
