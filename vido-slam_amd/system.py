"""`System` — ctypes mirror of VIDO_SLAM::System (vido_slam/include/System.h:72-114) over the C handle of include/vido_c.h
(vido_system_*).  Same three calls as the reference's callers make (vido_slam/demo/run_vido_slam.cc:82-135,
src/realtime_demo/src/run_vido.cc:229-235, 275):

    slam = System(); slam.Init("config.yaml", System.RGBD)
    Tcw = slam.TrackRGBD(im, depth, flow, mask, Tcw_gt, objpose_gt, timestamp, imTraj, nImage)
    slam.SaveResultsIJRR2020(prefix)

TrackRGBD releases the GIL for the whole call (ctypes), so a pipeline can run the three networks for frame k+1 from another
Python thread while frame k is being tracked (vido_slam_amd/pipeline.py::EndToEnd).
"""
import ctypes as C
import os
import numpy as np

from .host import load_library, VidoError, VIDO_OK


class SystemStats(C.Structure):
    """vido_system_stats."""
    _fields_ = [("frame_id", C.c_int32), ("n_keypoints", C.c_int32), ("n_static", C.c_int32), ("n_static_inliers", C.c_int32),
                ("n_objects", C.c_int32), ("n_object_points", C.c_int32), ("ba_window", C.c_int32), ("pad", C.c_int32),
                ("ms_total", C.c_float), ("ms_update_mask", C.c_float), ("ms_frame", C.c_float), ("ms_cam_pose", C.c_float),
                ("ms_obj_tracking", C.c_float), ("ms_obj_motion", C.c_float), ("ms_renew", C.c_float), ("ms_local_ba", C.c_float), ("ms_wait_inputs", C.c_float), ("ms_orb", C.c_float), ("ms_lists", C.c_float)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "pad"}


def _bind(lib):
    if getattr(lib, "_vido_system_bound", False):
        return lib
    lib.vido_system_create.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
    lib.vido_system_destroy.argtypes = [C.c_void_p]
    lib.vido_system_last_error.restype = C.c_char_p
    lib.vido_system_last_error.argtypes = [C.c_void_p]
    lib.vido_system_track_rgbd.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_void_p]
    lib.vido_system_track_rgbd_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_void_p]
    lib.vido_system_get_stats.argtypes = [C.c_void_p, C.POINTER(SystemStats)]
    lib.vido_system_save_results.argtypes = [C.c_void_p, C.c_char_p]
    lib.vido_system_context.restype = C.c_void_p
    lib.vido_system_context.argtypes = [C.c_void_p]
    lib.vido_system_set_depth_noise_seed.argtypes = [C.c_void_p, C.c_uint]
    lib.vido_system_set_zero_copy_maps.argtypes = [C.c_void_p, C.c_int]
    lib.vido_system_prefetch_image_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib._vido_system_bound = True
    return lib


class System:
    MONOCULAR, STEREO, RGBD, IMU_RGBD = 0, 1, 2, 3      # System::eSensor (System.h:76-81)

    def __init__(self):
        self.lib = _bind(load_library())
        self.h = None
        self._keep = None                                 # the arrays of the last call (the tracker holds shallow references to them)

    def Init(self, strSettingsFile, sensor=2):
        if sensor != self.RGBD:
            raise VidoError(-1, "System.Init: only the RGBD sensor path is built (IMU_RGBD / VIO is out of scope)")
        h = C.c_void_p()
        rc = self.lib.vido_system_create(os.fsencode(strSettingsFile), C.byref(h))
        if rc != VIDO_OK:
            raise VidoError(rc, self.lib.vido_system_last_error(None).decode())
        self.h = h

    def TrackRGBD(self, im, depthmap, flowmap, masksem, mTcw_gt=None, vObjPose_gt=None, timestamp=0.0, imTraj=None, nImage=10000):
        """im u8 (H,W) / (H,W,3|4); depthmap f32 (H,W) — REWRITTEN IN PLACE with the pre-scaled depth (Tracking.cc:299-322);
        flowmap f32 (H,W,2); masksem i32 (H,W).  Returns Tcw (4,4) f32.  mTcw_gt / vObjPose_gt / imTraj: accepted and ignored
        (ground-truth metrics and the trajectory canvas are viewer / evaluation features)."""
        if self.h is None:
            raise VidoError(-1, "System.TrackRGBD before Init")
        if im.dtype != np.uint8 or not im.flags.c_contiguous:
            im = np.ascontiguousarray(im, np.uint8)
        if depthmap.dtype != np.float32 or not depthmap.flags.c_contiguous or not depthmap.flags.writeable:
            raise VidoError(-1, "TrackRGBD: depthmap must be a writable C-contiguous float32 array (it is rescaled in place)")
        if flowmap.dtype != np.float32 or not flowmap.flags.c_contiguous:
            flowmap = np.ascontiguousarray(flowmap, np.float32)
        if masksem.dtype != np.int32 or not masksem.flags.c_contiguous:
            masksem = np.ascontiguousarray(masksem, np.int32)
        h, w = im.shape[:2]
        cn = 1 if im.ndim == 2 else im.shape[2]
        T = np.empty((4, 4), np.float32)
        keep = (im, depthmap, flowmap, masksem)
        rc = self.lib.vido_system_track_rgbd(self.h, im.ctypes.data, cn, w, h, depthmap.ctypes.data, flowmap.ctypes.data, masksem.ctypes.data,
                                             float(timestamp), int(nImage), T.ctypes.data)
        self._keep = keep
        if rc != VIDO_OK:
            raise VidoError(rc, self.lib.vido_system_last_error(self.h).decode())
        return T

    def PrefetchImageDevice(self, im_dev, channels, width, height, image_ready_event=None):
        """Put the ORB extraction of the next TrackRGBDDevice call's image on the tracker's stream now (it needs nothing but the image): call it, on the thread that tracks,
        while still waiting for the frame's networks; the track call for the same pointer then only collects (vido_system_prefetch_image_device)."""
        rc = self.lib.vido_system_prefetch_image_device(self.h, C.c_void_p(im_dev), int(channels), int(width), int(height), C.c_void_p(image_ready_event) if image_ready_event else None)
        if rc != 0:
            raise VidoError(rc, self.lib.vido_system_last_error(self.h).decode())

    def SetZeroCopyMaps(self, on=True):
        """TrackRGBDDevice adopts the three device map buffers instead of copying them into the tracker's slots (6.1 MB of device-to-device copies per 640x480 frame): the
        caller keeps a frame's maps alive and untouched until the call after the next one has returned (vido_system_set_zero_copy_maps)."""
        rc = self.lib.vido_system_set_zero_copy_maps(self.h, int(bool(on)))
        if rc != 0:
            raise VidoError(rc, "vido_system_set_zero_copy_maps")

    def TrackRGBDDevice(self, im_dev, channels, width, height, depth_dev, flow_dev, mask_dev, ready_event=None, timestamp=0.0, nImage=10000):
        """System::TrackRGBDDevice (extension): the same call on DEVICE-resident buffers given as raw pointers (e.g. tensor.data_ptr()): u8 image with
        `channels` interleaved channels, depth f32 (rescaled in place on the device), flow f32 x2, mask i32 of a width x height frame; ready_event: raw hipEvent_t
        (torch.cuda.Event.cuda_event) recorded by the producer of the buffers, or None.  The caller keeps the buffers alive for two frames.  Returns Tcw (4,4) f32."""
        if self.h is None:
            raise VidoError(-1, "System.TrackRGBDDevice before Init")
        T = np.empty((4, 4), np.float32)
        rc = self.lib.vido_system_track_rgbd_device(self.h, C.c_void_p(int(im_dev)), int(channels), int(width), int(height), C.c_void_p(int(depth_dev)), C.c_void_p(int(flow_dev)),
                                                    C.c_void_p(int(mask_dev)), C.c_void_p(int(ready_event)) if ready_event else None, float(timestamp), int(nImage), T.ctypes.data)
        if rc != VIDO_OK:
            raise VidoError(rc, self.lib.vido_system_last_error(self.h).decode())
        return T

    def set_depth_noise_seed(self, seed):
        """Pins the seed of the reference's time(NULL)-seeded depth noise (Frame.cc:711-716; 0 = the reference behaviour): reproducible PoseOptimizationNew results."""
        rc = self.lib.vido_system_set_depth_noise_seed(self.h, int(seed) & 0xffffffff)
        if rc != 0:
            raise VidoError(rc, "set_depth_noise_seed: no system")

    def stats(self):
        s = SystemStats()
        self.lib.vido_system_get_stats(self.h, C.byref(s))
        return s.as_dict()

    def SaveResultsIJRR2020(self, filename=""):
        rc = self.lib.vido_system_save_results(self.h, os.fsencode(filename))
        if rc != VIDO_OK:
            raise VidoError(rc, self.lib.vido_system_last_error(self.h).decode())

    def close(self):
        if self.h is not None:
            self.lib.vido_system_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
