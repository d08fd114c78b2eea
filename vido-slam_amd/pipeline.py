"""In-process network -> tracker hand-off.

The reference's realtime demo (src/realtime_demo/src/run_vido.cc:57-171) asks three ROS services for the optical flow (TYPE_32FC2), the
depth (MONO16) and the instance mask (MONO8) of every frame: the image goes to each node over the wire, each node uploads it, downloads its
result and sends it back, and TrackRGBD uploads the three maps again.  Here the networks and the tracker share one device: the frame is
uploaded once, the three networks write device tensors, and `vido_frontend_batch` takes those buffers as they are (maps_on_device = 2: the
tracker's map slots alias the caller's tensors, which this object keeps alive for the two frames the tracker may still read them).
Nothing is quantised on the way except what the reference's interfaces quantise themselves (depth to 16 bit, mask to 8 bit), so the
front-end results are identical to the round trip through host arrays (tests/test_pipeline_gpu.py).
"""
import numpy as np
import torch

from . import nets as _nets
from .host import FrameFeatures


def bgr_to_gray(bgr):
    """cvtColor(BGR2GRAY) in 14-bit fixed point on the device: (B*1868 + G*9617 + R*4899 + 8192) >> 14 (Tracking.cc:327-340; same formula as
    the facade's to_gray)."""
    x = bgr.to(torch.int32)
    return ((x[..., 0] * 1868 + x[..., 1] * 9617 + x[..., 2] * 4899 + 8192) >> 14).to(torch.uint8).contiguous()


class NetFrontEnd:
    """RunNet + the Frame construction of the frame that follows (run_vido.cc:138-171, Frame.cc:41-230).

    push(bgr) -> None for the first frame, then a dict with the front-end lists of the frame (FrameFeatures.frontend_batch views: keypoints,
    descriptors, static candidates, dense object samples) plus `labels` (Mask R-CNN class indices) and `slot` (the tracker map slot the
    frame's depth / flow / mask now live in, for vido_gather_* / vido_update_mask)."""

    def __init__(self, ctx, frame_params, flow_net, depth_net, mask_net, mask_feed=(1088, 800), depth_feed=(192, 640), confidence=0.8, keep_raw=False):
        self.ctx = ctx; self.flow_net, self.depth_net, self.mask_net = flow_net, depth_net, mask_net
        self.ff = FrameFeatures(ctx, frame_params)
        self.mask_feed, self.depth_feed, self.confidence = mask_feed, depth_feed, confidence
        self.dev = next(flow_net.parameters()).device
        self.prev = None
        self.slot = 0
        self._hold = {}                    # slot -> tensors the tracker's slot aliases
        self.keep_raw, self.raw = keep_raw, None      # keep_raw: copies of the network outputs of the last frame (the tracker rescales the depth map in place)

    @torch.no_grad()
    def infer(self, prev_bgr, cur_bgr):
        """The three service calls of RunNet on device tensors: flow HxWx2 f32, depth HxW f32 (MONO16 values), mask HxW i32, labels."""
        flow = _nets.analyse_flow(self.flow_net, prev_bgr, cur_bgr)
        mask_u8, labels = _nets.analyse_image(self.mask_net, cur_bgr, feed=self.mask_feed, confidence=self.confidence)
        depth_u16 = _nets.analyse_depth(self.depth_net, cur_bgr, feed=self.depth_feed)
        return flow.contiguous(), depth_u16.to(torch.float32).contiguous(), mask_u8.to(torch.int32).contiguous(), labels

    @torch.no_grad()
    def push(self, bgr):
        cur = torch.as_tensor(np.ascontiguousarray(bgr, np.uint8)).to(self.dev, non_blocking=True)      # the only upload of the frame
        if self.prev is None:
            self.prev = cur
            return None
        # RunNet's queue entry: the CURRENT image with the flow of the pair (previous, current) and the depth / mask of the current image
        flow, depth, mask, labels = self.infer(self.prev, cur)
        gray = bgr_to_gray(cur)
        if self.keep_raw: self.raw = (flow.clone(), depth.clone(), mask.clone())
        h, w = gray.shape
        torch.cuda.current_stream().synchronize()          # the tracker runs on the ctx's own stream
        slot = self.slot
        out = self.ff.frontend_batch(slot, (gray.data_ptr(), 1, h, w, h * w, w), depth.data_ptr(), flow.data_ptr(), mask.data_ptr(), alias=True)
        self._hold[slot] = (gray, depth, flow, mask)        # alive until the slot is overwritten (two frames later)
        self.slot ^= 1
        self.prev = cur
        out = dict(out); out["labels"] = labels; out["slot"] = slot
        return out
