"""Forward-only network nodes of the reference (src/thirdparty/{flow_net,mono_depth2}) on PyTorch-ROCm, with the
reference's hand-written CUDA ops replaced by the HIP ops of libvido_slam_hip.so (vido_correlation, ...).
State-dict keys and tensor shapes are identical to the reference modules, so its checkpoints load unchanged."""
from .ops import HipOps, correlation_torch_reference  # noqa: F401
from .liteflownet import LiteFlowNet, analyse_flow  # noqa: F401
from .monodepth2 import ResnetEncoder18, DepthDecoder, MonoDepth2, analyse_depth  # noqa: F401
from .maskrcnn import MaskRCNN, MaskRCNNConfig, analyse_image, analyse_image_static, paste_masks  # noqa: F401
from .weights import fill_deterministic, deterministic_tensor, fill_maskrcnn, calibrate_detector_scores  # noqa: F401
from .fuse import fold_batchnorm, Graphed  # noqa: F401
