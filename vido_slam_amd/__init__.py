"""Import shim: the product package lives in the directory `vido-slam_amd/` (a hyphen is not a legal
Python identifier), so `import vido_slam_amd` resolves its sub-modules from there."""
import os as _os
__path__ = [_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "vido-slam_amd")]
from .host import *  # noqa: F401,F403,E402
from . import problems, synth, g2o_io  # noqa: F401,E402
