"""Experiment builds of the library: python tools/r6/build_variant.py NAME STEM "FLAGS" -> vido-slam_amd/variants/libvido_NAME.so (only csrc/STEM.* is recompiled with FLAGS);
run a tool against it with VIDO_LIB_VARIANT=<that path>."""
import os, sys, importlib.util
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
name, stem, flags = sys.argv[1], sys.argv[2], sys.argv[3]
os.environ["VIDO_FLAGS_" + stem] = flags
spec = importlib.util.spec_from_file_location("b", os.path.join(ROOT, "vido-slam_amd", "build.py")); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
os.makedirs(os.path.join(ROOT, "vido-slam_amd", "variants"), exist_ok=True)
print(b.build(force=False, lib=os.path.join(ROOT, "vido-slam_amd", "variants", "libvido_%s.so" % name)))
