"""Builds libvido_slam_hip.so (gfx950 only) in-tree with hipcc.  No JIT cache, no fallback."""
import os, subprocess, glob, hashlib

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libvido_slam_hip.so")
import shlex as _shlex
FLAGS = _shlex.split(os.environ.get("VIDO_EXTRA_FLAGS", "")) + ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-fno-fast-math", "-munsafe-fp-atomics", "-Wall", "-Wno-unused-function", "-Wno-unused-result", "-Wno-unused-value", "-pthread"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp")))


def _stamp():
    h = hashlib.sha1()
    for p in sources() + sorted(glob.glob(os.path.join(CSRC, "*.hpp"))) + sorted(glob.glob(os.path.join(HERE, "..", "include", "*.h"))):
        h.update(open(p, "rb").read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force=False, verbose=False):
    stamp_file = LIB + ".stamp"
    stamp = _stamp()
    if not force and os.path.exists(LIB) and os.path.exists(stamp_file) and open(stamp_file).read() == stamp:
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    srcs = sources()
    cmd = [hipcc] + FLAGS + ["-o", LIB] + sum((["-x", "hip", s] for s in srcs), [])
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    open(stamp_file, "w").write(stamp)
    return LIB


DRIVER = os.path.join(HERE, "run_vido_slam.bin")


def build_driver(force=False):
    """Offline driver (tools/run_vido_slam.cpp) against the in-tree library."""
    src = os.path.join(HERE, "..", "tools", "run_vido_slam.cpp")
    if not force and os.path.exists(DRIVER) and os.path.getmtime(DRIVER) > max(os.path.getmtime(src), os.path.getmtime(LIB)):
        return DRIVER
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    subprocess.check_call([hipcc, "-O2", "-std=c++17", "-x", "c++", src, "-o", DRIVER, "-L" + HERE, "-lvido_slam_hip", "-lz", "-Wl,-rpath," + HERE, "-Wl,-rpath,$ORIGIN"])
    return DRIVER


if __name__ == "__main__":
    print(build(force=True, verbose=True))
