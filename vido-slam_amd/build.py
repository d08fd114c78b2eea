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


def _obj_stamp(src):
    h = hashlib.sha1()
    h.update(open(src, "rb").read())
    for p in sorted(glob.glob(os.path.join(CSRC, "*.hpp"))) + sorted(glob.glob(os.path.join(HERE, "..", "include", "*.h"))) + sorted(glob.glob(os.path.join(HERE, "..", "include", "vido_slam", "*.h"))):
        h.update(open(p, "rb").read())
    h.update(" ".join(FLAGS + _file_flags(src)).encode())
    return h.hexdigest()


# per-source flags.  convdirect: matrix-instruction accumulators in VGPRs (the kernel needs < 128 registers; in AGPR form the compiler copies the 16-32 accumulators
# AGPR -> VGPR -> AGPR around every tap's loop: ~130 moves per tap)
FILE_FLAGS = {"convdirect": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


_PROBED = {}


def _flag_ok(flags):
    """True when this hipcc accepts `flags` (an LLVM that does not know a `-mllvm` cl::opt aborts with 'Unknown command line argument'): probed once on an empty TU."""
    key = " ".join(flags)
    if key not in _PROBED:
        import tempfile
        with tempfile.TemporaryDirectory() as d:
            src = os.path.join(d, "probe.hip"); open(src, "w").write("__global__ void k() {}\n")
            r = subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-c", "-x", "hip", src, "-o", os.path.join(d, "probe.o")] + flags,
                               stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        _PROBED[key] = r.returncode == 0
    return _PROBED[key]


def _file_flags(src):
    """FILE_FLAGS + VIDO_FLAGS_<stem> (e.g. VIDO_FLAGS_orb="-DFS_MAXIW=37"): experiment flags for one source only, so that a variant build recompiles one object"""
    stem = os.path.basename(src).split(".")[0]
    ff = FILE_FLAGS.get(stem, [])
    if ff and not _flag_ok(ff):
        ff = []                                                          # (an optimisation flag this compiler does not know: build without it)
    return ff + _shlex.split(os.environ.get("VIDO_FLAGS_" + stem, ""))


def build(force=False, verbose=False, lib=None):
    """One object per source (compiled in parallel, cached by content hash under build/), one link."""
    lib = lib or LIB
    stamp_file = lib + ".stamp"
    stamp = _stamp() + "".join(" ".join(_file_flags(x)) for x in sources())
    if not force and os.path.exists(lib) and os.path.exists(stamp_file) and open(stamp_file).read() == stamp:
        return lib
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    cflags = [f for f in FLAGS if f != "-shared"]
    jobs, objs = [], []
    for s in sources():
        o = os.path.join(objdir, os.path.basename(s) + "." + _obj_stamp(s)[:16] + ".o")
        objs.append(o)
        if force or not os.path.exists(o):
            jobs.append([hipcc] + cflags + _file_flags(s) + ["-c", "-x", "hip", s, "-o", o])
    if jobs:
        from concurrent.futures import ThreadPoolExecutor
        def run(cmd):
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            list(ex.map(run, jobs))
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-o", lib] + objs)
    live = set(objs)
    for o in glob.glob(os.path.join(objdir, "*.o")):            # drop the objects of older source versions
        if o not in live and lib == LIB:
            os.remove(o)
    open(stamp_file, "w").write(stamp)
    return lib


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
