"""CPU: the C-ABI library builds, loads and exports every symbol include/vido_c.h declares.
No compute call is made here (there is no GPU in the build container)."""
import ctypes, os, re, sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "vido_c.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(vido_[a-z0-9_]+)\s*\(", txt)))


def test_header_declares_something():
    syms = declared_symbols()
    assert "vido_create" in syms and "vido_orb_extract" in syms and len(syms) >= 10


def test_library_builds_and_exports_all_symbols():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    ge.build()
    import vido_slam_amd
    lib = ctypes.CDLL(vido_slam_amd.LIB_PATH)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing


def test_create_fails_loudly_without_gpu():
    import vido_slam_amd
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        pytest.skip("GPU present")
    with pytest.raises(vido_slam_amd.VidoError) as e:
        vido_slam_amd.Context()
    assert e.value.code in (-2, -3)


def test_product_never_touches_oracle():
    """The oracle is test infrastructure: nothing under vido-slam_amd/ or include/ may reference it."""
    bad = []
    for base in ("vido-slam_amd", "include", "vido_slam_amd"):
        for dp, _, fns in os.walk(os.path.join(ROOT, base)):
            for fn in fns:
                if fn.endswith((".py", ".hip", ".cpp", ".hpp", ".h")):
                    t = open(os.path.join(dp, fn), errors="ignore").read()
                    if re.search(r"pyoracle|vido_oracle|libvido_oracle|from oracle|import oracle|oracle/", t):
                        bad.append(os.path.join(dp, fn))
    assert not bad, bad
