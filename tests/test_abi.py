"""CPU-side checks of the drop-in boundary: the C-ABI library loads without a GPU and exports
exactly what include/xrs_hip.h declares; the host layer refuses to compute without a device."""
import ctypes
import os
import re

import numpy as np
import pytest

import __graft_entry__ as entry

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    entry.build()
    from xrspatial_amd import _lib
    return _lib


def _declared():
    text = open(os.path.join(ROOT, "include", "xrs_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(xrs_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(built):
    lib = ctypes.CDLL(built.LIB_PATH)
    names = _declared()
    assert len(names) >= 40
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/xrs_hip.h but not exported"


def test_python_binding_covers_the_header(built):
    assert sorted(built.EXPORTED) == _declared()


def test_version_and_error_channel(built):
    lib = built.load()
    assert lib.xrs_version() == 1
    # argument validation happens before any device work, so it is testable on CPU
    rc = lib.xrs_slope_f32(None, None, 4, 4, 4, 4, 1.0, 1.0, 0, 0, None)
    assert rc != 0 and "null pointer" in built.last_error()
    k = np.ones((2, 3))
    rc = lib.xrs_convolve2d_f32(ctypes.c_void_p(16), ctypes.c_void_p(16), 4, 4, 4, 4, k.ctypes.data, 2, 3,
                                ctypes.c_void_p(16), 0, 0, None)
    assert rc != 0 and "odd x odd" in built.last_error()
    assert lib.xrs_kxk_workspace_bytes(5, 5) == 200


def test_no_cpu_fallback(built):
    import xrspatial_amd as xs
    if xs.has_hip():
        pytest.skip("a GPU is present")
    agg = xs.DataArray(np.zeros((8, 8), np.float32), attrs={'res': (1, 1)})
    for call in (lambda: xs.slope(agg), lambda: xs.hillshade(agg), lambda: xs.ndvi(agg, agg),
                 lambda: xs.focal.mean(agg), lambda: xs.zonal_stats(agg, agg, stats_funcs=['mean'])):
        with pytest.raises(xs.XrsError):
            call()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "xrspatial_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "xrs_oracle" not in src, f
                assert "fake_hip" not in src, f             # (the CPU emulation of the C ABI is for tests only)


def test_oracle_is_used_only_by_tests_smoke_and_the_cpu_baseline():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    imp = re.compile(r"^\s*(from|import)\s+oracle\b", flags=re.M)
    for f in os.listdir(os.path.join(ROOT, "tools")):
        if f.endswith(".py"):
            assert not imp.search(open(os.path.join(ROOT, "tools", f)).read()), f
    bench = open(os.path.join(ROOT, "bench.py")).read()
    hits = [m.start() for m in imp.finditer(bench)]
    leg = bench.index("def cpu_baseline")
    assert hits and all(h > leg for h in hits)              # every import sits inside the cpu_baseline function
    entry = open(os.path.join(ROOT, "__graft_entry__.py")).read()
    assert all(h > entry.index("def smoke") for h in [m.start() for m in imp.finditer(entry)])


def test_build_id_covers_every_source_and_header():
    """xrs_build_id() is a hash of the files the Makefile lists (SRCS + SRCS_AB + HDRS): bench.py, the PMC table and every log key
    their numbers by it, so a kernel source or header that is not listed would change the library without changing the id."""
    import glob
    import re
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "xrspatial_amd", "csrc")
    mk = open(os.path.join(csrc, "Makefile")).read().replace("\\\n", " ")
    listed = set()
    for var in ("SRCS", "SRCS_AB", "HDRS"):
        m = re.search(r"^%s\s*=\s*(.*)$" % var, mk, re.M)
        assert m, var
        listed |= {os.path.basename(w) for w in m.group(1).split()}
    on_disk = {os.path.basename(p) for ext in ("*.hip", "*.h") for p in glob.glob(os.path.join(csrc, ext))}
    assert on_disk <= listed, f"not in the Makefile's SRCS / HDRS (so not in the build id): {sorted(on_disk - listed)}"
    # ... and every quoted include resolves to a listed file
    for name in on_disk:
        for inc in re.findall(r'^\s*#include\s+"([^"]+)"', open(os.path.join(csrc, name)).read(), re.M):
            base = os.path.basename(inc)
            assert base in listed or base == "build_id.h", f"{name} includes {inc}, which the Makefile does not list"
