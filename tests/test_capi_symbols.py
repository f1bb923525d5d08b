"""CPU test: the C-ABI library builds, loads, and exports every symbol that
include/gnnflow_hip.h declares (no compute calls without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "gnnflow_hip.h")


def declared_symbols():
    text = open(HEADER).read()
    return sorted(set(re.findall(r"GF_API[^;(]*?\b(gf_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def built_lib():
    from gnnflow_amd import _build
    return _build.build()


def test_header_declares_symbols():
    syms = declared_symbols()
    assert len(syms) >= 35
    assert "gf_sampler_sample" in syms and "gf_cache_fetch" in syms


def test_library_exports_every_declared_symbol(built_lib):
    lib = ctypes.CDLL(built_lib)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, "not exported: {}".format(missing)


def test_ctypes_prototypes_cover_header(built_lib):
    from gnnflow_amd import _capi
    assert sorted(_capi.PROTOTYPES) == declared_symbols()
    lib = _capi.load()
    assert lib.gf_version().startswith(b"gnnflow_amd")


def test_error_path_without_gpu(built_lib):
    """Null handles are rejected with a status code + message, not a crash."""
    from gnnflow_amd import _capi
    lib = _capi.load()
    n = ctypes.c_size_t(0)
    rc = lib.gf_graph_num_edges(None, ctypes.byref(n))
    assert rc == _capi.GF_ERR_INVALID_ARGUMENT
    assert b"null graph" in lib.gf_last_error()


def test_python_layer_rejects_bad_strings():
    """gnnflow/dynamic_graph.py:61-72, temporal_sampler.py:38-39 raise ValueError
    before touching native code."""
    import gnnflow_amd
    with pytest.raises(ValueError):
        gnnflow_amd.DynamicGraph(1 << 20, 1 << 21, "bogus", 64, 128, "insert")
    with pytest.raises(ValueError):
        gnnflow_amd.DynamicGraph(1 << 20, 1 << 21, "cuda", 64, 128, "bogus")


def test_headers_are_plain_c(tmp_path):
    """The boundary must be bindable from C / cgo / JNI: both headers compile as C99 with
    no C++ or torch types in any signature."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        import pytest
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "t.c"
    src.write_text('#include "gnnflow_hip.h"\n#include "gnnflow_rng.h"\n'
                   'int main(void) { return (int)gf_philox4x32_10_first(1, 2, 3) & 0; }\n')
    p = subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I",
                        os.path.join(root, "include"), "-fsyntax-only", str(src)],
                       capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
