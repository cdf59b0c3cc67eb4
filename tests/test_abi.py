"""The C-ABI library builds for gfx950, loads, and exports every symbol include/bioreason_hip.h declares
(no compute calls here: this runs without a GPU)."""
import ctypes
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_hip_library_exports_header_symbols():
    from bioreason_amd import _lib
    subprocess.run(["make", "-s", "-j8", "-C", os.path.join(ROOT, "bioreason_amd", "csrc"), "all"], check=True)
    protos = _lib.parse_header()
    assert len(protos) >= 40
    dll = ctypes.CDLL(os.path.join(ROOT, "bioreason_amd", "libbioreason_hip.so"))
    for name in protos:
        assert hasattr(dll, name), name
    for name, args in protos.items():           # every argument has a C type the binding understands
        for kind, _ in args:
            assert kind in ("ptr", "int", "long", "float", "unsigned"), (name, kind)


def test_product_library_has_no_knobs_and_the_debug_library_has_them():
    """probes, tile-variant knobs and the persistent decode step live in include/bioreason_hip_debug.h and exist only in
    libbioreason_hip_debug.so (-DBRA_DEBUG); the product library exports none of them — no process-wide mutable state to bind"""
    from bioreason_amd import _lib
    csrc = os.path.join(ROOT, "bioreason_amd", "csrc")
    subprocess.run(["make", "-s", "-j8", "-C", csrc, "all", "debug"], check=True)
    pub, dbg = _lib.parse_header(), _lib.parse_header(_lib._DEBUG_HEADER)
    assert len(dbg) >= 12 and not set(pub) & set(dbg)
    for name in pub:
        assert "probe" not in name and "_set_" not in name and "stamps" not in name and "persist" not in name, name
    product = ctypes.CDLL(os.path.join(ROOT, "bioreason_amd", "libbioreason_hip.so"))
    debug = ctypes.CDLL(os.path.join(ROOT, "bioreason_amd", "libbioreason_hip_debug.so"))
    for name in dbg:
        assert not hasattr(product, name), f"{name} leaked into the product library"
        assert hasattr(debug, name), name
    for name in pub:
        assert hasattr(debug, name), name
    lib = _lib.KernelLibrary(os.path.join(ROOT, "bioreason_amd", "libbioreason_hip_debug.so"), debug=True)
    assert set(lib.protos) == set(pub) | set(dbg)


def test_product_refuses_cpu_tensors_and_missing_library(tmp_path):
    import pytest
    import torch
    from bioreason_amd import _lib
    with pytest.raises(RuntimeError):
        _lib.KernelLibrary(str(tmp_path / "nope.so"))
    lib = _lib.KernelLibrary(os.path.join(ROOT, "bioreason_amd", "libbioreason_hip.so"))
    with pytest.raises(RuntimeError):          # no silent CPU path: host tensors are rejected before any launch
        lib.call("bra_rmsnorm_fwd", torch.zeros(4, 8), 8, torch.zeros(8), torch.zeros(4, 8), 8, None, 4, 8, 1e-6, 0)
