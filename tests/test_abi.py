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


def test_product_refuses_cpu_tensors_and_missing_library(tmp_path):
    import pytest
    import torch
    from bioreason_amd import _lib
    with pytest.raises(RuntimeError):
        _lib.KernelLibrary(str(tmp_path / "nope.so"))
    lib = _lib.KernelLibrary(os.path.join(ROOT, "bioreason_amd", "libbioreason_hip.so"))
    with pytest.raises(RuntimeError):          # no silent CPU path: host tensors are rejected before any launch
        lib.call("bra_rmsnorm_fwd", torch.zeros(4, 8), 8, torch.zeros(8), torch.zeros(4, 8), 8, None, 4, 8, 1e-6, 0)
