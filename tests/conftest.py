import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

EMU_LIB = os.path.join(ROOT, "tests", "emu", "libbioreason_emu.so")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_addoption(parser):
    parser.addoption("--reverse-modules", action="store_true", default=False,
                     help="run the test MODULES in reversed order (tests inside a module keep theirs): the suite must not depend on "
                          "which module imported `bioreason` / touched sys.path first")


def pytest_collection_modifyitems(config, items):
    if not config.getoption("--reverse-modules"):
        return
    groups, order = {}, []
    for it in items:
        key = it.nodeid.split("::", 1)[0]
        if key not in groups:
            groups[key] = []
            order.append(key)
        groups[key].append(it)
    items[:] = [it for key in reversed(order) for it in groups[key]]


def _build_emu():
    import fcntl
    os.makedirs(os.path.join(ROOT, "tests", "emu", "build"), exist_ok=True)
    with open(os.path.join(ROOT, "tests", "emu", "build", ".lock"), "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)      # xdist workers build once, the others wait
        subprocess.run(["make", "-s", "-j8", "-C", os.path.join(ROOT, "bioreason_amd", "csrc"), "emu"], check=True,
                       stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def emu_lib_path():
    """The kernel-source emulator (tests/emu): same .hip sources, host fibers. Test infrastructure only."""
    _build_emu()
    return EMU_LIB


@pytest.fixture(params=["emu", pytest.param("hip", marks=pytest.mark.gpu)])
def backend(request, emu_lib_path):
    """Runs a kernel test twice: on the CPU emulator (CI without GPU) and on the real HIP library (-m gpu)."""
    from bioreason_amd import _lib
    if request.param == "emu":
        _lib.use_library_for_tests(emu_lib_path)
        yield torch.device("cpu")
        _lib.reset_library()
    else:
        if not torch.cuda.is_available():
            pytest.skip("no GPU")
        _lib.reset_library()
        lib = _lib.get_lib()          # raises if libbioreason_hip.so is missing
        assert not lib.emulated
        yield torch.device("cuda:0")
        torch.cuda.synchronize()


@pytest.fixture(params=["emu", pytest.param("hip", marks=pytest.mark.gpu)])
def debug_backend(request, emu_lib_path):
    """like `backend`, for tests that pin a tile variant / use a probe (include/bioreason_hip_debug.h): on the GPU these run against
    libbioreason_hip_debug.so — the same kernel sources built with -DBRA_DEBUG — never against the product library, which has no knobs"""
    from bioreason_amd import _lib
    if request.param == "emu":
        _lib.use_library_for_tests(emu_lib_path)
        yield torch.device("cpu")
        _lib.reset_library()
    else:
        if not torch.cuda.is_available():
            pytest.skip("no GPU")
        _lib.reset_library()
        lib = _lib.use_debug_library()
        assert lib.debug and not lib.emulated
        yield torch.device("cuda:0")
        torch.cuda.synchronize()
        _lib.reset_library()


@pytest.fixture
def hip_debug_device():
    """Real-GPU-only tests of the debug build (knobs, probes, the persistent decode step)."""
    from bioreason_amd import _lib
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    _lib.reset_library()
    _lib.use_debug_library()
    yield torch.device("cuda:0")
    torch.cuda.synchronize()
    _lib.reset_library()


@pytest.fixture
def hip_device():
    """Real-GPU-only tests."""
    from bioreason_amd import _lib
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    _lib.reset_library()
    _lib.get_lib()
    yield torch.device("cuda:0")
    torch.cuda.synchronize()
