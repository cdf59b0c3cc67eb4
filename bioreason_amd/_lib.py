"""ctypes binding of the C-ABI kernel library (include/bioreason_hip.h).

The prototypes are read from the header itself, so the header stays the single
statement of the boundary.  The product path opens ``libbioreason_hip.so`` (built
by ``__graft_entry__.build()`` / ``make -C bioreason_amd/csrc``) and raises if it
is missing — there is no CPU or PyTorch fallback.  ``use_library_for_tests`` lets
the CPU test-suite point the same Python host code at the kernel-source emulator
in ``tests/emu`` (test infrastructure; never used by bench.py / smoke / the
package itself).
"""
from __future__ import annotations

import ctypes
import os
import re
from typing import Dict, List, Optional, Tuple

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_HEADER = os.path.join(os.path.dirname(_HERE), "include", "bioreason_hip.h")
_DEBUG_HEADER = os.path.join(os.path.dirname(_HERE), "include", "bioreason_hip_debug.h")
_LIB_PATH = os.path.join(_HERE, "libbioreason_hip.so")
_DEBUG_LIB_PATH = os.path.join(_HERE, "libbioreason_hip_debug.so")

_CTYPES = {
    "int": ctypes.c_int,
    "unsigned": ctypes.c_uint,
    "long": ctypes.c_long,
    "float": ctypes.c_float,
}


def parse_header(path: str = _HEADER) -> Dict[str, List[Tuple[str, str]]]:
    """-> {function name: [(ctype kind, arg name), ...]} for every `int bra_*(...)` prototype."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos: Dict[str, List[Tuple[str, str]]] = {}
    for m in re.finditer(r"\bint\s+(bra_\w+)\s*\(([^)]*)\)\s*;", src):
        name, args = m.group(1), m.group(2).strip()
        out: List[Tuple[str, str]] = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                if "*" in a:
                    out.append(("ptr", a.split("*")[-1].strip()))
                else:
                    toks = a.split()
                    kind = toks[-2] if toks[-2] != "const" else toks[-3]
                    out.append((kind, toks[-1]))
        protos[name] = out
    return protos


BRA_ERR_ARG, BRA_ERR_UNSUPPORTED = -1, -2          # include/bioreason_hip.h


class KernelError(RuntimeError):
    """a C-ABI entry point returned a non-zero status; `.status` is the integer (negative: BRA_ERR_*, positive: hipError_t)"""

    def __init__(self, name: str, status: int):
        kind = {BRA_ERR_ARG: " (argument error)", BRA_ERR_UNSUPPORTED: " (unsupported shape / option)"}.get(
            status, " (argument error)" if status < 0 else " (hipError_t)")
        super().__init__(f"{name} failed with status {status}{kind}")
        self.name, self.status = name, int(status)


class KernelLibrary:
    def __init__(self, path: str, emulated: bool = False, debug: bool = False):
        """`debug`: the library was built with -DBRA_DEBUG and also exports include/bioreason_hip_debug.h (knobs, probes, the
        persistent decode step); the emulator build of tests/emu always is."""
        if not os.path.exists(path):
            raise RuntimeError(
                f"bioreason_amd: kernel library {path} is missing. Build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950). "
                "There is no CPU / PyTorch fallback for the hot path."
            )
        self.path = path
        self.emulated = emulated
        self.debug = bool(debug or emulated)
        self._dll = ctypes.CDLL(path)
        self.protos = parse_header()
        if self.debug:
            self.protos.update(parse_header(_DEBUG_HEADER))
        self._fn = {}
        for name, args in self.protos.items():
            try:
                f = getattr(self._dll, name)
            except AttributeError as e:  # header and library out of sync
                raise RuntimeError(f"{path} does not export {name} declared in {_HEADER}") from e
            f.restype = ctypes.c_int
            f.argtypes = [ctypes.c_void_p if k == "ptr" else _CTYPES[k] for k, _ in args]
            self._fn[name] = f
        # per function: (foreign function, number of arguments, indices of the pointer arguments) — the step issues thousands of
        # these calls from Python, so the per-call work is kept to one pass over the pointer arguments only
        self._fast = {name: (self._fn[name], len(args), tuple(i for i, (k, _) in enumerate(args) if k == "ptr"))
                      for name, args in self.protos.items()}

    def call_rc(self, name: str, *args) -> int:
        """like call(), but hands BRA_ERR_UNSUPPORTED (-2) back to the caller instead of raising"""
        try:
            return self.call(name, *args)
        except KernelError as e:
            if e.status == BRA_ERR_UNSUPPORTED:
                return BRA_ERR_UNSUPPORTED
            raise

    def call(self, name: str, *args) -> int:
        f, n, ptrs = self._fast[name]
        if len(args) != n:
            raise TypeError(f"{name}: expected {n} arguments, got {len(args)}")
        conv = list(args)                      # ints / floats go to ctypes as they are (argtypes convert them)
        check = not self.emulated
        for i in ptrs:
            a = conv[i]
            if a is None:
                continue
            if isinstance(a, torch.Tensor):
                if check and not a.is_cuda:
                    raise RuntimeError(f"{name}: argument {self.protos[name][i][1]} is a CPU tensor; the HIP library needs device memory")
                conv[i] = a.data_ptr()
            else:
                conv[i] = int(a)
        try:
            rc = f(*conv)
        except (ctypes.ArgumentError, TypeError):
            # an argument ctypes cannot coerce by itself (a float where the prototype says int, a 0-d tensor, ...): convert by kind
            for i, (kind, _) in enumerate(self.protos[name]):
                if kind != "ptr":
                    conv[i] = float(conv[i]) if kind == "float" else int(conv[i])
            rc = f(*conv)
        if rc != 0:
            raise KernelError(name, rc)
        return rc


_lib: Optional[KernelLibrary] = None


def get_lib() -> KernelLibrary:
    global _lib
    if _lib is None:
        _lib = KernelLibrary(_LIB_PATH, emulated=False)
    return _lib


def use_debug_library() -> KernelLibrary:
    """Opt into libbioreason_hip_debug.so (the same kernels + include/bioreason_hip_debug.h: tile-variant knobs, probes, the
    persistent decode step) for this process: tests that pin a variant, tools/, bench.py's A/B flags.  Never the default."""
    global _lib
    if _lib is not None and _lib.debug:
        return _lib
    if not os.path.exists(_DEBUG_LIB_PATH):
        raise RuntimeError(f"bioreason_amd: {_DEBUG_LIB_PATH} is missing; build it with `make -C bioreason_amd/csrc debug` "
                           "(__graft_entry__.build() does)")
    _lib = KernelLibrary(_DEBUG_LIB_PATH, emulated=False, debug=True)
    return _lib


def use_library_for_tests(path: str) -> KernelLibrary:
    """TEST HOOK: run the host code against the kernel-source emulator (tests/emu)."""
    global _lib
    _lib = KernelLibrary(path, emulated=True)
    return _lib


def reset_library() -> None:
    global _lib
    _lib = None


def current_stream(t: Optional[torch.Tensor] = None) -> int:
    if t is not None and t.is_cuda:
        return torch.cuda.current_stream(t.device).cuda_stream
    if torch.cuda.is_available() and (_lib is None or not _lib.emulated):
        return torch.cuda.current_stream().cuda_stream
    return 0
