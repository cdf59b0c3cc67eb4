"""row split of the ring GEMM (k_gemm.hip ring_split_rows): TFLOP/s of the per-shape choice with and without it on the
big-M shapes of a Qwen3-1.7B layer, and bit-equality of the two results."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bioreason_amd import ops
from bioreason_amd import _lib as _bra_lib
_bra_lib.use_debug_library()          # knobs / probes / persistent step: libbioreason_hip_debug.so (include/bioreason_hip_debug.h)
from bioreason_amd._lib import get_lib
dev = torch.device("cuda:0"); BF = torch.bfloat16


def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / iters


T = 8 * 2436
shapes = [("qkv", T, 4096, 2048, 128), ("o", T, 2048, 2048, 64), ("gate_up", T, 12288, 2048, 64), ("down", T, 2048, 6144, 64),
          ("d_gate_up", T, 2048, 12288, 128), ("d_down", T, 6144, 2048, 64), ("d_qkv", T, 2048, 4096, 64), ("d_o", T, 2048, 2048, 64)]
tot = {0: 0.0, 1: 0.0}
for name, M, N, K, K2 in shapes:
    a = torch.randn(M, K, device=dev).to(BF); b = torch.randn(N, K, device=dev).to(BF)
    a2 = torch.randn(M, K2, device=dev).to(BF); b2 = torch.randn(N, K2, device=dev).to(BF)
    res = torch.randn(M, N, device=dev).to(BF)
    out = {}
    tf = {}
    for on in (0, 1):
        get_lib().call("bra_gemm_set_row_split", on)
        out[on] = ops.gemm_nt(a, b, a2=a2, b2=b2, res=res)
        c = torch.empty(M, N, dtype=BF, device=dev)
        ms = timeit(lambda: ops.gemm_nt(a, b, a2=a2, b2=b2, res=res, out=c))
        tf[on] = 2.0 * M * N * (K + K2) / ms / 1e9
        tot[on] += ms
    get_lib().call("bra_gemm_set_row_split", 1)
    print(f"{name:10s} M={M} N={N} K={K}+{K2}: no split {tf[0]:.0f} TF  split {tf[1]:.0f} TF   identical: {bool(torch.equal(out[0], out[1]))}", flush=True)
print("sum of the eight: %.3f ms -> %.3f ms" % (tot[0], tot[1]))
