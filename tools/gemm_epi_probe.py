"""interior bf16 epilogue of the tiled GEMMs: direct 8-byte stores (rounds 2-5) against the LDS-turned full-line form (round 6), per shape
with the per-shape kernel choice; outputs compared bit for bit.   python tools/gemm_epi_probe.py   (GPU box, debug library)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bioreason_amd import ops
from bioreason_amd import _lib as _bra_lib
_bra_lib.use_debug_library()
from bioreason_amd._lib import get_lib
dev = torch.device("cuda:0"); BF = torch.bfloat16
def timeit(fn, iters=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / iters
shapes = []
for tag, M in (("p", 2180), ("c", 2048), ("s", 17440)):
    shapes += [(tag + "_qkv", M, 4096, 2048, 128, 0), (tag + "_o", M, 2048, 2048, 64, 1), (tag + "_gate_up", M, 12288, 2048, 64, 0),
               (tag + "_down", M, 2048, 6144, 64, 1), (tag + "_d_gate_up", M, 2048, 12288, 128, 1), (tag + "_d_down", M, 6144, 2048, 64, 0),
               (tag + "_d_qkv", M, 2048, 4096, 64, 1)]
shapes += [("e_qkv", 16384, 3072, 1024, 0, 0), ("e_ffn_up", 16384, 8192, 1024, 0, 0), ("e_ffn_dn", 16384, 1024, 4096, 0, 1)]
tot = [0.0, 0.0]
for name, M, N, K, K2, res in shapes:
    a = torch.randn(M, K, device=dev).to(BF); b = torch.randn(N, K, device=dev).to(BF)
    a2 = torch.randn(M, K2, device=dev).to(BF) if K2 else None; b2 = torch.randn(N, K2, device=dev).to(BF) if K2 else None
    rs = torch.randn(M, N, device=dev).to(BF) if res else None
    out = []
    ts = []
    for on in (0, 1):
        get_lib().call("bra_gemm_set_epi_lds", on)
        c = torch.empty(M, N, dtype=BF, device=dev)
        ts.append(timeit(lambda: ops.gemm_nt(a, b, a2=a2, b2=b2, out=c, res=rs)))
        out.append(c.clone())
    fl = 2.0 * M * N * (K + K2)
    tot[0] += ts[0]; tot[1] += ts[1]
    print(f"{name:12s} M {M:6d} N {N:6d} K {K:6d} res {res}  direct {ts[0] * 1e3:7.1f} us {fl / ts[0] / 1e9:6.0f} TF/s   via LDS {ts[1] * 1e3:7.1f} us {fl / ts[1] / 1e9:6.0f} TF/s   "
          f"x{ts[0] / ts[1]:5.3f}   identical {bool(torch.equal(out[0], out[1]))}", flush=True)
get_lib().call("bra_gemm_set_epi_lds", 1)
print(f"sum direct {tot[0]:.3f} ms, via LDS {tot[1]:.3f} ms")
