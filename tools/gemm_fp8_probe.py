"""fp8 x fp8 GEMM (bra_gemm_fp8_nt, v_mfma_scale_f32_16x16x128_f8f6f4) against the bf16 GEMM on the step's forward shapes, and the
per-row quantisation kernels:   python tools/gemm_fp8_probe.py     (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bioreason_amd import ops
dev = torch.device("cuda:0"); BF = torch.bfloat16
def timeit(fn, iters=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / iters
for tag, M in (("p", 2180), ("c", 2048), ("s", 17440)):
    for name, N, K in (("qkv", 4096, 2048), ("o", 2048, 2048), ("gate_up", 12288, 2048), ("down", 2048, 6144)):
        x = torch.randn(M, K, device=dev).to(BF); w = (torch.randn(N, K, device=dev) * 0.02).to(BF)
        xq, xs = ops.quant_rows_fp8(x); wq, ws = ops.quant_rows_fp8(w)
        c = torch.empty(M, N, dtype=BF, device=dev)
        t8 = timeit(lambda: ops.gemm_fp8_nt(xq, xs, wq, ws, out=c))
        t16 = timeit(lambda: ops.gemm_nt(x, w, out=c))
        tq = timeit(lambda: ops.quant_rows_fp8(x))
        fl = 2.0 * M * N * K
        print(f"{tag}_{name:8s} M {M:6d} N {N:6d} K {K:5d}  fp8 {t8 * 1e3:7.1f} us {fl / t8 / 1e9:7.0f} TF/s   bf16 {t16 * 1e3:7.1f} us {fl / t16 / 1e9:7.0f} TF/s   "
              f"x{t16 / t8:4.2f}   quant rows {tq * 1e3:6.1f} us ({M * K * 3 / tq / 1e6:5.0f} GB/s)", flush=True)
gu = torch.randn(2180, 12288, device=dev).to(BF)
print(f"swiglu_quant 2180 x 6144: {timeit(lambda: ops.swiglu_quant_fp8(gu)) * 1e3:.1f} us   swiglu_fwd {timeit(lambda: ops.swiglu_fwd(gu)) * 1e3:.1f} us")
