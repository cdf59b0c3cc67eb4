"""Fixed (K-independent) part of the step's GEMM launches: time(K) for the one-prompt shapes at K = 64 .. 2048 -> intercept and slope.
What does not scale with K is launch ramp + prologue + EPILOGUE (accumulators -> bf16 -> stores) + drain."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bioreason_amd import ops
dev = torch.device("cuda:0")
BF = torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: (torch.randn(*s, generator=g, device=dev) * 0.1).to(BF)

def timeit(fn, n=30):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

for (M, N, name) in [(2180, 4096, "p_qkv"), (2180, 12288, "p_gate_up"), (2180, 2048, "p_o/down"), (2048, 12288, "c_gate_up"), (19488, 4096, "full_qkv")]:
    row = []
    for K in (64, 256, 512, 1024, 2048, 6144):
        a, b = rn(M, K), rn(N, K)
        us = timeit(lambda: ops.gemm_nt(a, b))
        row.append((K, us))
    (k0, t0), (k1, t1) = row[2], row[4]
    slope = (t1 - t0) / (k1 - k0)
    fixed = t1 - slope * k1
    tf = 2.0 * M * N * 2048 / row[4][1] / 1e6
    print(f"{name:10s} M={M:5d} N={N:5d}  " + "  ".join(f"K={k}: {t:6.1f}us" for k, t in row) + f"   -> fixed {fixed:5.1f} us, {slope * 1024:5.1f} us per 1024 of K; K=2048: {tf:6.0f} TF/s, asymptote {2.0 * M * N / slope / 1e6:6.0f} TF/s")
