"""LoRA-branch kernels under training-mode dropout at the cfg-3 shapes (M = 8 x 2436 rows): us per call and the rate at which
the large operand streams.  Groups of a Qwen3-1.7B layer: qkv (K 2048, 3 targets -> R 128 padded), o (K 2048, R pad 64?),
gate/up (K 2048, 2 targets), down (K 6144).  LORA_BENCH_LIB=<path> times another build of the library."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bioreason_amd import ops, _lib
if os.environ.get("LORA_BENCH_LIB"):
    _lib._lib = _lib.KernelLibrary(os.environ["LORA_BENCH_LIB"], emulated=False)
dev = torch.device("cuda:0")
BF = torch.bfloat16
M = 8 * 2436


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(*s, generator=g, device=dev).to(BF)
for name, K, N, R, nt in (("qkv", 2048, 4096, 128, 3), ("o", 2048, 2048, 64, 1), ("gate_up", 2048, 12288, 64, 2), ("down", 6144, 2048, 64, 1)):
    seeds = (11, 22, 33, 44)[:nt]                 # one mask stream per target module; the rest of R is padding
    x, dy = rn(M, K), rn(M, N)
    A, AT = rn(R, K), rn(K, R)
    t, dts = rn(M, R), rn(M, R)
    dA = torch.zeros(R, K, dtype=torch.float32, device=dev)
    dB = torch.zeros(N, R, dtype=torch.float32, device=dev)
    mb_x, mb_dy = M * K * 2 / 1e6, M * N * 2 / 1e6
    r = {}
    r["down_drop"] = timeit(lambda: ops.lora_down_drop(x, A, 2.0, 0.05, seeds))
    r["down_nodrop(gemm)"] = timeit(lambda: ops.gemm_nt(x, A, alpha=2.0))
    r["up_drop"] = timeit(lambda: ops.lora_up_drop(dts, AT, 0.05, seeds))
    r["wgrad_dA_drop"] = timeit(lambda: ops.wgrad_tn(x, dts, dA, transposed_out=True, drop=(0.05, seeds)))
    r["wgrad_dA_nodrop"] = timeit(lambda: ops.wgrad_tn(x, dts, dA, transposed_out=True))
    r["wgrad_dB"] = timeit(lambda: ops.wgrad_tn(dy, t, dB))
    BT = rn(R, N)
    r["dts_gemm(dy*B)"] = timeit(lambda: ops.gemm_nt(dy, BT, alpha=2.0))
    print(f"{name:8s} K={K} N={N} R={R}: " + "  ".join(f"{k} {v:.0f}us ({(mb_dy if ('dB' in k or 'dts' in k) else mb_x) / v:.2f} TB/s)" for k, v in r.items()), flush=True)
