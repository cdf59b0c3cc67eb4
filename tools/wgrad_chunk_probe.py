"""wgrad_tn (LoRA weight gradients): rows of m per workgroup.  The launcher aims at ~512 workgroups of >= 256 rows; each walks its rows in 32-row
steps with one step prefetched — how many workgroups in flight does the chip want?   python tools/wgrad_chunk_probe.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bioreason_amd import ops
dev = torch.device("cuda:0"); BF = torch.bfloat16
def timeit(fn, iters=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / iters * 1e3
for M in (17440, 2180):
    for name, N, R, nt, drop in (("dB qkv", 4096, 128, 3, False), ("dB o/down", 2048, 64, 1, False), ("dB gate_up", 12288, 64, 2, False),
                                 ("dA qkv", 2048, 128, 3, True), ("dA o", 2048, 64, 1, True), ("dA gate_up", 2048, 64, 2, True), ("dA down", 6144, 64, 1, True)):
        y = torch.randn(M, N, device=dev).to(BF); t = torch.randn(M, R, device=dev).to(BF)
        out = torch.zeros((R, N) if drop else (N, R), dtype=torch.float32, device=dev)
        res = {}
        for mc in (0, 2048, 1024, 512, 256, 128, 64):
            res[mc] = timeit(lambda: ops.wgrad_tn(y, t, out, transposed_out=drop, m_chunk=mc, drop=(0.05, [11, 22, 33][:nt]) if drop else None))
        print(f"M {M:6d} {name:10s} N {N:5d} R {R:3d}: " + "  ".join(f"m_chunk {'auto' if mc == 0 else mc}: {v:5.1f}us" for mc, v in res.items())
              + f"   ({M * N * 2 / 1e6:.0f} MB)", flush=True)
