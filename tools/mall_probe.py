"""Does a decode projection run faster when its weights were touched just before (Infinity-Cache resident)?
cold: 32 distinct weight matrices in rotation (>1.6 GB, nothing survives); hot: the same matrix every launch."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bioreason_amd import ops
dev = torch.device("cuda:0")
H, F = 2048, 6144
def rnd(*s): return (torch.randn(*s, device=dev) * 0.02).to(torch.bfloat16)
x = rnd(8, H); nw = torch.ones(H, dtype=torch.bfloat16, device=dev)
ss = ops.row_sumsq(x, 256)
def run(ws, n, **kw):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): ops.dec_gemm2(kw.get("xin", x), ws[i % len(ws)], ss_in=kw.get("ss"), norm_w=kw.get("nw"), act=kw.get("act", False))
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, N, K, kw in (("gu", 2 * F, H, dict(ss=ss, nw=nw, act=True)), ("qkv", 4096, H, dict(ss=ss, nw=nw)), ("o", H, H, {}), ("down", H, F, dict(xin=rnd(8, F)))):
    ws = [rnd(N, K) for _ in range(32)]
    run(ws, 64, **kw)
    cold = run(ws, 256, **kw)
    hot = run(ws[:1], 256, **kw)
    print(f"{name:5s} N={N} K={K}  cold {cold:6.2f} us  hot {hot:6.2f} us   ({N*K*2/1e6:.1f} MB)", flush=True)
    del ws
