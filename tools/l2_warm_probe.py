"""VERDICT r5 #2a, upper bound: how much faster does a decode projection run when its weights are already in the L2 of the XCDs that
will read them?  `hot` = the same weight matrix every launch (same grid, same workgroup -> XCD map: the best case a successor-specific
warm-up could reach), `cold` = 32 matrices in rotation.  The launches are replayed from a hipGraph (64 per replay) so that the host's
issue rate (~10 us per ctypes call: tools/mall_probe.py measures that, not the kernels) is out of the picture.
Also `mixed`: every launch of the probed projection is preceded by a 25 MB streaming launch (down_proj of other weights), as in the token
loop, where the lines a warm-up would have left have to survive the next kernels' traffic."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bioreason_amd import ops
dev = torch.device("cuda:0")
H, F = 2048, 6144
def rnd(*s): return (torch.randn(*s, device=dev) * 0.02).to(torch.bfloat16)
x = rnd(8, H); nw = torch.ones(H, dtype=torch.bfloat16, device=dev)
ss = ops.row_sumsq(x, 256)
xd = rnd(8, F)
wd = [rnd(H, F) for _ in range(8)]
NL = 64
def graph_time(fn, reps=20):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(reps): g.replay()
        e1.record(s); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3 / NL
for name, N, K, kw in (("qkv", 4096, H, dict(ss_in=ss, norm_w=nw)), ("o", H, H, {}), ("down", H, F, dict(xin=xd)), ("gu", 2 * F, H, dict(ss_in=ss, norm_w=nw, act=True))):
    ws = [rnd(N, K) for _ in range(32)]
    xin = kw.pop("xin", x)
    def seq(wsel, mixed):
        def f():
            for i in range(NL):
                if mixed: ops.dec_gemm2(xd, wd[i % len(wd)])
                ops.dec_gemm2(xin, wsel[i % len(wsel)], **kw)
        return f
    cold, hot = graph_time(seq(ws, False)), graph_time(seq(ws[:1], False))
    base = graph_time(lambda: [ops.dec_gemm2(xd, wd[i % len(wd)]) for i in range(NL)])
    mcold, mhot = graph_time(seq(ws, True)) - base, graph_time(seq(ws[:1], True)) - base
    print(f"{name:5s} N={N:5d} K={K:4d} ({N*K*2/1e6:5.1f} MB)  alone: cold {cold:6.2f} us  hot {hot:6.2f} us  ({cold-hot:+.2f})   behind a 25 MB launch: cold {mcold:6.2f} us  hot {mhot:6.2f} us  ({mcold-mhot:+.2f})", flush=True)
    del ws
