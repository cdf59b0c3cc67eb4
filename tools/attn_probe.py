"""Attention kernels at the bench shape (B=8, S=2436, 16 q-heads / 8 kv-heads, hd 128, causal): ms and TFLOP/s of
bra_attn_fwd and bra_attn_bwd (dQ kernel + dK/dV kernel); run under rocprofv3 for per-kernel numbers."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bioreason_amd import ops

dev = torch.device("cuda:0")
B, S, Hq, Hkv, hd = int(os.environ.get("AP_B", 8)), int(os.environ.get("AP_S", 2436)), 16, 8, 128
g = torch.Generator(device="cpu").manual_seed(0)
def rnd(*s): return (torch.randn(*s, generator=g) * 0.5).to(torch.bfloat16).to(dev)
q, k, v, do = rnd(B, S, Hq, hd), rnd(B, S, Hkv, hd), rnd(B, S, Hkv, hd), rnd(B, S, Hq, hd)
kmask = torch.ones((B, S), dtype=torch.uint8, device=dev)
scale = hd ** -0.5
vt = ops.head_transpose(v)
fl_fwd = 4.0 * B * Hq * S * S * hd / 2
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
o, lse = ops.attn_fwd(q, k, vt, kmask, True, scale)
ms = timeit(lambda: ops.attn_fwd(q, k, vt, kmask, True, scale))
print("fwd ms %.3f  TFLOP/s %.1f" % (ms, fl_fwd / ms / 1e9), flush=True)
ms = timeit(lambda: ops.attn_bwd(q, k, v, o, do, lse, kmask, True, scale))
print("bwd (delta + 3 transposes + dq + dkv) ms %.3f  TFLOP/s (2.5x fwd flops) %.1f" % (ms, 2.5 * fl_fwd / ms / 1e9), flush=True)
dq, dk, dv = ops.attn_bwd(q, k, v, o, do, lse, kmask, True, scale)
print("checksums dq %.6e dk %.6e dv %.6e" % (dq.float().abs().sum().item(), dk.float().abs().sum().item(), dv.float().abs().sum().item()))
