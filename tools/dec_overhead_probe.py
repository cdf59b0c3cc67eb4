"""where do the microseconds of a decode projection go?  bra_dec_gemm2 at the qkv / o / gate_up / down shapes with
(a) weights rotated through > 600 MB (HBM-served), (b) one weight matrix re-used (Infinity-Cache / L2 served),
(c) K shrunk to 64 (fixed cost of launch + prologue + reduction + epilogue, almost no bytes).
GPU-side durations: run under `rocprofv3 --kernel-trace --stats` (tools/dec_overhead.sh), one case per process:
    dec_overhead_probe.py <name> <hbm|cached|k64>"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bioreason_amd import ops
dev = torch.device("cuda:0"); BF = torch.bfloat16
M = 8
CASES = {"qkv": (4096, 2048, 0, 1, 0), "o": (2048, 2048, 0, 0, 1), "gate_up": (12288, 2048, 1, 1, 0), "down": (2048, 6144, 0, 0, 1)}
name, tag = sys.argv[1], sys.argv[2]
N, K, act, norm, res = CASES[name]
Kx = 64 if tag == "k64" else K
rotate = tag == "hbm"
ncopy = max(2, int(600e6 // (N * Kx * 2)) + 1) if rotate else 1
Ws = [torch.randn(N, Kx, device=dev).to(BF) for _ in range(min(ncopy, 40))]
x = torch.randn(M, Kx, device=dev).to(BF); nw = torch.ones(Kx, device=dev).to(BF)
ss = ops.row_sumsq(x, 256) if norm else None
r = torch.randn(M, N, device=dev).to(BF) if res else None
for i in range(300):
    ops.dec_gemm2(x, Ws[i % len(Ws)], ss_in=ss, norm_w=nw if norm else None, res=r, act=bool(act), want_ss=bool(res))
torch.cuda.synchronize()
