"""lora_down_drop at big M (SFT: 8 x 2180 rows; full-row policy pass: 8 x 2436): one workgroup per 32 rows walks K alone, one prefetched step
ahead — does the split-K form (more workgroups in flight, fp32 partial tiles + fixed-order reduce) pay there too?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bioreason_amd import ops
from bioreason_amd._lib import get_lib, current_stream
dev = torch.device("cuda:0"); BF = torch.bfloat16
def timeit(fn, iters=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / iters * 1e3
lib = get_lib()
for M in (17440, 19488, 2180):
    for name, K, R, nt in (("qkv", 2048, 128, 3), ("o", 2048, 64, 1), ("gate_up", 2048, 64, 2), ("down", 6144, 64, 1)):
        x = torch.randn(M, K, device=dev).to(BF); A = torch.randn(R, K, device=dev).to(BF)
        t = torch.empty(M, R, dtype=BF, device=dev)
        seeds = [11, 22, 33, 44][:nt] + [0] * (4 - nt)
        res = {}
        ref = None
        for p in (0.05, 0.0):
            for ks in (1, 2, 3, 4, 6):
                part = torch.empty((ks, M, R), dtype=torch.float32, device=dev)
                def run():
                    lib.call("bra_lora_down_drop_splitk", x, K, A, K, t, R, M, K, R, 2.0, p, *seeds, nt, part, ks, current_stream(x))
                res[(p, ks)] = timeit(run)
        print(f"M {M:6d} {name:8s} K {K} R {R}: " + "  ".join(f"p={p} ks={ks}: {v:5.1f}us" for (p, ks), v in res.items()), flush=True)
