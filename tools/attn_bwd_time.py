import os, sys, torch
sys.path.insert(0, os.getcwd())
from bioreason_amd import ops, _lib
_lib._DEBUG_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(_lib.__file__)), os.environ.get("AS_LIB", "libbioreason_hip_debug.so"))
lib = _lib.use_debug_library()
dev = torch.device("cuda:0")
B, S, Hq, Hkv, hd = 8, 2436, 16, 8, 128
g = torch.Generator().manual_seed(0)
def rnd(*s): return (torch.randn(*s, generator=g) * 0.5).to(torch.bfloat16).to(dev)
q, k, v, do = rnd(B, S, Hq, hd), rnd(B, S, Hkv, hd), rnd(B, S, Hkv, hd), rnd(B, S, Hq, hd)
kmask = torch.ones((B, S), dtype=torch.uint8, device=dev)
vt = ops.head_transpose(v)
o, lse = ops.attn_fwd(q, k, vt, kmask, True, hd ** -0.5)
import time
for it in range(2):
    for _ in range(3): ops.attn_bwd(q, k, v, o, do, lse, kmask, True, hd ** -0.5)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.attn_bwd(q, k, v, o, do, lse, kmask, True, hd ** -0.5)
    e1.record(); torch.cuda.synchronize()
print(os.environ.get("AS_LIB"), f"bwd total {e0.elapsed_time(e1) / 20:.4f} ms")
