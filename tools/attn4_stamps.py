"""Cycle stamps of the 4-wave attention forward's hot loop (debug library, k_attn4.hip BRA_STAMP): one mid-sequence workgroup, per wave.
AS_SHAPE=full|enc."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bioreason_amd import ops, _lib
if os.environ.get("AS_LIB"):                      # a variant build (tools/build_attn4_variant.sh)
    _lib._DEBUG_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(_lib.__file__)), os.environ["AS_LIB"])
lib = _lib.use_debug_library()
dev = torch.device("cuda:0")
name = os.environ.get("AS_SHAPE", "full")
B, S, Hq, Hkv, hd, causal = (16, 1024, 16, 16, 64, False) if name == "enc" else (8, 2436, 16, 8, 128, True)
g = torch.Generator(device="cpu").manual_seed(0)
def rnd(*s): return (torch.randn(*s, generator=g) * 0.5).to(torch.bfloat16).to(dev)
q, k, v = rnd(B, S, Hq, hd), rnd(B, S, Hkv, hd), rnd(B, S, Hkv, hd)
kmask = torch.ones((B, S), dtype=torch.uint8, device=dev)
vt = ops.head_transpose(v)
probe = torch.zeros(80, dtype=torch.int64, device=dev)
if os.environ.get('AS_FWD4'): lib.call('bra_attn_set_fwd4', int(os.environ['AS_FWD4']))
for _ in range(3): ops.attn_fwd(q, k, vt, kmask, causal, hd ** -0.5)
lib.call("bra_attn_set_probe", probe)
ops.attn_fwd(q, k, vt, kmask, causal, hd ** -0.5)
torch.cuda.synchronize()
lib.call("bra_attn_set_probe", None)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): ops.attn_fwd(q, k, vt, kmask, causal, hd ** -0.5)
e1.record(); torch.cuda.synchronize()
print(f"{name} [{os.environ.get('AS_LIB', 'debug')}]: {e0.elapsed_time(e1) / 20:.4f} ms per call")
pm = probe.cpu()[64:].view(8, 2)
p = probe.cpu()[:64].view(8, 8)
print(f"{name}: per wave [dma wait, barrier, step1, step2, tail] cycles per iteration; iterations; kernel cycles; steps")
for w in range(8):
    if int(p[w, 6]) == 0: continue
    it = max(1, int(p[w, 5]))
    print(f"  wave {w}: " + " ".join(f"{int(p[w, i]) / it:8.1f}" for i in range(5)) + f"   min step1/2 {int(pm[w, 0])} {int(pm[w, 1])}   sum {sum(int(p[w, i]) for i in range(5)) / it:8.1f}  iters {it}  kernel {int(p[w, 6])}  steps {int(p[w, 7])}")
